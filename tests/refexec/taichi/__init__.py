"""tests/refexec/taichi -- TEST INFRASTRUCTURE: a host-side emulation of the part of the Taichi DSL that the reference's renderer
modules use, so that the reference's OWN SOURCE FILES (/root/reference/renderer/*.py, imported from where they lie, never copied)
execute as plain Python and can generate golden vectors for the oracle (tests/golden/make_reference_vectors.py).  This is NOT
Taichi and it is not the product's `taichi/` shim (which only runs the example scripts' voxel-authoring kernels): it is one
reading of Taichi's documented semantics --

  * @ti.func / @ti.kernel bodies are re-compiled through an AST pass (_Scoping): every binary operator goes through Taichi's type
    promotion (_sbin: f32 * i32 is an f32 product, not numpy's float64; a Python number is a compile-time constant that takes the
    type of the typed operand it meets; two Python numbers stay Python numbers), every assignment through _assign (the first one
    fixes a variable's type, later ones convert to it; vectors / matrices / structs are copied: value semantics), arguments are
    passed by value except ti.template() parameters, which the callee may assign to (_BYREF: _trace_sdf, _trace_voxel);
  * default_fp = f32, default_ip = i32; ti.cast(float -> int) truncates, to f16 rounds to nearest even; u32 arithmetic wraps;
  * in PYTHON scope (outside kernels) vectors hold Python numbers and compute in double, as Taichi's do (Atmos.__init__'s
    coefficients): they are rounded to f32 once, where a kernel uses them (_scope);
  * v.normalized() = v * (1 / sqrt(sum of squares)), sums left to right; mat3(a, b, c) stacks rows; a constant whole-number
    power is repeated multiplication by squaring (Taichi's algebraic simplification); ti.max / ti.min are maxnum / minnum;
  * fields: dense, blocked (ti.root.dense(...).dense(...).place) and offset layouts, vector / matrix / struct fields, 0-d fields
    through [None]; rgba8 / rgba32f textures (unorm8 store rounds to nearest, fetch returns byte / 255);
  * the outermost for of a kernel over a field or an ndrange is a PARALLEL loop (_parallel): an iteration sees the value an
    element had before the loop wherever another iteration has written it meanwhile;
  * what Taichi leaves undefined is the CALLER's to supply: ti.random() (set_random_source: receives the index of the running
    parallel iteration), the elementary functions (set_elementary; numpy's by default), reads outside a field
    (set_out_of_bounds_reads: "clamp" / "zero" / "error", per field); reads past the end of a 1-D field give 0 and writes grow
    it (the reference's LOD base formula, raytracer.py:32, indexes past its own allocation).

What the vectors pin: the oracle against the reference's source text executed under these semantics -- not against Taichi's
code generation (fast-math, its own sin / cos / pow), which stays unobservable here.
"""
import ast as _ast
import builtins as _b
import inspect as _inspect
import itertools as _it
import math as _m
import textwrap as _tw
import warnings as _w

import numpy as _np

_w.filterwarnings("ignore", category=RuntimeWarning)   # u32 wrap-around, 1 / 0, 0 * inf are part of the semantics

f16, f32, f64 = _np.float16, _np.float32, _np.float64
i8, i16, i32, i64 = _np.int8, _np.int16, _np.int32, _np.int64
u8, u16, u32, u64 = _np.uint8, _np.uint16, _np.uint32, _np.uint64
cpu, gpu, vulkan, cuda = "cpu", "gpu", "vulkan", "cuda"


def init(*a, **k):
    return None


def _norm(x):
    """default_fp / default_ip: nothing wider than 32 bits comes out of an operation on typed values."""
    t = type(x)
    if t is _np.float64:
        return _np.float32(x)
    if t is _np.int64:
        return _np.int32(x)
    if t is _np.bool_:
        return _b.bool(x)
    return x


_scope = [0]    # > 0 inside a ti.kernel / ti.func.  In Python scope Taichi's vectors hold plain Python numbers and their arithmetic is
                # Python's (double): Atmos.__init__ derives its coefficients that way (atmos.py:36-50) and they are rounded to f32
                # once, where a kernel uses them


def _typed(x):
    """A Python number entering a vector / a typed slot."""
    if _scope[0] == 0 and isinstance(x, (_b.int, _b.float)):
        return x
    if isinstance(x, _b.bool):
        return _np.int32(x)
    if isinstance(x, _b.int):
        return _np.int32(x) if -2**31 <= x < 2**31 else _np.uint32(x & 0xFFFFFFFF)
    if isinstance(x, _b.float):
        return _np.float32(x)
    return _norm(x)


def _is_float(x):
    return isinstance(x, (_b.float, _np.floating))


# ---- scalar arithmetic: Taichi's type promotion, not numpy's ---------------------------------------------------------------
# (numpy would compute float32 * int32 in float64; Taichi converts the integer to f32 and multiplies in f32.)  Python numbers
# are compile-time constants: they take the type of the typed operand they meet (a float constant beside an integer value makes
# both default_fp = f32); two Python numbers stay Python numbers.
_FLOATS = (_np.float16, _np.float32, _np.float64)


def _rank(t):
    return (2 if issubclass(t, _np.floating) else 1 if issubclass(t, _np.unsignedinteger) else 0, _np.dtype(t).itemsize)


def _promote(a, c):
    ta, tc = type(a), type(c)
    wa, wc = not isinstance(a, _np.generic), not isinstance(c, _np.generic)
    if wa and wc:
        return None
    if wa or wc:
        w, st = (a, tc) if wa else (c, ta)
        if st is _np.bool_:
            st = _np.int32
        if isinstance(w, _b.float) and not issubclass(st, _np.floating):
            return _np.float32
        return st
    if ta is _np.bool_:
        ta = _np.int32
    if tc is _np.bool_:
        tc = _np.int32
    if ta is tc:
        return ta
    fa, fc = issubclass(ta, _np.floating), issubclass(tc, _np.floating)
    if fa != fc:
        return ta if fa else tc
    sa, sc = _np.dtype(ta).itemsize, _np.dtype(tc).itemsize
    if sa != sc:
        return ta if sa > sc else tc
    return ta if issubclass(ta, _np.unsignedinteger) else tc


def _conv(x, t):
    if type(x) is t:
        return x
    if issubclass(t, _np.floating):
        return t(x)
    if isinstance(x, (_b.float, _np.floating)):
        return _cast1(x, t)
    return _np.int64(_b.int(x) & 0xFFFFFFFFFFFFFFFF if _b.int(x) >= 2**63 else _b.int(x)).astype(t) if -2**63 <= _b.int(x) < 2**64 else t(0)


_PYOPS = {"add": lambda a, c: a + c, "sub": lambda a, c: a - c, "mul": lambda a, c: a * c, "truediv": lambda a, c: a / c,
          "floordiv": lambda a, c: a // c, "mod": lambda a, c: a % c, "pow": lambda a, c: a ** c, "lshift": lambda a, c: a << c,
          "rshift": lambda a, c: a >> c, "and": lambda a, c: a & c, "or": lambda a, c: a | c, "xor": lambda a, c: a ^ c}


def _sbin(op, a, c):
    """One scalar operation."""
    if isinstance(a, _b.bool):
        a = _b.int(a)
    if isinstance(c, _b.bool):
        c = _b.int(c)
    if op == "pow":
        return _pow(a, c)
    t = _promote(a, c)
    if t is None:
        return _PYOPS[op](a, c)
    if op == "truediv" and not issubclass(t, _np.floating):
        t = _np.float32
    if t is _np.float64:
        t = _np.float32
    if t is _np.int64:
        t = _np.int32
    if op in ("lshift", "rshift") and not isinstance(a, _np.generic):
        t = _np.int32 if not isinstance(c, _np.unsignedinteger) else type(c)   # 1 << n: the constant is an i32
    return _PYOPS[op](_conv(a, t), _conv(c, t))


def _binop(op, a, c):
    """`a op c` inside a ti.func / ti.kernel (the AST pass routes every binary operator here)."""
    if isinstance(a, (Vector, Matrix)) or isinstance(c, (Vector, Matrix)):
        return _PYOPS[op](a, c) if op != "matmul" else a @ c
    if isinstance(a, (_b.int, _b.float, _np.generic)) and isinstance(c, (_b.int, _b.float, _np.generic)):
        return _sbin(op, a, c)
    return _PYOPS[op](a, c)    # lists, strings, ... outside the DSL's value types



class Vector:
    __slots__ = ("_v",)
    __hash__ = None
    __array_ufunc__ = None    # numpy scalars leave `scalar * vector` to the vector's reflected operators
    _AXES = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}

    def __init__(self, vals, dt=None):
        out = []
        for x in (vals._v if isinstance(vals, Vector) else vals):
            if isinstance(x, Vector):
                out.extend(x._v)
            elif isinstance(x, (list, tuple)):
                out.extend(_typed(y) for y in x)
            else:
                out.append(_typed(x))
        if _scope[0] == 0 and _b.all(isinstance(x, (_b.int, _b.float)) for x in out):
            pass    # Python scope, Python numbers: kept as they are
        elif dt is not None:
            out = [_cast1(x, dt) for x in out]
        elif any(_is_float(x) for x in out):   # a vector has ONE element type: float wins
            out = [x if _is_float(x) else _np.float32(x) for x in out]
        object.__setattr__(self, "_v", out)

    @staticmethod
    def _new(vals):
        v = object.__new__(Vector)
        object.__setattr__(v, "_v", vals)
        return v

    def _copy(self):
        return Vector._new(list(self._v))

    def __len__(self): return len(self._v)
    def __iter__(self): return iter(self._v)
    def __getitem__(self, i): return self._v[_b.int(i)]
    def __setitem__(self, i, val):
        i = _b.int(i)
        self._v[i] = _cast1(val, type(self._v[i]))
    def __repr__(self): return f"Vector({[x.item() if hasattr(x, 'item') else x for x in self._v]})"
    def to_list(self): return [x.item() if hasattr(x, "item") else x for x in self._v]

    def __getattr__(self, name):
        try:
            idx = [Vector._AXES[c] for c in name]
        except KeyError:
            raise AttributeError(name) from None
        if len(idx) == 1:
            return self._v[idx[0]]
        return Vector._new([self._v[i] for i in idx])

    def __setattr__(self, name, value):
        idx = [Vector._AXES[c] for c in name]
        if len(idx) == 1:
            self._v[idx[0]] = _cast1(value, type(self._v[idx[0]]))
        else:
            vals = value._v if isinstance(value, Vector) else list(value)
            for i, val in zip(idx, vals):
                self._v[i] = _cast1(val, type(self._v[i]))

    def _bin(self, o, fn, rev=False):
        if isinstance(o, Vector):
            if len(o._v) != len(self._v):
                raise ValueError("vector sizes differ")
            ov = o._v
        elif isinstance(o, (list, tuple)):
            ov = [_typed(x) for x in o]
        else:
            ov = [o] * len(self._v)
        if rev:
            return Vector._new([_norm(fn(c, a)) for a, c in zip(self._v, ov)])
        return Vector._new([_norm(fn(a, c)) for a, c in zip(self._v, ov)])

    def __add__(self, o): return self._bin(o, lambda a, c: _sbin("add", a, c))
    def __radd__(self, o): return self._bin(o, lambda a, c: _sbin("add", a, c), True)
    def __sub__(self, o): return self._bin(o, lambda a, c: _sbin("sub", a, c))
    def __rsub__(self, o): return self._bin(o, lambda a, c: _sbin("sub", a, c), True)
    def __mul__(self, o): return self._bin(o, lambda a, c: _sbin("mul", a, c))
    def __rmul__(self, o): return self._bin(o, lambda a, c: _sbin("mul", a, c), True)
    def __truediv__(self, o): return self._bin(o, lambda a, c: _sbin("truediv", a, c))
    def __rtruediv__(self, o): return self._bin(o, lambda a, c: _sbin("truediv", a, c), True)
    def __pow__(self, o): return self._bin(o, _pow)
    def __rpow__(self, o): return self._bin(o, _pow, True)
    def __lshift__(self, o): return self._bin(o, lambda a, c: _sbin("lshift", a, c))
    def __rshift__(self, o): return self._bin(o, lambda a, c: _sbin("rshift", a, c))
    def __rrshift__(self, o): return self._bin(o, lambda a, c: _sbin("rshift", a, c), True)
    def __rlshift__(self, o): return self._bin(o, lambda a, c: _sbin("lshift", a, c), True)
    def __and__(self, o): return self._bin(o, lambda a, c: _sbin("and", a, c))
    def __or__(self, o): return self._bin(o, lambda a, c: _sbin("or", a, c))
    def __xor__(self, o): return self._bin(o, lambda a, c: _sbin("xor", a, c))
    def __mod__(self, o): return self._bin(o, lambda a, c: _sbin("mod", a, c))
    def __floordiv__(self, o): return self._bin(o, lambda a, c: _sbin("floordiv", a, c))
    def __neg__(self): return Vector._new([-a for a in self._v])
    def __pos__(self): return self
    def __abs__(self): return Vector._new([_b.abs(a) for a in self._v])
    def __eq__(self, o): return self._bin(o, lambda a, c: _np.int32(a == c))
    def __ne__(self, o): return self._bin(o, lambda a, c: _np.int32(a != c))
    def __lt__(self, o): return self._bin(o, lambda a, c: _np.int32(a < c))
    def __le__(self, o): return self._bin(o, lambda a, c: _np.int32(a <= c))
    def __gt__(self, o): return self._bin(o, lambda a, c: _np.int32(a > c))
    def __ge__(self, o): return self._bin(o, lambda a, c: _np.int32(a >= c))

    def dot(self, o):
        ov = o._v if isinstance(o, Vector) else [_typed(x) for x in o]
        acc = _sbin("mul", self._v[0], ov[0])
        for a, c in zip(self._v[1:], ov[1:]):
            acc = _sbin("add", acc, _sbin("mul", a, c))
        return acc
    def norm_sqr(self): return self.dot(self)
    def norm(self): return sqrt(self.norm_sqr())
    def normalized(self, eps=0):
        inv = _sbin("truediv", 1.0, _sbin("add", self.norm(), eps)) if eps else _sbin("truediv", 1.0, self.norm())
        return self * inv
    def cross(self, o):
        a, c = self._v, (o._v if isinstance(o, Vector) else [_typed(x) for x in o])
        m, sub = (lambda p, q: _sbin("mul", p, q)), (lambda p, q: _sbin("sub", p, q))
        return Vector._new([sub(m(a[1], c[2]), m(a[2], c[1])), sub(m(a[2], c[0]), m(a[0], c[2])), sub(m(a[0], c[1]), m(a[1], c[0]))])
    def sum(self):
        acc = self._v[0]
        for a in self._v[1:]:
            acc = _sbin("add", acc, a)
        return acc
    def max(self): return _b.max(self._v)
    def min(self): return _b.min(self._v)
    def cast(self, dt): return Vector._new([_cast1(x, dt) for x in self._v])
    def any(self): return _b.any(_b.bool(x) for x in self._v)
    def all(self): return _b.all(_b.bool(x) for x in self._v)


class Matrix:
    """Row-major small matrix; Matrix(rows) with rows = Vectors (ti.math.mat3(a, b, c) stacks its arguments as rows)."""
    __hash__ = None

    def __init__(self, rows, dt=None):
        self.rows = [r._copy() if isinstance(r, Vector) else Vector(r, dt) for r in rows]

    def _copy(self): return Matrix(self.rows)
    def transpose(self):
        n, m = len(self.rows), len(self.rows[0])
        return Matrix([Vector._new([self.rows[i]._v[j] for i in range(n)]) for j in range(m)])
    def __getitem__(self, ij):
        if isinstance(ij, tuple):
            return self.rows[_b.int(ij[0])]._v[_b.int(ij[1])]
        return self.rows[_b.int(ij)]
    def __setitem__(self, ij, val):
        r = self.rows[_b.int(ij[0])]
        r._v[_b.int(ij[1])] = _cast1(val, type(r._v[_b.int(ij[1])]))
    def __matmul__(self, o):
        if isinstance(o, Vector):
            return Vector._new([r.dot(o) for r in self.rows])
        cols = o.transpose().rows
        return Matrix([Vector._new([r.dot(c) for c in cols]) for r in self.rows])
    def inverse(self):
        a = _np.array([[x.item() for x in r._v] for r in self.rows], dtype=_np.float64)
        return Matrix([Vector(list(map(float, row))) for row in _np.linalg.inv(a)])


def _cast1(x, dt):
    if dt in (_b.float,):
        dt = _np.float32
    if dt in (_b.int,):
        dt = _np.int32
    if isinstance(x, Vector):
        return x.cast(dt)
    if issubclass(dt, _np.integer) and _is_float(x):
        if x != x or x in (_m.inf, -_m.inf):
            return dt(0)
        x = _m.trunc(_b.float(x))   # toward zero
        if issubclass(dt, _np.unsignedinteger) or dt is _np.int8 or dt is _np.uint8:
            return dt(_b.int(x) & ((1 << (8 * _np.dtype(dt).itemsize)) - 1)) if issubclass(dt, _np.unsignedinteger) else _np.int64(x).astype(dt)
        return dt(_b.int(x)) if -2**31 <= x < 2**31 else dt(-2**31)
    if issubclass(dt, _np.integer) and not _is_float(x):
        return _np.int64(_b.int(x)).astype(dt) if _b.int(x) < 2**63 else dt(_b.int(x) & 0xFFFFFFFF)
    return dt(x)


def cast(x, dt):
    return _cast1(x, dt)


def _div(a, c):
    if not (_is_float(a) or _is_float(c)):   # Taichi's / on integers is true division in default_fp
        a, c = _np.float32(a), _np.float32(c)
    if isinstance(a, (_b.float, _b.int)) and isinstance(c, (_b.float, _b.int)):
        a = _np.float32(a)
    return _norm(_np.divide(a, c))


def _pow(a, c):
    if isinstance(a, (_b.int, _b.float)) and isinstance(c, (_b.int, _b.float)):
        return a ** c
    if isinstance(c, (_b.int, _b.float)) and _b.float(c).is_integer() and 0 < c <= 32 and _is_float(a):
        # a constant whole-number exponent: Taichi's algebraic simplification (fast_math, the default) turns the power into
        # repeated multiplication by squaring, least significant bit first
        n, result, sq = _b.int(c), None, _np.float32(a)
        while n:
            if n & 1:
                result = sq if result is None else _np.float32(result * sq)
            n >>= 1
            if n:
                sq = _np.float32(sq * sq)
        return result
    if _is_float(a) or _is_float(c):
        return _np.float32(_elem["pow"](_np.float32(a), _np.float32(c)))
    return _norm(_np.power(a, c))


def _map(fn, *args):
    n = next((len(a._v) for a in args if isinstance(a, Vector)), None)
    if n is None:
        return _norm(fn(*args))
    cols = [a._v if isinstance(a, Vector) else [a] * n for a in args]
    return Vector._new([_norm(fn(*vals)) for vals in zip(*cols)])


def _f(x):   # a float argument of an intrinsic: f32
    return _np.float32(x)


# elementary functions: numpy's binary32 routines unless the caller installs others (set_elementary); Taichi's own are not
# observable here either way
_elem = {"sin": _np.sin, "cos": _np.cos, "tan": _np.tan, "asin": _np.arcsin, "acos": _np.arccos, "atan2": _np.arctan2,
         "exp": _np.exp, "log": _np.log, "pow": _np.power}


def set_elementary(**fns):
    _elem.update(fns)


def sin(x): return _map(lambda v: _np.float32(_elem["sin"](_f(v))), x)
def cos(x): return _map(lambda v: _np.float32(_elem["cos"](_f(v))), x)
def tan(x): return _map(lambda v: _np.float32(_elem["tan"](_f(v))), x)
def asin(x): return _map(lambda v: _np.float32(_elem["asin"](_f(v))), x)
def acos(x): return _map(lambda v: _np.float32(_elem["acos"](_f(v))), x)
def atan2(y, x): return _map(lambda a, c: _np.float32(_elem["atan2"](_f(a), _f(c))), y, x)
def exp(x): return _map(lambda v: _np.float32(_elem["exp"](_f(v))), x)
def log(x): return _map(lambda v: _np.float32(_elem["log"](_f(v))), x)
def _py(v):
    """A Python number in Python scope: Taichi's functions fall back to Python's math there (double)."""
    return _scope[0] == 0 and isinstance(v, (_b.int, _b.float)) and not isinstance(v, _b.bool)


def sqrt(x): return _map(lambda v: _m.sqrt(v) if _py(v) else _np.sqrt(_f(v)), x)
def floor(x): return _map(lambda v: _np.floor(_f(v)), x)
def ceil(x): return _map(lambda v: _np.ceil(_f(v)), x)
def abs(x): return _map(lambda v: _b.abs(v), x)  # noqa: A001
def pow(x, y): return _map(_pow, x, y)  # noqa: A001
def round(x): return _map(lambda v: _np.float32(_m.floor(_b.float(v) + 0.5) if v >= 0 else _m.ceil(_b.float(v) - 0.5)), x)  # noqa: A001  (half away from zero, exactly: the sum in double)


def _pick(a, c, take_a):
    """The chosen operand of ti.max / ti.min, IN THE OPERANDS' COMMON TYPE: they are binary operations with Taichi's promotion
    (ti.max(x, 0) with an f32 x is an f32 zero -- a variable first assigned from it is an f32 variable)."""
    r = a if take_a else c
    if isinstance(a, (_b.int, _b.float, _np.generic)) and isinstance(c, (_b.int, _b.float, _np.generic)):
        t = _promote(_b.int(a) if isinstance(a, _b.bool) else a, _b.int(c) if isinstance(c, _b.bool) else c)
        if t is not None:
            return _conv(r, _np.float32 if t is _np.float64 else _np.int32 if t is _np.int64 else t)
    return _typed(r) if isinstance(r, (_b.int, _b.float)) else r


def _fold(fn, args):
    out = args[0]
    for a in args[1:]:
        out = _map(fn, out, a)
    return out


def _nan(x): return x != x


# ti.max / ti.min on floats: LLVM's maxnum / minnum (what Taichi's CPU and CUDA code generators emit): a NaN operand loses
def max(*args): return _fold(lambda a, c: _pick(c, a, _nan(a)) if (_nan(a) or _nan(c)) else _pick(a, c, not (a < c)), args)  # noqa: A001
def min(*args): return _fold(lambda a, c: _pick(c, a, _nan(a)) if (_nan(a) or _nan(c)) else _pick(a, c, not (a > c)), args)  # noqa: A001
def select(c, a, f):
    if isinstance(c, Vector):
        return _map(lambda cc, aa, ff: aa if cc else ff, c, a, f)
    r = a if c else f
    return r._copy() if isinstance(r, Vector) else r


# ---- random numbers: injected by the caller -----------------------------------------------------------------------------
# ti.random() has no reproducible definition; the build defines per-pixel counter streams.  The caller
# installs a function that receives the index of the struct-for iteration the draw happens in (None outside one).
_random_source = [None]
_loop_index = [None]
_touched = []       # fields written during the running struct-for (their _pre / _writer are dropped when it ends)


def set_random_source(fn_or_values):
    if callable(fn_or_values):
        _random_source[0] = fn_or_values
    else:
        it = iter(list(fn_or_values))
        _random_source[0] = lambda index: next(it)


def random(dtype=float):
    return _np.float32(_random_source[0](_loop_index[0]))


# ---- fields -------------------------------------------------------------------------------------------------------------
_oob_reads = ["error"]


def set_out_of_bounds_reads(policy, *fields):
    """"clamp": the nearest element; "zero": memory nobody wrote; "error" (default).  For the given fields, or for all."""
    if not fields:
        _oob_reads[0] = policy
    for f in fields:
        f._oob = policy


class _Axes:
    def __init__(self, names): self.names = names


i, j, k = _Axes("i"), _Axes("j"), _Axes("k")
ij, ijk = _Axes("ij"), _Axes("ijk")


class _SNode:
    def __init__(self, dims=None): self.dims = dict(dims or {})

    def dense(self, axes, sizes):
        sizes = [sizes] * len(axes.names) if isinstance(sizes, (_b.int, _np.integer)) else list(sizes)
        d = dict(self.dims)
        for a, n in zip(axes.names, sizes):
            d[a] = d.get(a, 1) * _b.int(n)
        return _SNode(d)

    def place(self, *fields, offset=None):
        shape = tuple(self.dims[a] for a in "ijk" if a in self.dims)
        for f in fields:
            f._allocate(shape, offset)


root = _SNode()


def _idx_tuple(idx):
    if idx is None:
        return ()
    if isinstance(idx, Vector):
        return tuple(_b.int(x) for x in idx._v)
    if isinstance(idx, (tuple, list)):
        out = []
        for x in idx:
            out.extend(_idx_tuple(x) if isinstance(x, (Vector, tuple, list)) else (_b.int(x),))
        return tuple(out)
    return (_b.int(idx),)


def _parallel(indices):
    """The outermost for of a kernel over a field or an ndrange is a PARALLEL loop: iterations run here one after another, but
    each sees the value an element had before the loop wherever ANOTHER iteration has written it meanwhile (its own writes it reads
    back) -- the outcome of threads that all read before any of them writes, which is also what the build decided for the
    reference's racy kernels.  Fields keep the overwritten values in _pre until the loop ends.  The index
    of the running iteration is what ti.random()'s source receives.  Loops nested inside are plain serial loops."""
    outer = _loop_index[0]
    try:
        for idx in indices:
            if outer is None:
                _loop_index[0] = idx
            yield idx
    finally:
        _loop_index[0] = outer
        if outer is None:
            for f in _touched:
                f._pre, f._writer = {}, {}
            del _touched[:]


class _FieldBase:
    """Dense field; struct-for iteration (`for u, v in f`, `for I in ti.grouped(f)`) visits every index once, row-major (the
    order is unspecified in Taichi), and publishes the index to ti.random()'s source."""
    def __init__(self, shape=None, offset=None):
        self.shape, self.offset = None, None
        if shape is not None:
            self._allocate((shape,) if isinstance(shape, (_b.int, _np.integer)) else tuple(shape), offset)

    def _allocate(self, shape, offset):
        self.shape = tuple(_b.int(s) for s in shape)
        self.offset = tuple(_b.int(o) for o in offset) if offset is not None else (0,) * len(self.shape)
        self._alloc()

    def _key(self, idx):
        k = _idx_tuple(idx)
        return tuple(a - o for a, o in zip(k, self.offset)) if any(self.offset) else k

    def _read_key(self, idx):
        """Reads outside a multi-dimensional field are undefined in release-mode Taichi; the caller picks what they see:
        the nearest element ("clamp") or an error (default)."""
        k = self._key(idx)
        if len(k) > 1 and not _b.all(0 <= a < n for a, n in zip(k, self.shape)):
            policy = getattr(self, "_oob", None) or _oob_reads[0]
            if policy == "clamp":
                return tuple(_b.min(_b.max(a, 0), n - 1) for a, n in zip(k, self.shape))
            if policy == "zero":
                return None
            raise IndexError(f"read at {k} outside a field of shape {self.shape}")
        return k

    def _indices(self):
        return _parallel(_it.product(*[range(o, o + n) for n, o in zip(self.shape, self.offset)]))

    _pre, _writer = {}, {}

    def _note_write(self, k, old):
        me = _loop_index[0]
        if me is None:
            return
        if not self._pre:
            self._pre, self._writer = {}, {}
            _touched.append(self)
        if k not in self._pre:
            self._pre[k] = old
        self._writer[k] = me

    def _foreign(self, k):
        """True when element k was written in this parallel loop by another iteration: the reader gets _pre[k]."""
        return bool(self._pre) and k in self._pre and self._writer[k] != _loop_index[0]

    def __iter__(self):
        if len(self.shape) == 1:
            for (a,) in self._indices():
                yield _np.int32(a)
        else:
            for idx in self._indices():
                yield tuple(_np.int32(a) for a in idx)


class _Field(_FieldBase):
    def __init__(self, dtype, shape=None, offset=None):
        self.dtype = _np.float32 if dtype is _b.float else _np.int32 if dtype is _b.int else dtype
        super().__init__(shape, offset)

    def _alloc(self): self.a = _np.zeros(self.shape, dtype=self.dtype)

    def __getitem__(self, idx):
        k = self._read_key(idx)
        if k is None:
            return self.dtype(0)
        if len(k) == 1 and self.a.ndim == 1 and not 0 <= k[0] < self.a.shape[0]:
            return self.dtype(0)          # past the end: release-mode Taichi does not check; unwritten memory reads as zero here
        if self._foreign(k):
            return self._pre[k]
        return self.a[k]

    def __setitem__(self, idx, val):
        k = self._key(idx)
        if len(k) == 1 and self.a.ndim == 1 and k[0] >= self.a.shape[0]:
            grown = _np.zeros(k[0] + 1 + k[0] // 4, dtype=self.dtype)
            grown[: self.a.shape[0]] = self.a
            self.a = grown
        if len(k) > 1:      # (1-D fields are only ever written through atomics here: every iteration must see those)
            self._note_write(k, self.a[k])
        self.a[k] = _cast1(val, self.dtype)

    def fill(self, v): self.a[...] = v
    def from_numpy(self, arr): self.a[...] = arr
    def to_numpy(self): return self.a.copy()


def field(dtype, shape=None, offset=None):
    return _Field(dtype, shape, offset)


class _BoundVector(Vector):
    """An element of a vector field: writes to components go through to the field (self.f[u, v].x += ...)."""
    __slots__ = ("_f", "_k")

    def _sync(self):
        self._f._note_write(self._k, self._f.a[self._k].copy())
        self._f.a[self._k] = self._v
    def __setitem__(self, i, val):
        Vector.__setitem__(self, i, val); self._sync()
    def __setattr__(self, name, value):
        if name in ("_f", "_k", "_v"):
            return object.__setattr__(self, name, value)
        Vector.__setattr__(self, name, value); self._sync()


class _VectorField(_FieldBase):
    def __init__(self, n, dtype, shape=None, offset=None):
        self.n = n
        self.dtype = _np.float32 if dtype is _b.float else _np.int32 if dtype is _b.int else dtype
        super().__init__(shape, offset)

    def _alloc(self): self.a = _np.zeros(self.shape + (self.n,), dtype=self.dtype)

    def __getitem__(self, idx):
        k = self._read_key(idx)
        if k is None:
            return Vector._new([self.dtype(0)] * self.n)
        v = object.__new__(_BoundVector)
        object.__setattr__(v, "_v", [self.dtype(x) for x in (self._pre[k] if self._foreign(k) else self.a[k])])
        object.__setattr__(v, "_f", self)
        object.__setattr__(v, "_k", k)
        return v

    def __setitem__(self, idx, val):
        vals = val._v if isinstance(val, Vector) else list(val)
        k = self._key(idx)
        self._note_write(k, self.a[k].copy())
        self.a[k] = [_cast1(x, self.dtype) for x in vals]

    def fill(self, v): self.a[...] = _np.asarray(v._v if isinstance(v, Vector) else v, dtype=self.dtype)
    def from_numpy(self, arr): self.a[...] = arr
    def to_numpy(self): return self.a.copy()


class _ObjectField(_FieldBase):
    """Matrix and struct fields: one live Python object per element, created on first touch."""
    def __init__(self, make, shape=None, offset=None):
        self._make = make
        super().__init__(shape, offset)

    def _alloc(self): self.objs = {}

    def __getitem__(self, idx):
        k = self._read_key(idx)
        if k is None:
            return self._make()
        if self._foreign(k):
            return self._pre[k]
        o = self.objs.get(k)
        if o is None:
            o = self.objs[k] = self._make()
        return o

    def __setitem__(self, idx, val):
        k = self._key(idx)
        if self.shape:      # (shape-() fields are parameters, not per-pixel data)
            self._note_write(k, self.objs.get(k) or self._make())
        self.objs[k] = val._copy()

    def fill(self, v):
        self.objs = {}
        self._make = (lambda v=v: v._copy())


Vector.field = staticmethod(lambda n, dtype, shape=None, offset=None: _VectorField(n, dtype, shape, offset))
Matrix.field = staticmethod(lambda n, m, dtype, shape=None: _ObjectField(
    lambda: Matrix([[_cast1(0, dtype)] * m for _ in range(n)]), shape))


class _NdArray:
    """ti.types.ndarray(element_dim=1) argument: `for i in data` / data[i] is a vector."""
    def __init__(self, arr): self.arr = arr
    def __iter__(self): return (_np.int32(a) for a in range(self.arr.shape[0]))
    def __getitem__(self, i): return Vector([x for x in self.arr[_b.int(i)]])


class _NdArrayType:
    def __init__(self, element_dim=0): self.element_dim = element_dim


class Format:
    rgba8, rgba32f = "rgba8", "rgba32f"


class Texture:
    """rgba8: store converts float -> unorm8 with round-to-nearest, fetch returns byte / 255 (Appendix A-11); rgba32f keeps f32."""
    def __init__(self, fmt, shape):
        self.fmt, self.shape = fmt, tuple(_b.int(s) for s in shape)
        self.a = _np.zeros(self.shape + (4,), dtype=_np.uint8 if fmt == Format.rgba8 else _np.float32)

    def store(self, idx, val):
        vals = val._v
        if self.fmt == Format.rgba8:
            vals = [_b.int(_m.floor(_b.min(_b.max(_b.float(x), 0.0), 1.0) * 255.0 + 0.5)) for x in vals]
        self.a[_idx_tuple(idx)] = vals

    def fetch(self, idx, lod=0):
        t = self.a[_idx_tuple(idx)]
        if self.fmt == Format.rgba8:
            return Vector._new([_np.float32(x) / _np.float32(255.0) for x in t])
        return Vector._new([_np.float32(x) for x in t])

    def to_numpy(self): return self.a.copy()


class _Tools:
    @staticmethod
    def imread(path):   # Appendix A-9: array index [x, y], y = 0 at the bottom row
        from PIL import Image
        img = _np.asarray(Image.open(path).convert("RGB"))
        return _np.ascontiguousarray(img.swapaxes(0, 1)[:, ::-1, :])


tools = _Tools()


def loop_config(**kw):
    return None


class _Subgroup:
    reduce_max = staticmethod(lambda x: x)
    reduce_min = staticmethod(lambda x: x)


class _Simt:
    subgroup = _Subgroup()


simt = _Simt()


def _atomic(op, container, index, value):
    old = container[index]
    new = {"or": lambda a, c: a | c, "and": lambda a, c: a & c, "add": lambda a, c: a + c, "max": lambda a, c: a if a > c else c,
           "min": lambda a, c: a if a < c else c}[op](old, _cast1(value, type(old)) if not isinstance(value, Vector) else value)
    container[index] = new
    return old


def atomic_add(x, y): return x + y   # only meaningful through the rewritten subscript form


# ---- decorators: Taichi's scoping rules by an AST pass ---------------------------------------------------------------------
def _copy(x):
    return x._copy() if hasattr(x, "_copy") else x


class _Unset:
    def __repr__(self): return "<unset>"


_UNSET = _Unset()


def _assign(old, new):
    """`x = value`: the first assignment fixes the variable's type, later ones convert to it (a float stored into an integer
    variable truncates); vectors, matrices and structs are copied."""
    if old is _UNSET or old is None:
        if isinstance(new, _b.bool):
            return new
        if isinstance(new, (_b.int, _b.float)):
            return _typed(new)
        return _copy(new)
    if isinstance(old, Vector):
        if isinstance(new, Vector) and len(new._v) == len(old._v):
            t = type(old._v[0])
            return Vector._new([x if type(x) is t else _cast1(x, t) for x in new._v])
        return _copy(new)
    if isinstance(old, _np.generic) and not isinstance(old, _np.bool_):
        if isinstance(new, (_b.bool, _np.bool_)):
            return type(old)(new)
        if isinstance(new, (_b.int, _b.float, _np.generic)):
            return new if type(new) is type(old) else _cast1(new, type(old))
    return _copy(new)


def _assign_tuple(olds, news):
    news = tuple(news)
    if len(news) != len(olds):
        raise ValueError("unpacking sizes differ")
    return tuple(_assign(o, n) for o, n in zip(olds, news))


_BYREF = {}   # function name -> positions (self not counted) of ti.template() parameters the body assigns to


class _Template:
    pass


def template():
    return _Template()


def _store_names(nodes):
    out = []
    for n in nodes:
        for sub in _ast.walk(n):
            if isinstance(sub, _ast.Name) and isinstance(sub.ctx, _ast.Store) and sub.id not in out:
                out.append(sub.id)
    return out


class _Scoping(_ast.NodeTransformer):
    def __init__(self, byref_params, remap):
        self.byref_params = byref_params
        self.remap = remap
        self.tmp = 0

    def visit_Name(self, node):
        if isinstance(node.ctx, _ast.Load) and node.id in self.remap:
            return _ast.Name(id=self.remap[node.id], ctx=_ast.Load())
        return node

    # x = value / a, b = value / x += value
    def visit_Assign(self, node):
        self.generic_visit(node)
        call = self._byref_call(node.value)
        if call is not None:
            return self._expand_byref(call, node.targets)
        if len(node.targets) == 1:
            t = node.targets[0]
            if isinstance(t, _ast.Name):
                node.value = _call("__ti_assign", [_ast.Name(id=t.id, ctx=_ast.Load()), node.value])
            elif isinstance(t, _ast.Tuple) and all(isinstance(e, _ast.Name) for e in t.elts):
                olds = _ast.Tuple(elts=[_ast.Name(id=e.id, ctx=_ast.Load()) for e in t.elts], ctx=_ast.Load())
                node.value = _call("__ti_assign_tuple", [olds, node.value])
            elif isinstance(node.value, (_ast.Name, _ast.Attribute, _ast.Subscript)):
                node.value = _call("__ti_copy", [node.value])
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.target, _ast.Name):
            load = _ast.Name(id=node.target.id, ctx=_ast.Load())
            return _ast.Assign(targets=[node.target], value=_call("__ti_assign", [load, _call("__ti_binop", [_ast.Constant(self._OPS[type(node.op)]), load, node.value])]))
        import copy as _cp
        tl = _cp.deepcopy(node.target)
        for sub in _ast.walk(tl):
            if hasattr(sub, "ctx"):
                sub.ctx = _ast.Load()
        return _ast.Assign(targets=[node.target], value=_call("__ti_binop", [_ast.Constant(self._OPS[type(node.op)]), tl, node.value]))

    def visit_Expr(self, node):
        self.generic_visit(node)
        call = self._byref_call(node.value)
        if call is not None:
            return self._expand_byref(call, None)
        return node

    def visit_Return(self, node):
        self.generic_visit(node)
        if self.byref_params:
            vals = _ast.Tuple(elts=[_ast.Name(id=p, ctx=_ast.Load()) for p in self.byref_params], ctx=_ast.Load())
            node.value = _ast.Tuple(elts=[node.value or _ast.Constant(None), vals], ctx=_ast.Load())
        return node

    _OPS = {_ast.Add: "add", _ast.Sub: "sub", _ast.Mult: "mul", _ast.Div: "truediv", _ast.FloorDiv: "floordiv", _ast.Mod: "mod",
            _ast.Pow: "pow", _ast.LShift: "lshift", _ast.RShift: "rshift", _ast.BitAnd: "and", _ast.BitOr: "or", _ast.BitXor: "xor",
            _ast.MatMult: "matmul"}

    def visit_BinOp(self, node):
        self.generic_visit(node)
        return _call("__ti_binop", [_ast.Constant(self._OPS[type(node.op)]), node.left, node.right])

    def visit_Call(self, node):
        self.generic_visit(node)
        f = node.func
        if isinstance(f, _ast.Attribute) and f.attr.startswith("atomic_") and node.args and isinstance(node.args[0], _ast.Subscript):
            sub = node.args[0]
            return _call("__ti_atomic", [_ast.Constant(f.attr[len("atomic_"):]), sub.value, sub.slice] + node.args[1:])
        return node

    def _byref_call(self, value):
        if isinstance(value, _ast.Call):
            f = value.func
            name = f.attr if isinstance(f, _ast.Attribute) else f.id if isinstance(f, _ast.Name) else None
            if name in _BYREF:
                return value
        return None

    def _expand_byref(self, call, targets):
        f = call.func
        pos = _BYREF[f.attr if isinstance(f, _ast.Attribute) else f.id]
        self.tmp += 1
        tmp = f"__ti_r{self.tmp}"
        outs = []
        for p in pos:
            if not isinstance(call.args[p], _ast.Name):
                raise SyntaxError("a ti.template() argument that the callee assigns must be a plain variable here")
            outs.append(_ast.Name(id=call.args[p].id, ctx=_ast.Store()))
        stmts = [_ast.Assign(targets=[_ast.Name(id=tmp, ctx=_ast.Store())], value=call),
                 _ast.Assign(targets=[_ast.Tuple(elts=outs, ctx=_ast.Store())],
                             value=_ast.Subscript(value=_ast.Name(id=tmp, ctx=_ast.Load()), slice=_ast.Constant(1), ctx=_ast.Load()))]
        if targets:
            ret = _ast.Subscript(value=_ast.Name(id=tmp, ctx=_ast.Load()), slice=_ast.Constant(0), ctx=_ast.Load())
            stmts.append(self.visit_Assign(_ast.Assign(targets=targets, value=ret)) if False else _ast.Assign(targets=targets, value=ret))
        return stmts


def _call(name, args):
    return _ast.Call(func=_ast.Name(id=name, ctx=_ast.Load()), args=args, keywords=[])


def _param_info(fn):
    """(parameter names, names of ti.template() parameters the body assigns to)."""
    fdef = _ast.parse(_tw.dedent(_inspect.getsource(fn))).body[0]
    names = [a.arg for a in fdef.args.args]
    tmpl = []
    for a in fdef.args.args:
        an = a.annotation
        if an is not None and isinstance(an, _ast.Call) and isinstance(an.func, _ast.Attribute) and an.func.attr == "template":
            tmpl.append(a.arg)
    stored = set(_store_names(fdef.body))
    return fdef, names, [t for t in tmpl if t in stored]


_KERNEL_BUILTINS = {"abs": abs, "max": max, "min": min, "pow": pow, "round": round, "int": lambda x=0: _cast1(x, _np.int32),
                    "float": lambda x=0.0: _cast1(x, _np.float32), "all": lambda x: x.all() if isinstance(x, Vector) else _b.all(x),
                    "any": lambda x: x.any() if isinstance(x, Vector) else _b.any(x),
                    "range": lambda *a: _krange(*a)}


def _compile(fn, fdef, byref):
    fdef.decorator_list = []
    _ast.increment_lineno(fdef, fn.__code__.co_firstlineno - 1)
    for a in fdef.args.args:
        a.annotation = None
    fdef.returns = None
    params = {a.arg for a in fdef.args.args}
    body_locals = [n for n in _store_names(fdef.body) if n not in params]
    # inside kernels abs / max / min / pow / round / int / float / all / any act on typed values and vectors the Taichi way: their
    # uses are renamed (the module's own globals stay what they are for its plain Python code), unless the module or the function
    # binds the name to something else (from taichi.math import * brings its own pow / max / min)
    g = fn.__globals__
    remap = {key: "__ti_b_" + key for key in _KERNEL_BUILTINS
             if (key not in g or g[key] is getattr(_b, key, None)) and key not in params and key not in body_locals}
    fdef = _Scoping(byref, remap).visit(fdef)
    if byref:
        vals = _ast.Tuple(elts=[_ast.Name(id=p, ctx=_ast.Load()) for p in byref], ctx=_ast.Load())
        fdef.body.append(_ast.Return(value=_ast.Tuple(elts=[_ast.Constant(None), vals], ctx=_ast.Load())))
    # every local exists from the start (Taichi variables are typed by their first assignment: _assign)
    pre = [_ast.Assign(targets=[_ast.Name(id=n, ctx=_ast.Store())], value=_ast.Name(id="__ti_unset", ctx=_ast.Load())) for n in body_locals]
    doc = []
    if fdef.body and isinstance(fdef.body[0], _ast.Expr) and isinstance(getattr(fdef.body[0], "value", None), _ast.Constant):
        doc, fdef.body = fdef.body[:1], fdef.body[1:]
    fdef.body = doc + pre + fdef.body
    tree = _ast.Module(body=[fdef], type_ignores=[])
    _ast.fix_missing_locations(tree)
    g["__ti_copy"], g["__ti_atomic"], g["__ti_assign"], g["__ti_assign_tuple"], g["__ti_unset"] = _copy, _atomic, _assign, _assign_tuple, _UNSET
    g["__ti_binop"] = _binop
    for key, v in remap.items():
        g[v] = _KERNEL_BUILTINS[key]
    ns = {}
    exec(compile(tree, _inspect.getsourcefile(fn) or "<ti.func>", "exec"), g, ns)
    return ns[fdef.name]


def _dsl(fn):
    fdef, names, byref = _param_info(fn)
    if byref:
        first = 1 if names and names[0] == "self" else 0
        _BYREF[fn.__name__] = [names.index(p) - first for p in byref]
    ann = {k: v for k, v in getattr(fn, "__annotations__", {}).items() if k != "return"}
    state = {}

    def wrapper(*args, **kwargs):
        raw = state.get("raw")
        if raw is None:   # compiled at the first call: every decorator of every module has run by then (_BYREF is complete)
            raw = state["raw"] = _compile(fn, fdef, byref)
        bound = list(args)
        for n, a in enumerate(bound):
            name = names[n] if n < len(names) else None
            t = ann.get(name)
            if isinstance(t, _Template) or name == "self":
                continue
            bound[n] = _arg(a, t)
        kw = {key: (v if isinstance(ann.get(key), _Template) else _arg(v, ann.get(key))) for key, v in kwargs.items()}
        _scope[0] += 1
        try:
            return raw(*bound, **kw)
        finally:
            _scope[0] -= 1
    wrapper.__name__ = fn.__name__
    wrapper.__wrapped__ = fn
    return wrapper


def _arg(a, t):
    if t is _b.float:
        t = _np.float32
    if t is _b.int:
        t = _np.int32
    if t is not None and isinstance(t, type) and issubclass(t, _np.generic) and not isinstance(a, Vector):
        return _cast1(a, t)
    if isinstance(t, _VecType):
        return Vector(a, t.dt) if not isinstance(a, Vector) else a.cast(t.dt)
    if isinstance(t, _NdArrayType):
        return _NdArray(a) if t.element_dim else a
    if isinstance(a, (_b.int, _b.float)) and not isinstance(a, _b.bool):
        return _typed(a)    # a Python number passed to a ti.func becomes a typed value
    return _copy(a)   # by value


def func(fn): return _dsl(fn)
def kernel(fn): return _dsl(fn)
def data_oriented(cls): return cls


class _Range:
    """The index set of a kernel-scope `for`: range(...) or ti.ndrange(...).  A loop over it is a RUNTIME loop whose indices are i32
    values (`i * 0.1` is an f32 product) -- unless it is wrapped in ti.static(...), which unrolls it over Python integers, i.e.
    compile-time constants (`i * 0.1` is folded in double and rounded once where it meets a typed value)."""
    def __init__(self, ranges): self.ranges = ranges

    def python(self):
        return self.ranges[0] if len(self.ranges) == 1 else _it.product(*self.ranges)

    def __iter__(self):
        if _scope[0] == 0:
            return iter(self.python())
        if len(self.ranges) == 1:
            return (_np.int32(a) for a in self.ranges[0])
        return (tuple(_np.int32(a) for a in idx) for idx in _it.product(*self.ranges))

    def __len__(self): return _b.int(_np.prod([len(r) for r in self.ranges]))


def _krange(*a):
    """`range` inside a kernel or func."""
    return _Range([range(*[_b.int(x) for x in a])])


def static(x, *rest):
    if rest:
        return (static(x),) + tuple(static(r) for r in rest)
    return x.python() if isinstance(x, _Range) else x


def ndrange(*dims):
    rs = []
    for d in dims:
        lo, hi = (d[0], d[1]) if isinstance(d, (tuple, list, Vector)) else (0, d)
        rs.append(range(_b.int(lo), _b.int(hi)))
    return _Range(rs)


def grouped(it):
    if isinstance(it, _FieldBase):
        for idx in it._indices():
            yield Vector(list(idx))
        return
    for idx in _parallel(it.python() if isinstance(it, _Range) else it):
        yield Vector(list(idx) if isinstance(idx, tuple) else [idx])


# ---- types ----------------------------------------------------------------------------------------------------------------
class _VecType:
    def __init__(self, n, dt):
        self.n, self.dt = n, (_np.float32 if dt is _b.float else _np.int32 if dt is _b.int else dt)

    def __call__(self, *args):
        flat = Vector(list(args))._v if args else [self.dt(0)] * self.n
        if len(flat) == 1:
            flat = flat * self.n
        if len(flat) != self.n:
            raise ValueError(f"vector({self.n}) got {len(flat)} components")
        if _scope[0] == 0:
            return Vector._new([x if isinstance(x, (_b.int, _b.float)) else _cast1(x, self.dt) for x in flat])
        return Vector._new([_cast1(x, self.dt) for x in flat])

    def zero(self): return Vector._new([self.dt(0)] * self.n)


def _zero_of(t):
    if isinstance(t, _VecType):
        return t.zero()
    if isinstance(t, type) and hasattr(t, "_ti_fields"):
        return t()
    if t is _b.float:
        return _np.float32(0)
    if t is _b.int:
        return _np.int32(0)
    return t(0)


def _make_struct(name, fields, body=None):
    ns = dict(body or {})

    def __init__(self, **kw):
        for key, t in fields.items():
            object.__setattr__(self, key, _zero_of(t))
        for key, v in kw.items():
            setattr(self, key, v)

    def __setattr__(self, key, v):
        t = fields.get(key)
        if t is None:
            raise AttributeError(key)
        if isinstance(t, _VecType):
            v = Vector(v, t.dt) if not isinstance(v, Vector) else v.cast(t.dt)
        elif isinstance(t, type) and hasattr(t, "_ti_fields"):
            v = v._copy()
        else:
            v = _cast1(v, _np.float32 if t is _b.float else _np.int32 if t is _b.int else t)
        object.__setattr__(self, key, v)

    def _copy_(self):
        c = type(self)()
        for key in fields:
            object.__setattr__(c, key, _copy(getattr(self, key)))
        return c

    ns.update(__init__=__init__, __setattr__=__setattr__, _copy=_copy_, _ti_fields=fields)
    cls = type(name, (), ns)
    cls.field = staticmethod(lambda shape=None: _ObjectField(cls, shape))
    return cls


def dataclass(cls):
    fields = dict(getattr(cls, "__annotations__", {}))
    body = {key: v for key, v in vars(cls).items() if callable(v) and not key.startswith("__")}
    return _make_struct(cls.__name__, fields, body)


class _Types:
    @staticmethod
    def vector(n, dt=float): return _VecType(n, dt)
    @staticmethod
    def struct(**fields): return _make_struct("struct", fields)
    @staticmethod
    def ndarray(element_dim=0, **kw): return _NdArrayType(element_dim)
    @staticmethod
    def texture(num_dimensions=0, **kw): return None
    @staticmethod
    def rw_texture(**kw): return None


types = _Types()

from . import math  # noqa: E402,F401
