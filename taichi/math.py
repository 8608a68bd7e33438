"""`taichi.math` look-alike for the example scripts: vec2/3/4, ivec2/3/4, swizzles, GLSL-style helpers -- and the scalar
arithmetic of the DSL shim (taichi/__init__.py).

Two scopes, as in Taichi.  In PYTHON scope (module level of a script) vectors hold Python numbers and compute in double.
In KERNEL scope (inside @ti.kernel / @ti.func, `_scope[0] > 0`) values are Taichi's default types: f32 and i32.  They are
carried as Python `float`s that are always exactly representable in binary32 and Python `int`s: Python's own dispatch then IS
Taichi's promotion (int op int -> int, anything with a float -> float, true division -> float), and every float result is
rounded to binary32 once (`_f32`) -- for + - * / and sqrt a double result rounded once equals the correctly rounded f32 result
(53 >= 2 * 24 + 2), so kernel arithmetic is bit-for-bit binary32 (reference example6.py:39, 54: `if ti.random() < prob`).
The elementary functions (sin, cos, atan2, exp, log, pow with a fractional exponent) are libm's double routines rounded once:
Taichi's own are not observable here, and this choice does not depend on the host's vector units.
"""
import builtins as _b
import math as _m
import struct as _struct

pi = _m.pi
e = _m.e
inf = float("inf")
nan = float("nan")

_AXES = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}
_scope = [0]          # > 0 while a @ti.kernel runs

_F = _struct.Struct("f")
_pk, _up = _F.pack, _F.unpack


def _f32(x):
    """x rounded to binary32, as a Python float."""
    try:
        return _up(_pk(x))[0]
    except OverflowError:
        return inf if x > 0 else -inf


def _i2f(n):
    """i32 -> f32 (exact up to 2^24, rounded beyond)."""
    return _b.float(n) if -16777216 <= n <= 16777216 else _f32(n)


def _wrap(i):
    return i if -2147483648 <= i <= 2147483647 else ((i + 2147483648) & 0xFFFFFFFF) - 2147483648


def _trunc(x):
    """f32 -> i32 as ti.cast does it: toward zero."""
    if x != x or x in (inf, -inf):
        return 0
    return _wrap(_b.int(x))


def _is_vec(x):
    return x.__class__ is Vector


# ---- one scalar operation in kernel scope -----------------------------------------------------------------------------
def _div0(a, b):
    if a != a or a == 0:
        return nan
    return inf if (a > 0) == (_m.copysign(1.0, b) > 0) else -inf


def _kadd(a, b):
    ta, tb = a.__class__, b.__class__
    if ta is float:
        if tb is float:
            try: return _up(_pk(a + b))[0]
            except OverflowError: return _f32(a + b)
        if tb is int or tb is bool: return _f32(a + _i2f(b))
    elif ta is int or ta is bool:
        if tb is float: return _f32(_i2f(a) + b)
        if tb is int or tb is bool: return _wrap(a + b)
    return a + b        # a vector on either side: its own methods apply this function element-wise


def _ksub(a, b):
    ta, tb = a.__class__, b.__class__
    if ta is float:
        if tb is float:
            try: return _up(_pk(a - b))[0]
            except OverflowError: return _f32(a - b)
        if tb is int or tb is bool: return _f32(a - _i2f(b))
    elif ta is int or ta is bool:
        if tb is float: return _f32(_i2f(a) - b)
        if tb is int or tb is bool: return _wrap(a - b)
    return a - b


def _kmul(a, b):
    ta, tb = a.__class__, b.__class__
    if ta is float:
        if tb is float:
            try: return _up(_pk(a * b))[0]
            except OverflowError: return _f32(a * b)
        if tb is int or tb is bool: return _f32(a * _i2f(b))
    elif ta is int or ta is bool:
        if tb is float: return _f32(_i2f(a) * b)
        if tb is int or tb is bool: return _wrap(a * b)
    return a * b


def _kdiv(a, b):
    """`/`: true division in default_fp, whatever the operands."""
    ta, tb = a.__class__, b.__class__
    if (ta is float or ta is int or ta is bool) and (tb is float or tb is int or tb is bool):
        if ta is not float: a = _b.float(a) if -16777216 <= a <= 16777216 else _f32(a)      # i32 -> f32
        if tb is not float: b = _b.float(b) if -16777216 <= b <= 16777216 else _f32(b)
        try:
            return _up(_pk(a / b))[0]
        except ZeroDivisionError:
            return _div0(a, b)
        except OverflowError:
            return _f32(a / b)
    return a / b


def _kfloordiv(a, b):
    ta, tb = a.__class__, b.__class__
    if (ta is int or ta is bool) and (tb is int or tb is bool):
        return _wrap(a // b)
    if (ta is float or ta is int or ta is bool) and (tb is float or tb is int or tb is bool):
        q = _kdiv(a, b)
        return _b.float(_m.floor(q)) if q == q and q not in (inf, -inf) else q
    return a // b


def _kmod(a, b):
    """`%`: Python's on integers; on floats a - b * floor(a / b), every step in f32 (Taichi builds it from those operations)."""
    ta, tb = a.__class__, b.__class__
    if (ta is int or ta is bool) and (tb is int or tb is bool):
        return _wrap(a % b)
    if (ta is float or ta is int or ta is bool) and (tb is float or tb is int or tb is bool):
        return _ksub(a, _kmul(b, _kfloordiv(a, b)))
    return a % b


def _kpow(a, b):
    ta, tb = a.__class__, b.__class__
    if ta is bool: a, ta = _b.int(a), int
    if tb is bool: b, tb = _b.int(b), int
    if tb is int and ta is int:
        return _wrap(a ** b) if b >= 0 else _kpow(_i2f(a), b)
    if tb is int and ta is float and 0 < b <= 32:
        # a whole-number exponent: repeated multiplication by squaring, least significant bit first (Taichi's algebraic
        # simplification of pow with a constant exponent); x ** 2 is x * x
        n, result, sq = b, None, a
        while n:
            if n & 1:
                result = sq if result is None else _f32(result * sq)
            n >>= 1
            if n:
                sq = _f32(sq * sq)
        return result
    if (ta is float or ta is int) and (tb is float or tb is int):
        x, y = (_i2f(a) if ta is int else a), (_i2f(b) if tb is int else b)
        try:
            return _f32(_m.pow(x, y))
        except OverflowError:
            return inf
        except (ValueError, ZeroDivisionError):
            return nan if x < 0 and x == x else inf
    return a ** b


def _bitop(fn):
    def op(a, b):
        ta, tb = a.__class__, b.__class__
        if (ta is int or ta is bool) and (tb is int or tb is bool):
            return _wrap(fn(_b.int(a), _b.int(b)))
        return fn(a, b)
    return op


_kand, _kor, _kxor = _bitop(lambda a, b: a & b), _bitop(lambda a, b: a | b), _bitop(lambda a, b: a ^ b)
_klshift, _krshift = _bitop(lambda a, b: a << b), _bitop(lambda a, b: a >> b)


def _unify(a, b):
    """The operands of min / max / select take a common type: an integer beside a float becomes a float."""
    ta, tb = a.__class__, b.__class__
    if ta is float and (tb is int or tb is bool):
        return a, _i2f(b)
    if tb is float and (ta is int or ta is bool):
        return _i2f(a), b
    return a, b


def _map(fn, *args):
    """Apply fn element-wise, broadcasting scalars against vectors."""
    n = None
    for a in args:
        if a.__class__ is Vector:
            n = len(a._v)
            break
    if n is None:
        return fn(*args)
    cols = [a._v if a.__class__ is Vector else [a] * n for a in args]
    return Vector._new([fn(*vals) for vals in zip(*cols)])


def _num(x):
    """A number entering a vector through the general constructor: f32 / i32 in kernel scope, as it is in Python scope."""
    if _scope[0] and x.__class__ is float:
        return _f32(x)
    return x


class Vector:
    """Value-semantics small vector with element-wise arithmetic, comparisons and swizzles."""
    __slots__ = ("_v",)
    __hash__ = None

    def __init__(self, vals, dt=None):
        if vals.__class__ is Vector:
            vals = vals._v
        v = []
        for x in vals:
            if x.__class__ is Vector:
                v.extend(x._v)
            elif isinstance(x, (list, tuple)):
                v.extend(x)
            else:
                v.append(x)
        if dt is int:
            v = [_trunc(x) if x.__class__ is float else _b.int(x) for x in v]
        elif dt is float:
            v = [_num(_b.float(x)) if x.__class__ is not float else _num(x) for x in v]
        elif _b.any(x.__class__ is float for x in v):      # ti.Vector([i, 0.5]): one element type, the widest
            v = [_num(_b.float(x)) for x in v]
        else:
            v = [_b.int(x) if x.__class__ is bool else x for x in v]
        object.__setattr__(self, "_v", v)

    # container protocol
    def __len__(self): return len(self._v)
    def __iter__(self): return iter(self._v)
    def __getitem__(self, i): return self._v[i]
    def __repr__(self): return f"Vector({self._v})"
    def to_list(self): return list(self._v)
    def _copy(self): return Vector._new(list(self._v))

    def __setitem__(self, i, val):
        self._v[i] = _like(self._v[i], val) if _scope[0] else val

    # swizzles
    def __getattr__(self, name):
        try:
            idx = [_AXES[c] for c in name]
        except KeyError:
            raise AttributeError(name) from None
        if len(idx) == 1:
            return self._v[idx[0]]
        return Vector._new([self._v[i] for i in idx])

    def __setattr__(self, name, value):
        idx = [_AXES[c] for c in name]
        if len(idx) == 1:
            self[idx[0]] = value
        else:
            for i, val in zip(idx, value):
                self[i] = val

    @staticmethod
    def _new(vals):
        v = object.__new__(Vector)
        object.__setattr__(v, "_v", vals)
        return v

    # arithmetic: Python's in Python scope, f32 / i32 in kernel scope (methods are attached below: _vec_op)
    # methods the examples call
    def dot(self, o):
        if not _scope[0]:
            return _b.sum(a * c for a, c in zip(self._v, o))
        a = self._v
        if o.__class__ is Vector and len(a) == 3:
            c = o._v
            a0, a1, a2 = a
            c0, c1, c2 = c
            if (a0.__class__ is float and a1.__class__ is float and a2.__class__ is float and
                    c0.__class__ is float and c1.__class__ is float and c2.__class__ is float):
                try:      # ((a0 c0 + a1 c1) + a2 c2), every product and sum rounded to binary32
                    return _up(_pk(_up(_pk(_up(_pk(a0 * c0))[0] + _up(_pk(a1 * c1))[0]))[0] + _up(_pk(a2 * c2))[0]))[0]
                except OverflowError:
                    pass
        acc = None
        for a, c in zip(self._v, o):      # left to right
            p = _kmul(a, c)
            acc = p if acc is None else _kadd(acc, p)
        return acc if acc is not None else 0

    def sum(self):
        if not _scope[0]:
            return _b.sum(self._v)
        acc = None
        for a in self._v:
            acc = a if acc is None else _kadd(acc, a)
        return acc

    def norm_sqr(self): return self.dot(self)
    def norm(self): return sqrt(self.norm_sqr())

    def normalized(self, eps=0):
        if not _scope[0]:
            n = self.norm()
            return Vector._new([a / (n + eps) for a in self._v])
        return self * _kdiv(1.0, _kadd(self.norm(), eps))      # multiplied by the reciprocal, as Taichi's does

    def cross(self, o):
        a, c = self._v, list(o)
        if not _scope[0]:
            return Vector._new([a[1] * c[2] - a[2] * c[1], a[2] * c[0] - a[0] * c[2], a[0] * c[1] - a[1] * c[0]])
        return Vector._new([_ksub(_kmul(a[1], c[2]), _kmul(a[2], c[1])), _ksub(_kmul(a[2], c[0]), _kmul(a[0], c[2])),
                            _ksub(_kmul(a[0], c[1]), _kmul(a[1], c[0]))])

    def cast(self, dt): return Vector(self._v, int if dt is int else float)
    def max(self): return _b.max(self._v)
    def min(self): return _b.min(self._v)
    def any(self): return _b.any(_b.bool(x) for x in self._v)
    def all(self): return _b.all(_b.bool(x) for x in self._v)


def _vec_op(kfn, pfn, rev=False):
    """A binary operator method of Vector: `kfn` element-wise in kernel scope, `pfn` in Python scope; scalars broadcast."""
    new = Vector._new
    if rev:
        def method(self, o):
            fn = kfn if _scope[0] else pfn
            a = self._v
            if o.__class__ is Vector:
                c = o._v
            elif o.__class__ is list or o.__class__ is tuple:
                c = o
            else:
                return new([fn(o, x) for x in a])
            if len(c) != len(a):
                raise ValueError(f"vector sizes differ: {len(a)} and {len(c)}")
            return new([fn(y, x) for x, y in zip(a, c)])
    else:
        def method(self, o):
            fn = kfn if _scope[0] else pfn
            a = self._v
            if o.__class__ is Vector:
                c = o._v
            elif o.__class__ is list or o.__class__ is tuple:
                c = o
            else:
                return new([fn(x, o) for x in a])
            if len(c) != len(a):
                raise ValueError(f"vector sizes differ: {len(a)} and {len(c)}")
            return new([fn(x, y) for x, y in zip(a, c)])
    return method


def _attach():
    import operator as _o
    V = Vector
    for name, kfn, pfn in (("add", _kadd, _o.add), ("sub", _ksub, _o.sub), ("mul", _kmul, _o.mul), ("truediv", _kdiv, _o.truediv),
                           ("floordiv", _kfloordiv, _o.floordiv), ("mod", _kmod, _o.mod), ("pow", _kpow, _o.pow)):
        setattr(V, f"__{name}__", _vec_op(kfn, pfn))
        setattr(V, f"__r{name}__", _vec_op(kfn, pfn, True))
    for name, kfn, pfn in (("and", _kand, lambda a, c: _b.int(a) & _b.int(c)), ("or", _kor, lambda a, c: _b.int(a) | _b.int(c)),
                           ("xor", _kxor, lambda a, c: _b.int(a) ^ _b.int(c))):
        setattr(V, f"__{name}__", _vec_op(kfn, pfn))
        setattr(V, f"__r{name}__", _vec_op(kfn, pfn))
    V.__lshift__, V.__rshift__ = _vec_op(_klshift, _o.lshift), _vec_op(_krshift, _o.rshift)
    # comparisons give 0/1 vectors like Taichi
    for name, fn in (("eq", lambda a, c: _b.int(a == c)), ("ne", lambda a, c: _b.int(a != c)), ("lt", lambda a, c: _b.int(a < c)),
                     ("le", lambda a, c: _b.int(a <= c)), ("gt", lambda a, c: _b.int(a > c)), ("ge", lambda a, c: _b.int(a >= c))):
        setattr(V, f"__{name}__", _vec_op(fn, fn))
    V.__neg__ = lambda self: V._new([-a for a in self._v])
    V.__pos__ = lambda self: self
    V.__abs__ = lambda self: V._new([_b.abs(a) for a in self._v])


_attach()


def _like(old, new):
    """`new` converted to the type `old` has (a typed variable or vector element being stored into)."""
    co, cn = old.__class__, new.__class__
    if co is float:
        if cn is float: return new
        if cn is int or cn is bool: return _i2f(new)
    elif co is int or co is bool:
        if cn is float: return _trunc(new)
        if cn is bool and co is int: return _b.int(new)
    return new


_SCALARS = (float, int, bool)


def _ctor(n, dt):
    new = Vector._new

    def make(*args):
        if n == 3 and len(args) == 3 and dt is float and _scope[0]:
            a, c, d = args
            if a.__class__ is float and c.__class__ is float and d.__class__ is float:
                return new([a, c, d])
        if len(args) == n:
            v = []
            for a in args:
                c = a.__class__
                if c is float:
                    v.append(a if dt is float else _trunc(a))
                elif c is int or c is bool:
                    v.append(_i2f(a) if dt is float else _b.int(a))
                else:
                    v = None
                    break
            if v is not None:
                if dt is float and not _scope[0]:
                    return new([_b.float(a) for a in args])      # Python scope: doubles
                return new(v)
        flat = []
        for a in args:
            if isinstance(a, (Vector, list, tuple)):
                flat.extend(a)
            else:
                flat.append(a)
        if len(flat) == 1:
            flat = flat * n
        if len(flat) != n:
            raise ValueError(f"vec{n} needs 1 or {n} components, got {len(flat)}")
        return Vector(flat, dt)
    return make


vec2, vec3, vec4 = _ctor(2, float), _ctor(3, float), _ctor(4, float)
ivec2, ivec3, ivec4 = _ctor(2, int), _ctor(3, int), _ctor(4, int)
uvec2, uvec3, uvec4 = ivec2, ivec3, ivec4


# ---- functions: double in Python scope, f32 in kernel scope ---------------------------------------------------------------
def _fn1(f):
    """A real function of one argument: libm's in double; in kernel scope the argument is an f32 and so is the result."""
    def one(v):
        if not _scope[0]:
            return f(v)
        try:
            return _f32(f(v if v.__class__ is float else _i2f(v)))
        except (ValueError, OverflowError):
            return nan
    return lambda x: _map(one, x)


sin, cos, tan = _fn1(_m.sin), _fn1(_m.cos), _fn1(_m.tan)
asin, acos, atan = _fn1(_m.asin), _fn1(_m.acos), _fn1(_m.atan)
exp, log, sqrt = _fn1(_m.exp), _fn1(_m.log), _fn1(_m.sqrt)
tanh = _fn1(_m.tanh)
floor = _fn1(lambda v: _b.float(_m.floor(v)))
ceil = _fn1(lambda v: _b.float(_m.ceil(v)))
radians = _fn1(_m.radians)
degrees = _fn1(_m.degrees)


def atan2(y, x):
    def one(a, c):
        if not _scope[0]:
            return _m.atan2(a, c)
        return _f32(_m.atan2(a if a.__class__ is float else _i2f(a), c if c.__class__ is float else _i2f(c)))
    return _map(one, y, x)


def _op(kfn, pfn):
    return lambda *a: (kfn if _scope[0] else pfn)(*a)


_add, _sub, _mul = _op(_kadd, lambda a, c: a + c), _op(_ksub, lambda a, c: a - c), _op(_kmul, lambda a, c: a * c)
_div = _op(_kdiv, lambda a, c: a / c)


def _max2(a, c):
    a, c = _unify(a, c)
    return c if (a != a) else a if (c != c) else (a if not (a < c) else c)     # maxnum: a NaN is ignored


def _min2(a, c):
    a, c = _unify(a, c)
    return c if (a != a) else a if (c != c) else (a if not (a > c) else c)


def _fold(fn, args):
    out = args[0]
    for a in args[1:]:
        out = _map(fn, out, a)
    return out


def max(*args): return _fold(_max2, args)  # noqa: A001
def min(*args): return _fold(_min2, args)  # noqa: A001
def abs(x): return _map(_b.abs, x)  # noqa: A001
def pow(x, y): return _map(_op(_kpow, lambda a, c: a ** c), x, y)  # noqa: A001


def _round1(x):  # ti.round: half away from zero, returns a float (the sum is exact in double)
    x = _b.float(x)
    return _b.float(_m.floor(x + 0.5)) if x >= 0 else _b.float(_m.ceil(x - 0.5))


def round(x): return _map(_round1, x)  # noqa: A001
def mix(x, y, a): return _add(_mul(x, _sub(1, a)), _mul(y, a))          # x * (1 - a) + y * a
def fract(x): return _sub(x, floor(x))
def clamp(x, lo, hi): return min(max(x, lo), hi)
def step(edge, x): return _map(lambda ed, v: 0.0 if v < ed else 1.0, edge, x)
def sign(x): return _map(lambda v: _like(v, (v > 0) - (v < 0)) if _scope[0] else (v > 0) - (v < 0), x)


def smoothstep(e0, e1, x):
    t = clamp(_div(_sub(x, e0), _sub(e1, e0)), 0.0, 1.0)
    return _mul(_mul(t, t), _sub(3.0, _mul(2.0, t)))


def dot(a, c): return (a if a.__class__ is Vector else Vector(a)).dot(c)
def cross(a, c): return Vector(a).cross(c)
def length(a): return (a if a.__class__ is Vector else Vector(a)).norm()
def distance(a, c): return length((a if a.__class__ is Vector else Vector(a)) - c)
def normalize(a): return Vector(a).normalized()
def mod(x, y): return _sub(x, _mul(y, floor(_div(x, y))))
def isnan(x): return _map(lambda v: _b.int(v != v), x)
def isinf(x): return _map(lambda v: _b.int(v in (inf, -inf)), x)


__all__ = ["pi", "e", "inf", "nan", "vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4", "mix", "fract",
           "clamp", "step", "sign", "smoothstep", "dot", "cross", "length", "distance", "normalize", "mod", "radians",
           "degrees", "sin", "cos", "tan", "tanh", "atan2", "acos", "asin", "sqrt", "exp", "log", "floor", "ceil", "isnan", "isinf"]
