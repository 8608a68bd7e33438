"""`taichi.math` look-alike for the example scripts: vec2/3/4, ivec2/3/4, swizzles, GLSL-style helpers."""
import builtins as _b
import math as _m

pi = _m.pi
e = _m.e
inf = float("inf")

_AXES = {"x": 0, "y": 1, "z": 2, "w": 3, "r": 0, "g": 1, "b": 2, "a": 3}


def _is_vec(x):
    return isinstance(x, Vector)


def _map(fn, *args):
    """Apply fn element-wise, broadcasting scalars against vectors."""
    n = None
    for a in args:
        if isinstance(a, Vector):
            n = len(a._v)
            break
    if n is None:
        return fn(*args)
    cols = [a._v if isinstance(a, Vector) else [a] * n for a in args]
    return Vector([fn(*vals) for vals in zip(*cols)])


class Vector:
    """Value-semantics small vector with element-wise arithmetic, comparisons and swizzles."""
    __slots__ = ("_v",)
    __hash__ = None

    def __init__(self, vals, dt=None):
        if isinstance(vals, Vector):
            vals = vals._v
        v = []
        for x in vals:
            if isinstance(x, Vector):
                v.extend(x._v)
            elif isinstance(x, (list, tuple)):
                v.extend(x)
            else:
                v.append(x)
        if dt is int:
            v = [_b.int(x) for x in v]
        elif dt is float:
            v = [_b.float(x) for x in v]
        object.__setattr__(self, "_v", v)

    # container protocol
    def __len__(self): return len(self._v)
    def __iter__(self): return iter(self._v)
    def __getitem__(self, i): return self._v[i]
    def __setitem__(self, i, val): self._v[i] = val
    def __repr__(self): return f"Vector({self._v})"
    def to_list(self): return list(self._v)

    # swizzles
    def __getattr__(self, name):
        try:
            idx = [_AXES[c] for c in name]
        except KeyError:
            raise AttributeError(name) from None
        if len(idx) == 1:
            return self._v[idx[0]]
        return Vector([self._v[i] for i in idx])

    def __setattr__(self, name, value):
        idx = [_AXES[c] for c in name]
        if len(idx) == 1:
            self._v[idx[0]] = value
        else:
            for i, val in zip(idx, value):
                self._v[i] = val

    # arithmetic (fast paths for vector (op) vector and vector (op) scalar)
    @staticmethod
    def _new(vals):
        v = object.__new__(Vector)
        object.__setattr__(v, "_v", vals)
        return v

    def __add__(self, o):
        if type(o) is Vector: return Vector._new([a + c for a, c in zip(self._v, o._v)])
        if isinstance(o, (list, tuple)): return Vector._new([a + c for a, c in zip(self._v, o)])
        return Vector._new([a + o for a in self._v])
    __radd__ = __add__
    def __sub__(self, o):
        if type(o) is Vector: return Vector._new([a - c for a, c in zip(self._v, o._v)])
        if isinstance(o, (list, tuple)): return Vector._new([a - c for a, c in zip(self._v, o)])
        return Vector._new([a - o for a in self._v])
    def __rsub__(self, o):
        if isinstance(o, (list, tuple)): return Vector._new([c - a for a, c in zip(self._v, o)])
        return Vector._new([o - a for a in self._v])
    def __mul__(self, o):
        if type(o) is Vector: return Vector._new([a * c for a, c in zip(self._v, o._v)])
        if isinstance(o, (list, tuple)): return Vector._new([a * c for a, c in zip(self._v, o)])
        return Vector._new([a * o for a in self._v])
    __rmul__ = __mul__
    def __truediv__(self, o):
        if type(o) is Vector: return Vector._new([a / c for a, c in zip(self._v, o._v)])
        if isinstance(o, (list, tuple)): return Vector._new([a / c for a, c in zip(self._v, o)])
        return Vector._new([a / o for a in self._v])
    def __rtruediv__(self, o): return _map(lambda a, c: c / a, self, o)
    def __floordiv__(self, o): return _map(lambda a, c: a // c, self, o)
    def __rfloordiv__(self, o): return _map(lambda a, c: c // a, self, o)
    def __mod__(self, o): return _map(lambda a, c: a % c, self, o)
    def __rmod__(self, o): return _map(lambda a, c: c % a, self, o)
    def __pow__(self, o): return _map(lambda a, c: a ** c, self, o)
    def __neg__(self): return Vector([-a for a in self._v])
    def __pos__(self): return self
    def __abs__(self): return Vector([_b.abs(a) for a in self._v])
    def __and__(self, o): return _map(lambda a, c: _b.int(a) & _b.int(c), self, o)
    def __or__(self, o): return _map(lambda a, c: _b.int(a) | _b.int(c), self, o)
    def __xor__(self, o): return _map(lambda a, c: _b.int(a) ^ _b.int(c), self, o)
    # comparisons give 0/1 vectors like Taichi
    def __eq__(self, o): return _map(lambda a, c: _b.int(a == c), self, o)
    def __ne__(self, o): return _map(lambda a, c: _b.int(a != c), self, o)
    def __lt__(self, o): return _map(lambda a, c: _b.int(a < c), self, o)
    def __le__(self, o): return _map(lambda a, c: _b.int(a <= c), self, o)
    def __gt__(self, o): return _map(lambda a, c: _b.int(a > c), self, o)
    def __ge__(self, o): return _map(lambda a, c: _b.int(a >= c), self, o)

    # methods the examples call
    def dot(self, o): return _b.sum(a * c for a, c in zip(self._v, o))
    def sum(self): return _b.sum(self._v)
    def norm(self): return _m.sqrt(_b.sum(a * a for a in self._v))
    def norm_sqr(self): return _b.sum(a * a for a in self._v)
    def normalized(self):
        n = self.norm()
        return Vector([a / n for a in self._v])
    def cross(self, o):
        a, c = self._v, list(o)
        return Vector([a[1] * c[2] - a[2] * c[1], a[2] * c[0] - a[0] * c[2], a[0] * c[1] - a[1] * c[0]])
    def cast(self, dt): return Vector(self._v, int if dt is int else float)
    def max(self): return _b.max(self._v)
    def min(self): return _b.min(self._v)


def _ctor(n, dt):
    def make(*args):
        if len(args) == n and not any(isinstance(a, (Vector, list, tuple)) for a in args):
            return Vector._new([dt(a) for a in args])
        flat = Vector(list(args))._v
        if len(flat) == 1:
            flat = flat * n
        if len(flat) != n:
            raise ValueError(f"vec{n} needs 1 or {n} components, got {len(flat)}")
        return Vector(flat, dt)
    return make


vec2, vec3, vec4 = _ctor(2, float), _ctor(3, float), _ctor(4, float)
ivec2, ivec3, ivec4 = _ctor(2, int), _ctor(3, int), _ctor(4, int)
uvec2, uvec3, uvec4 = ivec2, ivec3, ivec4


def mix(x, y, a): return _map(lambda p, q, t: p * (1 - t) + q * t, x, y, a)
def fract(x): return _map(lambda v: v - _m.floor(v), x)
def clamp(x, lo, hi): return _map(lambda v, a, c: a if v < a else (c if v > c else v), x, lo, hi)
def step(edge, x): return _map(lambda ed, v: 0.0 if v < ed else 1.0, edge, x)
def sign(x): return _map(lambda v: (v > 0) - (v < 0), x)
def smoothstep(e0, e1, x):
    t = clamp((x - e0) / (e1 - e0), 0.0, 1.0)
    return t * t * (3.0 - 2.0 * t)
def dot(a, c): return _b.sum(p * q for p, q in zip(a, c))
def cross(a, c): return Vector(a).cross(c)
def length(a): return _m.sqrt(_b.sum(p * p for p in a))
def distance(a, c): return _m.sqrt(_b.sum((p - q) ** 2 for p, q in zip(a, c)))
def normalize(a): return Vector(a).normalized()
def mod(x, y): return _map(lambda p, q: p - q * _m.floor(p / q), x, y)
def radians(x): return _map(_m.radians, x)
def degrees(x): return _map(_m.degrees, x)
def sin(x): return _map(_m.sin, x)
def cos(x): return _map(_m.cos, x)
def tan(x): return _map(_m.tan, x)
def atan2(y, x): return _map(_m.atan2, y, x)
def acos(x): return _map(_m.acos, x)
def asin(x): return _map(_m.asin, x)
def sqrt(x): return _map(_m.sqrt, x)
def exp(x): return _map(_m.exp, x)
def log(x): return _map(_m.log, x)
def floor(x): return _map(lambda v: float(_m.floor(v)), x)
def ceil(x): return _map(lambda v: float(_m.ceil(v)), x)


__all__ = ["pi", "e", "inf", "vec2", "vec3", "vec4", "ivec2", "ivec3", "ivec4", "uvec2", "uvec3", "uvec4", "mix", "fract",
           "clamp", "step", "sign", "smoothstep", "dot", "cross", "length", "distance", "normalize", "mod", "radians",
           "degrees", "sin", "cos", "tan", "atan2", "acos", "asin", "sqrt", "exp", "log", "floor", "ceil"]
