"""`taichi.math` of the reference-execution emulation (tests/refexec/taichi/__init__.py)."""
import builtins as _b
import math as _m

import numpy as _np

from . import (Vector, Matrix, _VecType, _map, _norm, _sbin, _binop, _pow, _typed, sin, cos, tan, asin, acos, atan2, exp, log, sqrt, floor, ceil,  # noqa: F401
               pow, max, min)

pi = _m.pi
e = _m.e
inf = float("inf")
vec2, vec3, vec4 = _VecType(2, _np.float32), _VecType(3, _np.float32), _VecType(4, _np.float32)
ivec2, ivec3, ivec4 = _VecType(2, _np.int32), _VecType(3, _np.int32), _VecType(4, _np.int32)
uvec2, uvec3, uvec4 = _VecType(2, _np.uint32), _VecType(3, _np.uint32), _VecType(4, _np.uint32)


def mat3(*rows): return Matrix(rows)
def mat4(*rows): return Matrix(rows)
def mix(x, y, a): return _map(lambda p, q, t: _sbin("add", _sbin("mul", p, _sbin("sub", 1.0, t)), _sbin("mul", q, t)), x, y, a)
def clamp(x, lo, hi): return _map(lambda v, a, c: min(max(v, a), c), x, lo, hi)
def fract(x): return _map(lambda v: _sbin("sub", v, _np.floor(_np.float32(v))), x)
def step(edge, x): return _map(lambda ed, v: _np.float32(0.0 if v < ed else 1.0), edge, x)
def sign(x): return _map(lambda v: _np.float32(_b.int(v > 0) - _b.int(v < 0)), x)
def smoothstep(e0, e1, x):
    B = _binop
    t = clamp(B("truediv", B("sub", x, e0), B("sub", e1, e0)), 0.0, 1.0)
    return B("mul", B("mul", t, t), B("sub", 3.0, B("mul", 2.0, t)))
def dot(a, c): return a.dot(c)
def cross(a, c): return a.cross(c)
def length(a): return a.norm()
def distance(a, c): return (a - c).norm()
def normalize(a): return a.normalized()
def reflect(i, n): return i - _sbin("mul", 2.0, n.dot(i)) * n
def refract(i, n, eta):
    B = _sbin
    k = B("sub", 1.0, B("mul", B("mul", eta, eta), B("sub", 1.0, B("mul", n.dot(i), n.dot(i)))))
    return vec3(0.0) if k < 0.0 else eta * i - B("add", B("mul", eta, n.dot(i)), sqrt(k)) * n
def mod(x, y): return _map(lambda p, q: _sbin("sub", p, _sbin("mul", q, _np.floor(_np.float32(_sbin("truediv", p, q))))), x, y)
def isnan(x): return _map(lambda v: _np.int32(v != v), x)
def isinf(x): return _map(lambda v: _np.int32(v in (inf, -inf)), x)
def inverse(m): return m.inverse()
