"""Minimal host-side `taichi` look-alike: just enough of the Taichi DSL for the voxel-authoring
kernels of voxel-rt2's example scripts (`example1..10.py`, `main.py`) to run unmodified as plain
Python.  NOT the Taichi runtime: nothing here compiles or renders -- the example kernels only fill
the voxel grid through `scene.set_voxel`, and rendering goes through libvrt_hip.so.

What the examples use: @ti.kernel / @ti.func, ti.static, ti.ndrange,
ti.grouped, ti.random, ti.min/max/sin/cos/atan2/pow/round/abs/sqrt, ti.Vector, ti.math.*, and --
because Taichi rewrites the kernel AST -- the builtins int()/float()/abs()/max()/min()/round()/
pow()/any()/all() applied to vectors.  The decorators make those names resolve to vector-aware
versions inside the decorated function's module.
"""
import builtins as _b
import math as _m
import os as _os

from . import math  # noqa: F401  (taichi.math)
from .math import Vector, _map, _is_vec

# dtypes / arch tokens that scripts may mention
i8 = i16 = i32 = i64 = u8 = u16 = u32 = u64 = int
f16 = f32 = f64 = float
cpu, gpu, vulkan, cuda, opengl, metal = "cpu", "gpu", "vulkan", "cuda", "opengl", "metal"


def init(*args, **kwargs):
    return None


# ---- random: deterministic PCG stream (VRT_SEED), uniform [0,1) with 24-bit resolution like ti.random -------------
_state = [(int(_os.environ.get("VRT_SEED", "0")) * 747796405 + 2891336453) & 0xFFFFFFFF]


def seed(s):
    _state[0] = (int(s) * 747796405 + 2891336453) & 0xFFFFFFFF


def random(dtype=float):
    s = (_state[0] * 747796405 + 2891336453) & 0xFFFFFFFF
    _state[0] = s
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    w = (w >> 22) ^ w
    if dtype is int:
        return w
    return (w >> 8) * (1.0 / 16777216.0)


# ---- vector-aware scalar functions -------------------------------------------------------------------------------
def _round1(x):  # ti.round: half away from zero, returns float
    return float(_m.floor(x + 0.5)) if x >= 0 else float(_m.ceil(x - 0.5))


def sin(x): return _map(_m.sin, x)
def cos(x): return _map(_m.cos, x)
def tan(x): return _map(_m.tan, x)
def asin(x): return _map(_m.asin, x)
def acos(x): return _map(_m.acos, x)
def exp(x): return _map(_m.exp, x)
def log(x): return _map(_m.log, x)
def sqrt(x): return _map(_m.sqrt, x)
def floor(x): return _map(lambda v: float(_m.floor(v)), x)
def ceil(x): return _map(lambda v: float(_m.ceil(v)), x)
def round(x): return _map(_round1, x)  # noqa: A001
def abs(x): return _map(_b.abs, x)  # noqa: A001
def atan2(y, x): return _map(_m.atan2, y, x)
def pow(x, y): return _map(lambda a, c: a ** c, x, y)  # noqa: A001


def _fold(fn, args):
    out = args[0]
    for a in args[1:]:
        out = _map(fn, out, a)
    return out


def max(*args): return _fold(lambda a, c: a if a > c else c, args)  # noqa: A001
def min(*args): return _fold(lambda a, c: a if a < c else c, args)  # noqa: A001


def cast(x, dtype):
    return _map((lambda v: int(v)) if dtype is int else (lambda v: float(v)), x)


def select(c, a, f):
    return _map(lambda cc, aa, ff: aa if cc else ff, c, a, f)


def _int(x=0, *a):  # int() as Taichi applies it inside kernels: truncating, element-wise on vectors
    if _is_vec(x):
        return Vector([_b.int(v) for v in x])
    return _b.int(x, *a)


def _float(x=0.0):
    if _is_vec(x):
        return Vector([_b.float(v) for v in x])
    return _b.float(x)


def _any(x):
    return _b.any(bool(v) for v in x) if _is_vec(x) or isinstance(x, (list, tuple)) else bool(x)


def _all(x):
    return _b.all(bool(v) for v in x) if _is_vec(x) or isinstance(x, (list, tuple)) else bool(x)


_KERNEL_BUILTINS = {"int": _int, "float": _float, "abs": abs, "max": max, "min": min, "round": round, "pow": pow,
                    "any": _any, "all": _all}


def _dsl(fn):
    """Inside Taichi kernels the builtins above act element-wise on vectors; make the decorated
    function's module see them that way (the example scripts are DSL code, not general Python)."""
    g = getattr(fn, "__globals__", None)
    if g is not None:
        for k, v in _KERNEL_BUILTINS.items():
            if k not in g or g[k] is getattr(_b, k, None):
                g[k] = v
    return fn


def kernel(fn):
    return _dsl(fn)


def func(fn):
    return _dsl(fn)


def data_oriented(cls):
    return cls


def static(x, *rest):
    return x if not rest else (x,) + rest


def template():
    return None


class _Types:
    @staticmethod
    def vector(n, dtype=float):
        return lambda *a: Vector(list(a) if len(a) != 1 else a[0])

    ndarray = staticmethod(lambda *a, **k: None)


types = _Types()


def ndrange(*dims):
    """Cartesian product of ranges; each dim is n or (lo, hi).  One dim yields ints, more yield tuples."""
    rs = []
    for d in dims:
        if isinstance(d, (tuple, list)) or _is_vec(d):
            lo, hi = d[0], d[1]
        else:
            lo, hi = 0, d
        rs.append(range(_b.int(lo), _b.int(hi)))
    if len(rs) == 1:
        return rs[0]
    import itertools
    return itertools.product(*rs)


def grouped(it):
    for idx in it:
        yield Vector(list(idx) if isinstance(idx, tuple) else [idx])
