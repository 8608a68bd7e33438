"""Minimal host-side `taichi` look-alike: just enough of the Taichi DSL for the voxel-authoring
kernels of voxel-rt2's example scripts (`example1..10.py`, `main.py`) to run unmodified.
NOT the Taichi runtime: nothing here renders -- the example kernels only fill
the voxel grid through `scene.set_voxel`, and rendering goes through libvrt_hip.so.

What the examples use: @ti.kernel / @ti.func, ti.static, ti.ndrange,
ti.grouped, ti.random, ti.min/max/sin/cos/atan2/pow/round/abs/sqrt, ti.Vector, ti.math.*, and --
because Taichi rewrites the kernel AST -- the builtins int()/float()/abs()/max()/min()/round()/
pow()/any()/all() applied to vectors.

Kernel bodies compute in binary32 / int32 like Taichi's (default_fp = f32, default_ip = i32): the decorators re-compile the
function (taichi/_kernel.py) so that every operation rounds to f32, variables keep the type of their first assignment,
int() truncates and ti.round rounds half away from zero.  Module-level code of a script is plain Python (double), as it is
under Taichi.  `if ti.random() < prob` (reference example6.py:39, 54; example7.py:20-22) therefore takes the branch an f32
`prob` takes.
"""
import builtins as _b
import os as _os

from . import math  # noqa: F401  (taichi.math)
from . import _kernel
from .math import Vector, _map, _is_vec, _scope, _f32, _i2f, _trunc, _unify
from .math import sin, cos, tan, asin, acos, exp, log, sqrt, floor, ceil, atan2, abs, pow, max, min, round, tanh  # noqa: F401,A004

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
    return (w >> 8) * (1.0 / 16777216.0)      # exactly representable in binary32


def cast(x, dtype):
    return _kernel._k_int(x) if dtype is int else _kernel._k_float(x)


def select(c, a, f):
    def one(cc, aa, ff):
        aa, ff = _unify(aa, ff)
        return aa if cc else ff
    return _map(one, c, a, f)


# ---- decorators ----------------------------------------------------------------------------------------------------
def _lazy(fn, is_kernel):
    """Compiled at the first call: every module-level name the body uses exists by then (and Taichi, too, compiles then)."""
    state = []

    def compiled():
        if not state:
            state.append(_kernel.compile_dsl(fn))
        return state[0]

    if is_kernel:
        def run(*args, **kwargs):
            f = compiled()
            _scope[0] += 1
            try:
                return f(*args, **kwargs)
            finally:
                _scope[0] -= 1
    else:
        def run(*args, **kwargs):
            return (state[0] if state else compiled())(*args, **kwargs)
    run.__name__, run.__doc__, run.__wrapped__ = fn.__name__, fn.__doc__, fn
    return run


def kernel(fn):
    return _lazy(fn, True)


def func(fn):
    return _lazy(fn, False)


def data_oriented(cls):
    return cls


def static(x, *rest):
    return x if not rest else (x,) + rest


def template():
    return None


class _Types:
    @staticmethod
    def vector(n, dtype=float):
        return lambda *a: Vector(list(a) if len(a) != 1 else a[0], dtype)

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
    new = Vector._new
    for idx in it:
        yield new(list(idx) if isinstance(idx, tuple) else [idx])
