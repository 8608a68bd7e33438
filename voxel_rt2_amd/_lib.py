"""Loader of libvrt_hip.so.  Fails loudly: there is no CPU implementation to fall back to."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libvrt_hip.so")
_lib = None


class LibraryMissing(RuntimeError):
    pass


def load(build_if_missing=True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        if not build_if_missing:
            raise LibraryMissing(f"{SO_PATH} not found; run `python -m voxel_rt2_amd.build` (needs hipcc)")
        from . import build
        build.build()
    try:
        lib = C.CDLL(SO_PATH)
    except OSError as e:
        raise LibraryMissing(f"cannot load {SO_PATH}: {e}. The renderer only runs through the HIP library "
                             "(ROCm runtime + an MI355X / gfx950 device); there is no CPU fallback.") from e
    _abi.declare(lib, "vrt_")
    lib.vrt_create.restype = C.c_void_p
    lib.vrt_create.argtypes = [C.POINTER(_abi.VrtConfig)]
    lib.vrt_set_instrumented.restype = C.c_int
    lib.vrt_set_instrumented.argtypes = [C.c_void_p, C.c_int]
    lib.vrt_reset_stats.restype = C.c_int
    lib.vrt_reset_stats.argtypes = [C.c_void_p]
    lib.vrt_detmath_probe.restype = C.c_int
    lib.vrt_detmath_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def exported_symbols():
    """Names include/vrt_api.h declares (used by the CPU-side ABI test)."""
    hdr = os.path.join(os.path.dirname(_HERE), "include", "vrt_api.h")
    import re
    text = open(hdr).read()
    return sorted(set(re.findall(r"\b(vrt_[a-z_]+)\s*\(", text)))
