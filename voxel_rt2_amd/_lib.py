"""Loader of libvrt_hip.so.  Fails loudly: there is no CPU implementation to fall back to."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("VRT_LIB_PATH") or os.path.join(_HERE, "libvrt_hip.so")  # VRT_LIB_PATH: A/B builds of the same library
_lib = None


class LibraryMissing(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in
    one process cannot both own the device (the second sees no GPU), and torch.distributed (RCCL)
    lives in torch's.  So if torch is installed, map ITS runtime before libvrt_hip.so is loaded:
    the loader then resolves our DT_NEEDED libamdhip64.so.7 to it by SONAME, and a later
    `import torch` finds its library already mapped.  Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def _want_hw_queues(n=16):
    """The library renders on up to eight streams of its own beside the caller's (vrt_api.hip, ensure_overlap); a multi-GPU
    rank adds a gather stream and RCCL's.  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues
    (default 4) and two streams that share a queue serialise: a wait queued for the gather then holds back the next render
    launch (one rank's share of an 8-way split: 0.418 ms per step with 4 queues, 0.317 with 8; the eight-launch pipeline the
    library uses on such small frames wants 16).  The variable is read when the runtime starts, so it is set here -- a
    process-wide side effect: torch and child processes see it too -- before libvrt_hip.so (or torch) initialises HIP, unless
    the user chose a value (any value: set GPU_MAX_HW_QUEUES yourself to opt out)."""
    import sys
    import warnings
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    torch = sys.modules.get("torch")
    try:
        up = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        up = False
    if up:
        warnings.warn("voxel_rt2_amd: the HIP runtime was initialised before the library was loaded, so it keeps its default of "
                      "4 hardware queues; export GPU_MAX_HW_QUEUES=16 (or import voxel_rt2_amd first) for multi-GPU runs",
                      RuntimeWarning, stacklevel=3)


DEV_SO_PATH = os.path.join(os.path.dirname(_HERE), "build_variants", "libvrt_dev.so")
_dev = None


def load_dev():
    """The same library built with -DVRT_DEV_KNOBS (build_variants/libvrt_dev.so): it also reads the development switches --
    the fault-injection hook and the A/B switches (csrc/vrt_api.hip, read_knobs) -- which the shipped library does not carry.
    For tests/test_gpu_pipeline.py and the A/B runs of tools/; built here when missing or older than the sources."""
    global _dev
    if _dev is None:
        from . import build
        if not os.path.exists(DEV_SO_PATH) or any(os.path.getmtime(p) > os.path.getmtime(DEV_SO_PATH) for p in build._deps()):
            build.build(force=True, extra_flags=["-DVRT_DEV_KNOBS"], out=DEV_SO_PATH)
        _dev = _open(DEV_SO_PATH)
    return _dev


def load(build_if_missing=True):
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        if not build_if_missing:
            raise LibraryMissing(f"{SO_PATH} not found; run `python -m voxel_rt2_amd.build` (needs hipcc)")
        from . import build
        build.build()
    _lib = _open(SO_PATH)
    return _lib


def _open(path):
    _want_hw_queues()
    _share_hip_runtime_with_torch()
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise LibraryMissing(f"cannot load {path}: {e}. The renderer only runs through the HIP library "
                             "(ROCm runtime + an MI355X / gfx950 device); there is no CPU fallback.") from e
    _abi.declare(lib, "vrt_")
    lib.vrt_create.restype = C.c_void_p
    lib.vrt_create.argtypes = [C.POINTER(_abi.VrtConfig)]
    lib.vrt_set_instrumented.restype = C.c_int
    lib.vrt_set_instrumented.argtypes = [C.c_void_p, C.c_int]
    lib.vrt_build_id.restype = C.c_char_p
    lib.vrt_build_id.argtypes = []
    lib.vrt_reserve_cus.restype = C.c_int
    lib.vrt_reserve_cus.argtypes = [C.c_void_p, C.c_int]
    lib.vrt_reset_stats.restype = C.c_int
    lib.vrt_reset_stats.argtypes = [C.c_void_p]
    lib.vrt_detmath_probe.restype = C.c_int
    lib.vrt_detmath_probe.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.vrt_sky_probe.restype = C.c_int
    lib.vrt_sky_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def build_id():
    return load().vrt_build_id().decode()


def exported_symbols():
    """Names include/vrt_api.h declares (used by the CPU-side ABI test)."""
    hdr = os.path.join(os.path.dirname(_HERE), "include", "vrt_api.h")
    import re
    text = open(hdr).read()
    return sorted(set(re.findall(r"\b(vrt_[a-z_]+)\s*\(", text)))
