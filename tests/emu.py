"""ctypes binding of tests/emul/_emul.so: the product's device headers compiled for the host
(test tooling only; see tests/emul/emul.cpp)."""
import ctypes as C
import os
import subprocess

from voxel_rt2_amd._session import NativeSession

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_SO = os.path.join(HERE, "emul", "_emul.so")
_lib = None


def build(force=False):
    src = os.path.join(HERE, "emul", "emul.cpp")
    deps = [src] + [os.path.join(ROOT, "voxel_rt2_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "voxel_rt2_amd", "csrc"))
                    if f.endswith(".h")] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(d) > os.path.getmtime(_SO) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-shared",
                        "-Wno-unknown-pragmas", "-o", _SO, src], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
    return _lib


class Emulated(NativeSession):
    def __init__(self, cfg):
        super().__init__(lib(), "emu_", cfg)

    def upload_sky(self, scat, trans):
        import numpy as np
        scat = np.ascontiguousarray(scat, dtype=np.float32)
        trans = np.ascontiguousarray(trans, dtype=np.float32)
        self._lib.emu_upload_sky(C.c_void_p(self._ctx), scat.ctypes.data_as(C.c_void_p), trans.ctypes.data_as(C.c_void_p))
