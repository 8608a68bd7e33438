"""ctypes binding of the CPU oracle (oracle/_build/libvrt_oracle.so) -- test infrastructure.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import ctypes as C
import os
import subprocess
import numpy as np

from voxel_rt2_amd import _abi, host, materials, scenes
from voxel_rt2_amd._session import NativeSession

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "_build", "libvrt_oracle.so")
_lib = None


def build(force=False):
    src_dir = os.path.join(ROOT, "oracle")
    if os.path.exists(os.path.join(src_dir, "Makefile")) and (force or not os.path.exists(_SO) or _stale(src_dir)):
        subprocess.run(["make", "-C", src_dir], check=True, capture_output=True)
    return _SO


def _stale(src_dir):
    t = os.path.getmtime(_SO)
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir) if f.endswith((".h", ".cpp"))]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return any(os.path.getmtime(d) > t for d in deps)


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_unit_encode_material.restype = C.c_uint32
        _lib.orc_unit_hash3.restype = C.c_uint32
        _lib.orc_unit_hash3.argtypes = [C.c_uint32] * 3
    return _lib


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle(NativeSession):
    def __init__(self, cfg, threads=None):
        threads = threads or min(8, os.cpu_count() or 1)
        super().__init__(lib(), "orc_", cfg, create_extra=(C.c_int(threads),))
        self.threads = threads

    def upload_sky(self, scat, trans):
        scat = np.ascontiguousarray(scat, dtype=np.float32)
        trans = np.ascontiguousarray(trans, dtype=np.float32)
        assert scat.shape == trans.shape == (self.cfg.sky_res, self.cfg.sky_res, 3)
        self._lib.orc_upload_sky(C.c_void_p(self._ctx), fptr(scat), fptr(trans))

    # single-function probes ---------------------------------------------------------------
    def query_occupancy(self, x, y, z, lod):
        return bool(self._lib.orc_unit_query_occupancy(C.c_void_p(self._ctx), int(x), int(y), int(z), int(lod)))

    def raytrace(self, origin, direction, tmin=1e-6, tmax=np.inf):
        o = np.asarray(origin, dtype=np.float32)
        d = np.asarray(direction, dtype=np.float32)
        out = np.zeros(8, dtype=np.float32)
        self._lib.orc_unit_raytrace(C.c_void_p(self._ctx), fptr(o), fptr(d), C.c_float(tmin), C.c_float(tmax), fptr(out))
        return dict(distance=out[0], cell=out[1:4].astype(int), normal=out[4:7].copy(), iters=int(out[7]))

    def next_hit(self, origin, direction, shadow=False):
        o = np.asarray(origin, dtype=np.float32)
        d = np.asarray(direction, dtype=np.float32)
        out = np.zeros(9, dtype=np.float32)
        self._lib.orc_unit_next_hit(C.c_void_p(self._ctx), fptr(o), fptr(d), int(shadow), fptr(out))
        return dict(closest=out[0], normal=out[1:4].copy(), albedo=out[4:7].copy(), hit_light=int(out[7]), mat_id=int(out[8]))

    def cast_dir(self, u, v):
        out = np.zeros(3, dtype=np.float32)
        self._lib.orc_unit_cast_dir(C.c_void_p(self._ctx), int(u), int(v), fptr(out))
        return out


def detmath(op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float32)
    out = np.empty_like(a)
    lib().orc_unit_detmath(int(op), int(a.size), fptr(a), fptr(b), fptr(out))
    return out


def setup(sess, mat, rgb, params, *, cam=None, table=None, cloud=None):
    """Drive a session (oracle or product: same calls) to the point where accumulate() may run."""
    sess.upload_voxels(mat, rgb)
    sess.upload_materials(table if table is not None else materials.load_table())
    if cloud is not None:
        sess.upload_cloud_texture(cloud)
    sess.set_scene(host.make_scene_params(**params))
    sess.set_camera(cam if cam is not None else host.default_camera(sess.W, sess.H))
    sess.prepare()
