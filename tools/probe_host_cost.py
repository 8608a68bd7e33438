#!/usr/bin/env python3
"""Host cost of one vrt_accumulate call: a frame so small that the device is never the limit (128 x 64, 2 bounces), many calls queued,
one synchronisation at the end.  Prints microseconds per call for the overlapped pipeline and for isolated launches, and the
share of it spent inside the C entry point (the rest is ctypes / Python)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import ctypes as C
from voxel_rt2_amd import host, scenes, materials, _lib
from voxel_rt2_amd._session import NativeSession

lib = _lib.load()
N = int(os.environ.get("N", 3000))
for overlap, spp in (("1", 4), ("1", 1), ("0", 4)):
    os.environ["VRT_OVERLAP"] = overlap
    mat, rgb, params = scenes.scene_s1(0)
    W, H = 128, 64
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=2, seed=0)
    s = NativeSession(lib, "vrt_", cfg)
    s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
    s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(W, H, jitter_index=1)); s.prepare()
    for _ in range(50):
        s.accumulate(spp)
    s.sync()
    fn, ctx = lib.vrt_accumulate, C.c_void_p(s._ctx)
    t0 = time.perf_counter()
    for _ in range(N):
        fn(ctx, spp)
    t1 = time.perf_counter()
    s.sync()
    t2 = time.perf_counter()
    print(f"VRT_OVERLAP={overlap} spp={spp}: {1e6 * (t1 - t0) / N:.1f} us per call to queue, {1e6 * (t2 - t0) / N:.1f} us per call incl. the final wait")
    s.close()
os.environ.pop("VRT_OVERLAP", None)
