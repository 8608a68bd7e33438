#!/usr/bin/env python3
"""Step the overlapped-launch pipeline a few times and say after each step that it returned: run under a profiler with a
short `timeout -k` to see whether (and where) a launch schedule stalls.  usage: probe_overlap.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from voxel_rt2_amd import host, scenes, materials, _lib
from voxel_rt2_amd._session import NativeSession
W, H = 960, 540
mat, rgb, params = scenes.scene_s1(0)
cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0)
s = NativeSession(_lib.load(), "vrt_", cfg)
s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(W, H, jitter_index=1)); s.prepare()
print("prepared", flush=True)
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    t = time.perf_counter()
    s.accumulate(4)
    print(f"step {k} queued {time.perf_counter() - t:.4f}s", flush=True)
    if k % 2 == 1:
        s.sync(); print(f"step {k} synced {time.perf_counter() - t:.4f}s", flush=True)
s.sync()
print("done mean", float(s.fetch_hdr().mean()), flush=True)
