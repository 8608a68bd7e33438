#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one bench step (vrt_accumulate + tile copy + events) against the device's
step period?  One rank's share of an 8-way split (rows 472..607 of config 2).  usage: tools/host_overhead.py [rows0 rows1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import torch
from voxel_rt2_amd import host, scenes, materials, _lib
from voxel_rt2_amd._session import NativeSession
r0, r1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (472, 607)
W, H = 1920, 1080
mat, rgb, params = scenes.scene_s1(0)
lib = _lib.load()
stream = torch.cuda.Stream()
cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0, rows=(r0, r1) if (r0, r1) != (0, H) else None)
s = NativeSession(lib, "vrt_", cfg)
s.set_stream(stream.cuda_stream)
s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table()); s.set_scene(host.make_scene_params(**params))
s.set_camera(host.default_camera(W, H, jitter_index=1)); s.prepare()
tiles = [torch.zeros((r1 - r0, W, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
side = torch.cuda.Stream()
MODE = os.environ.get("HO_MODE", "all")   # all | none | fetch | events
def step(k):
    with torch.cuda.stream(stream):
        s.accumulate(4)
        if MODE in ("all", "fetch"): s.fetch_hdr_device_async(tiles[k & 1].data_ptr())
        if MODE in ("all", "events"): ev = stream.record_event()
    if MODE in ("all", "events"):
        with torch.cuda.stream(side):
            side.wait_event(ev)
for k in range(10): step(k)
torch.cuda.synchronize()
for n in (20, 200):
    t0 = time.perf_counter()
    for k in range(n): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"steps {n}: enqueue {1e3 * (t1 - t0) / n:.3f} ms/step, total {1e3 * (t2 - t0) / n:.3f} ms/step")
s.close()
