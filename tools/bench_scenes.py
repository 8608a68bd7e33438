#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on one GPU (informational; bench.py is the judged line).
Prints one JSON line per case: Mpath-samples/s, per-kernel ms, traversal counters per path-sample."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from voxel_rt2_amd import host, scenes, materials, _lib
from voxel_rt2_amd._session import NativeSession

CASES = [
    dict(name="config2_s1_1080p_d8", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=10),
    dict(name="sunlit_1080p_d8", scene="sunlit", W=1920, H=1080, depth=8, spp=4, steps=10),
    dict(name="config4_dense_4k_d8_1gpu", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=4),
    dict(name="config3_s6_sky_clouds_restir_1080p_d8", scene="s6", W=1920, H=1080, depth=8, spp=4, steps=5, restir=True, sky_res=3840),
    dict(name="s6_sky_clouds_1080p_d8_norestir", scene="s6", W=1920, H=1080, depth=8, spp=4, steps=5, sky_res=3840),
    dict(name="sunlit_restir_1080p_d8", scene="sunlit", W=1920, H=1080, depth=8, spp=4, steps=5, restir=True),
    dict(name="s6_nosky_restir_1080p_d8", scene="s6", W=1920, H=1080, depth=8, spp=4, steps=5, restir=True),
    dict(name="s6_nosky_plain_1080p_d8", scene="s6", W=1920, H=1080, depth=8, spp=4, steps=5),
    # BASELINE config 5: the 256^3 grid (one GPU's form: the whole 4K frame, 4 fused samples per step)
    dict(name="config5_dense256_4k_d8_1gpu", scene="dense256", W=3840, H=2160, depth=8, spp=4, steps=4, grid=256),
    dict(name="sponge256_4k_d8", scene="sponge256", W=3840, H=2160, depth=8, spp=4, steps=4, grid=256),
    dict(name="s1_256_1080p_d8", scene="s1_256", W=1920, H=1080, depth=8, spp=4, steps=10, grid=256),
    # one rank's share of an 8- and a 2-GPU run of config 2 (rows through the middle of the picture)
    dict(name="shard_1of8_config2", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=20, rows=(472, 607)),
    dict(name="shard_1of2_config2", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=20, rows=(0, 540)),
    dict(name="shard_1of4_config2", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=20, rows=(405, 675)),
    # one rank's share of config 4 / config 5 on 2, 4 and 8 GPUs (the dense fill costs the same everywhere)
    dict(name="shard_1of2_config4", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=6, rows=(0, 1080)),
    dict(name="shard_1of4_config4", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=8, rows=(540, 1080)),
    dict(name="shard_1of8_config4", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=12, rows=(1080, 1350)),
    dict(name="shard_1of8_config5", scene="dense256", W=3840, H=2160, depth=8, spp=4, steps=12, rows=(1080, 1350), grid=256),
    # one rank's share of an 8-GPU run as INTERLEAVED STRIPES (vrt_set_row_stripes): part 3 of 8, stripes of 8 / 32 / 64 rows
    dict(name="stripes8_1of8_config2", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=20, stripes=(8, 8, 3)),
    dict(name="stripes32_1of8_config2", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=20, stripes=(32, 8, 3)),
    dict(name="stripes64_1of8_config2", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=20, stripes=(64, 8, 3)),
    dict(name="stripes32_1of8_config4", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=12, stripes=(32, 8, 3)),
    dict(name="stripes64_1of8_config4", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=12, stripes=(64, 8, 3)),
    # the reference's own loop shape: one sample per call, a new jitter and vrt_end_frame every frame (scene.py:177, 233-262)
    dict(name="scene_api_default_1080p", scene="s1", W=1920, H=1080, depth=8, spp=1, steps=120, per_frame_camera=True),
]


def run(case):
    if os.environ.get("VRT_BENCH_STEPS"):   # longer runs: a deep launch pipeline takes a few steps to fill and to drain
        case = dict(case, steps=int(os.environ["VRT_BENCH_STEPS"]))
    lib = _lib.load()
    mat, rgb, params = scenes.SCENES[case["scene"]](12345 if case["scene"].startswith("dense") else 0)
    sky_res = case.get("sky_res", 0)
    if not sky_res:
        params = dict(params, use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(case["W"], case["H"], voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=case["depth"],
                           seed=0, use_restir=case.get("restir", False), sky_res=sky_res, rows=case.get("rows"), grid_res=case.get("grid", 128))
    s = NativeSession(lib, "vrt_", cfg)
    if case.get("stripes"):
        s.set_row_stripes(*case["stripes"])
    if os.environ.get("VRT_BENCH_RESERVE"):   # one GPU's cost of the workgroup slots a multi-GPU rank leaves to RCCL
        s.reserve_cus(int(os.environ["VRT_BENCH_RESERVE"]))
    s.upload_voxels(mat, rgb)
    s.upload_materials(materials.load_table())
    if sky_res:
        s.upload_cloud_texture(np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy")))
    s.set_scene(host.make_scene_params(**params))
    s.set_camera(host.default_camera(case["W"], case["H"], jitter_index=1))
    s.prepare()
    sky_s = 0.0
    if sky_res:
        s.sync()
        t0 = time.perf_counter()
        for _ in range(32):
            s.sky_accumulate_clouds(32)
        s.sync()
        t1 = time.perf_counter()
        for sl in range(32):
            s.sky_compute_slice(sl, 32)
        s.sync()
        sky_s = time.perf_counter() - t0
        print(json.dumps(dict(name=case["name"], sky_clouds_s=round(t1 - t0, 3), sky_slices_s=round(time.perf_counter() - t1, 3))), flush=True)
    s.accumulate(case["spp"])
    s.sync()
    # the PCIe-inclusive rate: the frame copied to host memory after every step.  1: blocking vrt_fetch_hdr into pageable memory;
    # "async": vrt_fetch_ldr_async (tonemap + copy on the library's fetch stream) into two page-locked buffers, the caller
    # collecting frame k - 1 while frame k renders -- the reference's accumulate / fetch_image loop (scene.py:255-262)
    fetch_each = os.environ.get("VRT_BENCH_FETCH_EACH", "")
    sync_each = bool(os.environ.get("VRT_BENCH_SYNC_EACH"))   # a caller that looks at every frame: no two launches in flight
    # "async8": the same with the 8-bit image (vrt_fetch_ldr8_async), a quarter of the bytes.  VRT_BENCH_FETCH_LAG (default 1):
    # how many frames behind the caller collects (2: frame k - 2 while frames k - 1 and k render)
    lag = int(os.environ.get("VRT_BENCH_FETCH_LAG", 1))
    pinned = ([s.host_alloc((case["H"], case["W"], 4), np.uint8 if fetch_each == "async8" else np.float32) for _ in range(lag + 1)]
              if fetch_each in ("async", "async8") else None)
    cams = [host.default_camera(case["W"], case["H"], jitter_index=k + 1) for k in range(16)] if case.get("per_frame_camera") else None
    if pinned:   # the fetch path's one-time set-up (its stream, events and the 8-bit image) is not a frame's cost
        (s.fetch_ldr8_async if fetch_each == "async8" else s.fetch_ldr_async)(pinned[0], slot=0)
        s.fetch_wait(0)
        s.sync()
    lib.vrt_reset_stats(C.c_void_p(s._ctx))
    t0 = time.perf_counter()
    for k in range(case["steps"]):
        if cams:
            s.set_camera(cams[k % 16])
        s.accumulate(case["spp"])
        if cams:
            s.end_frame()
        if sync_each:
            s.sync()
        if pinned:
            (s.fetch_ldr8_async if fetch_each == "async8" else s.fetch_ldr_async)(pinned[k % (lag + 1)], slot=k % (lag + 1))
            if k >= lag:
                s.fetch_wait((k - lag) % (lag + 1))
        elif fetch_each:
            s.fetch_hdr()
    if pinned:
        for k in range(max(case["steps"] - lag, 0), case["steps"]):
            s.fetch_wait(k % (lag + 1))
    s.sync()
    dt = time.perf_counter() - t0
    st = s.stats()
    lib.vrt_set_instrumented(C.c_void_p(s._ctx), 1)
    lib.vrt_reset_stats(C.c_void_p(s._ctx))
    s.accumulate(1)
    ist = s.stats()
    n = ist["path_samples"]
    hdr = s.fetch_hdr()
    rows = case.get("rows") or (0, case["H"])
    n_rows = len(s.owned_rows()) if case.get("stripes") else rows[1] - rows[0]
    out = dict(name=case["name"], mpaths_per_s=round(case["W"] * n_rows * case["spp"] * case["steps"] / dt / 1e6, 1), ms_per_step=round(dt / case["steps"] * 1e3, 4), rows=int(n_rows),
               render_ms=round(st["render_ms"] / max(st["render_launches"], 1), 3),
               gris_ms=round(st["gris_ms"] / max(st["gris_launches"], 1), 3),
               temporal_ms=round(st["temporal_ms"] / max(st["temporal_launches"], 1), 3),
               rays_per_path=round(ist["rays"] / n, 2), iters_per_ray=round(ist["dda_iters"] / max(ist["rays"], 1), 2),
               queries_per_path=round(ist["occupancy_queries"] / n, 1), hits_per_path=round(ist["closest_hits"] / n, 3),
               sky_lookups_per_path=round(ist["sky_lookups"] / n, 3), sky_precompute_s=round(sky_s, 2),
               hdr_mean=float(hdr.mean()), finite=bool(np.isfinite(hdr).all()))
    print(json.dumps(out), flush=True)
    s.close()


if __name__ == "__main__":
    only = sys.argv[1:]
    for o in only:   # rows:A:B -> rows A..B of config 2 (S1, 1080p, 4 samples a step) as one rank's contiguous tile
        if o.startswith("rows:"):
            a, b = (int(x) for x in o.split(":")[1:3])
            run(dict(name=f"config2_rows_{a}_{b}", scene="s1", W=1920, H=1080, depth=8, spp=4, steps=100, rows=(a, b)))
    for c in CASES:
        if not only or any(o in c["name"] for o in only):
            run(c)
