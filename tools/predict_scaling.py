#!/usr/bin/env python3
"""What `bench.py --gpus N` does at N = 2, 4, 8, replayed on ONE GPU rank by rank: the equal split, the five balancing passes
of ShardedRun (each tile's step time with the pipeline in flight, parallel.rebalance_rows), then every balanced tile
timed alone with the pipeline in flight.  The slowest tile's step time is the predicted step time of the N-GPU run (the gather
overlaps the next step); against the whole frame's step time measured in the same process.  Also: interleaved stripes, the rank
that gets the most rows.

    python tools/predict_scaling.py [config2|config4|config5] [N ...]          one JSON line per N
"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import ctypes as C
import numpy as np
from voxel_rt2_amd import host, scenes, materials, parallel, _lib
from voxel_rt2_amd._session import NativeSession

CONFIGS = {"config2": dict(scene="s1", W=1920, H=1080, grid=128, steps=120),
           "config4": dict(scene="dense", W=3840, H=2160, grid=128, steps=12),
           "config5": dict(scene="dense256", W=3840, H=2160, grid=256, steps=12)}
lib = _lib.load()


def session(cf, rows=None, stripes=None, world=1):
    mat, rgb, params = scenes.SCENES[cf["scene"]](12345 if cf["scene"].startswith("dense") else 0)
    params = dict(params, use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(cf["W"], cf["H"], voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0, rows=rows, grid_res=cf["grid"])
    s = NativeSession(lib, "vrt_", cfg)
    if stripes:
        s.set_row_stripes(*stripes)
    if world > 1:
        parallel.configure_session(s, world)   # the workgroup slots a rank of a group leaves to RCCL
    s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
    s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(cf["W"], cf["H"], jitter_index=1)); s.prepare()
    return s


def step_ms(cf, rows=None, stripes=None, world=1):
    s = session(cf, rows, stripes, world)
    for _ in range(8):
        s.accumulate(4)
    s.sync()
    t0 = time.perf_counter()
    for _ in range(cf["steps"]):
        s.accumulate(4)
    s.sync()
    ms = (time.perf_counter() - t0) / cf["steps"] * 1e3
    n = len(s.owned_rows()) if stripes else (rows[1] - rows[0] if rows else cf["H"])
    s.close()
    return ms, n


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config2"
    cf = CONFIGS[name]
    worlds = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
    full, _ = step_ms(cf)
    print(json.dumps(dict(config=name, whole_frame_ms=round(full, 4))), flush=True)
    for world in worlds:
        bounds = parallel.split_rows(cf["H"], world)
        equal = [step_ms(cf, b, None, world)[0] for b in bounds]
        short = dict(cf, steps=24)
        for _ in range(5):   # ShardedRun's balancing passes: the pipelined step time of every tile, boundaries moved, again
            bounds = parallel.rebalance_rows(bounds, [step_ms(short, b, None, world)[0] for b in bounds], cf["H"])
        bal = [step_ms(cf, b, None, world)[0] for b in bounds]
        out = dict(config=name, n_gpus=world, whole_frame_ms=round(full, 4),
                   equal_tiles_ms=[round(x, 4) for x in equal], equal_speedup=round(full / max(equal), 2),
                   balanced_bounds=bounds, balanced_tiles_ms=[round(x, 4) for x in bal], balanced_speedup=round(full / max(bal), 2))
        for srows in ((16, 24, 32, 64) if name == "config2" else (64, 128)):
            # the rank with the most rows (part 0 owns the frame's first stripe and any remainder comes to the low parts first)
            worst = max(range(world), key=lambda r: sum(min(a + srows, cf["H"]) - a for a in range(r * srows, cf["H"], srows * world)))
            ms, n = step_ms(cf, None, (srows, world, worst), world)
            out[f"stripes{srows}"] = dict(part=worst, rows=n, ms=round(ms, 4), speedup=round(full / ms, 2))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
