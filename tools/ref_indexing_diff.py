#!/usr/bin/env python3
"""How far the two readings of cells OUTSIDE the grid are apart on a frame (vrt_set_reference_indexing, include/vrt_api.h).

Renders a BASELINE config twice -- default ("a query outside the grid is empty") and with the reference's own index arithmetic
(raytracer.py:17-44: the bit of another cell; a set one is a hit on a "voxel" outside the grid, painted black) -- and reports the
pixels whose HDR value differs and the relative L2 between the two frames.  Same seed, same random streams: a pixel differs only
if one of its paths met such a read.  The oracle also counts the reads themselves (orc_get_stats is per call, so the counts here
are per pixel of the image: pixels whose g-buffer differs = the camera ray itself hit outside).

    python tools/ref_indexing_diff.py --backend oracle --config 4 --spp 1 [--size 3840x2160] [--rows a:b]
    python tools/ref_indexing_diff.py --backend gpu --config 4           # 16 spp; config 5: 32 spp
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
from voxel_rt2_amd import host, scenes, materials, _abi  # noqa: E402

CONFIGS = {4: dict(scene="dense", grid=128, spp=16), 5: dict(scene="dense256", grid=256, spp=32)}


def session(backend, cfg):
    if backend == "oracle":
        import orc
        return orc.Oracle(cfg, threads=os.cpu_count())
    from voxel_rt2_amd import _lib
    from voxel_rt2_amd._session import NativeSession
    return NativeSession(_lib.load(), "vrt_", cfg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["oracle", "gpu"], default="gpu")
    ap.add_argument("--config", type=int, choices=[4, 5], default=4)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--rows", default="")
    ap.add_argument("--depth", type=int, default=8)
    a = ap.parse_args()
    c = CONFIGS[a.config]
    W, H = (int(x) for x in a.size.split("x"))
    spp = a.spp or c["spp"]
    rows = tuple(int(x) for x in a.rows.split(":")) if a.rows else None
    mat, rgb, params = scenes.SCENES[c["scene"]](12345)
    params = dict(params, use_physical_sky=0, use_clouds=0)
    frames, gbuf, secs = [], [], []
    for ref_idx in (False, True):
        cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=a.depth, seed=0,
                               grid_res=c["grid"], rows=rows)
        s = session(a.backend, cfg)
        s.set_reference_indexing(ref_idx)
        s.upload_voxels(mat, rgb)
        s.upload_materials(materials.load_table())
        s.set_scene(host.make_scene_params(**params))
        s.set_camera(host.default_camera(W, H))
        s.prepare()
        t = time.time()
        done = 0
        while done < spp:
            n = min(4, spp - done)
            s.accumulate(n)
            done += n
        frames.append(s.fetch_hdr())
        gbuf.append(s.fetch_buffer(_abi.BUF_GBUF_MAT))
        secs.append(time.time() - t)
        s.close()
    r0, r1 = rows if rows else (0, H)
    d, r = frames[0][r0:r1].astype(np.float64), frames[1][r0:r1].astype(np.float64)
    differ = (frames[0][r0:r1].view(np.uint32) != frames[1][r0:r1].view(np.uint32)).any(-1)
    primary = (gbuf[0][r0:r1] != gbuf[1][r0:r1]).any(-1)
    out = dict(config=a.config, backend=a.backend, scene=c["scene"], grid=c["grid"], size=[W, H], rows=[r0, r1], spp=spp, depth=a.depth,
               pixels=int(differ.size), pixels_differ=int(differ.sum()), share_differ=float(differ.mean()),
               pixels_camera_ray_hits_outside=int(primary.sum()),
               rel_l2=float(np.linalg.norm((d - r).ravel()) / np.linalg.norm(r.ravel())),
               mean_default=float(d.mean()), mean_reference_indexing=float(r.mean()),
               seconds=[round(x, 1) for x in secs])
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
