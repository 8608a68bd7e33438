#!/usr/bin/env python3
"""Bisect a GPU-vs-oracle mismatch on config 3's full-size shard: which ingredient (sky tables, clouds, ReSTIR, size) and which buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from voxel_rt2_amd import _abi, _lib, host, scenes
from voxel_rt2_amd._session import NativeSession
cloud = np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy"))
mat, rgb, params0 = scenes.scene_s6(0)

def run(W, H, rows, R, restir, sky, spp=2, depth=8, tables=None):
    params = dict(params0, use_physical_sky=int(sky), use_clouds=int(sky))
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=0, use_restir=restir,
                           sky_res=R if sky else 0, rows=rows)
    g = NativeSession(_lib.load(), "vrt_", cfg)
    orc.setup(g, mat, rgb, params, cloud=cloud if sky else None)
    o = orc.Oracle(cfg, threads=32)
    orc.setup(o, mat, rgb, params, cloud=cloud if sky else None)
    if sky:
        for _ in range(32): g.sky_accumulate_clouds(32)
        for sl in range(32): g.sky_compute_slice(sl, 32)
        scat, trans = g.fetch_buffer(_abi.BUF_SKY_SCATTERING), g.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE)
        o.upload_sky(scat, trans)
        den = np.abs(scat[np.isfinite(scat) & (scat != 0)])
        print(f"   tables: scat min |x| {den.min():.3e}, denormals {(den < 1.18e-38).sum()}, trans denormals {((np.abs(trans) < 1.18e-38) & (trans != 0)).sum()}")
    for s in (g, o): s.accumulate(spp)
    sl = slice(rows[0], rows[1]) if rows else slice(None)
    res = {}
    a, b = g.fetch_hdr()[sl], o.fetch_hdr()[sl]
    res["hdr"] = int((a.view(np.uint32) != b.view(np.uint32)).sum())
    for name, which in (("depth", _abi.BUF_GBUF_DEPTH), ("normal", _abi.BUF_GBUF_NORMAL), ("mat", _abi.BUF_GBUF_MAT), ("hist_d", _abi.BUF_HISTORY_DIFFUSE), ("hist_s", _abi.BUF_HISTORY_SPECULAR)):
        x, y = g.fetch_buffer(which)[sl], o.fetch_buffer(which)[sl]
        res[name] = int((x.view(np.uint8) != y.view(np.uint8)).any(axis=-1).sum())
    d = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
    if d.any():
        vv, uu = np.nonzero(d)
        print("   first differing pixels (row in shard, col):", list(zip(vv[:6].tolist(), uu[:6].tolist())), "cols span", uu.min(), uu.max(), "max ulp", int(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32)).max()))
    print(f"W={W} H={H} rows={rows} R={R} restir={restir} sky={sky} spp={spp}: differing -> {res}", flush=True)
    g.close(); o.close()

run(1920, 1080, (560, 576), 3840, False, True)
run(1920, 1080, (560, 576), 0, True, False)
run(1920, 1080, (560, 576), 256, True, True)
run(1920, 1080, (560, 576), 3840, True, True, spp=1)
run(480, 270, (140, 156), 3840, True, True)
