#!/usr/bin/env python3
"""One-off wider parity sweep on a GPU box: random sizes / seeds / depths / sample counts / scenes, libvrt_hip.so
against the CPU oracle, bit for bit (HDR, g-buffer, histories).  Not part of the test-suite (tests/test_gpu_parity.py
holds the fixed cases); run it after changing the render schedule.  usage: tools/soak_parity.py [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import orc
from voxel_rt2_amd import _abi, _lib, host, scenes
from voxel_rt2_amd._session import NativeSession

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
BUFS = (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_POSITION, _abi.BUF_GBUF_MAT, _abi.BUF_GBUF_REFL_DEPTH,
        _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR)
bad = 0
for k in range(n_cases):
    scene = ["s1", "sunlit", "dense", "s6"][int(rng.integers(0, 4))]
    W, H = int(rng.integers(40, 400)), int(rng.integers(24, 240))
    depth, seed = int([1, 2, 3, 4, 5, 6, 8, 12, 15, 16, 20][int(rng.integers(0, 11))]), int(rng.integers(0, 1 << 30))  # > 15: the fused kernel
    calls = [int(rng.integers(1, 9)) for _ in range(int(rng.integers(1, 5)))]  # accumulate(n) calls: fused 4 + remainder, several launches
    mat, rgb, params = scenes.SCENES[scene](int(rng.integers(0, 5)))
    params = dict(params, use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=seed)
    g, o = NativeSession(_lib.load(), "vrt_", cfg), orc.Oracle(cfg)
    for s in (g, o):
        orc.setup(s, mat, rgb, params)
        for n in calls:
            s.accumulate(n)
    ok = np.array_equal(g.fetch_hdr().view(np.uint32), o.fetch_hdr().view(np.uint32))
    for which in BUFS:
        ok = ok and np.array_equal(g.fetch_buffer(which).view(np.uint8), o.fetch_buffer(which).view(np.uint8))
    # the multi-GPU decomposition: contexts that own a random split of the rows reassemble the same frame
    full = g.fetch_hdr()
    cuts = sorted(set([0, H] + [int(x) for x in rng.integers(1, H, size=int(rng.integers(1, 4)))]))
    parts = np.zeros_like(full)
    for a, b in zip(cuts[:-1], cuts[1:]):
        cfg_s = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=seed, rows=(a, b))
        sh = NativeSession(_lib.load(), "vrt_", cfg_s)
        orc.setup(sh, mat, rgb, params)
        for n in calls:
            sh.accumulate(n)
        parts[a:b] = sh.fetch_hdr()[a:b]
        sh.close()
    shards_ok = np.array_equal(full.view(np.uint32), parts.view(np.uint32))
    ok = ok and shards_ok
    print(f"case {k}: {scene} {W}x{H} depth {depth} seed {seed} calls {calls} shards {cuts}: {'ok' if ok else 'MISMATCH'}{'' if shards_ok else ' (shards)'}", flush=True)
    bad += 0 if ok else 1
    g.close(); o.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
