#!/usr/bin/env python3
"""Randomised multi-step scenarios (static / moving camera, jitter changes, resets, light and floor changes, ReSTIR on or
off, fused calls) run through the CPU oracle and a second implementation, all buffers compared bit for bit after
every accumulate.  Second implementation: the emulated device code (default, no GPU needed) or libvrt_hip.so (--gpu).
A tool for hunting parity gaps, not part of the test-suite.  usage: tools/soak_scenarios.py [--gpu] [n_cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
import orc
from voxel_rt2_amd import _abi, camera, host, scenes

args = [a for a in sys.argv[1:] if a != "--gpu"]
use_gpu = "--gpu" in sys.argv
n_cases = int(args[0]) if args else 10
rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 1)
BUFS = (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_POSITION, _abi.BUF_GBUF_MAT, _abi.BUF_GBUF_REFL_DEPTH,
        _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR)

def second(cfg):
    if use_gpu:
        from voxel_rt2_amd import _lib
        from voxel_rt2_amd._session import NativeSession
        return NativeSession(_lib.load(), "vrt_", cfg)
    import emu
    return emu.Emulated(cfg)

SCALE = int(os.environ.get("SOAK_SCALE", "1"))   # frame sizes times this (bigger frames: more tiles, XCD bands, work ranges)
bad = 0
for k in range(n_cases):
    scene = ["s1", "sunlit", "dense", "s6"][int(rng.integers(0, 4))]
    W, H = int(rng.integers(24, 120)) * SCALE, int(rng.integers(16, 90)) * SCALE
    depth, seed, restir = int(rng.integers(1, 7)), int(rng.integers(0, 1 << 30)), bool(rng.integers(0, 3) == 0)
    mat, rgb, params = scenes.SCENES[scene](int(rng.integers(0, 4)))
    sky = scene == "s6" and bool(rng.integers(0, 2))  # the physical sky with clouds, tables at 32 x 32 (precomputed by both sides)
    if not sky: params = dict(params, use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=seed, use_restir=restir,
                           sky_res=32 if sky else 0)
    script, jitter = [], 1
    for _ in range(int(rng.integers(2, 8))):
        r = int(rng.integers(0, 10))
        if r < 4: script.append(("acc", int(rng.integers(1, 7))))
        elif r < 6:
            jitter += 1; script.append(("still", jitter, [3.0, 8.0, 999999999.0][int(rng.integers(0, 3))])); script.append(("acc", int(rng.integers(1, 6))))
        elif r < 8:
            jitter += 1; script.append(("move", float(rng.uniform(0.3, 0.5)), jitter, [0.5, 0.5, 0.75, 1.0][int(rng.integers(0, 4))])); script.append(("acc", 1))
        elif r == 8: script.append(("reset",) if rng.integers(0, 2) else ("voxels", int(rng.integers(0, 6))))
        else:
            c = int(rng.integers(0, 3))
            if c == 0: script.append(("scene", dict(light_color=[float(x) for x in rng.uniform(0, 2, 3)] if rng.integers(0, 2) else [0.0, 0.0, 0.0])))
            elif c == 1: script.append(("scene", dict(floor_height=float(rng.uniform(-0.9, 0.1)), floor_color=[float(x) for x in rng.uniform(0, 1, 3)],
                                                      floor_material=int([1, 1, 2, 10][int(rng.integers(0, 4))]))))
            else: script.append(("scene", dict(background_color=[float(x) for x in rng.uniform(0, 1, 3)],
                                               light_direction=[float(x) for x in rng.uniform(-1, 1, 3)], light_cone=float(rng.uniform(0.01, 0.6)))))
    o, e = orc.Oracle(cfg, threads=4), second(cfg)
    cloud = np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy")) if sky else None
    for s in (o, e):
        orc.setup(s, mat, rgb, params, cloud=cloud if (use_gpu or s is o) else None)
        if sky and (use_gpu or s is o):
            s.sky_accumulate_clouds(1)
            for sl in range(2): s.sky_compute_slice(sl, 2)
    if sky and not use_gpu:  # the emulation has no precompute kernels: it is handed the oracle's tables
        e.upload_sky(o.fetch_buffer(_abi.BUF_SKY_SCATTERING), o.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE))
    ok, where = True, None
    for i, step in enumerate(script):
        for s in (o, e):
            if step[0] == "acc": s.accumulate(step[1])
            elif step[0] == "still": s.end_frame(); s.set_camera(host.default_camera(W, H, jitter_index=step[1], max_accum_frames=float(step[2])))
            elif step[0] == "move":
                pos = (step[1], 0.5, 2.0); view, proj = camera.default_matrices(W, H, pos=pos)
                s.end_frame(); s.set_camera(host.make_camera(view, proj, pos, jitter_index=step[2], moving=True, render_scale=step[3], max_accum_frames=50.0))
            elif step[0] == "reset": s.reset()
            elif step[0] == "scene":
                if s is o: params = dict(params, **step[1])  # (o comes first: both sessions see the updated dict)
                s.set_scene(host.make_scene_params(**params))
            elif step[0] == "voxels":
                m2, r2, _ = scenes.SCENES[scene](step[1]); s.upload_voxels(m2, r2); s.prepare()
                if sky and not use_gpu and s is e:  # prepare() restarts the sky tables (pathtracer.py:322-323): hand the emulation the oracle's
                    e.upload_sky(o.fetch_buffer(_abi.BUF_SKY_SCATTERING), o.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE))
        if step[0] == "acc":
            same = np.array_equal(o.fetch_hdr().view(np.uint32), e.fetch_hdr().view(np.uint32))
            diff = [] if same else ["hdr"]
            for which in BUFS:
                if not np.array_equal(np.ascontiguousarray(o.fetch_buffer(which)).view(np.uint8), np.ascontiguousarray(e.fetch_buffer(which)).view(np.uint8)): diff.append(which)
            if diff and ok: ok, where = False, (i, step, diff)
    print(f"case {k}: {scene}{'+sky' if sky else ''} {W}x{H} depth {depth} seed {seed} restir {restir} script {script}: {'ok' if ok else 'MISMATCH at step %s' % (where,)}", flush=True)
    bad += 0 if ok else 1
    o.close(); e.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
