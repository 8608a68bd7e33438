#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz with the CPU oracle (oracle/ -- the restatement of the reference path).

These are NOT outputs of the reference (those are tests/golden/reference/, written by make_reference_vectors.py from the
reference's own source at sizes a Python loop can finish).  They freeze what the oracle computes today for a few small,
fixed cases, so that (a) a change to the oracle that alters results is noticed on the CPU and (b) the GPU box
can check libvrt_hip.so against committed data even without rebuilding the oracle (tests/test_golden.py).

    python tests/golden/make_golden.py        # from the repo root; overwrites the fixtures
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.dirname(HERE)):
    sys.path.insert(0, p)
import numpy as np
import orc
from voxel_rt2_amd import _abi, camera, host, scenes

BUFS = {"gbuf_depth": _abi.BUF_GBUF_DEPTH, "gbuf_normal": _abi.BUF_GBUF_NORMAL, "gbuf_mat": _abi.BUF_GBUF_MAT,
        "gbuf_position": _abi.BUF_GBUF_POSITION, "gbuf_refl_depth": _abi.BUF_GBUF_REFL_DEPTH,
        "history_diffuse": _abi.BUF_HISTORY_DIFFUSE, "history_specular": _abi.BUF_HISTORY_SPECULAR}

# name -> (scene, scene seed, W, H, max depth, rng seed, ReSTIR, script); script = list of steps
CASES = {
    "s1_64x64_d4": ("s1", 0, 64, 64, 4, 0, False, [("accumulate", 1)]),                       # BASELINE config 1 in small
    "s1_96x56_d8_fused": ("s1", 0, 96, 56, 8, 5, False, [("accumulate", 4), ("accumulate", 3)]),
    "sunlit_80x48_d6": ("sunlit", 0, 80, 48, 6, 11, False, [("accumulate", 2), ("accumulate", 4)]),
    "dense_48x40_d5": ("dense", 12345, 48, 40, 5, 3, False, [("accumulate", 3)]),
    "sunlit_restir_64x40_d4": ("sunlit", 0, 64, 40, 4, 7, True, [("accumulate", 2)]),
    "sunlit_moving_72x44_d4": ("sunlit", 0, 72, 44, 4, 9, False, [("accumulate", 2), ("end_frame",), ("move", 0.45), ("accumulate", 1),
                                                                   ("end_frame",), ("move", 0.5), ("accumulate", 1)]),
    "sunlit_restir_moving_56x40_d4": ("sunlit", 0, 56, 40, 4, 17, True, [("accumulate", 2), ("end_frame",), ("move", 0.44), ("accumulate", 1),
                                                                        ("end_frame",), ("still", 3), ("accumulate", 2)]),
    # two static frames with different jitter, then a quarter-frame pass: the pixels it leaves out must keep the
    # g-buffer of the LAST pass (the reference's g-buffer is one array), then static again
    "s1_jitter_then_moving_64x48_d5": ("s1", 0, 64, 48, 5, 13, False, [("accumulate", 4), ("end_frame",), ("still", 2), ("accumulate", 3),
                                                                       ("end_frame",), ("move", 0.42), ("accumulate", 1), ("end_frame",),
                                                                       ("still", 4), ("accumulate", 4)]),
}


SKY_BUFS = {"sky_scattering": _abi.BUF_SKY_SCATTERING, "sky_transmittance": _abi.BUF_SKY_TRANSMITTANCE, "trans_lut": _abi.BUF_TRANS_LUT}


def synthetic_sky(R, seed):
    """Two smooth positive tables standing in for the skybox (only the LOOKUP is under test: atmos.py:94-131)."""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.arange(R) / R, np.arange(R) / R, indexing="ij")
    def table(lo, hi):
        t = np.zeros((R, R, 3))
        for c in range(3):
            for _ in range(4):
                fu, fv, ph = rng.integers(1, 5), rng.integers(1, 5), rng.uniform(0, 6.28, 2)
                t[..., c] += rng.uniform(0.2, 1.0) * np.sin(6.2831853 * fu * u + ph[0]) * np.cos(6.2831853 * fv * v + ph[1])
        t = (t - t.min()) / (t.max() - t.min())
        return np.ascontiguousarray((lo + (hi - lo) * t).astype(np.float32))
    return table(0.0, 2.5), table(0.0, 1.0)


def upload_sky_tables(session, scat, trans):
    if hasattr(session, "upload_sky"):
        session.upload_sky(scat, trans)
        return
    import torch   # the HIP library: device memory through vrt_sky_table_io
    for which, t in ((_abi.BUF_SKY_SCATTERING, scat), (_abi.BUF_SKY_TRANSMITTANCE, trans)):
        d = torch.from_numpy(t).cuda()
        session.sky_table_io(which, 0, t.shape[0], d.data_ptr(), True)
        session.sync()
        torch.cuda.synchronize()


def run_case(session, case):
    """Drives any session object (oracle, emulation, GPU) through a case's script.  A ninth element turns the physical sky and its
    clouds on: (table size, cloud passes, slices) computes the tables before the first step, as Scene.finish() does;
    ("given", table size, seed) uploads synthetic tables instead (the lookup alone)."""
    scene, scene_seed, W, H, depth, seed, restir, script = case[:8]
    sky = case[8] if len(case) > 8 else None
    mat, rgb, params = scenes.SCENES[scene](scene_seed)
    params = dict(params, use_physical_sky=int(bool(sky)), use_clouds=int(bool(sky)))
    cloud = np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy")) if sky else None
    orc.setup(session, mat, rgb, params, cloud=cloud)
    if sky and sky[0] == "given":
        upload_sky_tables(session, *synthetic_sky(sky[1], sky[2]))
    elif sky:
        for _ in range(sky[1]):
            session.sky_accumulate_clouds(sky[1])
        for s in range(sky[2]):
            session.sky_compute_slice(s, sky[2])
    k = 1
    for step in script:
        if step[0] == "accumulate":
            session.accumulate(step[1])
        elif step[0] == "end_frame":
            session.end_frame()
        elif step[0] == "still":
            session.set_camera(host.default_camera(W, H, jitter_index=step[1]))
        elif step[0] == "move":
            pos = (step[1], 0.5, 2.0)
            view, proj = camera.default_matrices(W, H, pos=pos)
            k += 1
            session.set_camera(host.make_camera(view, proj, pos, jitter_index=k, moving=True, render_scale=0.5, max_accum_frames=50.0))
    out = {"hdr": session.fetch_hdr(), "ldr": session.fetch_ldr()}
    for name, which in BUFS.items():
        out[name] = session.fetch_buffer(which)
    if sky and sky[0] != "given":
        for name, which in SKY_BUFS.items():
            out[name] = session.fetch_buffer(which)
    return out


def config_of(case):
    scene, scene_seed, W, H, depth, seed, restir, script = case[:8]
    _, _, params = scenes.SCENES[scene](scene_seed)
    return host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=seed, use_restir=restir,
                            sky_res=(case[8][1] if case[8][0] == "given" else case[8][0]) if len(case) > 8 else 0)


if __name__ == "__main__":
    for name, case in CASES.items():
        o = orc.Oracle(config_of(case))
        out = run_case(o, case)
        o.close()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: (v.shape, str(v.dtype)) for k, v in out.items()}, os.path.getsize(os.path.join(HERE, name + ".npz")), "bytes")
