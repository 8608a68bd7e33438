"""The launch pipeline around the render kernels on the GPU: overlapped launches (fused and one-sample ones), the dispatch
gate and its host release, error roll-back, frames presented asynchronously, HDR tiles written by the temporal pass, the
workgroup-slot reservation of multi-GPU runs, and RCCL beside the library's own streams."""
import ctypes as C
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

import orc
from voxel_rt2_amd import _lib, host, scenes
from voxel_rt2_amd._session import NativeSession

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, DEPTH = 640, 360, 6


def render(calls=(4, 4, 4, 3, 4), reserve=None, dev=False):
    """dev: through build_variants/libvrt_dev.so, the build that reads the development switches (VRT_STREAMS, VRT_DRAIN_GATE,
    VRT_TEST_FAIL_LAUNCH, ...: csrc/vrt_api.hip read_knobs) -- the shipped library ignores them."""
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=5)
    s = NativeSession(_lib.load_dev() if dev else _lib.load(), "vrt_", cfg)
    if reserve is not None:
        s.reserve_cus(reserve)
    orc.setup(s, mat, rgb, params)
    for n in calls:
        s.accumulate(n)
    hdr, st = s.fetch_hdr(), s.stats()
    s.close()
    return hdr, st


@pytest.fixture(scope="module")
def reference_frame():
    os.environ["VRT_OVERLAP"] = "0"
    try:
        hdr, st = render()
    finally:
        os.environ.pop("VRT_OVERLAP", None)
    assert st["pipeline_flags"] & 1 == 0
    return hdr


def test_overlapped_launches_with_gate(reference_frame, monkeypatch):
    hdr, st = render()
    assert st["pipeline_flags"] & 3 == 3, "overlapped launches with the dispatch gate are the default on a plain GPU box"
    assert (st["pipeline_flags"] >> 8) & 0xFFFF <= 1   # only the release at context teardown may have happened
    assert np.array_equal(hdr.view(np.uint32), reference_frame.view(np.uint32))


def test_gate_left_out(reference_frame, monkeypatch):
    monkeypatch.setenv("VRT_DRAIN_GATE", "0")
    hdr, st = render()
    assert st["pipeline_flags"] & 3 == 3, "the shipped library does not read development switches"
    hdr, st = render(dev=True)
    assert st["pipeline_flags"] & 3 == 1
    assert np.array_equal(hdr.view(np.uint32), reference_frame.view(np.uint32))


def test_host_release_of_the_gate_changes_nothing(reference_frame, monkeypatch):
    """The watchdog that releases a stuck dispatch gate from the host, made to fire on every synchronisation: the gate
    only times dispatches (ordering is by events), so results are the same."""
    monkeypatch.setenv("VRT_GATE_WATCHDOG_MS", "0.0001")
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=5)
    s = NativeSession(_lib.load(), "vrt_", cfg)
    orc.setup(s, mat, rgb, params)
    for n in (4, 4, 4, 3, 4):
        s.accumulate(n)
        s.sync()
    hdr, st = s.fetch_hdr(), s.stats()
    s.close()
    assert (st["pipeline_flags"] >> 8) & 0xFFFF >= 2
    assert np.array_equal(hdr.view(np.uint32), reference_frame.view(np.uint32))


def test_pipeline_depths_over_many_launches(monkeypatch):
    """The depths of the launch pipeline -- two launches of every workgroup slot in flight, four launches of half the slots
    each (the default for frames up to 1080p), eight of a quarter each (the smallest frames, e.g. a rank's rows of an 8-GPU
    split) -- over enough launches that every rotating resource comes round more than once: up to 9 copies of the sample
    planes, 10 of the g-buffer normal / depth, 16 sets of work heads."""
    calls = (4, 4, 3, 4, 2, 1, 1) * 4 + (4,) * 3
    monkeypatch.setenv("VRT_OVERLAP", "0")
    ref, st = render(calls)
    assert st["pipeline_flags"] & 1 == 0
    monkeypatch.delenv("VRT_OVERLAP")
    for streams, div in (("4", "2"), ("2", "1"), ("4", "3"), ("8", "4"), ("8", "2")):
        monkeypatch.setenv("VRT_STREAMS", streams)
        monkeypatch.setenv("VRT_GRID_DIV", div)
        hdr, st = render(calls, dev=True)
        assert st["pipeline_flags"] & 1 == 1
        assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32)), (streams, div)
    monkeypatch.delenv("VRT_STREAMS")
    monkeypatch.delenv("VRT_GRID_DIV")
    monkeypatch.setenv("VRT_DEEP_ITEMS", "0")   # the policy's other side: this frame is "large"
    hdr, _ = render(calls, dev=True)
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))


def test_restir_samples_fused_in_the_render_launch(monkeypatch):
    """With ReSTIR on the render launch carries a call's samples (one reservoir plane each) and spatial reuse / accumulation
    run per sample over the planes: same HDR, g-buffer and histories as one launch per sample (VRT_FUSE_RESTIR=0), also when
    a pass that renders part of the frame (moving camera) follows and reads the last sample's reservoirs."""
    from voxel_rt2_amd import _abi, camera
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=9, use_restir=True)

    def run():
        s = NativeSession(_lib.load_dev(), "vrt_", cfg)     # (VRT_FUSE_RESTIR is a development switch)
        orc.setup(s, mat, rgb, params)
        for n in (4, 3):
            s.accumulate(n)
        s.end_frame()
        pos = (0.46, 0.5, 2.0)
        view, proj = camera.default_matrices(W, H, pos=pos)
        s.set_camera(host.make_camera(view, proj, pos, jitter_index=2, moving=True, render_scale=0.5, max_accum_frames=50.0))
        s.accumulate(1)
        s.end_frame()
        s.set_camera(host.make_camera(view, proj, pos, jitter_index=3))
        s.accumulate(4)
        out = [s.fetch_hdr()] + [s.fetch_buffer(b) for b in (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_MAT,
                                                            _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR)]
        s.close()
        return out

    monkeypatch.setenv("VRT_FUSE_RESTIR", "0")
    ref = run()
    monkeypatch.delenv("VRT_FUSE_RESTIR")
    got = run()
    assert np.isfinite(ref[0]).all() and ref[0].mean() > 0.01
    for a, b in zip(ref, got):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))


def test_reserved_workgroup_slots_change_nothing(reference_frame):
    hdr, _ = render(reserve=8)
    assert np.array_equal(hdr.view(np.uint32), reference_frame.view(np.uint32))
    hdr, _ = render(reserve=200)
    assert np.array_equal(hdr.view(np.uint32), reference_frame.view(np.uint32))


def test_camera_ray_records_under_contention():
    """Fused samples share camera-ray records across CUs (vrt_pool.h).  Short work ranges make readers race writers: a
    135-row shard (one GPU of eight at 1080p) and a 16-row one, each against the same rows of the unsharded frame."""
    mat, rgb, params = scenes.scene_s1(0)
    Wf, Hf = 1920, 1080

    def run(rows):
        cfg = host.make_config(Wf, Hf, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0, rows=rows)
        s = NativeSession(_lib.load(), "vrt_", cfg)
        orc.setup(s, mat, rgb, params)
        for _ in range(3):
            s.accumulate(4)
        out = s.fetch_hdr()
        s.close()
        return out

    full = run(None)
    for rows in ((405, 540), (536, 552), (0, 8)):
        part = run(rows)
        assert np.array_equal(part[rows[0]:rows[1]].view(np.uint32), full[rows[0]:rows[1]].view(np.uint32)), rows


def test_accumulate_failure_rolls_back(monkeypatch):
    """A launch that fails to queue (injected: VRT_TEST_FAIL_LAUNCH) returns an error, leaves nobody waiting at the dispatch
    gate, and the context renders on afterwards -- with the result of a context that never saw the failure."""
    lib = _lib.load_dev()     # the fault-injection hook is compiled into the development build only
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(320, 200, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=4, seed=2)
    good = NativeSession(lib, "vrt_", cfg)
    orc.setup(good, mat, rgb, params)
    for n in (4, 4, 4):
        good.accumulate(n)
    want = good.fetch_hdr()
    good.close()
    monkeypatch.setenv("VRT_TEST_FAIL_LAUNCH", "2")   # (read when the context is created)
    s = NativeSession(lib, "vrt_", cfg)
    monkeypatch.delenv("VRT_TEST_FAIL_LAUNCH")
    orc.setup(s, mat, rgb, params)
    s.accumulate(4)
    s.accumulate(4)
    assert lib.vrt_accumulate(C.c_void_p(s._ctx), 4) == -2 and b"injected" in lib.vrt_last_error()
    assert lib.vrt_accumulate(C.c_void_p(s._ctx), -1) == -1
    s.accumulate(4)
    assert (s.stats()["pipeline_flags"] >> 8) & 0xFFFF >= 1   # the failed launch's gate was released from the host
    assert np.array_equal(s.fetch_hdr().view(np.uint32), want.view(np.uint32))
    s.close()


def _session(cfg, mat, rgb, params, cam=None):
    s = NativeSession(_lib.load(), "vrt_", cfg)
    orc.setup(s, mat, rgb, params, cam=cam)
    return s


def test_pipeline_depth_follows_the_launch_size(monkeypatch):
    """A caller that changes the sample count of its calls: the depth of the pipeline follows the size of each launch (the
    pipeline is drained once per change), whatever the first call was.  Development build with the size limits moved so that
    this small frame has launches of all three kinds: one sample -> eight launches of a quarter of the slots, two -> four of a
    half, three or four -> two of every slot."""
    calls = (1, 4, 4, 2, 1, 1, 3, 4, 2, 2, 1) * 2
    monkeypatch.setenv("VRT_OVERLAP", "0")
    ref, st = render(calls)
    assert st["pipeline_flags"] & 1 == 0
    monkeypatch.delenv("VRT_OVERLAP")
    monkeypatch.setenv("VRT_DEEPER_ITEMS", str(W * H + 1))
    monkeypatch.setenv("VRT_DEEP_ITEMS", str(2 * W * H + 1))
    hdr, st = render(calls, dev=True)
    switches = st["pipeline_flags"] >> 24
    assert st["pipeline_flags"] & 1 == 1 and switches >= 12, st
    assert (st["pipeline_flags"] >> 2) & 7 == 4 and (st["pipeline_flags"] >> 5) & 7 == 4, "the last call was one sample: eight launches of a quarter"
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))


def test_pipeline_depth_of_the_shipped_library_at_900p(monkeypatch):
    """The shipped library's own limits: at 1600x900 a one-sample call is a small launch (1.4 M items: eight deep), a four-sample
    call a middle one (5.8 M, above the 4.5 M limit: four deep).  (1, 4, 4, 1, 1, 4): three changes of depth, the frame that of
    isolated launches."""
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(1600, 900, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=5)
    frames = []
    for overlap in ("0", None):
        if overlap is not None:
            monkeypatch.setenv("VRT_OVERLAP", overlap)
        else:
            monkeypatch.delenv("VRT_OVERLAP")
        s = _session(cfg, mat, rgb, params)
        for n in (1, 4, 4, 1, 1, 4):
            s.accumulate(n)
        frames.append((s.fetch_hdr(), s.stats()))
        s.close()
    (ref, st0), (hdr, st) = frames
    assert st0["pipeline_flags"] & 1 == 0 and st["pipeline_flags"] & 1 == 1
    if os.environ.get("GPU_MAX_HW_QUEUES") == "16":   # (the eight-deep pipeline needs sixteen hardware queues: _lib.load asks for them)
        assert st["pipeline_flags"] >> 24 == 3, st
        assert (st["pipeline_flags"] >> 2) & 7 == 2, "the last call was four samples: four launches of half the slots"
    assert np.array_equal(hdr.view(np.uint32), ref.view(np.uint32))


def test_stats_count_every_launch_and_scale_the_timed_ones():
    """Launches of the deep pipelines carry their timers one time in eight (include/vrt_api.h, vrt_stats): the counts are of ALL
    launches and passes, the times the timed launches' mean times those counts -- after a reset the first launch is a timed one,
    so a window of fewer than eight launches still has a time."""
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=5)
    s = _session(cfg, mat, rgb, params)
    for _ in range(20):
        s.accumulate(4)
    st = s.stats()
    assert st["render_launches"] == 20 and st["temporal_launches"] == 20 and st["path_samples"] == 20 * 4 * W * H, st
    per_launch = st["render_ms"] / st["render_launches"]
    assert 0.01 < per_launch < 5.0 and st["temporal_ms"] > 0.0, st
    _lib.load().vrt_reset_stats(C.c_void_p(s._ctx))
    for _ in range(3):
        s.accumulate(4)
    st = s.stats()
    assert st["render_launches"] == 3 and 0.01 < st["render_ms"] / 3 < 5.0, st
    assert 0.3 < (st["render_ms"] / 3) / per_launch < 3.0, (st, per_launch)   # the same kind of launch: the same time, roughly
    s.close()


def test_single_sample_calls_with_a_new_jitter_each():
    """The reference's own loop shape (scene.py:177, 233-262: samples_per_frame = 1, set_proj_mat draws a new jitter every frame,
    accumulate, copy_prev_matrices): twelve one-sample calls, a new jitter index before each, pipelined like fused launches
    (each renders into plane 0 of a rotating copy, eight copies: the rotation comes round).  Against the oracle."""
    Wq, Hq = 320, 180
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(Wq, Hq, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=11)
    g, o = _session(cfg, mat, rgb, params), orc.Oracle(cfg, threads=16)
    orc.setup(o, mat, rgb, params)
    for k in range(12):
        cam = host.default_camera(Wq, Hq, jitter_index=k + 1)
        for s in (g, o):
            s.set_camera(cam)
            s.accumulate(1)
            s.end_frame()
    st = g.stats()
    assert st["pipeline_flags"] & 1 and st["render_launches"] == 12, st   # one-sample launches overlap too
    a, b = g.fetch_hdr(), o.fetch_hdr()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} of {a.size} values differ"
    from voxel_rt2_amd import _abi
    for which in (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_MAT, _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR):
        assert np.array_equal(g.fetch_buffer(which).view(np.uint8), o.fetch_buffer(which).view(np.uint8)), which
    g.close(); o.close()


def test_frames_presented_asynchronously():
    """vrt_fetch_ldr_async / vrt_fetch_hdr_async into page-locked memory while further frames are queued: every presented
    frame equals the blocking fetch of a second context driven the same way (the reference presents every frame,
    scene.py:255-262)."""
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=8)
    a, b = _session(cfg, mat, rgb, params), _session(cfg, mat, rgb, params)
    ldr = [a.host_alloc((H, W, 4)) for _ in range(2)]
    hdr = [a.host_alloc((H, W, 3)) for _ in range(2)]
    ldr8 = a.host_alloc((H, W, 4), np.uint8)
    shown, shown8 = [], []
    for k in range(7):
        a.accumulate(4)
        a.fetch_ldr_async(ldr[k % 2], slot=k % 2)
        a.fetch_hdr_async(hdr[k % 2], slot=2 + k % 2)
        if k >= 1:   # collect the frame before while this one renders
            a.fetch_wait((k - 1) % 2); a.fetch_wait(2 + (k - 1) % 2)
            shown.append((ldr[(k - 1) % 2].copy(), hdr[(k - 1) % 2].copy()))
    a.fetch_wait(6 % 2); a.fetch_wait(2 + 6 % 2)
    shown.append((ldr[0].copy(), hdr[0].copy()))
    a.fetch_ldr8_async(ldr8, slot=0)      # the 8-bit image of the last frame: what scene.save_image makes of the f32 one
    a.fetch_wait(0)
    for k in range(7):
        b.accumulate(4)
        want_l, want_h = b.fetch_ldr(), b.fetch_hdr()
        assert np.array_equal(shown[k][0].view(np.uint32), want_l.view(np.uint32)), k
        assert np.array_equal(shown[k][1].view(np.uint32), want_h.view(np.uint32)), k
    assert np.array_equal(ldr8, (np.clip(want_l, 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))
    a.close(); b.close()


def test_hdr_tiles_written_by_the_temporal_pass():
    """vrt_set_hdr_targets: the pass that completes an accumulate call writes the context's HDR rows into the caller's ring of
    device tiles; vrt_hdr_targets_written says how many are queued.  A 90-row shard, ten calls, ring of eight: tile k equals
    the rows a second context fetches after its k-th call."""
    import torch
    mat, rgb, params = scenes.scene_sunlit(0)
    rows = (120, 210)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=DEPTH, seed=8, rows=rows)
    a, b = _session(cfg, mat, rgb, params), _session(cfg, mat, rgb, params)
    ring = [torch.zeros((rows[1] - rows[0], W, 3), dtype=torch.float32, device="cuda") for _ in range(8)]
    a.set_hdr_targets([t.data_ptr() for t in ring])
    got, seen = [], 0
    for k in range(10):
        a.accumulate(4 if k % 3 else 2)
        n = a.hdr_targets_written()
        assert n == k + 1                        # the call queues its own tile
        if n > seen and k % 4 == 3:
            a.sync()                             # (a gather would wait on an event of the context's stream instead)
            for j in range(seen, a.hdr_targets_written()):
                got.append(ring[j % 8].cpu().numpy().copy())
            seen = a.hdr_targets_written()
    a.sync()
    for j in range(seen, a.hdr_targets_written()):
        got.append(ring[j % 8].cpu().numpy().copy())
    assert len(got) == 10
    for k in range(10):
        b.accumulate(4 if k % 3 else 2)
        want = b.fetch_hdr()[rows[0]:rows[1]]
        assert np.array_equal(got[k].view(np.uint32), want.view(np.uint32)), k
    a.close(); b.close()


def test_rccl_beside_the_library():
    """backend='nccl' (RCCL) at world size 1 in a process that has loaded libvrt_hip.so FIRST and rendered: one HIP runtime
    shared with torch (_lib._share_hip_runtime_with_torch), RCCL's streams beside the library's, a gather + all_reduce between
    render launches, and the frame unchanged by it."""
    code = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
        import numpy as np
        from voxel_rt2_amd import _lib, host, scenes, parallel, materials
        from voxel_rt2_amd._session import NativeSession
        lib = _lib.load()
        import torch, torch.distributed as dist
        mat, rgb, params = scenes.scene_sunlit(0)
        cfg = host.make_config(480, 270, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=5, seed=3)
        def session():
            s = NativeSession(lib, "vrt_", cfg)
            s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
            s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(480, 270, jitter_index=1)); s.prepare()
            return s
        a = session()
        for _ in range(3): a.accumulate(4)
        want = a.fetch_hdr(); a.close()
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%%d" %% int(sys.argv[1]), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        b = session()
        stream = torch.cuda.Stream()
        b.set_stream(stream.cuda_stream)
        parallel.configure_session(b, 2)   # as a rank of a larger group would
        ring = [torch.zeros((270, 480, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
        b.set_hdr_targets([t.data_ptr() for t in ring])   # the temporal pass writes the tile the gather reads (bench.py's hand-over)
        got = [torch.zeros_like(ring[0])]
        for k in range(3):
            with torch.cuda.stream(stream):
                b.accumulate(4)
                assert b.hdr_targets_written() == k + 1
                dist.gather(ring[k & 1], got, dst=0)
                t = torch.ones(8, device="cuda"); dist.all_reduce(t)
        stream.synchronize(); torch.cuda.synchronize()
        same = bool(np.array_equal(got[0].cpu().numpy().view(np.uint32), want.view(np.uint32)))
        print(json.dumps(dict(same=same, backend=dist.get_backend(), queues=os.environ.get("GPU_MAX_HW_QUEUES"))))
        dist.destroy_process_group(); b.close()
    """) % (ROOT, ROOT)
    import socket
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GPU_MAX_HW_QUEUES")}
    r = subprocess.run([sys.executable, "-c", code, str(port)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out == dict(same=True, backend="nccl", queues="16")


def test_sky_precompute_split_by_columns_on_the_gpu():
    """The sharded sky precompute (parallel.precompute_sky_columns + vrt_sky_table_io) with three "ranks" played by three
    contexts on the one GPU, columns exchanged through device tensors: every context ends with the tables of an unsharded
    precompute, bit for bit (the collective itself is covered on CPU by tests/test_dist_gloo.py)."""
    import torch
    from voxel_rt2_amd import _abi, parallel
    R, world = 96, 3
    mat, rgb, params = scenes.scene_s6(0)
    cloud = np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy"))
    cfg = host.make_config(64, 40, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=3, seed=4, sky_res=R)

    def context():
        s = NativeSession(_lib.load(), "vrt_", cfg)
        orc.setup(s, mat, rgb, params, cloud=cloud)
        return s

    ref = context()
    for _ in range(4):
        ref.sky_accumulate_clouds(4)
    for sl in range(6):
        ref.sky_compute_slice(sl, 6)
    want = [ref.fetch_buffer(_abi.BUF_SKY_SCATTERING), ref.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE)]
    assert np.isfinite(want[0]).all() and want[0].max() > 0
    ranks = [context() for _ in range(world)]
    for r, s in enumerate(ranks):
        parallel.precompute_sky_columns(s, r, world, cloud_passes=4, cloud_samples=4, atmosphere_slices=6)
    w = R // world
    for k, which in enumerate((_abi.BUF_SKY_SCATTERING, _abi.BUF_SKY_TRANSMITTANCE)):
        full = torch.empty((world, w, R, 3), dtype=torch.float32, device="cuda")
        for r, s in enumerate(ranks):   # "all-gather"
            s.sky_table_io(which, r * w, (r + 1) * w, full[r].data_ptr(), False)
            s.sync()
        for r, s in enumerate(ranks):
            for q in range(world):
                if q != r:
                    s.sky_table_io(which, q * w, (q + 1) * w, full[q].data_ptr(), True)
            s.sync()
            assert np.array_equal(s.fetch_buffer(which).view(np.uint32), want[k].view(np.uint32)), (which, r)
    # and a frame rendered from the exchanged tables equals one rendered from the unsharded ones
    for s in (ref, ranks[1]):
        s.accumulate(2)
    assert np.array_equal(ref.fetch_hdr().view(np.uint32), ranks[1].fetch_hdr().view(np.uint32))
