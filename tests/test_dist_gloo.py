"""The N > 1 path on CPU: world_size-2 (and 3) gloo process groups shard the frame by rows, each
rank renders ITS rows (here with the CPU oracle standing in for a GPU context -- the sharding
contract, halo handling and gather plumbing are the same code bench.py uses), rank 0 gathers and
compares with the unsharded frame bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 96, 50, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, restir, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import orc
    from voxel_rt2_amd import host, scenes, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mat, rgb, params = scenes.scene_sunlit(0)
    rows = parallel.split_rows(H, world)[rank]
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=5, seed=4,
                           use_restir=restir, rows=rows)
    o = orc.Oracle(cfg, threads=1)
    orc.setup(o, mat, rgb, params)
    o.accumulate(SPP)
    hdr = o.fetch_hdr()
    tile = torch.zeros((parallel.max_tile_rows(H, world), W, 3), dtype=torch.float32)
    tile[: rows[1] - rows[0]] = torch.from_numpy(hdr[rows[0]:rows[1]])
    frame = parallel.gather_frame(tile, H, W, rank, world)
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,restir", [(2, False), (3, False), (2, True)])
def test_sharded_frame_equals_full_frame(tmp_path, world, restir):
    import orc
    from voxel_rt2_amd import host, scenes
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), restir, out), nprocs=world, join=True)
    got = np.load(out)
    mat, rgb, params = scenes.scene_sunlit(0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=5, seed=4, use_restir=restir)
    o = orc.Oracle(cfg, threads=2)
    orc.setup(o, mat, rgb, params)
    o.accumulate(SPP)
    ref = o.fetch_hdr()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_rebalance_rows_equalises_a_step_time_with_a_fixed_part():
    """bench.py's balancing passes (ShardedRun): every rank reports what a step costs on its rows, the boundaries move, again.
    A cost model like the measured one -- a per-row cost that varies fivefold down the frame (S1's sky rows against its horizon
    rows) plus a fixed part per step that no row count removes -- must come out within a few per cent after bench.py's five passes, with the
    tiles still covering the frame once and none below the minimum."""
    from voxel_rt2_amd import parallel
    H, fixed = 1080, 0.03
    row_cost = np.where(np.arange(H) > 700, 0.0002, 0.001) + 0.0004 * np.exp(-((np.arange(H) - 520) / 60.0) ** 2)
    for world in (2, 3, 4, 8):
        bounds = parallel.split_rows(H, world)
        cost = lambda b: [fixed + float(row_cost[a:e].sum()) for a, e in b]
        first = max(cost(bounds)) / min(cost(bounds))
        for _ in range(5):
            bounds = parallel.rebalance_rows(bounds, cost(bounds), H)
            assert bounds[0][0] == 0 and bounds[-1][1] == H and all(bounds[i][1] == bounds[i + 1][0] for i in range(world - 1))
            assert all(e - a >= 8 for a, e in bounds)
        c = cost(bounds)
        assert max(c) / min(c) < min(1.08, first), (world, bounds, c)
    assert parallel.rebalance_rows([(0, 1080)], [1.0], 1080) == [(0, 1080)]
    # ranks that report nothing useful (zero cost) still get their minimum of rows
    b = parallel.rebalance_rows(parallel.split_rows(64, 4), [0.0, 0.0, 1.0, 0.0], 64)
    assert b[0][0] == 0 and b[-1][1] == 64 and all(e - a >= 8 for a, e in b)


def test_split_rows_covers_frame():
    from voxel_rt2_amd import parallel
    for h in (1080, 2160, 50, 7):
        for n in (1, 2, 3, 4, 8):
            parts = parallel.split_rows(h, n)
            assert parts[0][0] == 0 and parts[-1][1] == h
            assert all(parts[i][1] == parts[i + 1][0] for i in range(n - 1))
            assert max(b - a for a, b in parts) - min(b - a for a, b in parts) <= 1
    assert parallel.split_rows(1080, 8)[3] == (405, 540)  # 135-row tiles at 1080p


# ---- sky precompute split by table columns ---------------------------------------------------------------------
SKY_R = 32


def _sky_worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import orc
    from voxel_rt2_amd import _abi, host, scenes, parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mat, rgb, params = scenes.scene_s6(0)
    cfg = host.make_config(48, 32, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=3, seed=4, sky_res=SKY_R)
    o = orc.Oracle(cfg, threads=1)
    orc.setup(o, mat, rgb, params, cloud=_cloud_tile())
    parallel.precompute_sky_sharded(o, rank, world, device="cpu", cloud_passes=3, cloud_samples=3, atmosphere_slices=4)
    if rank == world - 1:   # any rank holds the full tables afterwards
        np.save(out_path, np.stack([o.fetch_buffer(_abi.BUF_SKY_SCATTERING), o.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE)]))
    dist.barrier()
    dist.destroy_process_group()


def _cloud_tile():
    x = np.arange(256)
    t = (np.sin(x[:, None] * 0.11) * np.cos(x[None, :] * 0.07) * 0.5 + 0.5) * 255
    return np.stack([t, t.T, np.full_like(t, 230)], axis=-1).astype(np.uint8)


@pytest.mark.parametrize("world", [2, 4])
def test_sky_precompute_sharded_by_columns_equals_unsharded(tmp_path, world):
    import orc
    from voxel_rt2_amd import _abi, host, scenes
    out = str(tmp_path / "sky.npy")
    mp.spawn(_sky_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    mat, rgb, params = scenes.scene_s6(0)
    cfg = host.make_config(48, 32, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=3, seed=4, sky_res=SKY_R)
    o = orc.Oracle(cfg, threads=2)
    orc.setup(o, mat, rgb, params, cloud=_cloud_tile())
    for _ in range(3):
        o.sky_accumulate_clouds(3)
    for sl in range(4):
        o.sky_compute_slice(sl, 4)
    ref = np.stack([o.fetch_buffer(_abi.BUF_SKY_SCATTERING), o.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE)])
    assert np.isfinite(ref).all() and ref[0].max() > 0 and ref[1].max() > 0
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
