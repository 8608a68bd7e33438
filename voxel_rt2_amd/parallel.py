"""Row-tile sharding of the HDR frame across ranks and the gather to rank 0.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the MI355X node; "gloo"
in the CPU tests).  Pixels are independent given the replicated scene and the per-pixel random
streams, so a rank renders its contiguous rows (plus a small halo it renders redundantly) with no
data-path communication; the only exchange is one gather of the finished tiles per presented frame:
at 1080p each of 7 peers sends 135 x 1920 x 12 B = 3.1 MB straight to rank 0 over its own xGMI link.
"""
import os

import numpy as np

# CUs' worth of workgroup slots a rank's persistent render grid leaves free when the process group has more than one rank,
# so that RCCL's gather kernel can be scheduled while a render launch is resident (include/vrt_api.h: vrt_reserve_cus).
# One GPU's cost of the reservation on config 2: DESIGN.md section 8.
DEFAULT_RESERVED_CUS = 8


def reserved_cus(world_size):
    if world_size <= 1:
        return 0
    return int(os.environ.get("VRT_RESERVE_CUS", DEFAULT_RESERVED_CUS))


def configure_session(sess, world_size):
    """What a rank's render context needs beyond a single-GPU one."""
    sess.reserve_cus(reserved_cus(world_size))


def split_rows(height, world_size):
    """Contiguous row ranges, one per rank, covering [0, height)."""
    edges = [(height * i) // world_size for i in range(world_size + 1)]
    return [(edges[i], edges[i + 1]) for i in range(world_size)]


def max_tile_rows(height, world_size):
    return max(b - a for a, b in split_rows(height, world_size))


def gather_frame(tile, height, width, rank, world_size, dst=0, out_list=None):
    """Gather per-rank tiles (torch tensors [max_tile_rows, width, 3], rows past the tile unused) to
    `dst` and return the assembled [height, width, 3] frame there (None elsewhere)."""
    import torch
    import torch.distributed as dist
    if world_size == 1:
        return tile[:height]
    if rank == dst and out_list is None:
        out_list = [torch.zeros_like(tile) for _ in range(world_size)]
    dist.gather(tile, out_list if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    frame = torch.empty((height, width, 3), dtype=tile.dtype, device=tile.device)
    for r, (a, b) in enumerate(split_rows(height, world_size)):
        frame[a:b] = out_list[r][: b - a]
    return frame


def rebalance_rows(bounds, costs, height, min_rows=8):
    """New contiguous row ranges with (approximately) equal cost, given the cost each rank measured on its current
    range (piecewise-constant cost density per row).  Rows are the unit; every rank keeps at least `min_rows`."""
    n = len(bounds)
    if n == 1:
        return [(0, height)]
    dens = np.zeros(height, dtype=np.float64)
    for (a, b), c in zip(bounds, costs):
        dens[a:b] = max(float(c), 1e-9) / max(b - a, 1)
    cum = np.concatenate([[0.0], np.cumsum(dens)])
    edges = [0]
    for k in range(1, n):
        target = cum[-1] * k / n
        e = int(np.searchsorted(cum, target))
        e = max(e, edges[-1] + min_rows)
        e = min(e, height - (n - k) * min_rows)
        edges.append(e)
    edges.append(height)
    return [(edges[i], edges[i + 1]) for i in range(n)]


# ---- sky precompute, columns split over the ranks ------------------------------------------------------
# Scene.finish() runs 32 cloud passes over the whole 3840^2 table and then the atmosphere pass in 32 column slices
# (reference scene.py:243-253, atmos.py:140-189): 6.4 s on one MI355X.  A texel of either pass depends on no other texel, so
# rank r computes the columns of slice r and the ranks all-gather the two tables (2 x 177 MB / world per rank).

def sky_columns(sky_res, rank, world_size):
    if sky_res % world_size:
        raise ValueError(f"sky_res {sky_res} is not a multiple of the world size {world_size}")
    w = sky_res // world_size
    return rank * w, (rank + 1) * w


def precompute_sky_columns(sess, rank, world_size, *, cloud_passes=32, cloud_samples=32, atmosphere_slices=32):
    """This rank's share of the precompute: its columns of every cloud pass, then of the atmosphere pass."""
    sky_columns(sess.cfg.sky_res, rank, world_size)
    for _ in range(cloud_passes):
        sess.sky_accumulate_clouds_slice(cloud_samples, rank, world_size)
    sub = max(1, atmosphere_slices // world_size)          # keep the reference's slice granularity where it divides
    if sess.cfg.sky_res % (world_size * sub):
        sub = 1
    for k in range(sub):
        sess.sky_compute_slice(rank * sub + k, world_size * sub)


def exchange_sky_columns(sess, rank, world_size, device):
    """All-gather both tables: every rank ends with the full tables in its context."""
    import torch
    import torch.distributed as dist
    from . import _abi
    R = sess.cfg.sky_res
    u0, u1 = sky_columns(R, rank, world_size)
    for which in (_abi.BUF_SKY_SCATTERING, _abi.BUF_SKY_TRANSMITTANCE):
        full = torch.empty((world_size, u1 - u0, R, 3), dtype=torch.float32, device=device)
        sess.sky_table_io(which, u0, u1, full[rank].data_ptr(), False)
        sess.sync()
        if world_size > 1:
            dist.all_gather(list(full.unbind(0)), full[rank].clone())
            if full.is_cuda:
                torch.cuda.current_stream().synchronize()
        for r in range(world_size):
            if r != rank:
                sess.sky_table_io(which, r * (u1 - u0), (r + 1) * (u1 - u0), full[r].data_ptr(), True)
        sess.sync()


def precompute_sky_sharded(sess, rank, world_size, device="cuda", **kw):
    precompute_sky_columns(sess, rank, world_size, **kw)
    exchange_sky_columns(sess, rank, world_size, device)
