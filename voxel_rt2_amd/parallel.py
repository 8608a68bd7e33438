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
