#!/usr/bin/env python3
"""Benchmark of the hot path: Mpath-samples/s of Renderer.accumulate() on MI355X.

Workload (BASELINE.json configs[1]): example-1-style scene S1 on the 128^3 grid, 1920x1080,
4 spp per frame, 8 bounces.  One "step" = one frame = 4 accumulate() passes over every pixel;
a path-sample = one pixel x one pass.  All inputs are resident in HBM
before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU: the frame is
split into N contiguous row tiles (strong scaling: the frame is fixed), every rank renders its
tile with no data-path communication, and the HDR tiles are gathered to rank 0 over RCCL at the
end of every step (inside the timed region).

Rank 0 prints ONE JSON line with the throughput, the roofline of the dominant kernel (k_render_pool)
and the CPU-oracle baseline timed on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

# The library renders on two streams of its own beside the caller's; a multi-GPU rank adds the gather stream and RCCL's.
# The HIP runtime multiplexes streams onto 4 hardware queues by default, and two streams that share a queue serialise: a
# wait queued for the gather then holds back the next render launch (measured on one rank's share of an 8-way split:
# 0.418 ms per step with 4 queues, 0.317 with 8).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

WIDTH, HEIGHT, SPP_PER_STEP, MAX_DEPTH, SEED = 1920, 1080, 4, 8, 0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def split_rows(height, n):
    edges = [(height * i) // n for i in range(n + 1)]
    return [(edges[i], edges[i + 1]) for i in range(n)]


def setup_session(sess, mat, rgb, params):
    from voxel_rt2_amd import host, materials
    sess.upload_voxels(mat, rgb)
    sess.upload_materials(materials.load_table())
    sess.set_scene(host.make_scene_params(**params))
    sess.set_camera(host.default_camera(sess.W, sess.H, jitter_index=1))
    sess.prepare()


def cpu_baseline(mat, rgb, params):
    """The CPU oracle (oracle/, a restatement of the reference's ti.cpu path) on a bounded sample
    of the same workload: a central band of rows of the same frame, same depth / spp / seed."""
    import orc
    from voxel_rt2_amd import host
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    rows = (0, HEIGHT)
    cfg = host.make_config(WIDTH, HEIGHT, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=MAX_DEPTH,
                           seed=SEED)
    o = orc.Oracle(cfg, threads=cores)
    setup_session(o, mat, rgb, params)
    o.accumulate(1)  # warm-up pass (page-in, thread start), also calibrates the sample size
    t0 = time.perf_counter()
    o.accumulate(1)
    one = time.perf_counter() - t0
    passes = int(min(64, max(2, round(12.0 / max(one, 1e-3)))))  # aim at ~12 s of wall time
    t0 = time.perf_counter()
    o.accumulate(passes)
    dt = time.perf_counter() - t0
    samples = WIDTH * HEIGHT * passes
    hdr = o.fetch_hdr()
    o.close()
    return dict(value=samples / dt / 1e6, unit="Mpath-samples/s", cores=cores, kind="port",
                sample=f"the whole {WIDTH}x{HEIGHT} frame, {passes} accumulate passes, {MAX_DEPTH} bounces, same scene and seed "
                       f"({samples} path-samples in {dt:.1f} s on {cores} threads; oracle/ = CPU restatement of the "
                       f"reference's ti.cpu path, reported only)"), hdr, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from voxel_rt2_amd import host, scenes, _lib
    from voxel_rt2_amd._session import NativeSession

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the renderer has no CPU path")
    # VRT_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a one-GPU box -- every rank on device 0, collectives over
    # gloo with host staging instead of RCCL.  Exercises the sharding, rebalancing, gather and reporting logic; its
    # numbers mean nothing.
    rehearse = os.environ.get("VRT_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if rehearse else "cuda"

    from voxel_rt2_amd import parallel
    mat, rgb, params = scenes.scene_s1(0)
    lib = _lib.load()
    stream = torch.cuda.Stream()  # the context's stream: temporal passes and the tile copy (render launches go to the library's own streams)

    def make_session(rows):
        cfg = host.make_config(WIDTH, HEIGHT, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=MAX_DEPTH,
                               seed=SEED, device=local_rank, rows=rows if world > 1 else None)
        s = NativeSession(lib, "vrt_", cfg)
        s.set_stream(stream.cuda_stream)
        setup_session(s, mat, rgb, params)
        return s

    # Row tiles: start from an equal split, then (untimed) let every rank measure its tile's device time and move the
    # tile boundaries so that all ranks carry the same cost -- the sky rows of this scene cost a fraction of the floor rows.
    bounds = split_rows(HEIGHT, world)
    sess = make_session(bounds[rank])
    user_overlap = os.environ.get("VRT_OVERLAP")
    os.environ["VRT_OVERLAP"] = "0"  # isolated launches while measuring: an overlapped launch's span includes its neighbours
    for _ in range(2 if world > 1 else 0):
        sess.accumulate(SPP_PER_STEP)
        lib.vrt_reset_stats(C.c_void_p(sess._ctx))
        sess.accumulate(SPP_PER_STEP)
        st0 = sess.stats()
        mine = torch.tensor([st0["render_ms"] + st0["temporal_ms"]], dtype=torch.float64, device=coll_dev)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        bounds = parallel.rebalance_rows(bounds, [float(t.item()) for t in allc], HEIGHT)
        sess.close()
        sess = make_session(bounds[rank])
    if user_overlap is None:
        os.environ.pop("VRT_OVERLAP", None)
    else:
        os.environ["VRT_OVERLAP"] = user_overlap
    rows = bounds[rank]

    # gather plumbing: equal-sized tiles (padded to the tallest tile), torch owns the staging tensors.  The gather runs on
    # its own stream behind an event, from one of three staging tiles, so that the next step's temporal pass (same stream as
    # the tile copy) does not queue behind a collective that waits for the slowest rank.
    max_rows = max(b - a for a, b in bounds)
    n_tiles = 3
    tiles = [torch.zeros((max_rows, WIDTH, 3), dtype=torch.float32, device="cuda") for _ in range(n_tiles)]
    gathered = [torch.zeros_like(tiles[0]) for _ in range(world)] if (world > 1 and rank == 0) else None
    host_gathered = [torch.zeros_like(tiles[0], device="cpu") for _ in range(world)] if (rehearse and gathered is not None) else None
    gather_stream = torch.cuda.Stream() if world > 1 else None
    tile_free = [None] * n_tiles   # recorded on gather_stream when the gather that read tiles[j] is done
    step_no = [0]

    def step():
        j = step_no[0] % n_tiles
        step_no[0] += 1
        with torch.cuda.stream(stream):
            sess.accumulate(SPP_PER_STEP)
            if world > 1:
                if tile_free[j] is not None:
                    stream.wait_event(tile_free[j])
                sess.fetch_hdr_device_async(tiles[j].data_ptr())  # D2D, queued behind the kernels
                ready = stream.record_event()
        if world > 1:
            with torch.cuda.stream(gather_stream):
                gather_stream.wait_event(ready)
                if rehearse:
                    dist.gather(tiles[j].cpu(), host_gathered, dst=0)
                    if rank == 0:
                        for g, h in zip(gathered, host_gathered):
                            g.copy_(h)
                else:
                    dist.gather(tiles[j], gathered, dst=0)
                tile_free[j] = gather_stream.record_event()

    def fence():
        stream.synchronize()
        if gather_stream is not None:
            gather_stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, not warm-up: the library allocates the buffers of its overlapped pipeline (second and third buffer set, the
    # camera-ray tables of both render streams) on the first launches that use them; keep that out of a short --warmup
    for _ in range(4):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    lib.vrt_reset_stats(C.c_void_p(sess._ctx))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = sess.stats()

    if rehearse and world > 1 and rank == 0:
        # rehearsal only: the frame assembled from the gathered tiles equals an unsharded render of the same passes
        full = np.concatenate([gathered[r][: bounds[r][1] - bounds[r][0]].cpu().numpy() for r in range(world)], axis=0)
        cfg1 = host.make_config(WIDTH, HEIGHT, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=MAX_DEPTH,
                                seed=SEED, device=local_rank)
        one = NativeSession(lib, "vrt_", cfg1)
        setup_session(one, mat, rgb, params)
        for _ in range(step_no[0]):   # every step() so far: setup, warm-up and timed
            one.accumulate(SPP_PER_STEP)
        ref = one.fetch_hdr()
        one.close()
        same = np.array_equal(full.view(np.uint32), ref.view(np.uint32))
        print(f"[rehearsal] gathered frame == unsharded frame: {same}", file=sys.stderr, flush=True)
        if not same:
            raise SystemExit("rehearsal: gathered frame differs from the unsharded render")

    # algorithmic bytes of the dominant kernel: one untimed instrumented pass counts what a path does
    lib.vrt_set_instrumented(C.c_void_p(sess._ctx), 1)
    lib.vrt_reset_stats(C.c_void_p(sess._ctx))
    sess.accumulate(SPP_PER_STEP)
    ist = sess.stats()
    lib.vrt_set_instrumented(C.c_void_p(sess._ctx), 0)

    if rank == 0:
        total_samples = WIDTH * HEIGHT * SPP_PER_STEP * args.steps
        value = total_samples / elapsed / 1e6
        own_px = WIDTH * (rows[1] - rows[0])
        # per path-sample counts on this rank's tile (SURVEY.md section 8d formula, render-kernel share):
        # 4 B per occupancy query + (4 B texel + 56 B material row) per closest hit + 96 B per sky lookup
        # + 52 B written per path (two colour values and the g-buffer)
        n_inst = max(ist["path_samples"], 1)
        q, hc, ls = ist["occupancy_queries"] / n_inst, ist["closest_hits"] / n_inst, ist["sky_lookups"] / n_inst
        bytes_per_sample = 4.0 * q + 60.0 * hc + 96.0 * ls + 52.0
        launches = max(st["render_launches"], 1)
        avg_ms = st["render_ms"] / launches
        samples_per_launch = st["path_samples"] / launches  # a launch renders own_px pixels x the fused samples
        achieved = bytes_per_sample * samples_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        valu_insts = None
        # the render stage runs the pooled schedule (k_render_pool) unless VRT_RENDER=fused asks for the fused one
        render_kernel = "k_render" if os.environ.get("VRT_RENDER") == "fused" else "k_render_pool"
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile) and world == 1:  # measured on the one-GPU launch (whole frame); a tile's launch moves less
            try:
                tj = json.load(open(tfile))
                traffic = tj.get(f"{render_kernel}_bytes_per_launch")
                valu_insts = tj.get(f"{render_kernel}_valu_wave_insts_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mpath-samples/sec at 1920x1080, 8 bounces", "value": round(value, 3), "unit": "Mpath-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "example1-style scene S1 (scenes.scene_s1), 128^3 grid, 1920x1080, 4 spp/step, 8 bounces, "
                                   "static camera, ReSTIR off", "width": WIDTH, "height": HEIGHT, "spp_per_step": SPP_PER_STEP,
                       "max_depth": MAX_DEPTH, "seed": SEED, "sharding": (f"{world} contiguous row tiles, boundaries balanced by measured tile cost, RCCL gather per step; "
                                    f"tile rows {[b - a for a, b in bounds]}") if world > 1 else "none"},
            "roofline": {"bound": "hbm", "kernel": render_kernel, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "algorithmic_bytes_per_path_sample": round(bytes_per_sample, 2),
                         "queries_per_path_sample": round(q, 2), "closest_hits_per_path_sample": round(hc, 3),
                         "path_samples_per_launch": int(samples_per_launch), "render_ms_per_launch": round(avg_ms, 4),
                         "temporal_ms_per_launch": round(st["temporal_ms"] / max(st["temporal_launches"], 1), 4),
                         # what actually bounds the kernel: wave-level VALU instructions per launch (PMC, profiles/traffic.json)
                         # over the live launch duration, against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction
                         "valu_issue": ({"achieved": round(valu_insts / (avg_ms * 1e-3) / 1e9, 2), "peak": 614.4, "unit": "G wave-instructions/s",
                                         "frac": round(valu_insts / (avg_ms * 1e-3) / 614.4e9, 4)} if (valu_insts and avg_ms > 0) else None),
                         "note": "HBM is the bound the tier names; the 128^3 working set is cache resident and the kernel is "
                                 "VALU-issue bound (profiles/r01_v10_pmc_k_render_pool.txt). Launches overlap: "
                                 "the duration is a launch's span, the step period is ms_per_step"},
        }
        if not args.no_cpu_baseline and world == 1:
            cb, ref_rows, rr = cpu_baseline(mat, rgb, params)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sess.close()


if __name__ == "__main__":
    main()
