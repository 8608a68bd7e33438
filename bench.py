#!/usr/bin/env python3
"""Benchmark of the hot path: Mpath-samples/s of Renderer.accumulate() on MI355X.

Headline workload (BASELINE.json configs[1]): example-1-style scene S1 on the 128^3 grid, 1920x1080,
4 spp per frame, 8 bounces.  One "step" = one frame = 4 accumulate() passes over every pixel;
a path-sample = one pixel x one pass.  All inputs are resident in HBM
before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-secondary] [--no-cpu-baseline]

N > 1 is launched by the driver through torch.distributed.run, one rank per GPU: the frame is
split into N contiguous row tiles (strong scaling: the frame is fixed), every rank renders its
tile with no data-path communication, and every step's HDR tiles are gathered to rank 0 over RCCL
(inside the timed region).

Rank 0 prints ONE JSON line: the headline throughput, the roofline of its dominant kernel
(k_render_pool), the CPU-oracle baseline timed on this box's host cores, a `secondary` list and, as the
LAST key, a compact `summary` of every workload's value (a record that keeps only the end of the line keeps these):
on one GPU the other BASELINE configs in their one-GPU form (config 3: sky + clouds + ReSTIR at
1080p; config 4: dense 128^3 at 3840x2160; config 5: dense 256^3 at 3840x2160), each with its own
dominant kernel, duration, algorithmic bytes and roofline fraction, plus the reference's own loop
shape (one sample per call, `scene_api_default`); on N GPUs configs 4 and 5 as BASELINE.json
defines them -- rows split N ways, the gather inside the timed region.
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

# More hardware queues than the HIP runtime's default 4 (voxel_rt2_amd/_lib.py explains; it sets the same default when
# the library is loaded first).  Must be in the environment before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

WIDTH, HEIGHT, SPP_PER_STEP, MAX_DEPTH, SEED = 1920, 1080, 4, 8, 0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue: a wave64 vector instruction takes 2 cycles on a SIMD-32 (MI355X_MICROARCH.md:54, 473): 256 CUs x 4 SIMDs x
# 2.4 GHz / 2.  Measured on this chip with an independent v_fma_f32 stream (tools/probes/valu_issue.cpp,
# profiles/r02_valu_issue.txt): 836 G wave-instructions/s with 4 waves per SIMD, 778 with the 2 waves per SIMD the
# render kernels run at (2.75 cycles per instruction; the chip clocks down to 1.7-2.1 GHz under that load).
VALU_PEAK_NOMINAL, VALU_PEAK_MEASURED_2WAVES = 1228.8, 778.0
# reference-algorithm bytes of one spatial-reuse pass per pixel: 32 taps x ~75 B of g-buffer and
# reservoir, 55 B reservoir written, 24 B of colour written
GRIS_BYTES_PER_PIXEL = 32 * 75 + 55 + 24


def split_rows(height, n):
    edges = [(height * i) // n for i in range(n + 1)]
    return [(edges[i], edges[i + 1]) for i in range(n)]


def setup_session(sess, mat, rgb, params, cloud=None):
    from voxel_rt2_amd import host, materials
    sess.upload_voxels(mat, rgb)
    sess.upload_materials(materials.load_table())
    if cloud is not None:
        sess.upload_cloud_texture(cloud)
    sess.set_scene(host.make_scene_params(**params))
    sess.set_camera(host.default_camera(sess.W, sess.H, jitter_index=1))
    sess.prepare()


def cpu_baseline(mat, rgb, params):
    """The CPU oracle (oracle/, a restatement of the reference's ti.cpu path) on a bounded sample
    of the same workload: the same frame, depth, spp and seed, a few accumulate passes."""
    import orc
    from voxel_rt2_amd import host
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
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
    o.close()
    return dict(value=samples / dt / 1e6, unit="Mpath-samples/s", cores=cores, kind="port",
                sample=f"the whole {WIDTH}x{HEIGHT} frame, {passes} accumulate passes, {MAX_DEPTH} bounces, same scene and seed "
                       f"({samples} path-samples in {dt:.1f} s on {cores} threads; oracle/ = CPU restatement of the "
                       f"reference's ti.cpu path, reported only)")


def measured_counters(lib):
    """profiles/traffic.json: HBM bytes and VALU instructions per launch from rocprofv3 --pmc passes -- only if they were
    taken on the build that is loaded now (vrt_build_id), else nothing."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tfile))
    except Exception:
        return {}, None
    have = lib.vrt_build_id().decode()
    if tj.get("build_id") != have:
        return {}, f"profiles/traffic.json was measured on build {tj.get('build_id')}, loaded library is {have}: not reported"
    return tj.get("kernels", {}), None


def count_work(lib, sess, spp):
    """Per path-sample counts from instrumented launches: the reference algorithm's (every camera ray walked, what the
    oracle counts too) and the timed schedule's own (fused samples share their camera rays)."""
    out = {}
    for mode, key in ((1, "reference"), (2, "timed")):
        lib.vrt_set_instrumented(C.c_void_p(sess._ctx), mode)
        lib.vrt_reset_stats(C.c_void_p(sess._ctx))
        sess.accumulate(spp)
        ist = sess.stats()
        n = max(ist["path_samples"], 1)
        out[key] = dict(q=ist["occupancy_queries"] / n, hc=ist["closest_hits"] / n, ls=ist["sky_lookups"] / n,
                        rays=ist["rays"] / n, iters=ist["dda_iters"] / n)
    lib.vrt_set_instrumented(C.c_void_p(sess._ctx), 0)
    return out


def render_bytes(c):
    # SURVEY.md section 8d, render-kernel share: 4 B per occupancy query + (4 B texel + 56 B material row) per closest hit
    # + 96 B per sky lookup + 52 B written per path (two colour values and the g-buffer)
    return 4.0 * c["q"] + 60.0 * c["hc"] + 96.0 * c["ls"] + 52.0


def roofline_block(kernel, kernel_ms, units_per_launch, bytes_ref, bytes_timed, counters, notes, period_ms=None):
    """HBM roofline of one kernel: algorithmic bytes per launch over its measured duration (`achieved`, `frac`: what the rocprofv3
    kernel trace shows per launch).  Launches of this kernel overlap (up to four in flight, half of the workgroup
    slots each), so the chip's rate is a launch's bytes over the PERIOD between launches: `all_launches_in_flight`."""
    ach = bytes_timed * units_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    ach_ref = bytes_ref * units_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    k = counters.get(kernel, {}) if counters else {}
    traffic = k.get("bytes_per_launch")
    valu = k.get("valu_wave_insts_per_launch")
    blk = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 6), "traffic": traffic,
           "algorithmic_bytes_per_unit": round(bytes_timed, 2),
           "reference_algorithm": {"bytes_per_unit": round(bytes_ref, 2), "achieved": round(ach_ref, 3), "frac": round(ach_ref / HBM_PEAK_GBS, 6)},
           "units_per_launch": int(units_per_launch), "kernel_ms_per_launch": round(kernel_ms, 4)}
    if traffic and kernel_ms > 0:
        blk["traffic_GBps"] = round(traffic / (kernel_ms * 1e-3) / 1e9, 1)
        blk["traffic_frac"] = round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
    if valu and kernel_ms > 0:
        g = valu / (kernel_ms * 1e-3) / 1e9
        blk["valu_issue"] = {"achieved": round(g, 1), "unit": "G wave-instructions/s", "peak_nominal": VALU_PEAK_NOMINAL,
                             "frac_nominal": round(g / VALU_PEAK_NOMINAL, 4), "peak_measured_2_waves_per_simd": VALU_PEAK_MEASURED_2WAVES,
                             "frac_measured": round(g / VALU_PEAK_MEASURED_2WAVES, 4)}
    if period_ms and kernel_ms > 0:
        k_in = kernel_ms / period_ms
        blk["all_launches_in_flight"] = {"launch_period_ms": round(period_ms, 4), "launches_in_flight": round(k_in, 2),
                                         "achieved": round(ach * k_in, 3), "frac": round(ach * k_in / HBM_PEAK_GBS, 6)}
        if "valu_issue" in blk:
            g = blk["valu_issue"]["achieved"] * k_in
            blk["all_launches_in_flight"]["valu_issue"] = {"achieved": round(g, 1), "frac_nominal": round(g / VALU_PEAK_NOMINAL, 4),
                                                           "frac_measured": round(g / VALU_PEAK_MEASURED_2WAVES, 4)}
    blk["note"] = notes
    return blk


SECONDARY = [
    dict(config="3", name="config3_s6_sky_clouds_restir_1080p", scene="s6", W=1920, H=1080, depth=8, spp=4, steps=8, restir=True, sky_res=3840,
         workload="example6-style scene S6 (scenes.scene_s6), physical sky + clouds (3840^2 tables precomputed untimed), ReSTIR spatial reuse on, "
                  "128^3 grid, 1920x1080, 4 spp/step, 8 bounces"),
    dict(config="4 (one GPU's form)", name="config4_dense_4k", scene="dense", W=3840, H=2160, depth=8, spp=4, steps=8,
         workload="dense random 128^3 fill (p = 0.5, seed 12345), 3840x2160, 4 spp/step (16 spp = 4 steps), 8 bounces, the whole frame on one GPU"),
    dict(config="5 (one GPU's form)", name="config5_dense256_4k", scene="dense256", W=3840, H=2160, depth=8, spp=4, steps=8, grid=256,
         workload="dense random 256^3 fill (p = 0.5, seed 12345; 64 MiB of texels, 2 MiB of fine brick words), 3840x2160, 4 spp/step "
                  "(32 spp = 8 steps), 8 bounces, the whole frame on one GPU"),
]


def run_secondary(lib, case, counters):
    from voxel_rt2_amd import host, scenes
    from voxel_rt2_amd._session import NativeSession
    mat, rgb, params = scenes.SCENES[case["scene"]](12345 if case["scene"].startswith("dense") else 0)
    sky_res = case.get("sky_res", 0)
    if not sky_res:
        params = dict(params, use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(case["W"], case["H"], voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=case["depth"],
                           seed=SEED, use_restir=case.get("restir", False), sky_res=sky_res, grid_res=case.get("grid", 128))
    s = NativeSession(lib, "vrt_", cfg)
    cloud = np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy")) if sky_res else None
    setup_session(s, mat, rgb, params, cloud)
    if sky_res:  # Scene.finish()'s precompute (scene.py:243-253), untimed
        for _ in range(32):
            s.sky_accumulate_clouds(32)
        for sl in range(32):
            s.sky_compute_slice(sl, 32)
    spp, steps = case["spp"], case["steps"]
    for _ in range(4):   # set-up of the overlapped pipeline + warm-up
        s.accumulate(spp)
    s.sync()
    lib.vrt_reset_stats(C.c_void_p(s._ctx))
    t0 = time.perf_counter()
    for _ in range(steps):
        s.accumulate(spp)
    s.sync()
    dt = time.perf_counter() - t0
    st = s.stats()
    work = count_work(lib, s, spp)
    hdr_ok = bool(np.isfinite(s.fetch_hdr()).all())
    s.close()
    px = case["W"] * case["H"]
    ms = {k: st[f"{k}_ms"] / max(st[f"{k}_launches"], 1) for k in ("render", "gris", "temporal")}
    per_launch = st["path_samples"] / max(st["render_launches"], 1)
    if case.get("restir"):
        # per sample: k_render<restir> + k_gris_prepare + k_gris_classify + k_gris (two kernels) + k_temporal; the spatial-reuse pass dominates
        kernel, kms, units = "k_gris", ms["gris"], px
        b_ref = b_timed = float(GRIS_BYTES_PER_PIXEL)
        note = ("dominant kernel of this config: the ReSTIR spatial-reuse pass (k_gris_prepare + k_gris_classify + k_gris<.,.,1> + k_gris<.,.,2>, timed together); unit = one pixel of "
                "one pass; bytes = the reference algorithm's reads/writes per pixel (SURVEY.md 8 a13); the pass is VALU bound (DESIGN.md 4)")
    else:
        kernel, kms, units = "k_render_pool", ms["render"], per_launch
        b_ref, b_timed = render_bytes(work["reference"]), render_bytes(work["timed"])
        note = "unit = one path-sample; launches overlap, so the kernel duration is a launch's span and the step period is ms_per_step"
    period = dt / steps * 1e3 / (spp if case.get("restir") else max(st["render_launches"], 1) / steps)
    value = px * spp * steps / dt / 1e6
    return {"config": case["config"], "name": case["name"], "workload": case["workload"], "metric": "Mpath-samples/sec", "value": round(value, 2),
            "unit": "Mpath-samples/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "finite": hdr_ok,
            "rays_per_s": rays_block(value, work),
            "kernel_ms_per_launch": {k: round(v, 4) for k, v in ms.items()},
            "work_per_path_sample": {k: {a: round(b, 3) for a, b in v.items()} for k, v in work.items()},
            "roofline": roofline_block(kernel, kms, units, b_ref, b_timed, counters.get(case["name"], {}), note,
                                       None if case.get("restir") else period)}


class ShardedRun:
    """One workload on `world` ranks: contiguous row tiles (optionally balanced by measured cost), per step one vrt_accumulate
    and the RCCL gather of the HDR tiles to rank 0.  The tile is written by the temporal pass itself into a ring of device
    tensors (vrt_set_hdr_targets: no device-to-device copy behind the pass); the gather of a step is issued, on a stream of
    its own behind an event, as soon as vrt_hdr_targets_written says its tile is queued -- so that the next step's temporal
    pass does not queue behind a collective that waits for the slowest rank.  finish() (inside the timed region) completes
    every gather on every rank.  world == 1: no tiles, no gather."""

    N_TILES = 8   # deeper than the library's launch pipeline (eight launches in flight on a shard of an eighth of 1080p)

    def __init__(self, lib, dist, torch, *, scene, W, H, depth, spp, grid=128, rank=0, world=1, local_rank=0, rehearse=False, balance=False, stripes=0):
        from voxel_rt2_amd import host, parallel
        from voxel_rt2_amd._session import NativeSession
        self.lib, self.dist, self.torch, self.parallel = lib, dist, torch, parallel
        self.W, self.H, self.spp, self.rank, self.world, self.rehearse = W, H, spp, rank, world, rehearse
        self.mat, self.rgb, self.params = scene
        self.stream = torch.cuda.Stream()   # the context's stream: temporal passes (the render launches go to the library's own)
        self.coll_dev = "cpu" if rehearse else "cuda"

        self.stripes = stripes if world > 1 else 0   # rows per interleaved stripe (vrt_set_row_stripes) instead of contiguous tiles

        def make_session(rows):
            cfg = host.make_config(W, H, voxel_edges=self.params["voxel_edges"], exposure=self.params["exposure"], max_depth=depth,
                                   seed=SEED, device=local_rank, rows=rows if (world > 1 and not self.stripes) else None, grid_res=grid)
            s = NativeSession(lib, "vrt_", cfg)
            if self.stripes and rows is not None:
                s.set_row_stripes(self.stripes, world, rank)
            s.set_stream(self.stream.cuda_stream)
            parallel.configure_session(s, world)   # N > 1: leave workgroup slots free for RCCL's kernels
            setup_session(s, self.mat, self.rgb, self.params)
            return s

        self.make_full_session = lambda: make_session(None)
        # Row tiles: start from an equal split, then (untimed) let every rank measure what a step costs on its tile and move the
        # tile boundaries so that all ranks carry the same cost -- the sky rows of S1 cost a fraction of the floor rows.  The cost
        # is the step time with the launch pipeline in flight, as the timed steps run (a tile's isolated kernel time undervalues
        # cheap rows: 365 sky rows "cost" what 80 floor rows do in isolation and a third more in the pipeline); five passes (a step has
        # a part that no row count removes, which the per-row model of rebalance_rows takes four or five passes to work around),
        # each on the boundaries of the one before (tools/predict_scaling.py replays this on one GPU).
        self.bounds = split_rows(H, world)
        if balance and world > 1 and not self.stripes:
            for _ in range(5):
                sess = make_session(self.bounds[rank])
                for _w in range(8):
                    sess.accumulate(spp)
                sess.sync()
                t0 = time.perf_counter()
                for _w in range(24):
                    sess.accumulate(spp)
                sess.sync()
                mine = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.coll_dev)
                allc = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allc, mine)
                self.bounds = parallel.rebalance_rows(self.bounds, [float(t.item()) for t in allc], H)
                sess.close()
        self.sess = make_session(self.bounds[rank])
        self.steps_done = 0
        self.tiles_gathered = 0
        # rows of the frame every rank produces, in the order its tile holds them
        self.rank_rows = [np.arange(a, b) for a, b in self.bounds]
        if self.stripes:
            period = self.stripes * world
            self.rank_rows = [np.concatenate([np.arange(a, min(a + self.stripes, H)) for a in range(r * self.stripes, H, period)]) for r in range(world)]
        if world > 1:
            max_rows = max(len(r) for r in self.rank_rows)
            # every rank's tile is padded to the tallest: the library writes the rank's own rows at the top of its tile
            self.tiles = [torch.zeros((max_rows, W, 3), dtype=torch.float32, device="cuda") for _ in range(self.N_TILES)]
            self.gathered = [torch.zeros_like(self.tiles[0]) for _ in range(world)] if rank == 0 else None
            self.host_gathered = [torch.zeros_like(self.tiles[0], device="cpu") for _ in range(world)] if (rehearse and rank == 0) else None
            self.gather_stream = torch.cuda.Stream()
            self.tile_free = [None] * self.N_TILES   # recorded on gather_stream when the gather that read tiles[j] is done
            self.sess.set_hdr_targets([t.data_ptr() for t in self.tiles])

    def _gather_ready_tiles(self):
        torch, dist = self.torch, self.dist
        n = self.sess.hdr_targets_written()
        if n == self.tiles_gathered:
            return
        ready = self.stream.record_event()   # behind the passes that write the tiles
        with torch.cuda.stream(self.gather_stream):
            self.gather_stream.wait_event(ready)
            while self.tiles_gathered < n:
                j = self.tiles_gathered % self.N_TILES
                if self.rehearse:
                    dist.gather(self.tiles[j].cpu(), self.host_gathered, dst=0)
                    if self.rank == 0:
                        for g, h in zip(self.gathered, self.host_gathered):
                            g.copy_(h)
                else:
                    dist.gather(self.tiles[j], self.gathered, dst=0)
                self.tile_free[j] = self.gather_stream.record_event()
                self.tiles_gathered += 1

    def step(self):
        torch = self.torch
        with torch.cuda.stream(self.stream):
            if self.world > 1:
                j = self.steps_done % self.N_TILES   # the tile this step's pass will write, whenever it is queued
                if self.tile_free[j] is not None:
                    self.stream.wait_event(self.tile_free[j])
            self.sess.accumulate(self.spp)
        self.steps_done += 1
        if self.world > 1:
            self._gather_ready_tiles()

    def finish(self):
        """Everything queued so far completes: the last passes, their tiles' gathers, all ranks."""
        torch = self.torch
        if self.world > 1:
            self.sess.sync()
            self._gather_ready_tiles()
            assert self.tiles_gathered == self.steps_done
            self.gather_stream.synchronize()
        else:
            self.sess.sync()
        torch.cuda.synchronize()
        if self.world > 1:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, warmup):
        for _ in range(4):   # set-up, not warm-up: the library allocates its pipeline's buffers on the first launches
            self.step()
        self.finish()
        for _ in range(warmup):
            self.step()
        self.finish()
        self.lib.vrt_reset_stats(C.c_void_p(self.sess._ctx))
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.finish()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.coll_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, self.sess.stats()

    def check_against_unsharded(self, label):
        """Rehearsal only (rank 0): the frame assembled from the LAST gathered tiles equals an unsharded render of the same steps."""
        full = np.zeros((self.H, self.W, 3), np.float32)
        for r in range(self.world):
            full[self.rank_rows[r]] = self.gathered[r][: len(self.rank_rows[r])].cpu().numpy()
        one = self.make_full_session()
        for _ in range(self.steps_done):
            one.accumulate(self.spp)
        ref = one.fetch_hdr()
        one.close()
        same = bool(np.array_equal(full.view(np.uint32), ref.view(np.uint32)))
        print(f"[rehearsal] {label}: gathered frame == unsharded frame: {same}", file=sys.stderr, flush=True)
        if not same:
            raise SystemExit(f"rehearsal ({label}): gathered frame differs from the unsharded render")

    def close(self):
        self.sess.close()


def rays_block(value_mpaths, work):
    """Rays per second beside path-samples per second: S1's paths average 1.8 rays (most end on the black sky after one bounce),
    the dense scenes' 6.1."""
    return {"unit": "Grays/s", "reference_algorithm": round(value_mpaths * work["reference"]["rays"] / 1e3, 3),
            "timed_schedule": round(value_mpaths * work["timed"]["rays"] / 1e3, 3),
            "note": "path-samples/s x rays per path-sample (closest-hit + shadow rays the reference would trace / the timed schedule walks)"}


def scene_api_default(lib):
    """The reference's own loop shape on one GPU (scene.py:177, 233-262: samples_per_frame = 1; every frame set_proj_mat draws a
    new jitter, accumulate() once, copy_prev_matrices()) -- what `Scene.finish()` and example1..10.py run by default."""
    from voxel_rt2_amd import host, scenes
    from voxel_rt2_amd._session import NativeSession
    mat, rgb, params = scenes.scene_s1(0)
    cfg = host.make_config(WIDTH, HEIGHT, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=MAX_DEPTH, seed=SEED)
    s = NativeSession(lib, "vrt_", cfg)
    setup_session(s, mat, rgb, params)
    cams = [host.default_camera(WIDTH, HEIGHT, jitter_index=k + 1) for k in range(16)]

    def frames(n, k0):
        for k in range(n):
            s.set_camera(cams[(k0 + k) % 16])
            s.accumulate(1)
            s.end_frame()
    frames(40, 0)
    s.sync()
    n = 240
    t0 = time.perf_counter()
    frames(n, 40)
    s.sync()
    dt = time.perf_counter() - t0
    ok = bool(np.isfinite(s.fetch_hdr()).all())
    s.close()
    return {"config": "2, as the Scene API drives it", "name": "scene_api_default", "metric": "Mpath-samples/sec", "unit": "Mpath-samples/s",
            "workload": "scene S1, 1920x1080, 8 bounces, ONE sample per vrt_accumulate call, a new TAA jitter (vrt_set_camera) and "
                        "vrt_end_frame per frame: the reference's samples_per_frame = 1 loop (scene.py:177, 233-262)",
            "value": round(WIDTH * HEIGHT * n / dt / 1e6, 2), "steps": n, "ms_per_step": round(dt / n * 1e3, 4), "finite": ok}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)   # at least 50 timed frames
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    from voxel_rt2_amd import scenes, _lib, parallel
    lib = _lib.load()   # before torch touches the device: the library picks the HIP runtime torch ships (and the queue count)
    import torch
    import torch.distributed as dist

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
    common = dict(rank=rank, world=world, local_rank=local_rank, rehearse=rehearse)

    scene = scenes.scene_s1(0)
    mat, rgb, params = scene
    # VRT_BENCH_STRIPES=S: interleaved stripes of S rows (vrt_set_row_stripes) instead of cost-balanced contiguous row tiles
    stripes = int(os.environ.get("VRT_BENCH_STRIPES", "0"))
    run = ShardedRun(lib, dist, torch, scene=scene, W=WIDTH, H=HEIGHT, depth=MAX_DEPTH, spp=SPP_PER_STEP, balance=True, stripes=stripes, **common)
    elapsed, st = run.timed(args.steps, args.warmup)
    if rehearse and world > 1 and rank == 0:
        run.check_against_unsharded("config 2")
    bounds = run.bounds
    # what a path-sample of this rank's tile does: untimed instrumented launches
    work = count_work(lib, run.sess, SPP_PER_STEP)
    run.sess.sync()

    if rank == 0:
        total_samples = WIDTH * HEIGHT * SPP_PER_STEP * args.steps
        value = total_samples / elapsed / 1e6
        launches = max(st["render_launches"], 1)
        avg_ms = st["render_ms"] / launches
        samples_per_launch = st["path_samples"] / launches  # a launch renders this rank's pixels x the fused samples
        counters, counters_note = measured_counters(lib) if world == 1 else ({}, None)
        render_kernel = "k_render" if os.environ.get("VRT_RENDER") == "fused" else "k_render_pool"
        flags = st.get("pipeline_flags", 0)
        note = ("HBM is the bound the tier names; the 128^3 working set (8.3 MB) is cache resident and the kernel is bound by vector-instruction "
                "issue under divergence at 2 waves per SIMD (DESIGN.md 7); `achieved` counts the bytes of the schedule that is timed (fused "
                "samples share their camera rays: queries counted once), `reference_algorithm` the bytes the reference would move for the same "
                "paths. Launches overlap: `achieved` / `frac` use a launch's span (what the kernel trace shows), `all_launches_in_flight` "
                "the period between launches (the chip's rate)")
        if counters_note:
            note += "; " + counters_note
        out = {
            "metric": "Mpath-samples/sec at 1920x1080, 8 bounces", "value": round(value, 3), "unit": "Mpath-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "example1-style scene S1 (scenes.scene_s1), 128^3 grid, 1920x1080, 4 spp/step, 8 bounces, "
                                   "static camera, ReSTIR off", "width": WIDTH, "height": HEIGHT, "spp_per_step": SPP_PER_STEP,
                       "max_depth": MAX_DEPTH, "seed": SEED, "build_id": lib.vrt_build_id().decode(),
                       "launch_pipeline": {"overlapped": bool(flags & 1), "dispatch_gate": bool(flags & 2), "gate_host_releases": int((flags >> 8) & 0xFFFF),
                                           "launches_in_flight": 2 * int((flags >> 2) & 7), "workgroup_slots_per_launch": f"1/{max(int((flags >> 5) & 7), 1)}"},
                       "sharding": ((f"{world} sets of interleaved {stripes}-row stripes (every rank renders row stripes spread over the whole frame, two more rows either side of each), "
                                     if stripes else f"{world} contiguous row tiles, boundaries balanced by measured tile cost, ") + "HDR tile written by the temporal pass into a ring "
                                    f"of {ShardedRun.N_TILES} device tiles, RCCL gather of every step's tile (issued when its pass is queued, all inside the "
                                    f"timed region), {parallel.reserved_cus(world)} CUs' worth of workgroup slots left free for the collective; "
                                    f"tile rows {[len(r) for r in run.rank_rows]}") if world > 1 else "none"},
            "roofline": roofline_block(render_kernel, avg_ms, samples_per_launch, render_bytes(work["reference"]), render_bytes(work["timed"]),
                                       counters.get("config2_s1_1080p", {}), note, elapsed / launches * 1e3),
            "work_per_path_sample": {k: {a: round(b, 3) for a, b in v.items()} for k, v in work.items()},
            "rays_per_s": rays_block(value, work),
            "kernel_ms_per_launch": {"render": round(avg_ms, 4), "temporal": round(st["temporal_ms"] / max(st["temporal_launches"], 1), 4)},
        }
    run.close()

    sec = None
    if not args.no_secondary:
        sec = []
        if world == 1:
            for case in SECONDARY:
                try:
                    sec.append(run_secondary(lib, case, counters))
                except Exception as e:  # a secondary leg must not take the headline down
                    sec.append({"config": case["config"], "name": case["name"], "error": f"{type(e).__name__}: {e}"})
            try:
                sec.append(scene_api_default(lib))
            except Exception as e:
                sec.append({"name": "scene_api_default", "error": f"{type(e).__name__}: {e}"})
        else:
            # the configs BASELINE.json defines ON N GPUs: rows split N ways, the gather inside the timed region
            for case in SECONDARY:
                if case.get("restir"):
                    continue
                sc_ = scenes.SCENES[case["scene"]](12345)
                sc_ = (sc_[0], sc_[1], dict(sc_[2], use_physical_sky=0, use_clouds=0))
                r2 = ShardedRun(lib, dist, torch, scene=sc_, W=case["W"], H=case["H"], depth=case["depth"], spp=case["spp"], grid=case.get("grid", 128), stripes=stripes, **common)
                steps = 2 if rehearse else case["steps"]
                el, st2 = r2.timed(steps, 1)
                if rehearse and rank == 0:
                    r2.check_against_unsharded(case["name"])
                if rank == 0:
                    px = case["W"] * case["H"]
                    sec.append({"config": case["config"].replace(" (one GPU's form)", f" on {world} GPUs"), "name": case["name"] + f"_{world}gpu",
                                "workload": case["workload"].replace("the whole frame on one GPU", f"rows split over {world} GPUs, RCCL gather of the HDR tiles every step inside the timed region"),
                                "metric": "Mpath-samples/sec", "unit": "Mpath-samples/s", "value": round(px * case["spp"] * steps / el / 1e6, 2),
                                "n_gpus": world, "steps": steps, "ms_per_step": round(el / steps * 1e3, 4),
                                "kernel_ms_per_launch": {"render": round(st2["render_ms"] / max(st2["render_launches"], 1), 4)},
                                "tile_rows": [len(r) for r in r2.rank_rows]})
                r2.close()
    if rank == 0:
        out["secondary"] = sec
        out["cpu_baseline"] = cpu_baseline(mat, rgb, params) if (not args.no_cpu_baseline and world == 1) else None
        # every workload's number once more, compactly, as the LAST key: a record that keeps only the end of this (long) line keeps these
        by_name = {s_.get("name", ""): s_ for s_ in (sec or [])}
        pick = lambda prefix: next((v.get("value", v.get("error")) for k, v in by_name.items() if k.startswith(prefix)), None)  # noqa: E731
        rl = out["roofline"]
        out["summary"] = {"unit": "Mpath-samples/s", "n_gpus": world, "config2": out["value"], "config3": pick("config3"), "config4": pick("config4"),
                          "config5": pick("config5"), "scene_api_default": pick("scene_api_default"),
                          "roofline_frac": rl["frac"], "roofline_frac_all_launches_in_flight": (rl.get("all_launches_in_flight") or {}).get("frac"),
                          "config3_roofline_frac": ((by_name.get("config3_s6_sky_clouds_restir_1080p") or {}).get("roofline") or {}).get("frac"),
                          "cpu_baseline": (out["cpu_baseline"] or {}).get("value")}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
