// vrt_api.hip -- the C ABI of include/vrt_api.h: context, device memory, launch sequencing.
//
// Replaces the host side of the reference's Renderer (renderer/pathtracer.py:28-136 field
// allocation, 139-150 / 246-287 setters, 314-329 prepare + sky steps, 664-668 reset, 1310-1323
// accumulate / fetch_image).  One context = one HIP device, one stream; all per-pixel buffers
// cover the context's rows plus a halo (row-tile sharding across GPUs renders the halo rows
// redundantly instead of exchanging them: per-pixel random streams make them bit-identical).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "../../include/vrt_api.h"
#include "vrt_kernels.h"

#define VRT_MAX_FUSED 4   // samples of one vrt_accumulate(n) call rendered by a single launch
#define VRT_MAX_STREAMS 8 // render launches in flight at most (a stream, a pool scratch and a camera-ray table each): 2, 4 or 8 are used
#define VRT_MAX_SETS 9    // copies of what a render launch writes (set 0 = the canonical buffers): streams + 1 are used
#define VRT_GB_ROT (VRT_MAX_SETS + 1)   // rotating g-buffer normal / depth copies: copy j is read by temporal passes j and j + 1, and the
                                       // launch that writes it again only waits for the pass VRT_MAX_SETS launches back
#define VRT_WORK_SETS 16  // rotating sets of work heads (vrt_kernels.hip: a launch zeroes the set eight launches ahead)
#define VRT_FETCH_SLOTS 4 // asynchronous fetches the caller may have outstanding (vrt_fetch_*_async)

using namespace vrt;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#if defined(VRT_HOST_PROFILE)
// Diagnostic build (tools/probe_host_cost.py): host time of every HIP_TRY call site, printed when the library is unloaded.
#include <map>
#include <algorithm>
static double prof_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static std::map<std::string, std::pair<double, long>> g_prof;
static struct ProfDump {
    ~ProfDump() {
        std::vector<std::pair<double, std::string>> rows;
        for (auto& kv : g_prof) rows.push_back({kv.second.first, kv.first});
        std::sort(rows.begin(), rows.end());
        for (auto it = rows.rbegin(); it != rows.rend() && it - rows.rbegin() < 25; ++it)
            fprintf(stderr, "[host] %9.1f ms %8ld calls %7.2f us  %s\n", it->first * 1e3, g_prof[it->second].second, it->first * 1e6 / (double)g_prof[it->second].second, it->second.substr(0, 110).c_str());
    }
} g_prof_dump;
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        const double t0_ = prof_now();                                                                  \
        hipError_t e_ = (expr);                                                                         \
        auto& p_ = g_prof[#expr]; p_.first += prof_now() - t0_; p_.second++;                            \
        if (e_ != hipSuccess)                                                                           \
            return fail(VRT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)
#else
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(VRT_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));               \
    } while (0)
#endif

struct EventPair { hipEvent_t a, b; int kind; };  // kind 0 render, 1 temporal, 2 gris

// Environment switches, read ONCE when a context is created (never on the launch path).
// The shipped library knows four: VRT_RENDER=pool|fused (which of the two schedules of the same per-path code renders),
// VRT_OVERLAP=0 (isolated launches: what the --pmc passes and the tile balancing of bench.py measure on),
// VRT_GATE_WATCHDOG_MS (how long a synchronisation waits at a gated launch before the host releases the gate) and the HIP
// runtime's own GPU_MAX_HW_QUEUES (how deep a pipeline the runtime's queues carry).
// A build with -DVRT_DEV_KNOBS (build_variants/libvrt_dev.so: `python -m voxel_rt2_amd.build --variant dev -DVRT_DEV_KNOBS`,
// loaded by tests/test_gpu_pipeline.py and the A/B runs of tools/) adds the development switches: the fault-injection hook
// VRT_TEST_FAIL_LAUNCH and the A/B switches VRT_CULL, VRT_DENSE, VRT_DEEP_ITEMS, VRT_DEEPER_ITEMS, VRT_STREAMS, VRT_GRID_DIV,
// VRT_DRAIN_GATE, VRT_FUSE, VRT_FUSE_RESTIR, VRT_OVERLAP_SINGLE, VRT_FULL_BELOW, VRT_CHUNK.
struct Knobs {
    int render = -1;               // -1: the library's choice, 0: fused, 1: pool; -2: a value VRT_RENDER does not know
    bool overlap = true;
    double gate_watchdog_s = 2.0;
    int hw_queues = 4;
    // development switches: the defaults below are what the shipped library always runs with
    int cull = -1, dense = -1;     // -1: decided from the scene (vrt_prepare)
    long long deep_items = (long long)12 << 20, deeper_items = (long long)9 << 19;
    int streams = 0, grid_div = 0; // 0: decided from the frame size (ensure_overlap)
    bool drain_gate = true, fuse_restir = true, overlap_single = true;
    int max_fused = VRT_MAX_FUSED, full_below = 2, chunk = 0, fail_launch = -1, gate_extra = 0, time_every = 0;
};
static Knobs read_knobs() {
    Knobs k;
    if (const char* e = getenv("VRT_RENDER")) k.render = strcmp(e, "fused") == 0 ? 0 : strcmp(e, "pool") == 0 ? 1 : -2;
    if (const char* e = getenv("VRT_OVERLAP")) k.overlap = atoi(e) != 0;
    if (const char* e = getenv("VRT_GATE_WATCHDOG_MS")) { const double v = atof(e); if (v > 0.0) k.gate_watchdog_s = v * 1e-3; }
    if (const char* e = getenv("GPU_MAX_HW_QUEUES")) k.hw_queues = atoi(e);
#if defined(VRT_DEV_KNOBS)
    if (const char* e = getenv("VRT_TEST_FAIL_LAUNCH")) k.fail_launch = atoi(e);
    if (const char* e = getenv("VRT_CULL")) k.cull = atoi(e) != 0;
    if (const char* e = getenv("VRT_DENSE")) k.dense = atoi(e) != 0;
    if (const char* e = getenv("VRT_DEEP_ITEMS")) k.deep_items = atoll(e);
    if (const char* e = getenv("VRT_DEEPER_ITEMS")) k.deeper_items = atoll(e);
    if (const char* e = getenv("VRT_STREAMS")) { const int v = atoi(e); if (v == 2 || v == 4 || v == 8) k.streams = v; }
    if (const char* e = getenv("VRT_GRID_DIV")) { const int v = atoi(e); if (v >= 1 && v <= 4) k.grid_div = v; }
    if (const char* e = getenv("VRT_DRAIN_GATE")) k.drain_gate = atoi(e) != 0;
    if (const char* e = getenv("VRT_FUSE")) { const int v = atoi(e); if (v >= 1 && v <= VRT_MAX_FUSED) k.max_fused = v; }
    if (const char* e = getenv("VRT_FUSE_RESTIR")) k.fuse_restir = atoi(e) != 0;
    if (const char* e = getenv("VRT_OVERLAP_SINGLE")) k.overlap_single = atoi(e) != 0;
    if (const char* e = getenv("VRT_TIME_EVERY")) { const int v = atoi(e); if (v >= 1 && v <= 1024) k.time_every = v; }   // 0 (default): by launch size
    if (const char* e = getenv("VRT_GATE_EXTRA")) { const int v = atoi(e); if (v >= 0 && v <= 4) k.gate_extra = v; }
    if (const char* e = getenv("VRT_FULL_BELOW")) { const int v = atoi(e); if (v >= 1 && v <= 3) k.full_below = v; }
    if (const char* e = getenv("VRT_CHUNK")) { const int v = atoi(e); if (v >= 64 && v <= 4096) k.chunk = v / 64 * 64; }
#endif
    return k;
}

struct vrt_ctx {
    vrt_config cfg;
    Knobs knobs;                      // environment switches as they stood when the context was created
    vrt_scene_params scene;
    vrt_camera cam;
    bool have_scene = false, have_cam = false, prepared = false, have_prev = false;
    bool instrumented = false;
    bool count_as_timed = false;      // instrumented launches keep the camera-ray reuse of the timed schedule (vrt_set_instrumented(ctx, 2))
    bool ref_oob = false;             // vrt_set_reference_indexing: cells outside the grid are read the reference's way (vrt_trace.h, ref_bit)
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;
    int n_cu = 0, render_blocks = 0;
    int render_blocks_d12 = 0;        // ... of the twelve-wave geometry a dense 128^3 grid renders on (k_render_pool_dense12)
    int reserved_cus = 0;             // CUs' worth of workgroup slots the persistent render grid leaves free (vrt_reserve_cus)
    bool pooled = false;              // render through k_render_pool (vrt_pool.h) instead of k_render
    uint32_t* d_pool_scratch = nullptr;
    PrimaryRecord* d_prim_cache[VRT_MAX_STREAMS] = {};  // camera-ray records of fused launches (one table per render stream)
    // rows
    int own0 = 0, own1 = 0;   // rows this context produces
    int stripe_rows = 0, stripe_parts = 0, stripe_part = 0;   // ... or, of them, every stripe_parts-th stripe of stripe_rows rows (vrt_set_row_stripes)
    int buf0 = 0, buf1 = 0;   // rows held in the buffers (own + halo)
    int halo = 2;
    size_t npix = 0;          // (buf1 - buf0) * W
    // scene data
    int8_t* d_mat = nullptr; uint8_t* d_rgb = nullptr; uint32_t* d_grid = nullptr;
    unsigned long long *d_l0 = nullptr, *d_l1 = nullptr, *d_l2 = nullptr, *d_l3 = nullptr, *d_l0c = nullptr;
    uint32_t* d_l0c_base = nullptr;  // [512] offsets + [1] count
    bool cull_active = false;        // the grown box leaves part of the grid out: there are rays to cull (read back by vrt_prepare)
    bool dense_grid = false;         // half of the bricks or more are non-empty (read back by vrt_prepare)
    float* d_cull = nullptr;         // [8] grown bounding box of the solid voxels + flag, [8] the same with the flag off (cull_ray, vrt_trace.h)
    float* d_mats = nullptr;
    Counters* d_counters = nullptr;
    unsigned* d_work = nullptr;
    // sky
    float *d_sky_scat = nullptr, *d_sky_trans = nullptr, *d_cloud_ambient = nullptr;
    uint16_t* d_trans_lut = nullptr;
    uint8_t* d_cloud_tex = nullptr;
    uint32_t cloud_pass = 0;
    // per-pixel
    f3 *d_cbuf[2] = {nullptr, nullptr};  // color_buffer: [cidx] = HDR of the last pass = render target of the next (pathtracer.py:39)
    int cidx = 0;
    f3 *d_color_s = nullptr, *d_color_d2 = nullptr, *d_color_s2 = nullptr, *d_gb_pos = nullptr;
    f3* d_multi_d = nullptr;        // diffuse colour planes of the samples fused into one launch (allocated on first use)
    f3* d_spec_planes = nullptr;    // VRT_MAX_FUSED specular planes; d_color_s = the last one
    float* d_refl_planes = nullptr; // likewise for the raw reflection depth; d_gb_refl = the last one
    uint32_t* d_gb_normal[VRT_GB_ROT] = {};  // rotating: [cur] is written by the next launch, [prev_gb] by the last
    float* d_gb_depth[VRT_GB_ROT] = {};
    uint32_t* d_gb_mat = nullptr;
    float *d_gb_refl = nullptr, *d_gb_refl_f = nullptr;
    f4 *d_hist_d[2] = {nullptr, nullptr}, *d_hist_s[2] = {nullptr, nullptr};
    f4* d_ldr = nullptr;
    uint32_t* d_ldr8 = nullptr;   // rgba8 image of vrt_fetch_ldr8_async (allocated on first use)
    ReservoirRec* d_res[2] = {nullptr, nullptr};
    ReservoirRec* d_res_planes = nullptr;   // input reservoirs: VRT_MAX_FUSED planes, d_res[0] is the last of them
    GrisGeo* d_gris_geo = nullptr;   // per-pixel records of k_gris's prepare pass (vrt_restir.h)
    GrisSrc* d_gris_src = nullptr;
    GrisTest* d_gris_tst = nullptr;
    float* d_mats_x = nullptr;       // [128][8] mat_derive() of every material row
    int cur = 0;      // g-buffer rotation: render writes [cur], temporal reads [prev_gb] as "prev"
    int prev_gb = VRT_GB_ROT - 1;  // the copy the most recent launch wrote
    // Overlapped launches (vrt_accumulate): n_streams + 1 copies (set 0 = the canonical buffers, alt_*[s - 1] the others) of
    // everything a render launch writes and its temporal pass reads, n_streams render streams and the events that order
    // them, so that the next launches start while launch k drains and temporal pass k runs beside them.
    f3* alt_multi_d[VRT_MAX_SETS - 1] = {}; f3* alt_spec_planes[VRT_MAX_SETS - 1] = {}; float* alt_refl_planes[VRT_MAX_SETS - 1] = {};
    f3* alt_gb_pos[VRT_MAX_SETS - 1] = {}; uint32_t* alt_gb_mat[VRT_MAX_SETS - 1] = {};
    uint32_t* alt_pool_scratch[VRT_MAX_STREAMS - 1] = {};  // the other render streams' scratch
    int n_streams = 2;   // depth of the launch pipeline (ensure_overlap): 2, 4 with launches of half the workgroup slots each, or 8 with quarters
    int grid_div = 1;    // an overlapped launch takes render_blocks / grid_div workgroups
    // A render launch queued behind another on another render stream would be dispatched at once and sit in the
    // queue until workgroups retire -- which the profiler and the events count as its run time.  Instead the kernel
    // raises this word (HSA signal memory, host visible) to launch_seq + 1 when it starts to drain, and the stream of the
    // launch that will take its workgroup slots (the next one; the one after with half-size launches) waits for that
    // value (hipStreamWaitValue32) before the dispatch.  The gate only TIMES dispatches
    // (ordering is by events), so raising the word early is always safe: release_gate() does it from the host on
    // error paths and when a synchronisation overstays (gate_watchdog_ms).  A stream wait is itself a queue operation:
    // under a tool that runs one queue operation at a time (rocprofv3 --pmc) a wait that is dispatched ahead of the
    // launch it waits for blocks that launch for ever (tools/probes/probe_gate.cpp reproduces it: the wait completes
    // by itself in 0.3 ms, never under --pmc, and a host store releases it) -- ensure_overlap() tests for exactly
    // that once and leaves the gate out where the test fails.
    uint32_t* drain_signal = nullptr;
    bool drain_signalled = false;  // the most recent render launch was given the signal
    bool prev_launch_full = true;  // ... and took every workgroup slot (the next dispatch waits for ITS drain, whatever the pipeline's depth)
    unsigned gate_releases = 0;    // host releases so far (error paths, watchdog): diagnostic
    hipStream_t rstream[VRT_MAX_STREAMS] = {};
    hipEvent_t ev_r[VRT_MAX_SETS] = {}, ev_t[VRT_MAX_SETS] = {}, ev_main = nullptr;
    bool ev_t_valid[VRT_MAX_SETS] = {};
    bool overlap_ready = false, overlap_failed = false;
    // Device time per kind of pass (0 render, 1 accumulation, 2 spatial reuse): every pass is counted, the ones that carry timers
    // are summed (small launches: one in eight, accumulate_impl) and vrt_get_stats scales the sum to all of them.
    double timed_ms[3] = {0.0, 0.0, 0.0};
    uint32_t timed_n[3] = {0u, 0u, 0u}, passes_n[3] = {0u, 0u, 0u};
    unsigned since_reset = 0;     // render launches since vrt_reset_stats: the first one carries timers
    unsigned mode_switches = 0;   // times the pipeline was drained to change its depth (ensure_overlap)
    bool main_dirty = true;   // work other than accumulate passes was queued on the main stream since the last overlapped launch
    unsigned pipe_seq = 0;    // overlapped launches so far
    int last_set = 0;         // copy (0 = the canonical buffers) the most recent render launch wrote
    int last_render_set = -1; // copy whose ev_r the most recent overlapped launch recorded (-1: none yet)
    const uint32_t* last_gb_normal = nullptr;   // g-buffer normal / depth of the most recent render launch (either schedule)
    const float* last_gb_depth = nullptr;
    // HDR tiles handed over device to device (vrt_set_hdr_targets): pass k also writes its HDR rows to ring[k % n]
    std::vector<void*> hdr_targets;
    unsigned long long hdr_targets_written = 0;
    unsigned long long hdr_targets_committed = 0;   // ... by calls that returned VRT_OK (abort_pipeline rolls back to it)
    // asynchronous fetches (vrt_fetch_*_async)
    hipStream_t fetch_stream = nullptr;
    hipEvent_t ev_fetch[VRT_FETCH_SLOTS] = {}, ev_fetch_src = nullptr, ev_cbuf_read[2] = {};
    bool fetch_valid[VRT_FETCH_SLOTS] = {}, cbuf_read_pending[2] = {};
    unsigned last_full_seq = 0;          // launch_seq + 1 of the most recent launch that took every workgroup slot (0: none)
    int hist_in = 0;  // history ping-pong
    mat4 prev_view{}, prev_proj{};
    uint32_t frame = 0;
    unsigned launch_seq = 0;  // render launches so far (selects which of the two work counters a launch uses)
    // stats
    std::vector<EventPair> pending;
    vrt_stats stats{};
};

template <class T>
static hipError_t dalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, n * sizeof(T));
    // hipMemset fills on the NULL stream and may return before the fill has run; the context's streams are non-blocking, so
    // nothing orders a launch queued next (a buffer allocated on its first use: the fused samples' planes) after that fill --
    // it zeroed the first tiles a render launch had just written (seen once the allocator handed back recycled memory:
    // tests/test_gpu_parity.py::test_row_shards_equal_full_frame after the large frames of test_gpu_fullsize.py).  Wait for it.
    if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T));
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    return e;
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Host store to the gate word: every launch queued so far counts as draining.  launch_seq is at least what any launch in
// flight will raise the word to, and later launches raise it further (atomic max), so no wait can be lost.
static void release_gate(vrt_ctx* c) {
    if (!c->drain_signal) return;
    __atomic_store_n(c->drain_signal, (uint32_t)c->launch_seq, __ATOMIC_RELEASE);
    c->gate_releases++;
}

// hipStreamSynchronize with a bound on how long a gated launch may hold the stream: past it the gate is released from
// the host (harmless when the launches are merely long; the way out when a dispatch never comes).
static hipError_t sync_guarded(vrt_ctx* c, hipStream_t st) {
    if (c->drain_signal && c->drain_signalled) {
        const double limit = c->knobs.gate_watchdog_s;
        const double t0 = now_s();
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q != hipErrorNotReady) { (void)hipGetLastError(); break; }
            const double waited = now_s() - t0;
            if (waited > limit) { release_gate(c); break; }
            // a frame is a millisecond: yield for the first of it (a 50 us sleep overshoots by 50-150 us with the kernel's timer
            // slack, 5-15 % of a fetch-every-frame loop), sleep only through launches that are really long
            if (waited < 2e-3) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    return hipStreamSynchronize(st);
}

// True when a stream wait queued BEFORE the operation that satisfies it (on another render stream) completes: the
// order in which a launch and the wait of its successor can reach the hardware.  Under a tool that serialises queue
// operations it does not -- then the word is released from the host and the caller leaves the gate out.
static bool gate_self_test(vrt_ctx* c) {
    if (hipStreamWaitValue32(c->rstream[1], c->drain_signal, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu) != hipSuccess) { (void)hipGetLastError(); return false; }
    bool wrote = hipStreamWriteValue32(c->rstream[0], c->drain_signal, 1u, 0) == hipSuccess;
    bool by_itself = false;
    const double t0 = now_s();
    while (wrote && now_s() - t0 < 0.1) {
        if (hipStreamQuery(c->rstream[1]) == hipSuccess) { by_itself = true; break; }
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    (void)hipGetLastError();
    if (!by_itself) __atomic_store_n(c->drain_signal, 1u, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(c->rstream[1]);
    (void)hipStreamSynchronize(c->rstream[0]);
    __atomic_store_n(c->drain_signal, 0u, __ATOMIC_RELEASE);
    (void)hipGetLastError();
    return by_itself;
}

static void account(vrt_ctx* c, const EventPair& ev) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { c->timed_ms[ev.kind] += ms; c->timed_n[ev.kind]++; }
    hipEventDestroy(ev.a);
    hipEventDestroy(ev.b);
}
static void resolve_events(vrt_ctx* c) {   // waits for every launch timed so far
    for (auto& ev : c->pending)
        if (hipEventSynchronize(ev.b) == hipSuccess) account(c, ev);
        else { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    c->pending.clear();
}
// On the launch path: the timers of launches that HAVE completed are read and freed, nothing is waited for (a wait here would
// empty the launch pipeline every hundred calls); a caller that never synchronises is held to 4096 outstanding timers.
static void resolve_completed(vrt_ctx* c) {
    if (c->pending.size() > 4096) { resolve_events(c); return; }
    size_t done = 0;
    while (done < c->pending.size() && hipEventQuery(c->pending[done].b) == hipSuccess) { account(c, c->pending[done]); done++; }
    (void)hipGetLastError();   // (hipErrorNotReady is the expected answer at the first launch still running)
    c->pending.erase(c->pending.begin(), c->pending.begin() + (long)done);
}

// The row ranges this context produces: one, or with vrt_set_row_stripes its stripes.
static std::vector<std::pair<int, int>> owned_ranges(const vrt_ctx* c) {
    std::vector<std::pair<int, int>> r;
    if (c->stripe_rows == 0) { r.emplace_back(c->own0, c->own1); return r; }
    const int period = c->stripe_rows * c->stripe_parts;
    for (int s0 = c->stripe_part * c->stripe_rows; s0 < c->cfg.height; s0 += period)
        r.emplace_back(s0, s0 + c->stripe_rows < c->cfg.height ? s0 + c->stripe_rows : c->cfg.height);
    return r;
}
static size_t owned_rows(const vrt_ctx* c) {
    size_t n = 0;
    for (const auto& r : owned_ranges(c)) n += (size_t)(r.second - r.first);
    return n;
}
static FrameParams make_frame_params(const vrt_ctx* c) {
    FrameParams fp;
    memset(&fp, 0, sizeof(fp));
    memcpy(fp.view.m, c->cam.view, 64);
    memcpy(fp.proj.m, c->cam.proj, 64);
    memcpy(fp.view_inv.m, c->cam.view_inv, 64);
    memcpy(fp.proj_inv.m, c->cam.proj_inv, 64);
    fp.camera_pos = mk3(c->cam.pos[0], c->cam.pos[1], c->cam.pos[2]);
    const int W = c->cfg.width, H = c->cfg.height;
    fp.inv_res = mk2((float)(1.0 / (double)W), (float)(1.0 / (double)H));
    // TAA jitter: two draws of random stream 3 per set_proj_mat call (pathtracer.py:264-265)
    dm_rng rng = dm_rng_init(c->cfg.seed, c->cam.jitter_index, 0u, 3u);
    float r0 = dm_rng_f32(&rng), r1 = dm_rng_f32(&rng);
    fp.taa_jitter = mk2((r0 * 2.0f - 1.0f) * fp.inv_res.x, (r1 * 2.0f - 1.0f) * fp.inv_res.y);
    fp.W = W; fp.H = H;
    fp.row0 = c->buf0; fp.row1 = c->buf1;
    fp.camera_is_moving = c->cam.camera_is_moving;
    fp.render_scale = c->cam.render_scale;
    fp.max_accum_frames = c->cam.max_accum_frames;
    fp.light_dir = mk3(c->scene.light_direction[0], c->scene.light_direction[1], c->scene.light_direction[2]);
    fp.light_color = mk3(c->scene.light_color[0], c->scene.light_color[1], c->scene.light_color[2]);
    fp.light_cos_max = c->scene.light_cos_theta_max;
    fp.light_weight = c->scene.light_weight;
    fp.floor_height = c->scene.floor_height;
    fp.floor_color = mk3(c->scene.floor_color[0], c->scene.floor_color[1], c->scene.floor_color[2]);
    fp.floor_material = c->scene.floor_material;
    fp.background = mk3(c->scene.background_color[0], c->scene.background_color[1], c->scene.background_color[2]);
    fp.use_sky = c->scene.use_physical_sky;
    fp.voxel_edges = c->cfg.voxel_edges;
    fp.exposure = c->cfg.exposure;
    fp.max_depth = c->cfg.max_depth;
    fp.seed = c->cfg.seed;
    fp.frame = c->frame;
    if (c->stripe_rows) {
        fp.stripe_rows = c->stripe_rows;
        fp.stripe_period = c->stripe_rows * c->stripe_parts;
        fp.stripe_first = c->stripe_part * c->stripe_rows;
        fp.stripe_tile_rows = (int)owned_ranges(c).size() * (c->stripe_rows / 8 + 2);
    }
    return fp;
}
// launches that count the reference's work walk every ray, as the reference and the oracle do; VRT_CULL=0 for A/B runs
static bool culling(const vrt_ctx* c) {
    // with the reference's indexing a ray clear of every solid voxel can still "hit" outside the grid: every ray is walked
    bool cull = c->cull_active && !(c->instrumented && !c->count_as_timed) && !c->ref_oob;
    if (c->knobs.cull == 0) cull = false;
    return cull;
}
static SceneData make_scene_data(const vrt_ctx* c) {
    SceneData sc;
    sc.pyr.l0 = c->d_l0; sc.pyr.l1 = c->d_l1; sc.pyr.l2 = c->d_l2; sc.pyr.l3 = c->d_l3;
    sc.pyr.l0c = c->d_l0c; sc.pyr.l0c_base = c->d_l0c_base; sc.pyr.l0c_count = c->d_l0c_base + 512;
    sc.pyr.ref_oob = c->ref_oob ? 1 : 0;
    sc.grid = c->d_grid;
    sc.mats = c->d_mats;
    sc.sky.scattering = c->d_sky_scat;
    sc.sky.transmittance = c->d_sky_trans;
    sc.sky.res = c->cfg.sky_res;
    sc.sky.fres = c->cfg.sky_res > 0 ? (float)(1.0 / (double)c->cfg.sky_res) : 0.0f;
    sc.counters = c->d_counters;
    sc.cull = c->d_cull + (culling(c) ? 0 : 8);
    return sc;
}
static SkyPrecompute make_sky(const vrt_ctx* c) {
    SkyPrecompute sp;
    sp.scattering = c->d_sky_scat; sp.transmittance = c->d_sky_trans; sp.trans_lut = c->d_trans_lut;
    sp.cloud_tex = c->d_cloud_tex; sp.cloud_ambient = c->d_cloud_ambient;
    sp.res = c->cfg.sky_res;
    sp.fres = (float)(1.0 / (double)c->cfg.sky_res);
    sp.use_clouds = c->scene.use_clouds;
    sp.seed = c->cfg.seed;
    return sp;
}
static void sun_of(const vrt_ctx* c, f3& dir, f3& col, float& cosm) {
    dir = mk3(c->scene.light_direction[0], c->scene.light_direction[1], c->scene.light_direction[2]);
    col = mk3(c->scene.light_color[0], c->scene.light_color[1], c->scene.light_color[2]) * c->scene.light_weight;
    cosm = c->scene.light_cos_theta_max;
}

extern "C" {

const char* vrt_last_error(void) { return g_err.c_str(); }
#ifndef VRT_BUILD_ID
#define VRT_BUILD_ID "unknown"
#endif
const char* vrt_build_id(void) { return VRT_BUILD_ID; }

vrt_ctx* vrt_create(const vrt_config* cfg) {
    if (!cfg) { fail(VRT_E_INVALID, "null config"); return nullptr; }
    if (cfg->grid_res != 128 && cfg->grid_res != 256) { fail(VRT_E_INVALID, "grid_res must be 128 (pathtracer.py:83) or 256"); return nullptr; }
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 16384 || cfg->height > 16384) { fail(VRT_E_INVALID, "bad image size"); return nullptr; }
    if (cfg->max_depth < 1 || cfg->max_depth > 64) { fail(VRT_E_INVALID, "max_depth must be in 1..64"); return nullptr; }
    if (cfg->sky_res < 0 || cfg->sky_res > 8192 || (cfg->sky_res > 0 && cfg->sky_res < 4)) { fail(VRT_E_INVALID, "sky_res must be 0 or 4..8192"); return nullptr; }
    if (cfg->dx != 2.0f / (float)cfg->grid_res) { fail(VRT_E_INVALID, "dx must be 2 / grid_res: 1/64 at 128 (scene.py:11), 1/128 at 256 -- the grid spans the world box [-1, 1]^3"); return nullptr; }
    int own0 = 0, own1 = cfg->height;
    if (cfg->row_end > cfg->row_begin) {
        if (cfg->row_begin < 0 || cfg->row_end > cfg->height) { fail(VRT_E_INVALID, "row range outside the image"); return nullptr; }
        own0 = cfg->row_begin; own1 = cfg->row_end;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fail(VRT_E_DEVICE, "no HIP device: libvrt_hip has no CPU path"); return nullptr; }
    if (cfg->device < 0 || cfg->device >= ndev) { fail(VRT_E_INVALID, "device ordinal out of range"); return nullptr; }
    hipDeviceProp_t prop;
    if (hipSetDevice(cfg->device) != hipSuccess || hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) {
        fail(VRT_E_DEVICE, "cannot select HIP device");
        return nullptr;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fail(VRT_E_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        return nullptr;
    }
    vrt_ctx* c = new vrt_ctx();
    c->cfg = *cfg;
    c->knobs = read_knobs();
    c->device = cfg->device;
    c->n_cu = prop.multiProcessorCount;
    c->own0 = own0; c->own1 = own1;
    c->halo = cfg->use_restir ? 26 : 2;  // bilinear + prepass taps; + spatial reuse radius 24 (pathtracer.py:1313)
    c->buf0 = own0 - c->halo < 0 ? 0 : own0 - c->halo;
    c->buf1 = own1 + c->halo > cfg->height ? cfg->height : own1 + c->halo;
    c->npix = (size_t)(c->buf1 - c->buf0) * cfg->width;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    const size_t G = (size_t)cfg->grid_res, nvox = G * G * G, n = c->npix;
    const size_t nw0 = nvox / 64, nw1 = nw0 / 64, nw2 = nw1 / 64;  // words of the brick levels
    ok = ok && dalloc(&c->d_mat, nvox) == hipSuccess && dalloc(&c->d_rgb, nvox * 3) == hipSuccess && dalloc(&c->d_grid, nvox) == hipSuccess;
    ok = ok && dalloc(&c->d_l0, nw0) == hipSuccess && dalloc(&c->d_l1, nw1) == hipSuccess && dalloc(&c->d_l2, nw2) == hipSuccess &&
         dalloc(&c->d_l3, 1) == hipSuccess && dalloc(&c->d_cull, 16) == hipSuccess && dalloc(&c->d_l0c, 32768) == hipSuccess && dalloc(&c->d_l0c_base, 513) == hipSuccess;
    ok = ok && dalloc(&c->d_mats, 128 * 14) == hipSuccess && dalloc(&c->d_counters, 1) == hipSuccess && dalloc(&c->d_work, VRT_WORK_SETS * VRT_WORK_HEADS * VRT_WORK_HEAD_STRIDE) == hipSuccess;
    ok = ok && dalloc(&c->d_cbuf[0], n) == hipSuccess && dalloc(&c->d_cbuf[1], n) == hipSuccess && dalloc(&c->d_spec_planes, n * VRT_MAX_FUSED) == hipSuccess && dalloc(&c->d_gb_pos, n) == hipSuccess;
    ok = ok && dalloc(&c->d_gb_mat, n) == hipSuccess && dalloc(&c->d_refl_planes, n * VRT_MAX_FUSED) == hipSuccess;
    if (ok) { c->d_color_s = c->d_spec_planes + (size_t)(VRT_MAX_FUSED - 1) * n; c->d_gb_refl = c->d_refl_planes + (size_t)(VRT_MAX_FUSED - 1) * n; }
    ok = ok && dalloc(&c->d_gb_refl_f, n) == hipSuccess && dalloc(&c->d_ldr, n) == hipSuccess;
    for (int s = 0; s < VRT_GB_ROT && ok; s++) ok = ok && dalloc(&c->d_gb_normal[s], n) == hipSuccess && dalloc(&c->d_gb_depth[s], n) == hipSuccess;
    for (int s = 0; s < 2 && ok; s++) {
        ok = ok && dalloc(&c->d_hist_d[s], n) == hipSuccess && dalloc(&c->d_hist_s[s], n) == hipSuccess;
    }
    if (cfg->use_restir) {
        // VRT_MAX_FUSED planes of input reservoirs, the LAST being the slot the reference knows (as with the specular planes):
        // a fused launch ends on it, so a later pass that renders part of the frame finds the last sample's reservoirs there
        ok = ok && dalloc(&c->d_res_planes, n * VRT_MAX_FUSED) == hipSuccess && dalloc(&c->d_res[1], n) == hipSuccess;
        if (ok) c->d_res[0] = c->d_res_planes + (size_t)(VRT_MAX_FUSED - 1) * n;
    }
    if (cfg->use_restir) ok = ok && dalloc(&c->d_color_d2, n) == hipSuccess && dalloc(&c->d_color_s2, n) == hipSuccess &&
                              dalloc(&c->d_gris_geo, n) == hipSuccess && dalloc(&c->d_gris_src, n) == hipSuccess && dalloc(&c->d_gris_tst, n) == hipSuccess;
    ok = ok && dalloc(&c->d_mats_x, 128 * 8) == hipSuccess;
    if (cfg->sky_res > 0) {
        size_t ns = (size_t)cfg->sky_res * cfg->sky_res * 3;
        ok = ok && dalloc(&c->d_sky_scat, ns) == hipSuccess && dalloc(&c->d_sky_trans, ns) == hipSuccess;
        ok = ok && dalloc(&c->d_trans_lut, 256 * 128 * 3) == hipSuccess && dalloc(&c->d_cloud_tex, 256 * 256 * 3) == hipSuccess;
        ok = ok && dalloc(&c->d_cloud_ambient, 4) == hipSuccess;
    }
    if (!ok) {
        fail(VRT_E_DEVICE, std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError()));
        vrt_destroy(c);
        return nullptr;
    }
    // default material table rows (materials.py:50-63) until vrt_upload_materials is called
    {
        std::vector<float> t(128 * 14);
        const float row[14] = {1, 1, 1, 0, 0, 0.04f, 0, 0.9f, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 128; i++) memcpy(&t[14 * i], row, sizeof(row));
        hipMemcpy(c->d_mats, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        launch_mat_derived(c->stream, c->d_mats, c->d_mats_x);
    }
    c->last_gb_normal = c->d_gb_normal[VRT_GB_ROT - 1]; c->last_gb_depth = c->d_gb_depth[VRT_GB_ROT - 1];
    memset(&c->scene, 0, sizeof(c->scene));
    c->scene.floor_color[0] = c->scene.floor_color[1] = c->scene.floor_color[2] = 1.0f;  // pathtracer.py:91-93
    c->scene.floor_material = 1;
    c->scene.light_cos_theta_max = 1.0f;
    return c;
}

void vrt_destroy(vrt_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    release_gate(c);   // nothing may be left waiting at a gate
    if (c->fetch_stream) hipStreamSynchronize(c->fetch_stream);
    for (int s = 0; s < VRT_MAX_STREAMS; s++) if (c->rstream[s]) hipStreamSynchronize(c->rstream[s]);
    if (c->stream) hipStreamSynchronize(c->stream);
    resolve_events(c);
    for (int s = 0; s < VRT_MAX_SETS; s++) {
        if (c->ev_r[s]) hipEventDestroy(c->ev_r[s]);
        if (c->ev_t[s]) hipEventDestroy(c->ev_t[s]);
    }
    for (int s = 0; s < VRT_MAX_STREAMS; s++) if (c->rstream[s]) hipStreamDestroy(c->rstream[s]);
    if (c->ev_main) hipEventDestroy(c->ev_main);
    for (int s = 0; s < VRT_FETCH_SLOTS; s++) if (c->ev_fetch[s]) hipEventDestroy(c->ev_fetch[s]);
    for (int s = 0; s < 2; s++) if (c->ev_cbuf_read[s]) hipEventDestroy(c->ev_cbuf_read[s]);
    if (c->ev_fetch_src) hipEventDestroy(c->ev_fetch_src);
    if (c->fetch_stream) hipStreamDestroy(c->fetch_stream);
    if (c->drain_signal) hipFree(c->drain_signal);
    for (int s = 0; s < VRT_MAX_STREAMS; s++) {
        if (c->d_prim_cache[s]) hipFree(c->d_prim_cache[s]);
        if (s < VRT_MAX_STREAMS - 1 && c->alt_pool_scratch[s]) hipFree(c->alt_pool_scratch[s]);
    }
    for (int s = 0; s < VRT_MAX_SETS - 1; s++) {
        void* copies[] = {c->alt_multi_d[s], c->alt_spec_planes[s], c->alt_refl_planes[s], c->alt_gb_pos[s], c->alt_gb_mat[s]};
        for (void* p : copies)
            if (p) hipFree(p);
    }
    for (int s = 2; s < VRT_GB_ROT; s++) {
        if (c->d_gb_normal[s]) hipFree(c->d_gb_normal[s]);
        if (c->d_gb_depth[s]) hipFree(c->d_gb_depth[s]);
    }
    void* ptrs[] = {
                    c->d_cull, c->d_mat, c->d_rgb, c->d_grid, c->d_l0, c->d_l1, c->d_l2, c->d_l3, c->d_l0c, c->d_l0c_base, c->d_mats, c->d_counters, c->d_work, c->d_sky_scat,
                    c->d_sky_trans, c->d_cloud_ambient, c->d_trans_lut, c->d_cloud_tex, c->d_cbuf[0], c->d_cbuf[1], c->d_spec_planes, c->d_color_d2,
                    c->d_color_s2, c->d_gb_pos, c->d_gb_normal[0], c->d_gb_normal[1], c->d_gb_depth[0], c->d_gb_depth[1],
                    c->d_gb_mat, c->d_refl_planes, c->d_gb_refl_f, c->d_hist_d[0], c->d_hist_d[1], c->d_hist_s[0], c->d_hist_s[1],
                    c->d_ldr, c->d_ldr8, c->d_res[1], c->d_res_planes, c->d_multi_d, c->d_pool_scratch, c->d_gris_geo, c->d_gris_src, c->d_gris_tst, c->d_mats_x};
    for (void* p : ptrs)
        if (p) hipFree(p);
    if (c->stream && c->owns_stream) hipStreamDestroy(c->stream);
    delete c;
}

int vrt_upload_voxels(vrt_ctx* c, const int8_t* mat, const uint8_t* rgb) {
    if (!c || !mat || !rgb) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t nvox = (size_t)c->cfg.grid_res * c->cfg.grid_res * c->cfg.grid_res;
    HIP_TRY(hipMemcpyAsync(c->d_mat, mat, nvox, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_rgb, rgb, nvox * 3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(sync_guarded(c, c->stream));  // host buffers are only borrowed for the call
    c->prepared = false;
    c->main_dirty = true;
    return VRT_OK;
}
int vrt_upload_materials(vrt_ctx* c, const float* table) {
    if (!c || !table) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    c->main_dirty = true;
    HIP_TRY(hipMemcpyAsync(c->d_mats, table, 128 * 14 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(launch_mat_derived(c->stream, c->d_mats, c->d_mats_x));
    HIP_TRY(sync_guarded(c, c->stream));
    return VRT_OK;
}
int vrt_upload_cloud_texture(vrt_ctx* c, const uint8_t* rgb) {
    if (!c || !rgb) return fail(VRT_E_INVALID, "null argument");
    if (c->cfg.sky_res <= 0) return fail(VRT_E_STATE, "context was created without sky tables (sky_res = 0)");
    HIP_TRY(hipSetDevice(c->device));
    c->main_dirty = true;
    HIP_TRY(hipMemcpyAsync(c->d_cloud_tex, rgb, 256 * 256 * 3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(sync_guarded(c, c->stream));
    return VRT_OK;
}
int vrt_set_scene(vrt_ctx* c, const vrt_scene_params* s) {
    if (!c || !s) return fail(VRT_E_INVALID, "null argument");
    if (s->use_physical_sky && c->cfg.sky_res <= 0) return fail(VRT_E_INVALID, "use_physical_sky needs sky_res > 0 at vrt_create");
    c->scene = *s;
    c->have_scene = true;
    return VRT_OK;
}
int vrt_set_camera(vrt_ctx* c, const vrt_camera* cam) {
    if (!c || !cam) return fail(VRT_E_INVALID, "null argument");
    if (!(cam->render_scale > 0.0f) || cam->render_scale > 1.0f) return fail(VRT_E_INVALID, "render_scale must be in (0, 1]");
    if (cam->camera_is_moving && (c->own0 != 0 || c->own1 != c->cfg.height || c->stripe_rows))
        return fail(VRT_E_INVALID, "row-sharded contexts support the static camera only (history resampling crosses tiles)");
    c->cam = *cam;
    c->have_cam = true;
    return VRT_OK;
}
int vrt_reserve_cus(vrt_ctx* c, int n_cus) {
    if (!c || n_cus < 0) return fail(VRT_E_INVALID, "bad argument");
    if (n_cus > c->n_cu - 8) n_cus = c->n_cu - 8 > 0 ? c->n_cu - 8 : 0;
    HIP_TRY(hipSetDevice(c->device));
    c->reserved_cus = n_cus;
    c->render_blocks = 0;   // the grid is sized again at the next vrt_accumulate
    return VRT_OK;
}
int vrt_set_row_stripes(vrt_ctx* c, int stripe_rows, int n_parts, int part) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    if (stripe_rows == 0) { c->stripe_rows = c->stripe_parts = c->stripe_part = 0; return VRT_OK; }
    if (stripe_rows < 8 || stripe_rows % 8 != 0 || n_parts < 1 || part < 0 || part >= n_parts) return fail(VRT_E_INVALID, "stripe_rows must be a multiple of 8, 0 <= part < n_parts");
    if (c->own0 != 0 || c->own1 != c->cfg.height) return fail(VRT_E_INVALID, "row stripes are a property of a whole-frame context (row_begin = row_end = 0)");
    if (c->cfg.use_restir) return fail(VRT_E_INVALID, "row stripes with ReSTIR would render a 24-row halo around every stripe: use contiguous row tiles");
    if (c->have_cam && c->cam.camera_is_moving) return fail(VRT_E_INVALID, "row stripes support the static camera only");
    if (c->frame != 0) return fail(VRT_E_STATE, "set the stripes before the first vrt_accumulate");
    c->stripe_rows = stripe_rows; c->stripe_parts = n_parts; c->stripe_part = part;
    return VRT_OK;
}
int vrt_set_instrumented(vrt_ctx* c, int on) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->instrumented = on != 0;
    c->count_as_timed = on == 2;
    c->render_blocks = 0;
    return VRT_OK;
}

int vrt_set_reference_indexing(vrt_ctx* c, int on) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->ref_oob = on != 0;
    c->render_blocks = 0;   // other kernel instantiations (the instrumented ones carry the code): the grid is sized again
    return VRT_OK;
}

int vrt_prepare(vrt_ctx* c) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    c->main_dirty = true;
    HIP_TRY(launch_prepare(c->stream, c->cfg.grid_res, c->d_mat, c->d_rgb, c->d_grid, c->d_l0, c->d_l1, c->d_l2, c->d_l3, c->d_l0c, c->d_l0c_base, c->d_cull));
    {
        const float off[8] = {-1e30f, -1e30f, -1e30f, 1e30f, 1e30f, 1e30f, 0.0f, 0.0f};   // nothing is culled
        HIP_TRY(hipMemcpyAsync(c->d_cull + 8, off, sizeof(off), hipMemcpyHostToDevice, c->stream));
        float box[8];
        HIP_TRY(hipMemcpyAsync(box, c->d_cull, sizeof(box), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));   // (source and destination are on this stack frame)
        c->cull_active = box[6] != 0.0f;
        // at least half of the 4x4x4 bricks hold a voxel: a dense grid (shadow rays end after a step or two: launch_render_pool)
        c->dense_grid = box[7] >= 0.5f;
        if (c->knobs.dense >= 0) c->dense_grid = c->knobs.dense != 0;   // A/B
    }
    if (c->scene.use_physical_sky == 1) {
        SkyPrecompute sp = make_sky(c);
        f3 sd, sc_;
        float cm;
        sun_of(c, sd, sc_, cm);
        size_t ns = (size_t)c->cfg.sky_res * c->cfg.sky_res * 3 * sizeof(float);
        HIP_TRY(launch_sky_prepare(c->stream, sp, sd, sc_, cm));
        HIP_TRY(hipMemsetAsync(c->d_sky_scat, 0, ns, c->stream));   // pathtracer.py:322-323
        HIP_TRY(hipMemsetAsync(c->d_sky_trans, 0, ns, c->stream));
        c->cloud_pass = 0;
    }
    c->prepared = true;
    return VRT_OK;
}
int vrt_sky_accumulate_clouds(vrt_ctx* c, int max_samples) {
    if (!c || max_samples <= 0) return fail(VRT_E_INVALID, "bad argument");
    if (!c->prepared || c->scene.use_physical_sky != 1) return fail(VRT_E_STATE, "needs vrt_prepare with use_physical_sky");
    HIP_TRY(hipSetDevice(c->device));
    f3 sd, sc_;
    float cm;
    sun_of(c, sd, sc_, cm);
    c->main_dirty = true;
    HIP_TRY(launch_sky_clouds(c->stream, make_sky(c), sd, sc_, cm, max_samples, c->cloud_pass, 0, c->cfg.sky_res));
    c->cloud_pass++;
    return VRT_OK;
}
// One cloud pass over the table columns of slice `slice_idx` of `max_slices` only (the split of vrt_sky_compute_slice,
// atmos.py:162).  A texel's passes depend on no other texel, so ranks that each own a slice run this max_samples times,
// then vrt_sky_compute_slice on their slice, and exchange columns (vrt_sky_table_io): voxel_rt2_amd/parallel.py.
int vrt_sky_accumulate_clouds_slice(vrt_ctx* c, int max_samples, int slice_idx, int max_slices) {
    if (!c || max_samples <= 0 || max_slices <= 0 || slice_idx < 0 || slice_idx >= max_slices) return fail(VRT_E_INVALID, "bad argument");
    if (!c->prepared || c->scene.use_physical_sky != 1) return fail(VRT_E_STATE, "needs vrt_prepare with use_physical_sky");
    HIP_TRY(hipSetDevice(c->device));
    f3 sd, sc_;
    float cm;
    sun_of(c, sd, sc_, cm);
    if (c->cfg.sky_res % max_slices != 0) return fail(VRT_E_INVALID, "sky_res is not a multiple of max_slices: trailing table columns would belong to no slice");
    const int w = c->cfg.sky_res / max_slices;
    c->main_dirty = true;
    HIP_TRY(launch_sky_clouds(c->stream, make_sky(c), sd, sc_, cm, max_samples, c->cloud_pass, w * slice_idx, w * (slice_idx + 1)));
    c->cloud_pass++;
    return VRT_OK;
}
// Copy table columns [u0, u1) between the library's sky tables and caller-owned DEVICE memory (f32[u1-u0][R][3], the table's
// own layout: a column slice is one contiguous block).  which = VRT_BUF_SKY_SCATTERING / VRT_BUF_SKY_TRANSMITTANCE;
// to_library = 0 reads, 1 writes.  Queued on the context's stream.
int vrt_sky_table_io(vrt_ctx* c, int which, int u0, int u1, void* device_ptr, int to_library) {
    if (!c || !device_ptr) return fail(VRT_E_INVALID, "null argument");
    if (c->cfg.sky_res <= 0) return fail(VRT_E_STATE, "no sky tables");
    if (which != VRT_BUF_SKY_SCATTERING && which != VRT_BUF_SKY_TRANSMITTANCE) return fail(VRT_E_INVALID, "not a sky table");
    if (u0 < 0 || u1 > c->cfg.sky_res || u1 <= u0) return fail(VRT_E_INVALID, "column range outside the table");
    HIP_TRY(hipSetDevice(c->device));
    float* table = which == VRT_BUF_SKY_SCATTERING ? c->d_sky_scat : c->d_sky_trans;
    const size_t col = (size_t)c->cfg.sky_res * 3 * sizeof(float);
    char* lib = (char*)table + (size_t)u0 * col;
    const size_t bytes = (size_t)(u1 - u0) * col;
    c->main_dirty = true;
    if (to_library) HIP_TRY(hipMemcpyAsync(lib, device_ptr, bytes, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(hipMemcpyAsync(device_ptr, lib, bytes, hipMemcpyDeviceToDevice, c->stream));
    return VRT_OK;
}
int vrt_sky_compute_slice(vrt_ctx* c, int slice_idx, int max_slices) {
    if (!c || max_slices <= 0 || slice_idx < 0 || slice_idx >= max_slices) return fail(VRT_E_INVALID, "bad slice");
    if (!c->prepared || c->scene.use_physical_sky != 1) return fail(VRT_E_STATE, "needs vrt_prepare with use_physical_sky");
    HIP_TRY(hipSetDevice(c->device));
    f3 sd, sc_;
    float cm;
    sun_of(c, sd, sc_, cm);
    int w = c->cfg.sky_res / max_slices;  // atmos.py:162: floor division -- trailing columns (3840 / 32 has none) belong to no slice, as in the reference
    if (w == 0) return VRT_OK;   // fewer columns than slices: every slice is empty
    c->main_dirty = true;
    HIP_TRY(launch_sky_slice(c->stream, make_sky(c), sd, sc_, cm, w * slice_idx, w * (slice_idx + 1)));
    return VRT_OK;
}

static int record(vrt_ctx* c, int kind, hipEvent_t* a, hipEvent_t* b) {
    HIP_TRY(hipEventCreate(a));
    HIP_TRY(hipEventCreate(b));
    c->pending.push_back(EventPair{*a, *b, kind});
    return VRT_OK;
}

// How deep a launch of `g` samples wants the pipeline.  A launch lasts at least as long as its deepest path takes alone (about
// 0.2 ms at 8 bounces), whatever its size, and a workgroup slot its wave has left stays empty until the NEXT launch may start.
// A launch of every slot can only be followed when it starts to drain (two in flight).  Launches of half the slots each follow
// one another at half that distance -- two run at full strength while a third drains and a fourth waits its turn.  Measured
// (profiles/r02_pipeline_depth.txt): 1080p x 4 samples +3.7 %, half of it +5 %, an eighth (one rank's rows of an 8-GPU
// run) +24 %; thirds and quarters of the slots are worse again; a 4K frame (33 M items a launch) loses 0-7 % and keeps
// the two-deep pipeline.
// Deeper still for the smaller launches -- one rank's rows of an 8-GPU split of 1080p are 1 M items: eight launches of a
// quarter of the slots each (+7.5 % on those rows; with the timers thinned out, below: +2 % on half a frame of 4.1 M items,
// +17 % on its cheap upper 480 rows, -2 % on a whole frame, -9 % on the sun-lit one: the limit is 4.5 M items).  Each render stream wants a
// hardware queue of its own (two streams on one queue serialise), so only where the runtime was started with sixteen
// (GPU_MAX_HW_QUEUES, which voxel_rt2_amd/_lib.py sets unless the user has).
// VRT_DEEP_ITEMS / VRT_DEEPER_ITEMS (development build): largest launch (pixels x fused samples) of each kind; VRT_STREAMS /
// VRT_GRID_DIV override.
static void pipeline_mode_for(const vrt_ctx* c, int g, bool heavy, int* n_streams, int* grid_div) {
    // (heavy: the dense-grid kernel -- six rays a path instead of two: an item is about twice the work, a rank's 4.1 M items of an
    // 8-way split of a dense 4K frame lose 10 % in the eight-deep pipeline that the same number of S1's items gain 2-17 % from)
    const size_t items = (size_t)c->cfg.width * owned_rows(c) * (size_t)g;
    const bool deep = items <= (size_t)c->knobs.deep_items;                                         // 12 M
    const bool deeper = deep && items * (heavy ? 2u : 1u) <= (size_t)c->knobs.deeper_items && c->knobs.hw_queues >= 16;  // 4.5 M
    *n_streams = deeper ? 8 : deep ? 4 : 2;
    *grid_div = deeper ? 4 : deep ? 2 : 1;
    if (c->knobs.streams) *n_streams = c->knobs.streams;
    if (c->knobs.grid_div) *grid_div = c->knobs.grid_div;
}
// Streams, copies and events for a pipeline `want` launches deep (what a shallower one already has is kept).
static bool grow_pipeline(vrt_ctx* c, int want) {
    const size_t n = c->npix;
    bool ok = true;
    for (int s = 0; s < want && ok; s++) {   // stream s, copy s + 1 (alt_*[s]) and, beyond the first, a pool scratch of its own
        if (c->rstream[s]) continue;
        ok = dalloc(&c->alt_multi_d[s], n * VRT_MAX_FUSED) == hipSuccess && dalloc(&c->alt_spec_planes[s], n * VRT_MAX_FUSED) == hipSuccess &&
             dalloc(&c->alt_refl_planes[s], n * VRT_MAX_FUSED) == hipSuccess && dalloc(&c->alt_gb_pos[s], n) == hipSuccess &&
             dalloc(&c->alt_gb_mat[s], n) == hipSuccess;
        if (ok && s > 0)
            ok = hipMalloc((void**)&c->alt_pool_scratch[s - 1], pool_scratch_bytes(c->cfg.grid_res, c->cfg.use_restir != 0, c->render_blocks, c->render_blocks_d12)) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&c->rstream[s], hipStreamNonBlocking) == hipSuccess;   // (last: a stream that exists has everything)
    }
    for (int s = 0; s < want + 1 && ok; s++)
        if (!c->ev_r[s])
            ok = hipEventCreateWithFlags(&c->ev_r[s], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&c->ev_t[s], hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}
// The pipeline for a launch of g samples; false (and never tried again) if its streams and copies cannot be had.  The depth
// follows the launch: a context whose caller changes habit (one sample per call, then four) is drained once and goes on in
// the other mode -- the set numbering and the gate distance of the two modes do not mix.
static bool ensure_overlap(vrt_ctx* c, int g, bool heavy) {
    if (c->overlap_failed) return false;
    int ns = 0, gd = 0;
    pipeline_mode_for(c, g, heavy, &ns, &gd);
    if (c->overlap_ready) {
        if (ns == c->n_streams && gd == c->grid_div) return true;
        if (!grow_pipeline(c, ns)) return true;   // no memory for the other mode: this one goes on
        if (sync_guarded(c, c->stream) != VRT_OK) return true;
        for (int s = 0; s < VRT_MAX_STREAMS; s++) if (c->rstream[s]) (void)hipStreamSynchronize(c->rstream[s]);
        if (sync_guarded(c, c->stream) != VRT_OK) return true;   // (the temporal passes behind those launches)
        (void)hipGetLastError();
        for (int s = 0; s < VRT_MAX_SETS; s++) c->ev_t_valid[s] = false;   // every pass has completed
        c->n_streams = ns;
        c->grid_div = gd;
        c->mode_switches++;
        return true;
    }
    bool ok = grow_pipeline(c, ns);
    c->n_streams = ns;
    c->grid_div = gd;
    ok = ok && hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming) == hipSuccess;
    int can_wait = 0;
    bool want_gate = ok;
    want_gate = want_gate && c->knobs.drain_gate;   // off: launches overlap all the same, only queue earlier
    c->drain_signal = nullptr;
    if (want_gate && hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, c->device) == hipSuccess && can_wait &&
        hipExtMallocWithFlags((void**)&c->drain_signal, 8, hipMallocSignalMemory) == hipSuccess) {
        hipPointerAttribute_t at;
        bool usable = hipPointerGetAttributes(&at, c->drain_signal) == hipSuccess && at.hostPointer == (void*)c->drain_signal;  // the host must be able to release it
        if (usable) { __atomic_store_n(c->drain_signal, 0u, __ATOMIC_RELEASE); usable = gate_self_test(c); }
        if (!usable) { (void)hipGetLastError(); hipFree(c->drain_signal); c->drain_signal = nullptr; }
    }
    (void)hipGetLastError();
    if (!ok) { (void)hipGetLastError(); c->overlap_failed = true; return false; }
    c->overlap_ready = true;
    return true;
}

// After a failed queue operation inside vrt_accumulate: nothing may be left waiting for a launch that did not happen, and
// the context must be usable again.  The rotation state (buffer roles, frame index, pipeline slot) only advances at the end
// of an iteration whose launches were all queued, so a context without ReSTIR on the overlapped or plain schedule is back at
// the pass before the failed one.  NOT so the accumulated history of a fused ReSTIR call: its per-sample reuse and
// accumulation passes ping-pong the histories in place, so the passes queued before the failure have advanced them while
// the roles were rolled back -- after a failed call on a ReSTIR context the caller must vrt_reset().
static void abort_pipeline(vrt_ctx* c) {
    const std::string keep = g_err;
    release_gate(c);
    c->drain_signalled = false;
    for (int s = 0; s < VRT_MAX_STREAMS; s++) if (c->rstream[s]) (void)hipStreamSynchronize(c->rstream[s]);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->fetch_stream) (void)hipStreamSynchronize(c->fetch_stream);   // asynchronous fetches queued before the failure have completed
    c->cbuf_read_pending[0] = c->cbuf_read_pending[1] = false;
    c->hdr_targets_written = c->hdr_targets_committed;   // a tile handed out for a pass that was never queued is handed out again
    (void)hipGetLastError();
    resolve_events(c);
    // the work heads rotate with the launch number and each launch zeroes the set eight launches ahead: a launch that did not
    // run leaves a used set behind -- nothing is in flight now, so all of them start clean
    (void)hipMemset(c->d_work, 0, VRT_WORK_SETS * VRT_WORK_HEADS * VRT_WORK_HEAD_STRIDE * sizeof(unsigned));
    (void)hipStreamSynchronize(nullptr);   // (the fill runs on the NULL stream: see dalloc)
    (void)hipGetLastError();
    for (int s = 0; s < VRT_MAX_SETS; s++) c->ev_t_valid[s] = false;
    c->main_dirty = true;
    c->render_blocks = 0;   // residency and scratch are looked at again
    g_err = keep;
}


// rows of the HDR frame a pass also writes to the caller's ring (vrt_set_hdr_targets)
static f3* next_hdr_target(vrt_ctx* c) {
    if (c->hdr_targets.empty()) return nullptr;
    return (f3*)c->hdr_targets[(size_t)(c->hdr_targets_written++ % c->hdr_targets.size())];
}
static int wait_cbuf_readers(vrt_ctx* c, int b) {   // an asynchronous fetch may still be reading the HDR buffer a pass is about to write
    if (c->cbuf_read_pending[b]) { HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_cbuf_read[b], 0)); c->cbuf_read_pending[b] = false; }
    return VRT_OK;
}

static int accumulate_impl(vrt_ctx* c, int n_samples);

int vrt_accumulate(vrt_ctx* c, int n_samples) {
    if (!c || n_samples < 0) return fail(VRT_E_INVALID, "bad argument");
    if (!c->prepared) return fail(VRT_E_STATE, "vrt_prepare has not run since the last voxel upload");
    if (!c->have_cam) return fail(VRT_E_STATE, "vrt_set_camera has not been called");
    HIP_TRY(hipSetDevice(c->device));
#if defined(VRT_HOST_PROFILE)
    const double t_acc = prof_now();
    const int rc = accumulate_impl(c, n_samples);
    { auto& p_ = g_prof["(the whole of accumulate_impl)"]; p_.first += prof_now() - t_acc; p_.second++; }
#else
    const int rc = accumulate_impl(c, n_samples);
#endif
    if (rc != VRT_OK) abort_pipeline(c);
    else c->hdr_targets_committed = c->hdr_targets_written;
    return rc;
}

static int accumulate_impl(vrt_ctx* c, int n_samples) {
    // (the instrumented instantiations are the ones that carry the reference's out-of-grid reading: vrt_set_reference_indexing)
    const bool restir = c->cfg.use_restir != 0, instr = c->instrumented || c->ref_oob;
    if (c->render_blocks == 0) {
        // Two schedules of the same per-path code: the fused one (a lane owns a path, vrt_path.h) and the pooled one
        // (a wave owns a pool of paths in LDS and works stage by stage, vrt_pool.h).  The pooled kernel packs pixel
        // coordinates in 12 bits and the depth in 4, so contexts outside that use the fused one (ReSTIR runs on either:
        // k_render_pool_restir keeps the reconnection state in the per-slot scratch line).  VRT_RENDER=fused selects the
        // fused kernel everywhere (A/B measurements, tests).
        bool pooled = c->cfg.width <= 4096 && c->cfg.height <= 4096 && c->cfg.max_depth <= 15;
        if (c->knobs.render == -2) return fail(VRT_E_INVALID, "VRT_RENDER must be 'fused' or 'pool'");
        if (c->knobs.render == 0) pooled = false;
        int per_cu = 0;
        if (pooled) HIP_TRY(query_render_pool_residency(c->cfg.grid_res, restir, instr, &per_cu));
        else HIP_TRY(query_render_residency(c->cfg.grid_res, restir, instr, &per_cu));
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 8) per_cu = 8;
        int cus = c->n_cu - c->reserved_cus;
        if (cus < 8) cus = c->n_cu < 8 ? c->n_cu : 8;
        c->render_blocks = per_cu * cus;   // (abort_pipeline zeroes it again if an allocation below fails)
        c->render_blocks_d12 = 0;
        if (pooled && !restir) {
            int per_cu12 = 0;
            HIP_TRY(query_render_pool_dense12_residency(c->cfg.grid_res, instr, &per_cu12));
            c->render_blocks_d12 = (per_cu12 < 1 ? 1 : per_cu12) * cus;
        }
        c->pooled = pooled;
        if (pooled) {
            HIP_TRY(sync_guarded(c, c->stream));
            if (c->d_pool_scratch) { HIP_TRY(hipFree(c->d_pool_scratch)); c->d_pool_scratch = nullptr; }
            HIP_TRY(hipMalloc((void**)&c->d_pool_scratch, pool_scratch_bytes(c->cfg.grid_res, c->cfg.use_restir != 0, c->render_blocks, c->render_blocks_d12)));
            if (c->overlap_ready) {  // the other render streams' scratch follows
                for (int s = 0; s < VRT_MAX_STREAMS - 1; s++) {   // (every stream the context has, whatever the depth in use)
                    if (!c->rstream[s + 1]) continue;
                    if (c->alt_pool_scratch[s]) { HIP_TRY(hipFree(c->alt_pool_scratch[s])); c->alt_pool_scratch[s] = nullptr; }
                    HIP_TRY(hipMalloc((void**)&c->alt_pool_scratch[s], pool_scratch_bytes(c->cfg.grid_res, c->cfg.use_restir != 0, c->render_blocks, c->render_blocks_d12)));
                }
            }
        }
    }
    // The samples of one call share camera, jitter and scene; with a still camera at full render scale and ReSTIR
    // off they only differ in their random streams, so up to VRT_MAX_FUSED of them go through ONE k_render launch
    // (work items = pixels x samples: 4x the parallelism per launch, one tail instead of four) into consecutive
    // colour planes, and ONE k_temporal launch advances the running means sample by sample in registers.
    const int max_fused = c->knobs.max_fused;
    // With ReSTIR on the samples fuse in the RENDER launch all the same (one reservoir plane per sample beside the colour
    // planes; the pooled kernel only): spatial reuse and accumulation then run sample by sample over the planes, as the
    // reference runs them -- the reuse pass of a sample reads nothing an earlier sample's pass wrote.  VRT_FUSE_RESTIR=0: off.
    const bool fuse_restir = c->pooled && c->knobs.fuse_restir;
    const bool can_fuse = (!restir || fuse_restir) && c->cam.camera_is_moving == 0 && c->cam.render_scale == 1.0f;
    // A persistent render launch ends in a tail: the last paths of every wave bounce on at low occupancy (about 0.16 ms
    // of a 1.5 ms launch at 1080p).  Fused launches of the pooled kernel are therefore OVERLAPPED: launch k+1 goes to
    // the next of n_streams render streams and writes the next copy of the colour planes / g-buffer while launch k drains
    // and its temporal pass (main stream, waits for launch k only) runs.  With n_streams + 1 copies launch k+n_streams+1
    // reuses launch k's and waits for temporal pass k, so render launches follow each other without a gap and the temporal
    // passes run beside them (ensure_overlap: how deep).  Results are unchanged; VRT_OVERLAP=0 turns it off.
    const bool may_overlap = c->pooled && can_fuse && !restir && c->knobs.overlap;
    for (int done = 0; done < n_samples;) {
        int g = (can_fuse && n_samples - done > 1) ? (n_samples - done < max_fused ? n_samples - done : max_fused) : 1;
        // One-sample launches are pipelined like fused ones (the reference's own loop is one sample per call: scene.py:177,
        // 255-256): they render into plane 0 of the rotating copies instead of the HDR buffer.  VRT_OVERLAP_SINGLE=0: only fused ones.
        bool want_overlap = may_overlap;
        if (g == 1 && !c->knobs.overlap_single) want_overlap = false;
        if ((g > 1 || want_overlap) && !c->d_multi_d) {
            if (dalloc(&c->d_multi_d, c->npix * VRT_MAX_FUSED) != hipSuccess) { (void)hipGetLastError(); c->d_multi_d = nullptr; g = 1; want_overlap = false; }  // no memory: one launch per sample
        }
        const bool heavy = c->pooled && c->render_blocks_d12 > 0 && pool_uses_dense12(c->cfg.grid_res, restir, c->dense_grid, make_frame_params(c));
        const bool overlapped = want_overlap && ensure_overlap(c, g, heavy);
        const bool planes = g > 1 || overlapped;   // the launch writes colour planes of its own, not the HDR buffer
        const int set = overlapped ? (int)(c->pipe_seq % (unsigned)(c->n_streams + 1)) : 0;
        const int lane_of = (int)(c->pipe_seq % (unsigned)c->n_streams);  // which render stream (and pool scratch): consecutive launches take turns
        hipStream_t rs = overlapped ? c->rstream[lane_of] : c->stream;
        // A launch of half the slots only pays with other launches beside it: one that finds the pipeline empty (the caller
        // fetches every frame, or this is the first of a run) takes every slot like a launch that is not overlapped.
        // The same holds for a pipeline that is nearly empty -- a caller that presents every frame and waits for frame k - 1
        // before it queues frame k + 1 keeps one or two launches in flight, which as half-size launches leave half the chip idle:
        // fewer than two launches still running means every slot.
        bool lone = !overlapped;
        if (overlapped && c->grid_div > 1) {
            int running = 0;
            const unsigned n_sets = (unsigned)c->n_streams + 1u;
            for (unsigned back = 1; back <= 3u && back <= c->pipe_seq; back++)
                if (hipEventQuery(c->ev_r[(c->pipe_seq - back) % n_sets]) == hipErrorNotReady) running++;
            (void)hipGetLastError();   // (hipErrorNotReady is the expected answer)
            lone = running < c->knobs.full_below;   // every slot while fewer than two launches are still running
        }
        if (overlapped) {
            if (c->main_dirty) {  // uploads / prepare / sky kernels queued on the main stream come first
                HIP_TRY(hipEventRecord(c->ev_main, c->stream));
                for (int s = 0; s < c->n_streams; s++) HIP_TRY(hipStreamWaitEvent(c->rstream[s], c->ev_main, 0));
                c->main_dirty = false;
            }
            if (c->ev_t_valid[set]) HIP_TRY(hipStreamWaitEvent(rs, c->ev_t[set], 0));  // the pass that last read this copy
            // dispatch when the launch whose workgroup slots this one will take starts to drain: the one before it, or with
            // launches of half the slots the one before that (the signal carries the number + 1 of the latest launch draining)
            // -- unless the one before it took EVERY slot (a lone launch): then that one has to drain first
            // and never for a launch OLDER than the last one that took every slot: until that one drains there is no slot at all
            const unsigned back = c->prev_launch_full ? 1u : (unsigned)(c->grid_div + c->knobs.gate_extra);
            unsigned target = c->launch_seq + 1u > back ? c->launch_seq + 1u - back : 0u;
            if (target < c->last_full_seq) target = c->last_full_seq;
            if (c->drain_signal && c->drain_signalled && target > 0u)
                HIP_TRY(hipStreamWaitValue32(rs, c->drain_signal, target, hipStreamWaitValueGte, 0xFFFFFFFFu));
        } else if (c->last_set != 0) {
            // back to the single copy: whoever reads pixels this launch does not write (moving camera at half render
            // scale) expects the last sample of the last launch in the canonical buffers
            const size_t last = (size_t)(VRT_MAX_FUSED - 1) * c->npix;
            const int a = c->last_set - 1;
            HIP_TRY(hipMemcpyAsync(c->d_color_s, c->alt_spec_planes[a] + last, c->npix * sizeof(f3), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_gb_refl, c->alt_refl_planes[a] + last, c->npix * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_gb_pos, c->alt_gb_pos[a], c->npix * sizeof(f3), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_gb_mat, c->alt_gb_mat[a], c->npix * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            c->last_set = 0;
        }
        if (c->pending.size() > 192) resolve_completed(c);
        // (an asynchronous fetch may still be reading the HDR buffer this launch renders into; the passes below check theirs)
        if (!planes && wait_cbuf_readers(c, c->cidx) != VRT_OK) return VRT_E_DEVICE;
        FrameParams fp = make_frame_params(c);
        SceneData sc = make_scene_data(c);
        PixelBuffers out;
        f3* rt = c->d_cbuf[c->cidx];       // render target: holds the previous HDR outside the render area
        // specular colour and raw reflection depth: VRT_MAX_FUSED planes each, the LAST plane being the buffer the
        // reference knows (color_buffer_specular, gbuff_depth_reflection); a fused launch ends on it, so whatever
        // later reads stale pixels (moving camera at half render scale) finds the last sample there, as in the reference
        const size_t last_plane = (size_t)(VRT_MAX_FUSED - 1) * c->npix;
        out.color_d = planes ? (set ? c->alt_multi_d[set - 1] : c->d_multi_d) : rt;
        out.color_s = (set ? c->alt_spec_planes[set - 1] + last_plane : c->d_color_s) - (size_t)(g - 1) * c->npix;
        out.gb_refl_depth = (set ? c->alt_refl_planes[set - 1] + last_plane : c->d_gb_refl) - (size_t)(g - 1) * c->npix;
        out.sample_stride = g > 1 ? (int)c->npix : 0;
        out.gb_normal = c->d_gb_normal[c->cur]; out.gb_depth = c->d_gb_depth[c->cur];
        if (c->cam.render_scale != 1.0f) {
            // a pass that renders part of the frame: the reference's g-buffer is ONE array, so the pixels it leaves out
            // keep what the last pass wrote -- the rotating copy starts as a copy of the last one
            HIP_TRY(hipMemcpyAsync(c->d_gb_normal[c->cur], c->last_gb_normal, c->npix * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->d_gb_depth[c->cur], c->last_gb_depth, c->npix * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        out.gb_position = set ? c->alt_gb_pos[set - 1] : c->d_gb_pos; out.gb_mat = set ? c->alt_gb_mat[set - 1] : c->d_gb_mat;
        out.reservoir = restir ? c->d_res[0] - (size_t)(g - 1) * c->npix : nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        const unsigned seq = c->launch_seq++;
        // Timers (two events around each kernel, vrt_stats' device times) cost a short step its rate: the barrier packets they put
        // around the kernel sit in the chain from one launch's drain to the next one's first wave -- a fixed 30-40 us of a step,
        // 3-25 % of the steps of 0.15-0.3 ms that a rank's rows of an 8-GPU split or the reference's one-sample calls take, nothing
        // of a 1 ms step.  What a step will take is not known here, its size is: launches of the deep pipelines (up to 12 M work
        // items, ReSTIR off) carry timers one time in eight; vrt_get_stats scales the timed launches' sum to all of them.
        const size_t launch_items = (size_t)c->cfg.width * owned_rows(c) * (size_t)g;
        const unsigned every = c->knobs.time_every > 0 ? (unsigned)c->knobs.time_every : ((!restir && launch_items <= (size_t)c->knobs.deep_items) ? 8u : 1u);
        const bool timed = c->since_reset++ % every == 0u;   // this launch and its passes carry timers
        c->passes_n[0]++;
        if (timed) {
            if (record(c, 0, &a, &b) != VRT_OK) return VRT_E_DEVICE;
            HIP_TRY(hipEventRecord(a, rs));
        }
        // test hook (tests/test_gpu_pipeline.py): launch number VRT_TEST_FAIL_LAUNCH (read at vrt_create) is reported as failed instead of queued
        if (c->knobs.fail_launch >= 0 && (unsigned)c->knobs.fail_launch == seq) return fail(VRT_E_DEVICE, "injected launch failure (VRT_TEST_FAIL_LAUNCH)");
        PrimaryRecord* prim = nullptr;  // fused samples share their camera rays through this table (vrt_pool.h)
        if (c->pooled && g > 1 && (!instr || c->count_as_timed)) {  // counting the reference's work: every camera ray is walked
            const int which = overlapped ? lane_of : 0;
            if (!c->d_prim_cache[which] && dalloc(&c->d_prim_cache[which], c->npix) != hipSuccess) { (void)hipGetLastError(); c->d_prim_cache[which] = nullptr; }
            prim = c->d_prim_cache[which];
        }
        const bool d12 = c->pooled && c->render_blocks_d12 > 0 && pool_uses_dense12(c->cfg.grid_res, restir, c->dense_grid, fp);
        const int all_blocks = d12 ? c->render_blocks_d12 : c->render_blocks;
        const int blocks = lone ? all_blocks : (all_blocks / c->grid_div + 7) & ~7;  // whole rounds of the 8 XCDs
        if (c->pooled) HIP_TRY(launch_render_pool(rs, c->cfg.grid_res, restir, instr, blocks, fp, sc, out, c->d_work, seq, g, (overlapped && lane_of) ? c->alt_pool_scratch[lane_of - 1] : c->d_pool_scratch, c->drain_signal, prim, culling(c), c->dense_grid, d12));
        else HIP_TRY(launch_render(rs, c->cfg.grid_res, restir, instr, c->render_blocks, fp, sc, out, c->d_work, seq, g, c->knobs.chunk));
        c->drain_signalled = c->pooled && c->drain_signal != nullptr;
        c->prev_launch_full = blocks == all_blocks;
        if (c->prev_launch_full && c->pooled) c->last_full_seq = seq + 1u;
        if (timed) HIP_TRY(hipEventRecord(b, rs));
        if (overlapped) {
            HIP_TRY(hipEventRecord(c->ev_r[set], rs));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_r[set], 0));
            c->last_render_set = set;
        }
        // ReSTIR: spatial reuse and accumulation sample by sample over the planes of the launch (one pass with one sample)
        const int passes = restir ? g : 1;
        int hist = c->hist_in, ci = c->cidx;   // (the context's own copies only move once every launch of the iteration is queued)
        for (int s = 0; s < passes; s++) {
            const size_t off = (size_t)s * (size_t)out.sample_stride;   // this sample's plane (ReSTIR; stride 0 with one sample)
            FrameParams fps = fp;
            fps.frame = fp.frame + (uint32_t)s;
            const f3* cd = out.color_d;
            const f3* cs = out.color_s;
            if (restir) {
                GrisBuffers gb;
                gb.color_d_in = out.color_d + off; gb.color_s_in = out.color_s + off; gb.color_d_out = c->d_color_d2; gb.color_s_out = c->d_color_s2;
                gb.gb_normal = out.gb_normal; gb.gb_depth = out.gb_depth; gb.gb_mat = out.gb_mat;
                gb.res_in = out.reservoir + off; gb.res_out = c->d_res[1];
                gb.geo = c->d_gris_geo; gb.src = c->d_gris_src; gb.tst = c->d_gris_tst; gb.mats_x = c->d_mats_x;
                int g0 = c->own0 - 2 < c->buf0 ? c->buf0 : c->own0 - 2, g1 = c->own1 + 2 > c->buf1 ? c->buf1 : c->own1 + 2;
                if (timed) {
                    if (record(c, 2, &a, &b) != VRT_OK) return VRT_E_DEVICE;
                    HIP_TRY(hipEventRecord(a, c->stream));
                }
                HIP_TRY(launch_gris(c->stream, c->cfg.grid_res, instr, fps, sc, gb, g0, g1));
                c->passes_n[2]++;
                if (timed) HIP_TRY(hipEventRecord(b, c->stream));
                cd = c->d_color_d2;
                cs = c->d_color_s2;
            }
            TemporalBuffers tb;
            tb.color_d = cd; tb.color_s = cs;
            tb.gb_normal = out.gb_normal; tb.gb_depth = out.gb_depth; tb.gb_mat = out.gb_mat;
            tb.gb_refl_raw = out.gb_refl_depth + (restir ? off : 0); tb.gb_refl_filtered = c->d_gb_refl_f;
            tb.hist_d_in = c->d_hist_d[hist]; tb.hist_d_out = c->d_hist_d[hist ^ 1];
            tb.hist_s_in = c->d_hist_s[hist]; tb.hist_s_out = c->d_hist_s[hist ^ 1];
            // (the "previous" g-buffer of a launch's later samples is the launch's own: a launch per sample would have written it again)
            tb.prev_normal = s == 0 ? c->last_gb_normal : c->d_gb_normal[c->cur];
            tb.prev_depth = s == 0 ? c->last_gb_depth : c->d_gb_depth[c->cur];
            tb.hdr = c->d_cbuf[ci ^ 1];
            tb.sample_stride = restir ? 0 : out.sample_stride;
            tb.prev_view = c->prev_view; tb.prev_proj = c->prev_proj;
            f3* const tile = (done + g >= n_samples && s == passes - 1) ? next_hdr_target(c) : nullptr;   // the pass that completes the call
            if (wait_cbuf_readers(c, ci ^ 1) != VRT_OK) return VRT_E_DEVICE;
            if (timed) {
                if (record(c, 1, &a, &b) != VRT_OK) return VRT_E_DEVICE;
                HIP_TRY(hipEventRecord(a, c->stream));
            }
            tb.tile = tile;
            tb.tile_row0 = c->own0;
            if (c->stripe_rows) {   // one launch over the context's own rows, stripe after stripe (the kernel maps them: k_temporal)
                const int n_own = (int)owned_ranges(c).size() * c->stripe_rows;   // (a last stripe cut short by the frame's edge is cut there)
                HIP_TRY(launch_temporal(c->stream, fps, tb, 0, n_own, g));
            } else {
                HIP_TRY(launch_temporal(c->stream, fps, tb, c->own0, c->own1, restir ? 1 : g));
            }
            if (timed) HIP_TRY(hipEventRecord(b, c->stream));
            c->passes_n[1]++;
            hist ^= 1; ci ^= 1;   // pathtracer.py:1298-1303 copy loop == pointer swaps, once per accumulation pass
        }
        if (overlapped) {
            HIP_TRY(hipEventRecord(c->ev_t[set], c->stream));
            c->ev_t_valid[set] = true;
            c->pipe_seq += 1;
        } else if (c->overlap_ready) {  // this pass used copy 0 and the single-copy buffers: later overlapped launches wait for it
            for (int s = 0; s < c->n_streams + 1; s++) { HIP_TRY(hipEventRecord(c->ev_t[s], c->stream)); c->ev_t_valid[s] = true; }
        }
        c->last_set = set;
        c->hist_in = hist;
        c->prev_gb = c->cur;
        c->last_gb_normal = c->d_gb_normal[c->cur]; c->last_gb_depth = c->d_gb_depth[c->cur];
        c->cur = (c->cur + 1) % VRT_GB_ROT;
        c->cidx = ci;
        c->frame += (uint32_t)g;
        c->stats.path_samples += (uint64_t)g * (uint64_t)c->cfg.width * (uint64_t)owned_rows(c);
        done += g;
    }
    return VRT_OK;
}

int vrt_reset(vrt_ctx* c) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    for (int s = 0; s < 2; s++) {
        HIP_TRY(hipMemsetAsync(c->d_hist_d[s], 0, c->npix * sizeof(f4), c->stream));
        HIP_TRY(hipMemsetAsync(c->d_hist_s[s], 0, c->npix * sizeof(f4), c->stream));
    }
    return VRT_OK;
}
int vrt_end_frame(vrt_ctx* c) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    memcpy(c->prev_view.m, c->cam.view, 64);
    memcpy(c->prev_proj.m, c->cam.proj, 64);
    c->have_prev = true;
    return VRT_OK;
}
int vrt_sync(vrt_ctx* c) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_guarded(c, c->stream));  // every render launch on the render streams has its temporal pass here
    return VRT_OK;
}

// copy rows [own0, own1) of a per-pixel device buffer into a full-image host array (other rows zero)
static int fetch_rows(vrt_ctx* c, const void* dbuf, size_t elem, void* out) {
    HIP_TRY(hipSetDevice(c->device));
    const size_t W = c->cfg.width;
    if (c->own0 != 0 || c->own1 != c->cfg.height || c->stripe_rows) memset(out, 0, (size_t)c->cfg.height * W * elem);   // a shard: the other rows are zero
    for (const auto& rr : owned_ranges(c)) {
        const char* src = (const char*)dbuf + (size_t)(rr.first - c->buf0) * W * elem;
        char* dst = (char*)out + (size_t)rr.first * W * elem;
        HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(rr.second - rr.first) * W * elem, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(sync_guarded(c, c->stream));
    return VRT_OK;
}
int vrt_fetch_hdr(vrt_ctx* c, float* out) {
    if (!c || !out) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    return fetch_rows(c, c->d_cbuf[c->cidx], sizeof(f3), out);
}
int vrt_fetch_hdr_device(vrt_ctx* c, void* device_ptr) {
    if (!c || !device_ptr) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t W = c->cfg.width;
    size_t done = 0;   // (a striped context's rows: one stripe after the other)
    for (const auto& rr : owned_ranges(c)) {
        const char* src = (const char*)c->d_cbuf[c->cidx] + (size_t)(rr.first - c->buf0) * W * sizeof(f3);
        HIP_TRY(hipMemcpyAsync((char*)device_ptr + done * W * sizeof(f3), src, (size_t)(rr.second - rr.first) * W * sizeof(f3), hipMemcpyDeviceToDevice, c->stream));
        done += (size_t)(rr.second - rr.first);
    }
    HIP_TRY(sync_guarded(c, c->stream));
    return VRT_OK;
}
int vrt_fetch_hdr_device_async(vrt_ctx* c, void* device_ptr) {
    if (!c || !device_ptr) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    const size_t W = c->cfg.width;
    size_t done = 0;   // (a striped context's rows: one stripe after the other)
    for (const auto& rr : owned_ranges(c)) {
        const char* src = (const char*)c->d_cbuf[c->cidx] + (size_t)(rr.first - c->buf0) * W * sizeof(f3);
        HIP_TRY(hipMemcpyAsync((char*)device_ptr + done * W * sizeof(f3), src, (size_t)(rr.second - rr.first) * W * sizeof(f3), hipMemcpyDeviceToDevice, c->stream));
        done += (size_t)(rr.second - rr.first);
    }
    return VRT_OK;
}
int vrt_set_stream(vrt_ctx* c, void* hip_stream) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_guarded(c, c->stream));
    resolve_events(c);
    for (int s = 0; s < VRT_MAX_SETS; s++) c->ev_t_valid[s] = false;  // everything recorded on the old stream has completed
    c->main_dirty = true;
    if (c->owns_stream && c->stream) hipStreamDestroy(c->stream);
    if (hip_stream) { c->stream = (hipStream_t)hip_stream; c->owns_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->owns_stream = true; }
    return VRT_OK;
}
int vrt_fetch_ldr(vrt_ctx* c, float* out) {
    if (!c || !out) return fail(VRT_E_INVALID, "null argument");
    if (!c->have_cam) return fail(VRT_E_STATE, "vrt_set_camera has not been called");
    HIP_TRY(hipSetDevice(c->device));
    if (c->ev_fetch_src) {   // asynchronous fetches share d_ldr: theirs first
        HIP_TRY(hipEventRecord(c->ev_fetch_src, c->fetch_stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_fetch_src, 0));
    }
    FrameParams fp = make_frame_params(c);
    HIP_TRY(launch_tonemap(c->stream, fp, c->d_cbuf[c->cidx], c->d_ldr, c->own0, c->own1));
    return fetch_rows(c, c->d_ldr, sizeof(f4), out);
}

// ---- presenting every frame without emptying the pipeline ---------------------------------------------------------------
// The reference presents each frame (scene.py:255-262: accumulate, fetch_image, copy_prev_matrices).  vrt_fetch_hdr / _ldr
// are blocking copies into pageable memory on the context's stream: the caller waits for every launch queued so far and the
// next launch starts on an idle chip.  The asynchronous forms queue tonemap and copy on a stream of their own behind the
// passes queued so far and return; the caller goes on queueing frames and collects the image with vrt_fetch_wait(slot).
// `out` should be page-locked (vrt_host_alloc): the copy then runs beside the following launches.  The frame is whole for
// an unsharded context; a shard's rows land in their place and the other rows of `out` are left alone.
static int ensure_fetch_stream(vrt_ctx* c) {
    if (c->fetch_stream) return VRT_OK;
    HIP_TRY(hipStreamCreateWithFlags(&c->fetch_stream, hipStreamNonBlocking));
    for (int i = 0; i < VRT_FETCH_SLOTS; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_fetch[i], hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_cbuf_read[i], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fetch_src, hipEventDisableTiming));
    return VRT_OK;
}
static int fetch_async(vrt_ctx* c, void* out, int slot, int what /* 0 HDR, 1 LDR f32 x 4, 2 LDR 8 bit x 4 */) {
    const bool ldr = what != 0;
    if (!c || !out || slot < 0 || slot >= VRT_FETCH_SLOTS) return fail(VRT_E_INVALID, "bad argument (slot must be 0..3)");
    if (ldr && !c->have_cam) return fail(VRT_E_STATE, "vrt_set_camera has not been called");
    if (c->fetch_valid[slot]) return fail(VRT_E_STATE, "this slot's previous fetch has not been collected (vrt_fetch_wait)");
    if (c->stripe_rows) return fail(VRT_E_STATE, "asynchronous fetches are not available on a context with row stripes");
    HIP_TRY(hipSetDevice(c->device));
    if (what == 2 && !c->d_ldr8 && dalloc(&c->d_ldr8, c->npix) != hipSuccess) { c->d_ldr8 = nullptr; return fail(VRT_E_DEVICE, "no memory for the 8-bit image"); }
    if (ensure_fetch_stream(c) != VRT_OK) return VRT_E_DEVICE;
    const size_t W = c->cfg.width, rows = (size_t)(c->own1 - c->own0), off = (size_t)(c->own0 - c->buf0) * W;
    const int b = c->cidx;
    HIP_TRY(hipEventRecord(c->ev_fetch_src, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->fetch_stream, c->ev_fetch_src, 0));
    if (what == 2) {
        HIP_TRY(launch_tonemap8(c->fetch_stream, make_frame_params(c), c->d_cbuf[b], c->d_ldr8, c->own0, c->own1));
        HIP_TRY(hipEventRecord(c->ev_cbuf_read[b], c->fetch_stream));   // the HDR buffer is free again once the tonemap has read it
        HIP_TRY(hipMemcpyAsync((char*)out + (size_t)c->own0 * W * 4, (const char*)c->d_ldr8 + off * 4, rows * W * 4, hipMemcpyDeviceToHost, c->fetch_stream));
    } else if (ldr) {
        HIP_TRY(launch_tonemap(c->fetch_stream, make_frame_params(c), c->d_cbuf[b], c->d_ldr, c->own0, c->own1));
        HIP_TRY(hipEventRecord(c->ev_cbuf_read[b], c->fetch_stream));
        HIP_TRY(hipMemcpyAsync((char*)out + (size_t)c->own0 * W * sizeof(f4), (const char*)c->d_ldr + off * sizeof(f4), rows * W * sizeof(f4), hipMemcpyDeviceToHost, c->fetch_stream));
    } else {
        HIP_TRY(hipMemcpyAsync((char*)out + (size_t)c->own0 * W * sizeof(f3), (const char*)c->d_cbuf[b] + off * sizeof(f3), rows * W * sizeof(f3), hipMemcpyDeviceToHost, c->fetch_stream));
        HIP_TRY(hipEventRecord(c->ev_cbuf_read[b], c->fetch_stream));
    }
    c->cbuf_read_pending[b] = true;
    HIP_TRY(hipEventRecord(c->ev_fetch[slot], c->fetch_stream));
    c->fetch_valid[slot] = true;
    return VRT_OK;
}
int vrt_fetch_hdr_async(vrt_ctx* c, float* out, int slot) { return fetch_async(c, out, slot, 0); }
int vrt_fetch_ldr_async(vrt_ctx* c, float* out, int slot) { return fetch_async(c, out, slot, 1); }
int vrt_fetch_ldr8_async(vrt_ctx* c, uint8_t* out, int slot) { return fetch_async(c, out, slot, 2); }
int vrt_fetch_wait(vrt_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= VRT_FETCH_SLOTS) return fail(VRT_E_INVALID, "bad argument (slot must be 0..3)");
    if (!c->fetch_valid[slot]) return VRT_OK;
    HIP_TRY(hipSetDevice(c->device));
    // (the copy waits for launches that may be held at the dispatch gate: same bounded wait as every other synchronisation)
    if (c->drain_signal && c->drain_signalled) {
        const double t0 = now_s();
        while (hipEventQuery(c->ev_fetch[slot]) == hipErrorNotReady) {
            if (now_s() - t0 > c->knobs.gate_watchdog_s) { release_gate(c); break; }
            std::this_thread::yield();
        }
        (void)hipGetLastError();
    }
    HIP_TRY(hipEventSynchronize(c->ev_fetch[slot]));
    c->fetch_valid[slot] = false;
    return VRT_OK;
}
int vrt_host_alloc(vrt_ctx* c, uint64_t bytes, void** out) {
    if (!c || !out || bytes == 0) return fail(VRT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return VRT_OK;
}
int vrt_host_free(vrt_ctx* c, void* p) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    if (!p) return VRT_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipHostFree(p));
    return VRT_OK;
}
// The tile a multi-GPU rank hands to the gather, written by the temporal pass itself (12 bytes more per pixel of a pass that
// moves ~110) instead of a device-to-device copy behind it: the last pass of every vrt_accumulate call also stores its HDR
// rows [row_begin, row_end) in device_ptrs[k % n], k = the number of such tiles written so far (vrt_hdr_targets_written);
// the pass is queued on the context's stream by the call itself, so work the caller queues on that stream afterwards (an
// event for the gather's stream) is ordered behind the tile.  The caller keeps a tile untouched until its gather has read
// it.  n = 0 ends it.
int vrt_set_hdr_targets(vrt_ctx* c, void* const* device_ptrs, int n) {
    if (!c || n < 0 || (n > 0 && !device_ptrs)) return fail(VRT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    c->hdr_targets.assign(device_ptrs, device_ptrs + n);
    c->hdr_targets_written = c->hdr_targets_committed = 0;
    return VRT_OK;
}
int vrt_hdr_targets_written(vrt_ctx* c, uint64_t* count) {
    if (!c || !count) return fail(VRT_E_INVALID, "null argument");
    *count = c->hdr_targets_written;
    return VRT_OK;
}
int vrt_fetch_buffer(vrt_ctx* c, int which, void* out) {
    if (!c || !out) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    // the g-buffer written by the most recent accumulate (whichever schedule rendered it)
    const f3* pos = c->last_set ? c->alt_gb_pos[c->last_set - 1] : c->d_gb_pos;
    const uint32_t* gmat = c->last_set ? c->alt_gb_mat[c->last_set - 1] : c->d_gb_mat;
    switch (which) {
        case VRT_BUF_GBUF_DEPTH: return fetch_rows(c, c->last_gb_depth, 4, out);
        case VRT_BUF_GBUF_NORMAL: return fetch_rows(c, c->last_gb_normal, 4, out);
        case VRT_BUF_GBUF_POSITION: return fetch_rows(c, pos, 12, out);
        case VRT_BUF_GBUF_MAT: return fetch_rows(c, gmat, 4, out);
        case VRT_BUF_GBUF_REFL_DEPTH: return fetch_rows(c, c->d_gb_refl_f, 4, out);
        case VRT_BUF_HISTORY_DIFFUSE: return fetch_rows(c, c->d_hist_d[c->hist_in], 16, out);
        case VRT_BUF_HISTORY_SPECULAR: return fetch_rows(c, c->d_hist_s[c->hist_in], 16, out);
        default: break;
    }
    HIP_TRY(hipSetDevice(c->device));
    if (which == VRT_BUF_SKY_SCATTERING || which == VRT_BUF_SKY_TRANSMITTANCE) {
        if (c->cfg.sky_res <= 0) return fail(VRT_E_STATE, "no sky tables");
        size_t ns = (size_t)c->cfg.sky_res * c->cfg.sky_res * 3 * sizeof(float);
        HIP_TRY(hipMemcpyAsync(out, which == VRT_BUF_SKY_SCATTERING ? c->d_sky_scat : c->d_sky_trans, ns, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(sync_guarded(c, c->stream));
        return VRT_OK;
    }
    if (which == VRT_BUF_TRANS_LUT) {
        if (c->cfg.sky_res <= 0) return fail(VRT_E_STATE, "no sky tables");
        HIP_TRY(hipMemcpyAsync(out, c->d_trans_lut, 256 * 128 * 3 * 2, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(sync_guarded(c, c->stream));
        return VRT_OK;
    }
    return fail(VRT_E_INVALID, "unknown buffer id");
}
int vrt_get_stats(vrt_ctx* c, vrt_stats* out) {
    if (!c || !out) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_guarded(c, c->stream));
    resolve_events(c);
    Counters h;
    HIP_TRY(hipMemcpy(&h, c->d_counters, sizeof(h), hipMemcpyDeviceToHost));
    c->stats.rays = h.rays; c->stats.dda_iters = h.iters; c->stats.occupancy_queries = h.queries;
    c->stats.closest_hits = h.closest_hits; c->stats.sky_lookups = h.sky_lookups;
    {   // every pass counted, the timed ones' mean for all of them
        auto scaled = [&](int k) { return c->timed_n[k] ? c->timed_ms[k] * ((double)c->passes_n[k] / (double)c->timed_n[k]) : 0.0; };
        c->stats.render_ms = scaled(0); c->stats.temporal_ms = scaled(1); c->stats.gris_ms = scaled(2);
        c->stats.render_launches = c->passes_n[0]; c->stats.temporal_launches = c->passes_n[1]; c->stats.gris_launches = c->passes_n[2];
    }
    c->stats.pipeline_flags = (c->overlap_ready ? 1u : 0u) | (c->drain_signal ? 2u : 0u) | (((uint32_t)c->gate_releases & 0xFFFFu) << 8) |
                              ((c->mode_switches < 255u ? c->mode_switches : 255u) << 24) |
                              (c->overlap_ready ? ((uint32_t)(c->n_streams >> 1) << 2) | ((uint32_t)c->grid_div << 5) : 0u);
    *out = c->stats;
    return VRT_OK;
}
int vrt_reset_stats(vrt_ctx* c) {
    if (!c) return fail(VRT_E_INVALID, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(sync_guarded(c, c->stream));
    resolve_events(c);
    HIP_TRY(hipMemsetAsync(c->d_counters, 0, sizeof(Counters), c->stream));   // on the stream the counting launches follow on
    c->main_dirty = true;
    memset(&c->stats, 0, sizeof(c->stats));
    for (int k = 0; k < 3; k++) { c->timed_ms[k] = 0.0; c->timed_n[k] = c->passes_n[k] = 0u; }
    c->since_reset = 0u;
    return VRT_OK;
}
// Test hook: rows of arguments through single functions of the sky precompute (vrt_sky_kernels.hip, k_sky_probe).
int vrt_sky_probe(vrt_ctx* c, int op, int n, const float* in, int in_stride, float* out, int out_stride, const uint16_t* trans_lut, const float* cloud_ambient) {
    if (!c || !in || !out || n <= 0 || in_stride <= 0 || out_stride <= 0 || op < 0 || op > 9) return fail(VRT_E_INVALID, "bad argument");
    if (c->cfg.sky_res <= 0) return fail(VRT_E_STATE, "context was created without sky tables (sky_res = 0)");
    HIP_TRY(hipSetDevice(c->device));
    c->main_dirty = true;
    if (trans_lut) HIP_TRY(hipMemcpyAsync(c->d_trans_lut, trans_lut, 256 * 128 * 3 * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_in, (size_t)n * in_stride * sizeof(float)));
    if (hipMalloc((void**)&d_out, (size_t)n * out_stride * sizeof(float)) != hipSuccess) { hipFree(d_in); return fail(VRT_E_DEVICE, "no memory for the probe"); }
    const f3 amb = cloud_ambient ? mk3(cloud_ambient[0], cloud_ambient[1], cloud_ambient[2]) : mk3(0.0f);
    hipError_t e = hipMemcpyAsync(d_in, in, (size_t)n * in_stride * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, (size_t)n * out_stride * sizeof(float), c->stream);
    if (e == hipSuccess) e = launch_sky_probe(c->stream, make_sky(c), op, n, d_in, in_stride, d_out, out_stride, amb);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, (size_t)n * out_stride * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_in); hipFree(d_out);
    if (e != hipSuccess) return fail(VRT_E_DEVICE, std::string("sky probe: ") + hipGetErrorString(e));
    return VRT_OK;
}
// diagnostic builds (-DVRT_DIAG_REGIONS) only: 32 x {wave entries, active lanes} per instrumented code region
int vrt_diag_regions(vrt_ctx* c, unsigned long long* out64, int reset) {
    if (!c || !out64) return fail(VRT_E_INVALID, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 64 * sizeof(unsigned long long)));
    hipError_t e = launch_diag_read(c->stream, d, reset);
    if (e != hipSuccess) { hipFree(d); return fail(VRT_E_STATE, "library was not built with VRT_DIAG_REGIONS"); }
    HIP_TRY(sync_guarded(c, c->stream));
    HIP_TRY(hipMemcpy(out64, d, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    hipFree(d);
    return VRT_OK;
}
// runs op over n floats on the device: checks the numeric contract of vrt_detmath.h on gfx950
int vrt_detmath_probe(int device, int op, int n, const float* a, const float* b, float* out) {
    if (n <= 0 || !a || !b || !out) return fail(VRT_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void**)&da, n * 4));
    HIP_TRY(hipMalloc((void**)&db, n * 4));
    HIP_TRY(hipMalloc((void**)&dout, n * 4));
    HIP_TRY(hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice));
    HIP_TRY(launch_detmath_probe(0, op, n, da, db, dout));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost));
    hipFree(da); hipFree(db); hipFree(dout);
    return VRT_OK;
}

}  // extern "C"
