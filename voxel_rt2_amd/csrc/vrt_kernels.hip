// vrt_kernels.hip -- gfx950 kernels of libvrt_hip.so.
//
//   k_pack_grid / k_build_l0 / k_build_coarse / k_build_l0c   prepare_data: packed voxel texels, the bit-brick
//                                                         occupancy pyramid and its compacted fine level
//   k_render_pool<INSTR>                                  persistent wave64 path tracer, pooled schedule (vrt_pool.h):
//                                                         a wave owns 128 path records in LDS and runs them stage by
//                                                         stage (WALK / SHADE / ESCAPE / BEGIN); default without ReSTIR
//   k_render<RESTIR, INSTR>                               persistent wave64 path tracer, one path per lane (vrt_path.h)
//   k_gris_prepare / k_gris                               ReSTIR spatial reuse: per-pixel records (decoded sample, shading
//                                                         basis), then 32 taps x 2 reconnection shifts per pixel
//   k_mat_derived                                         per-material-id part of a shading point (after material uploads)
//   k_temporal                                            fused temporal accumulation -> HDR (sized to run beside
//                                                         the next render launch, see vrt_api.hip)
//   k_tonemap                                             LDR presentation
//
// k_render is a persistent-thread kernel: the grid is sized to the device's residency, each wave
// keeps 64 path records in registers and pulls pixels (8x8 tiles, tile-major order) from a global
// counter.  Between path segments lanes whose path has ended are compacted out with
// __ballot/__popcll and refilled, so the wave's lanes are at different bounce depths but all busy.
// The two coarse brick levels of the pyramid (4 KiB + 64 B) and the material table (7 KiB) are
// staged in LDS once per workgroup; the fine brick level (256 KiB) and the texel grid (8 MiB) stay
// in L2 / Infinity Cache.  Every wave reaches the exit: the loop ends when the counter is exhausted
// and no lane holds a path.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "vrt_kernels.h"

namespace vrt {

// ---- prepare_data ----------------------------------------------------------------------------
// voxel_world.py:69-87: rgba8 texel per voxel; a negative material byte stores 0 (unorm clamp)
template <int G>
__global__ void k_pack_grid(const int8_t* __restrict__ mat, const uint8_t* __restrict__ rgb, uint32_t* __restrict__ grid) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;  // [x][y][z], the order of the uploaded arrays
    if (i >= G * G * G) return;
    int m = mat[i];
    uint32_t a = (m < 0) ? 0u : (uint32_t)m;
    const int z = i & (G - 1), y = (i / G) & (G - 1), x = i / (G * G);
    grid[texel_index<G>(x, y, z)] = (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16) | (a << 24);
}
// raytracer.py:46-53: LOD-0 bit = voxel_material > 0 (signed), gathered into one 4x4x4 brick word per thread
__global__ void k_build_l0(const int8_t* __restrict__ mat, unsigned long long* __restrict__ l0, int G) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = G >> 2;
    if (b >= n0 * n0 * n0) return;
    int bx = b % n0, by = (b / n0) % n0, bz = b / (n0 * n0);
    unsigned long long w = 0ULL;
    for (int z = 0; z < 4; z++)
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int vx = bx * 4 + x, vy = by * 4 + y, vz = bz * 4 + z;
                if (mat[(vx * G + vy) * G + vz] > 0) w |= 1ULL << (z * 16 + y * 4 + x);
            }
    l0[b] = w;
}
// raytracer.py:54-70: a coarser cell is occupied if any child is; here one bit per non-zero child word
__global__ void k_build_coarse(const unsigned long long* __restrict__ fine, unsigned long long* __restrict__ coarse, int n_coarse) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    int total = n_coarse * n_coarse * n_coarse;
    if (b >= total) return;
    int n_fine = n_coarse * 4;
    int bx = b % n_coarse, by = (b / n_coarse) % n_coarse, bz = b / (n_coarse * n_coarse);
    unsigned long long w = 0ULL;
    for (int z = 0; z < 4; z++)
        for (int y = 0; y < 4; y++)
            for (int x = 0; x < 4; x++) {
                int fx = bx * 4 + x, fy = by * 4 + y, fz = bz * 4 + z;
                if (fine[(fz * n_fine + fy) * n_fine + fx] != 0ULL) w |= 1ULL << (z * 16 + y * 4 + x);
            }
    coarse[b] = w;
}

// Bounding box of the solid voxels (brick granular) grown by VRT_CULL_MARGIN: what cull_ray() (vrt_trace.h) tests rays against.
// One block; an empty grid gives lo > hi.
__global__ void k_cull_box(const unsigned long long* __restrict__ l0, int n0, float* __restrict__ out) {
    __shared__ int s_lo[3][256], s_hi[3][256];
    __shared__ int s_cnt[256];
    int cnt = 0;   // non-empty bricks
    int lo[3] = {1 << 20, 1 << 20, 1 << 20}, hi[3] = {-(1 << 20), -(1 << 20), -(1 << 20)};
    for (int b = threadIdx.x; b < n0 * n0 * n0; b += blockDim.x) {
        if (l0[b] == 0ULL) continue;
        cnt++;
        const int c[3] = {b % n0, (b / n0) % n0, b / (n0 * n0)};
        for (int a = 0; a < 3; a++) { lo[a] = min(lo[a], c[a] * 4); hi[a] = max(hi[a], c[a] * 4 + 4); }
    }
    for (int a = 0; a < 3; a++) { s_lo[a][threadIdx.x] = lo[a]; s_hi[a][threadIdx.x] = hi[a]; }
    s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x < 3) {
        int l = 1 << 20, h = -(1 << 20);
        for (int i = 0; i < 256; i++) { l = min(l, s_lo[threadIdx.x][i]); h = max(h, s_hi[threadIdx.x][i]); }
        out[threadIdx.x] = (float)l - VRT_CULL_MARGIN;
        out[3 + threadIdx.x] = (float)h + VRT_CULL_MARGIN;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // is there anything to cull: does the grown box leave part of the grid out
        const float G = (float)(n0 * 4);
        bool some = false;
        for (int a = 0; a < 3; a++) some = some || out[a] > 0.0f || out[3 + a] < G;
        out[6] = some ? 1.0f : 0.0f;
        int total = 0;
        for (int i = 0; i < 256; i++) total += s_cnt[i];
        out[7] = (float)total / (float)(n0 * n0 * n0);   // share of non-empty 4x4x4 bricks (read by the host: which SHADE variant; cull_ray does not look at it)
    }
}

// The fine level without its empty words (Pyramid::l0c, vrt_types.h).  One block of 512 threads, thread i = l1 brick i:
// exclusive prefix sum of the popcounts, then each thread copies the words behind its set bits in bit order.
__global__ void k_build_l0c(const unsigned long long* __restrict__ l0, const unsigned long long* __restrict__ l1,
                            unsigned long long* __restrict__ l0c, uint32_t* __restrict__ base) {
    __shared__ uint32_t s_scan[512];
    const int i = threadIdx.x;
    const unsigned long long w = l1[i];
    const uint32_t cnt = (uint32_t)__popcll(w);
    s_scan[i] = cnt;
    __syncthreads();
    for (int off = 1; off < 512; off <<= 1) {  // Hillis-Steele inclusive scan
        const uint32_t v = (i >= off) ? s_scan[i - off] : 0u;
        __syncthreads();
        s_scan[i] += v;
        __syncthreads();
    }
    const uint32_t first = s_scan[i] - cnt;
    base[i] = first;
    if (i == 511) base[512] = s_scan[511];
    const int bx1 = i & 7, by1 = (i >> 3) & 7, bz1 = i >> 6;
    uint32_t k = first;
    for (int b = 0; b < 64; b++) {
        if ((w >> b) & 1ULL) {
            const int bx = bx1 * 4 + (b & 3), by = by1 * 4 + ((b >> 2) & 3), bz = bz1 * 4 + (b >> 4);
            l0c[k++] = l0[(bz * 32 + by) * 32 + bx];
        }
    }
}

// ---- render ----------------------------------------------------------------------------------
// OOB: the instantiation can read cells outside the grid the reference's way (oob_capable_of, vrt_trace.h).  It is the INSTRUMENTED
// instantiations that can (a context with vrt_set_reference_indexing launches those): the timed kernels carry no test for it.
template <int G_, bool OOB_ = false>
struct LdsPyramid {  // coarse levels in LDS, fine level through L2
    static constexpr int G = G_;
    static constexpr bool flat_descend = false;  // its kernels walk with few active lanes: descend()'s early outs win
    static constexpr bool cull = true;
    static constexpr bool oob_capable = OOB_;
    bool oob;   // wave-uniform: Pyramid::ref_oob
    __device__ __forceinline__ bool oob_ref() const { return oob; }
    const unsigned long long* l0;
    const unsigned long long* l1;
    const unsigned long long* l2;
    unsigned long long w3;   // the top word (G = 256), wave-uniform
    __device__ __forceinline__ unsigned long long load_l0(int i) const { return l0[i]; }
    __device__ __forceinline__ unsigned long long load_l1(int i) const { return l1[i]; }
    __device__ __forceinline__ unsigned long long load_l2(int i) const { return l2[i]; }
    __device__ __forceinline__ unsigned long long load_l3() const { return w3; }
};
// The pooled kernel's view.  128^3: each l1 word beside its parent l2 word ({w1, w2}[512], one 16-byte LDS read per
// cell), and the head of the compacted fine level (Pyramid::l0c) in LDS too -- a sparse scene's fine level is a few
// hundred words, and the fine word is the load every other DDA step depends on (L2: ~700 cycles, LDS: ~130).
// CULL: rays that cannot hit a voxel are not walked (cull_ray, vrt_trace.h).  A launch over a scene whose solids fill the grid
// has nothing to cull and runs the instantiation without the test (its registers cost a dense 4K frame 2.5 %).
template <int G_, bool CULL_, bool SHBR_ = false, bool OOB_ = false>
struct LdsPyramid2 {
    static constexpr int G = G_;
    static constexpr bool cull = CULL_;
    static constexpr bool oob_capable = OOB_;
    bool oob;
    __device__ __forceinline__ bool oob_ref() const { return oob; }
    static constexpr bool flat_descend = true;   // the pooled kernel walks with nearly full waves
    static constexpr bool shadow_branchy = SHBR_;   // ... except SHADE's inline shadow rays in a dense grid (raytrace, vrt_trace.h)
    const unsigned long long* l0;
    const ulonglong2* l12;
    const unsigned long long* l2;
    const uint32_t* fine_base;         // LDS copy of Pyramid::l0c_base
    const unsigned long long* fine;    // LDS copy of l0c[0 .. n_fine)
    uint32_t n_fine;
    __device__ __forceinline__ unsigned long long load_l0(int i) const { return l0[i]; }
    __device__ __forceinline__ unsigned long long load_l1(int i) const { return l12[i].x; }
    __device__ __forceinline__ unsigned long long load_l2(int i) const { return l2[i]; }
    __device__ __forceinline__ unsigned long long load_l3() const { return 0ULL; }
    __device__ __forceinline__ unsigned long long load_fine(int key, uint32_t idx) const { return idx < n_fine ? fine[idx] : l0[key]; }
    __device__ __forceinline__ void load_coarse(int i1, int i2, unsigned long long& w1, unsigned long long& w2, uint32_t& base) const {
        (void)i2;
        const ulonglong2 v = l12[i1];
        w1 = v.x; w2 = v.y;
        base = fine_base[i1];
    }
};
// 256^3: the whole l1 level (16^3 words, 32 KiB) and l2 (4^3 words) resident in LDS -- the brick cache of this grid:
// every coarse query of every DDA step is an LDS read -- the top word in scalar registers, and the fine words (2 MiB:
// they miss LDS by a factor of 13) through L1 / L2 with the current brick's word cached in registers per ray.  A
// workgroup is eight waves (one per CU: eight 13.5 KB path pools + 32.5 KB of pyramid + the material table = 148 KB of
// the CU's 160 KB), so the level is staged once per CU.
template <bool CULL_, bool SHBR_, bool OOB_>
struct LdsPyramid2<256, CULL_, SHBR_, OOB_> {
    static constexpr int G = 256;
    static constexpr bool cull = CULL_;
    static constexpr bool oob_capable = OOB_;
    bool oob;
    __device__ __forceinline__ bool oob_ref() const { return oob; }
    static constexpr bool flat_descend = true;
    static constexpr bool shadow_branchy = SHBR_;
    const unsigned long long* l0;
    const unsigned long long* l1;   // LDS [4096]
    const unsigned long long* l2;   // LDS [64]
    unsigned long long w3;
    __device__ __forceinline__ unsigned long long load_l0(int i) const { return l0[i]; }
    __device__ __forceinline__ unsigned long long load_l1(int i) const { return l1[i]; }
    __device__ __forceinline__ unsigned long long load_l2(int i) const { return l2[i]; }
    __device__ __forceinline__ unsigned long long load_l3() const { return w3; }
    __device__ __forceinline__ unsigned long long load_fine(int key, uint32_t idx) const { (void)idx; return l0[key]; }
    __device__ __forceinline__ void load_coarse(int i1, int i2, unsigned long long& w1, unsigned long long& w2, uint32_t& base) const {
        w1 = l1[i1]; w2 = l2[i2];
        base = 0u;
    }
};
template <int G> struct PoolGeom { static constexpr int waves = VRT_POOL_WAVES; };   // waves (= path pools) per workgroup
template <> struct PoolGeom<256> { static constexpr int waves = 8; };
#ifndef VRT_POOL_SLOTS_128
#define VRT_POOL_SLOTS_128 VRT_POOL_SLOTS   // (experiments: another pool size at 128^3 only -- the 256^3 workgroup has no LDS to spare)
#endif
template <int G> struct PoolSlots { static constexpr int value = G == 256 ? VRT_POOL_SLOTS : VRT_POOL_SLOTS_128; };

__device__ __forceinline__ void flush_stats(const TraceStats& ts, Counters* c) {
    // wave-level sum, one atomic per counter per wave
    unsigned v[5] = {ts.rays, ts.iters, ts.queries, ts.closest_hits, ts.sky_lookups};
    unsigned long long* dst[5] = {&c->rays, &c->iters, &c->queries, &c->closest_hits, &c->sky_lookups};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        unsigned long long s = v[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(dst[k], s);
    }
}

template <int G, bool RESTIR, bool INSTR>
__global__ __launch_bounds__(VRT_RENDER_THREADS, VRT_RENDER_MIN_WAVES) void k_render(FrameParams fp, SceneData sc, PixelBuffers out, unsigned* work_counter, unsigned* next_counter, unsigned chunk, int n_samples) {
    constexpr int N1 = GridDim<G>::n1 * GridDim<G>::n1 * GridDim<G>::n1, N2 = GridDim<G>::n2 * GridDim<G>::n2 * GridDim<G>::n2;
    __shared__ unsigned long long s_l1[N1];
    __shared__ unsigned long long s_l2[N2];
    __shared__ float s_mats[128 * 14];
    __shared__ float s_cull[8];
    for (int i = threadIdx.x; i < N1; i += blockDim.x) s_l1[i] = sc.pyr.l1[i];
    if (threadIdx.x < N2) s_l2[threadIdx.x] = sc.pyr.l2[threadIdx.x];
    if (threadIdx.x < 8) s_cull[threadIdx.x] = sc.cull[threadIdx.x];
    for (int i = threadIdx.x; i < 128 * 14; i += blockDim.x) s_mats[i] = sc.mats[i];
    if (blockIdx.x == 0 && threadIdx.x < VRT_WORK_HEADS) next_counter[threadIdx.x * VRT_WORK_HEAD_STRIDE] = 0u;  // the next launch's heads (idle during this launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) next_counter[1] = 0u;  // and its "drain announced" word
    __syncthreads();
    LdsPyramid<G, INSTR> P;
    P.l0 = sc.pyr.l0; P.l1 = s_l1; P.l2 = s_l2;
    P.w3 = (G == 256) ? sc.pyr.l3[0] : 0ULL;
    P.oob = sc.pyr.ref_oob != 0;
    SceneData scl = sc;
    scl.mats = s_mats;
    scl.cull = s_cull;

    const int lane = threadIdx.x & 63;
    const int tiles_x = (fp.W + 7) >> 3;
    const int tiles_y = launch_tile_rows(fp);
    // work items are (tile, sample, pixel-in-tile), 64 consecutive items = one 8x8 tile of one sample
    const unsigned total = (unsigned)(tiles_x * tiles_y) * 64u * (unsigned)n_samples;

    Path<RESTIR> p;
    p.depth = -1;
    int local_idx = 0;
    TraceStats ts;
    stats_zero(ts);
    bool exhausted = false;               // wave-uniform
    unsigned chunk_next = 0u, chunk_end = 0u;  // wave-uniform: pixels [chunk_next, chunk_end) are reserved for this wave

    for (;;) {
        const bool need = p.depth < 0;
        const unsigned long long mask = __ballot(need);
        if (mask != 0ULL && !exhausted) {
            // One returning atomic on a single word saturates near 88 dequeues/us chip-wide (MI355X_MICROARCH.md,
            // row `dequeue`): a wave therefore reserves `chunk` pixels at a time and hands them to its lanes itself.
            // (Splitting the items over per-XCD heads, which the pooled kernel does, made THIS kernel 3-10 % slower.)
            if (chunk_next == chunk_end) {
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(work_counter, chunk);
                base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                chunk_next = base < total ? base : total;
                chunk_end = (base + chunk < total) ? base + chunk : total;
                if (base >= total) { exhausted = true; chunk_end = chunk_next; }
            }
            const unsigned avail = chunk_end - chunk_next;
            const unsigned n = (unsigned)__popcll(mask);
            const unsigned rank = (unsigned)__popcll(mask & ((1ULL << lane) - 1ULL));
            const unsigned my = chunk_next + rank;
            chunk_next += (n < avail) ? n : avail;
            if (need && rank < avail) {
                const unsigned group = my >> 6, in = my & 63u;
                const unsigned tile = group / (unsigned)n_samples;
                const int sample = (int)(group % (unsigned)n_samples);
                const int u = (int)(tile % (unsigned)tiles_x) * 8 + (int)(in & 7u);
                const int v = launch_tile_row(fp, (int)(tile / (unsigned)tiles_x)) + (int)(in >> 3);
                if (u < fp.W && v < fp.row1 && launch_renders_row(fp, v) && !outside_render_area(fp, (float)u, (float)v)) {
                    VRT_REGION(0);
                    path_begin(fp, p, u, v, sample);
                    local_idx = (v - fp.row0) * fp.W + u;
                }
            }
        }
        if (__ballot(p.depth >= 0) == 0ULL) {
            if (exhausted) break;
            continue;
        }
        if (p.depth >= 0) {
            const bool done = path_segment<RESTIR>(fp, scl, P, out, local_idx, p, ts);
            if (done) {
                path_finish<RESTIR>(fp, scl, out, local_idx, p, ts);
                p.depth = -1;
            }
        }
    }
    if (INSTR) flush_stats(ts, sc.counters);
}

// ---- render, pooled schedule (vrt_pool.h) ------------------------------------------------------
// One wave = one pool of VRT_POOL_SLOTS path records in LDS.  Everything below the stage functions is wave-uniform
// bookkeeping: a census of slot states (ballots), a compacted list of the slots the chosen stage will work on, and
// the WALK loop's in-loop refill.  Waves never talk to each other; the only workgroup-level object is the staged
// pyramid / material table.  The wave leaves when the work counter is exhausted and its pool has drained.
#ifndef VRT_POOL_REFILL
#define VRT_POOL_REFILL 16   // WALK: idle lanes that trigger a refill from the pending list
#endif
#ifndef VRT_POOL_PARK
#define VRT_POOL_PARK 16     // WALK: with the list empty, suspend the walks still going once this few are left
#endif
#ifndef VRT_POOL_FINE_WORDS
#define VRT_POOL_FINE_WORDS 1024   // fine brick words of the scene kept in LDS (8 KB): all of a sparse scene's
#endif
// (state words per lane = (SLOTS + 63) / 64; slots past SLOTS are void: state 4)

__device__ __forceinline__ void wave_lds_sync() {  // LDS written by some lanes of this wave, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int lane_rank(unsigned long long m) {  // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// WAVES path pools of SLOTS slots per workgroup: 4 x 128 with two workgroups per CU (two waves per SIMD) by default; the whole l1
// level of a 256^3 grid leaves room for one workgroup of 8 x 128 per CU; a DENSE 128^3 grid runs one workgroup of 12 x 96 per CU
// -- three waves per SIMD at 168 registers (k_render_pool_dense12 below).
template <int G, bool RESTIR, bool INSTR, bool BLACK_SUN, bool CULL, bool SHBR = false, int WAVES = PoolGeom<G>::waves, int SLOTS = PoolSlots<G>::value>
// 208 registers per wave (the attribute counts half of the unified file): two waves per SIMD then leave the 96 that
// k_temporal runs in beside them (see there).  The allocator would take 238; the cap costs 28 bytes of scratch.
// The ReSTIR instantiation (no overlapped launches, so nothing runs beside it) takes the two-wave maximum of 256.
#ifndef VRT_POOL_HALF_VGPRS
#define VRT_POOL_HALF_VGPRS 104
#endif
__device__ __forceinline__ void render_pool_body(const FrameParams& fp, const SceneData& sc, const PixelBuffers& out, unsigned* work_counter, unsigned* next_counter, int n_samples, uint32_t* cold, uint32_t* drain_signal, uint32_t drain_value, PrimaryRecord* prim_cache) {
    constexpr int WORDS = (SLOTS + 63) / 64;
    constexpr bool BIG = (G == 256);   // which coarse levels are staged how: see LdsPyramid2
    __shared__ ulonglong2 s_l12[BIG ? 1 : 512];
    __shared__ unsigned long long s_l1[BIG ? 4096 : 1];
    __shared__ unsigned long long s_l2[BIG ? 64 : 8];
    __shared__ unsigned long long s_fine[BIG ? 1 : VRT_POOL_FINE_WORDS];
    __shared__ uint32_t s_fine_base[BIG ? 1 : 512];
    __shared__ float s_mats[128 * 14];
    __shared__ float s_cull[8];
    __shared__ uint32_t s_pool[WAVES][PF_COUNT * SLOTS];
    __shared__ uint32_t s_state[WAVES][WORDS * 64];
    __shared__ uint32_t s_list[WAVES][SLOTS];
    LdsPyramid2<G, CULL, SHBR, INSTR> P;
    P.l0 = sc.pyr.l0; P.l2 = s_l2;
    P.oob = sc.pyr.ref_oob != 0;
    if constexpr (BIG) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) s_l1[i] = sc.pyr.l1[i];
        if (threadIdx.x < 64) s_l2[threadIdx.x] = sc.pyr.l2[threadIdx.x];
        P.l1 = s_l1;
        P.w3 = sc.pyr.l3[0];
    } else {
        for (int i = threadIdx.x; i < 512; i += blockDim.x) {
            ulonglong2 v;
            v.x = sc.pyr.l1[i];
            v.y = sc.pyr.l2[(((i >> 8) & 1) << 2) | (((i >> 5) & 1) << 1) | ((i >> 2) & 1)];
            s_l12[i] = v;
            s_fine_base[i] = sc.pyr.l0c_base[i];
        }
        if (threadIdx.x < 8) s_l2[threadIdx.x] = sc.pyr.l2[threadIdx.x];
        uint32_t n_fine = *sc.pyr.l0c_count;  // non-empty fine words of the scene; the first VRT_POOL_FINE_WORDS live in LDS
        if (n_fine > VRT_POOL_FINE_WORDS) n_fine = VRT_POOL_FINE_WORDS;
        for (uint32_t i = threadIdx.x; i < n_fine; i += blockDim.x) s_fine[i] = sc.pyr.l0c[i];
        P.l12 = s_l12; P.fine_base = s_fine_base; P.fine = s_fine; P.n_fine = n_fine;
    }
    for (int i = threadIdx.x; i < 128 * 14; i += blockDim.x) s_mats[i] = sc.mats[i];
    if (threadIdx.x < 8) s_cull[threadIdx.x] = sc.cull[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x < VRT_WORK_HEADS) next_counter[threadIdx.x * VRT_WORK_HEAD_STRIDE] = 0u;  // the next launch's heads (idle during this launch)
    if (blockIdx.x == 0 && threadIdx.x == 0) next_counter[1] = 0u;  // and its "drain announced" word
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* const pool = s_pool[wave];
    uint32_t* const state = s_state[wave];
    uint32_t* const list = s_list[wave];
    for (int k = 0; k < WORDS; k++) state[k * 64 + lane] = (k * 64 + lane < SLOTS) ? (uint32_t)SLOT_EMPTY : 4u;
    __syncthreads();
    SceneData scl = sc;
    scl.mats = s_mats;
    scl.cull = s_cull;
    constexpr int COLD = ColdLine<RESTIR>::count;
    uint32_t* const cold_wave = cold + (size_t)(blockIdx.x * WAVES + wave) * SLOTS * COLD;

    const int tiles_x = (fp.W + 7) >> 3;
    const int tiles_y = launch_tile_rows(fp);
    const unsigned total = (unsigned)(tiles_x * tiles_y) * 64u * (unsigned)n_samples;  // (tile, sample, pixel-in-tile) items
    TraceStats ts;
    stats_zero(ts);
    bool exhausted = false;  // wave-uniform
    // items are split into VRT_WORK_HEADS contiguous ranges of whole tiles; this wave starts on its XCD's range
    // a range is a whole number of tiles with all their samples; inside it items run sample-major (all tiles' sample 0,
    // then all tiles' sample 1, ...) so that the camera-ray records of sample 0 are there when the others begin
    const unsigned tiles_per_range = ((unsigned)(tiles_x * tiles_y) + VRT_WORK_HEADS - 1u) / VRT_WORK_HEADS;
    const unsigned range = tiles_per_range * 64u * (unsigned)n_samples;
    const uint32_t prim_tag = drain_value;  // launch_seq + 1: unique per launch, never 0
    unsigned head = blockIdx.x & (VRT_WORK_HEADS - 1u);  // workgroups go round-robin over the 8 XCDs
    int heads_left = VRT_WORK_HEADS;

#if defined(VRT_DIAG_REGIONS)
    unsigned long long t_prev = __builtin_readcyclecounter();
    int prev_stage = 4;  // 4 = start-up
    // summed in registers and flushed once at the end: an atomic per stage switch from 2048 waves saturates the
    // words it lands on and slows every other memory operation (it made the first version of this clock useless)
    unsigned long long t_stage[6] = {0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL};
    unsigned long long t_sub[4] = {0ULL, 0ULL, 0ULL, 0ULL};   // BEGIN: reservation, record read, set-up from a record, set-up with a ray
#define VRT_SUB_CLOCK(q) do { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_ = __builtin_readcyclecounter(); t_sub[q] += t_ - t_sub_prev; t_sub_prev = t_; } while (0)
#define VRT_SUB_START() do { __asm__ volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); t_sub_prev = __builtin_readcyclecounter(); } while (0)
    unsigned long long t_sub_prev = 0ULL;
#define VRT_POOL_CLOCK(next_stage)                                                                              \
    do {                                                                                                        \
        const unsigned long long t_now = __builtin_readcyclecounter();                                          \
        _Pragma("unroll") for (int q_ = 0; q_ < 6; q_++) if (prev_stage == q_) t_stage[q_] += t_now - t_prev;   \
        t_prev = t_now; prev_stage = (next_stage);                                                              \
    } while (0)
#else
#define VRT_POOL_CLOCK(next_stage) ((void)0)
#define VRT_SUB_CLOCK(q) ((void)0)
#define VRT_SUB_START() ((void)0)
#endif
    for (;;) {
        wave_lds_sync();
        VRT_POOL_CLOCK(5);  // 5 = census + list
        // census
        uint32_t st[WORDS];
        int cnt[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < WORDS; k++) {
            st[k] = state[k * 64 + lane];
#pragma unroll
            for (int q = 0; q < 4; q++) cnt[q] += __popcll(__ballot(st[k] == (uint32_t)q));
        }
        if (exhausted) cnt[SLOT_EMPTY] = 0;
        // the stage with the most slots waiting; ties: SHADE, ESCAPE, WALK, BEGIN.  (Rules that hold WALK back until
        // more rays are pending -- a minimum count, or a lead over the other stages -- measured 0 to -4 %.)
        int stage = SLOT_SHADE, best = cnt[SLOT_SHADE];
        if (cnt[SLOT_ESCAPE] > best) { stage = SLOT_ESCAPE; best = cnt[SLOT_ESCAPE]; }
        if (cnt[SLOT_RAY] > best) { stage = SLOT_RAY; best = cnt[SLOT_RAY]; }
        if (cnt[SLOT_EMPTY] > best) { stage = SLOT_EMPTY; best = cnt[SLOT_EMPTY]; }
        if (best == 0) break;  // nothing pending and no work left
        // compacted list of that stage's slots
        int n = 0;
#pragma unroll
        for (int k = 0; k < WORDS; k++) {
            const bool mine = st[k] == (uint32_t)stage;
            const unsigned long long m = __ballot(mine);
            if (mine) list[n + lane_rank(m)] = (uint32_t)(k * 64 + lane);
            n += __popcll(m);
        }
        wave_lds_sync();
        VRT_POOL_CLOCK(stage);

        if (stage == SLOT_RAY) {
            VRT_REGION(14);
#ifndef VRT_POOL_PARK_MIN
#define VRT_POOL_PARK_MIN 64
#endif
            const int park = (n >= VRT_POOL_PARK_MIN) ? VRT_POOL_PARK : 0;  // suspending only pays when other stages then have work
            int head = 0;
            bool active = false, ended = false;
            RayWalk w;
            BrickCache bc;
            bc.key = -1; bc.word = 0ULL;
            CoarseWords cw;
            cw.w1 = 0ULL; cw.w2 = 0ULL;
            SlotRef s;
            s.base = pool; s.stride = SLOTS;
            int slot = 0, iters0 = 0;
            for (;;) {
                // event: finished walks go to their slots, idle lanes take the next pending rays -- or, with the
                // list empty and few walks left, the rest is suspended
                const unsigned long long idle = __ballot(!active);
                const int n_idle = __popcll(idle);
                if (ended) {
                    VRT_REGION(16);
                    walk_store(s, w);
                    state[slot] = (uint32_t)slot_state_after_walk<G>(w.t, s.f(PF_FLOOR_T));
                    if (prim_cache) {  // the camera ray of a pixel's sample 0: leave its record for the other samples
                        const uint32_t ids = s.u(PF_IDS);
                        if ((ids >> 24) == 0u)  // depth 0, sample 0 (the check word tells a reader whether it saw all of the record)
                            primary_record_store(&prim_cache[((int)((ids >> 12) & 0xfffu) - fp.row0) * fp.W + (int)(ids & 0xfffu)], primary_record(s, prim_tag));
                    }
                    ts.iters += (unsigned)(w.iters - iters0);
                    ended = false;
                }
                if (head >= n) {
                    if (64 - n_idle <= park) {
                        if (active) {  // suspend: the slot stays SLOT_RAY
                            VRT_REGION(17);
                            walk_store(s, w);
                            ts.iters += (unsigned)(w.iters - iters0);
                        }
                        break;
                    }
                } else {
                    const int idx = head + lane_rank(idle);
                    if (!active && idx < n) {
                        VRT_REGION(15);
                        slot = (int)list[idx];
                        s.base = pool + slot;
                        walk_load<G>(s, w);
                        coarse_fetch(P, w.ix, w.iy, w.iz, cw);
                        bc.key = -1;
                        iters0 = w.iters;
                        active = true;
                    }
                    head += (n_idle < n - head) ? n_idle : n - head;
                }
                // the DDA loop proper: nothing but steps until enough lanes have gone idle for the next event
                const int target = (head < n) ? VRT_POOL_REFILL : 64 - park;
                do {
                    if (active) {
                        int nq;
                        if (walk_trip(P, w, bc, cw, nq)) { active = false; ended = true; }
                        ts.queries += (unsigned)nq;
                    }
                } while (__popcll(__ballot(!active)) < target);
            }
        } else {
            const int take = n < 64 ? n : 64;
            unsigned base = 0u, limit = 0u;  // items [base, limit) go to lanes 0.. of this BEGIN
            unsigned range0 = 0u, per_sample = 1u;  // first item of the range they belong to, items per sample in it
            if (stage == SLOT_EMPTY) {
                VRT_SUB_START();
                // pull from the current head; a used-up range sends the wave on to the next head (and this BEGIN
                // hands out nothing: the census runs again)
                unsigned got = 0u;
                if (lane == 0) got = atomicAdd(work_counter + head * VRT_WORK_HEAD_STRIDE, (unsigned)take);
                got = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
                const unsigned r0 = head * range, r1 = (r0 + range < total) ? r0 + range : total;
                base = r0 + got;
                limit = (base + (unsigned)take < r1) ? base + (unsigned)take : r1;
                range0 = r0;
                per_sample = (r1 > r0) ? (r1 - r0) / (unsigned)n_samples : 1u;
                if (r0 >= total || base + (unsigned)take >= r1) {
                    heads_left -= 1;
                    head = (head + 1u) & (VRT_WORK_HEADS - 1u);
                    if (heads_left == 0) {
                        exhausted = true;
                        // the launch starts to drain: tell the stream holding the NEXT launch back until now (vrt_api.hip)
                        // (the first wave to get here does; word 1 of this launch's head line says whether one has)
                        if (drain_signal && lane == 0 && atomicExch(work_counter + 1, 1u) == 0u)
                            __hip_atomic_fetch_max(drain_signal, drain_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
                if (base > limit) base = limit;
                VRT_SUB_CLOCK(0);
            }
            if (lane < take) {
                SlotRef s;
                s.stride = SLOTS;
                if (stage == SLOT_SHADE) {
                    VRT_REGION(11);
                    const int slot = (int)list[lane];
                    s.base = pool + slot;
                    state[slot] = (uint32_t)pool_shade<HIT_SOMETHING, BLACK_SUN, RESTIR>(fp, scl, P, out, s, cold_wave + slot * COLD, ts);
                } else if (stage == SLOT_ESCAPE) {
                    VRT_REGION(12);
                    const int slot = (int)list[lane];
                    s.base = pool + slot;
                    state[slot] = (uint32_t)pool_shade<HIT_NOTHING, false, RESTIR>(fp, scl, P, out, s, cold_wave + slot * COLD, ts);
                } else {
                    VRT_REGION(13);
                    const unsigned my = base + (unsigned)lane;
                    if (my < limit) {
                        const int slot = (int)list[lane];
                        s.base = pool + slot;
                        const unsigned j = my - range0;  // sample-major inside the range (n_samples <= 4)
                        const int sample = (int)(j >= per_sample) + (int)(j >= 2u * per_sample) + (int)(j >= 3u * per_sample);
                        const unsigned rem = j - (unsigned)sample * per_sample;
                        const unsigned tile = range0 / (64u * (unsigned)n_samples) + (rem >> 6), in = rem & 63u;
                        const int u = (int)(tile % (unsigned)tiles_x) * 8 + (int)(in & 7u);
                        const int v = launch_tile_row(fp, (int)(tile / (unsigned)tiles_x)) + (int)(in >> 3);
                        if (u < fp.W && v < fp.row1 && launch_renders_row(fp, v) && !outside_render_area(fp, (float)u, (float)v)) {
                            bool known = false;
                            PrimaryRecord rec;
                            rec.x = rec.y = rec.z = rec.dx = rec.dy = rec.dz = rec.ft = rec.w = 0u;
                            if (prim_cache && sample > 0) {
                                rec = primary_record_load(&prim_cache[(v - fp.row0) * fp.W + u]);
                                known = primary_record_valid(rec, prim_tag);
                            }
                            VRT_SUB_CLOCK(1);
                            int st;
                            if (known) st = pool_begin_known<G>(fp, s, u, v, sample, rec);
                            else {
                                st = pool_begin<G, CULL>(fp, scl.cull, s, u, v, sample, ts);
                                // a camera ray with nothing to walk (it misses the grid or every solid voxel) is finished here: its
                                // record is left here too
                                if (prim_cache && sample == 0 && st != SLOT_RAY) primary_record_store(&prim_cache[(v - fp.row0) * fp.W + u], primary_record(s, prim_tag));
                            }
                            state[slot] = (uint32_t)st;
                            VRT_SUB_CLOCK(sample > 0 ? 2 : 3);
                        }
                    }
                }
            }
        }
    }
    VRT_POOL_CLOCK(4);
#if defined(VRT_DIAG_REGIONS)
    if (lane == 0)
        for (int q = 0; q < 6; q++) atomicAdd(&g_vrt_region[2 * (20 + q)], t_stage[q]);
    if (lane == 0)
        for (int q = 0; q < 4; q++) atomicAdd(&g_vrt_region[52 + q], t_sub[q]);
#endif
    if (INSTR) flush_stats(ts, sc.counters);
}
// (the register attribute takes a literal, hence one kernel per budget around the shared body)
template <int G, bool INSTR, bool BLACK_SUN, bool CULL>
__global__ __launch_bounds__(64 * PoolGeom<G>::waves, VRT_POOL_MIN_WAVES) __attribute__((amdgpu_num_vgpr(VRT_POOL_HALF_VGPRS))) void k_render_pool(FrameParams fp, SceneData sc, PixelBuffers out, unsigned* work_counter, unsigned* next_counter, int n_samples, uint32_t* cold, uint32_t* drain_signal, uint32_t drain_value, PrimaryRecord* prim_cache) {
    render_pool_body<G, false, INSTR, BLACK_SUN, CULL>(fp, sc, out, work_counter, next_counter, n_samples, cold, drain_signal, drain_value, prim_cache);
}
// a dense grid under an emitting sun: the shadow rays of SHADE with the branchy descent (shadow_branchy_of, vrt_trace.h)
template <int G, bool INSTR, bool CULL>
__global__ __launch_bounds__(64 * PoolGeom<G>::waves, VRT_POOL_MIN_WAVES) __attribute__((amdgpu_num_vgpr(VRT_POOL_HALF_VGPRS))) void k_render_pool_dense(FrameParams fp, SceneData sc, PixelBuffers out, unsigned* work_counter, unsigned* next_counter, int n_samples, uint32_t* cold, uint32_t* drain_signal, uint32_t drain_value, PrimaryRecord* prim_cache) {
    render_pool_body<G, false, INSTR, false, CULL, true>(fp, sc, out, work_counter, next_counter, n_samples, cold, drain_signal, drain_value, prim_cache);
}
// ... and at 128^3 the same with THREE waves per SIMD: one workgroup of twelve waves per CU, 96 slots per pool (12 x 10.4 KB of pools
// + 25 KB of pyramid and materials = 148 KB of LDS), 168 registers (21-36 spilled).  A dense grid's rays end after a step or two, so
// its waves spend their time in SHADE waiting on texel and shadow-ray loads, which a third wave covers: the dense 4K frame
// 2 673 -> 2 953 Mpath-samples/s.  A sparse scene loses as much with it (config 2 -3.6 %, sun-lit -6 %: fewer slots per pool
// thin the WALK stage out, and the spills cost), so only launches the dense variant would take anyway use it.
// At 256^3 the staged l1 level (32 KB) leaves the twelve pools 88 slots each (153 KB of LDS in all).
#define VRT_D12_WAVES 12
#define VRT_D12_SLOTS 96
#ifndef VRT_D12_SLOTS_256
#define VRT_D12_SLOTS_256 88
#endif
template <int G> struct D12Slots { static constexpr int value = G == 256 ? VRT_D12_SLOTS_256 : VRT_D12_SLOTS; };
template <int G, bool INSTR, bool CULL>
__global__ __launch_bounds__(64 * VRT_D12_WAVES, 3) __attribute__((amdgpu_num_vgpr(84))) void k_render_pool_dense12(FrameParams fp, SceneData sc, PixelBuffers out, unsigned* work_counter, unsigned* next_counter, int n_samples, uint32_t* cold, uint32_t* drain_signal, uint32_t drain_value, PrimaryRecord* prim_cache) {
    render_pool_body<G, false, INSTR, false, CULL, true, VRT_D12_WAVES, D12Slots<G>::value>(fp, sc, out, work_counter, next_counter, n_samples, cold, drain_signal, drain_value, prim_cache);
}
template <int G, bool INSTR, bool CULL>
__global__ __launch_bounds__(64 * PoolGeom<G>::waves, VRT_POOL_MIN_WAVES) __attribute__((amdgpu_num_vgpr(128))) void k_render_pool_restir(FrameParams fp, SceneData sc, PixelBuffers out, unsigned* work_counter, unsigned* next_counter, int n_samples, uint32_t* cold, uint32_t* drain_signal, uint32_t drain_value, PrimaryRecord* prim_cache) {
    render_pool_body<G, true, INSTR, false, CULL>(fp, sc, out, work_counter, next_counter, n_samples, cold, drain_signal, drain_value, prim_cache);
}

// ---- spatial reuse ---------------------------------------------------------------------------
#ifndef VRT_GRIS_SPLIT
#define VRT_GRIS_SPLIT 1   // the pass as two kernels of three waves per SIMD each (gris_pixel, vrt_restir.h); 0: one kernel of two
                           // (config 3: 358 against 317 Mpath-samples/s; the first half alone at three waves: 335)
#endif
#ifndef VRT_GRIS_MIN_WAVES
#define VRT_GRIS_MIN_WAVES 2   // whole pass: 256 registers, no spills; one wave per SIMD left the VALU idle a third of the time (6.6 vs 10.2 ms)
#endif
#ifndef VRT_GRIS_MIN_WAVES_A
#define VRT_GRIS_MIN_WAVES_A 3 // first half: 168 registers, no spills
#endif
#ifndef VRT_GRIS_MIN_WAVES_B
#define VRT_GRIS_MIN_WAVES_B 3 // second half: 168 registers hold it once the output reservoir carries the chosen TAP instead of its sample
#endif
template <int G, bool INSTR, int PHASE = 0>
__global__ __launch_bounds__(256, (PHASE == 1 ? VRT_GRIS_MIN_WAVES_A : PHASE == 2 ? VRT_GRIS_MIN_WAVES_B : VRT_GRIS_MIN_WAVES)) void k_gris(FrameParams fp, SceneData sc, GrisBuffers gb, int r0, int r_first, int r1, int tiles_x, int band_w) {
    constexpr int N1 = GridDim<G>::n1 * GridDim<G>::n1 * GridDim<G>::n1, N2 = GridDim<G>::n2 * GridDim<G>::n2 * GridDim<G>::n2;
    __shared__ unsigned long long s_l1[N1];
    __shared__ unsigned long long s_l2[N2];
    __shared__ float s_mats[128 * 14];
    __shared__ float s_mats_x[128 * 8];
    __shared__ float s_cs[4][64];          // per wave (= 8x8 pixel tile): cos / sin of the 32 tap angles
    __shared__ uint16_t s_off[32][256];    // per tap, per thread: the tap's pixel offset
    for (int i = threadIdx.x; i < N1; i += blockDim.x) s_l1[i] = sc.pyr.l1[i];
    if (threadIdx.x < N2) s_l2[threadIdx.x] = sc.pyr.l2[threadIdx.x];
    for (int i = threadIdx.x; i < 128 * 14; i += blockDim.x) s_mats[i] = sc.mats[i];
    for (int i = threadIdx.x; i < 128 * 8; i += blockDim.x) s_mats_x[i] = gb.mats_x[i];
    __shared__ float s_cull[8];
    if (threadIdx.x < 8) s_cull[threadIdx.x] = sc.cull[threadIdx.x];
    LdsPyramid<G, INSTR> P;
    P.l0 = sc.pyr.l0; P.l1 = s_l1; P.l2 = s_l2;
    P.w3 = (G == 256) ? sc.pyr.l3[0] : 0ULL;
    P.oob = sc.pyr.ref_oob != 0;
    SceneData scl = sc;
    scl.mats = s_mats;
    scl.cull = s_cull;
    GrisBuffers gbl = gb;
    gbl.mats_x = s_mats_x;
    // 16x16 pixel tile per workgroup = four 8x8 wave tiles.  Workgroups go to the 8 XCDs round robin: XCD k takes the
    // k-th vertical band of tiles, row by row, so that the tiles resident on one XCD at a time are neighbours and their
    // 64x64-pixel tap windows overlap in that XCD's L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = xcd * band_w + slot % band_w, by = slot / band_w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u = bx * 16 + (wave & 1) * 8 + (lane & 7);
    // r0 is a multiple of 8: a wave's 8x8 pixels are then one tile of the tap-angle hash (u >> 3, v >> 3), whatever row the
    // shard starts at; rows [r0, r_first) are not this launch's
    const int v = r0 + by * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (lane < 32) gris_tap_cs(u, v, 0, lane, s_cs[wave]);
    __syncthreads();
    GrisTaps taps;
    taps.cs = s_cs[wave]; taps.off = &s_off[0][threadIdx.x]; taps.off_stride = 256;
    TraceStats ts;
    stats_zero(ts);
    if (bx < tiles_x && u < fp.W && v >= r_first && v < r1) gris_pixel<PHASE>(fp, scl, P, gbl, taps, u, v, 0, 24.0f, 32, 1, ts);
    if (INSTR) flush_stats(ts, sc.counters);
}

// The first kernel of the split pass (pathtracer.py:917-931: the centre's sample shifted into every accepted neighbour's domain; the
// sum of the terms is the canonical MIS weight), on a WAVE-LEVEL schedule.  Whether a pixel's shifts need evaluating is a
// property of ITS sample (gris_classify_pixel: a dead sample makes every one of them a constant), so the counts are bimodal
// -- 0 or ~25 taps per pixel, 7.8 on average in a scene open to the sky -- and a loop per lane runs as long as the wave's busiest
// pixel (measured: 16.6 trips at 30 of 64 lanes).  Here the wave's 8x8 pixels pool their taps: every lane lists its live taps in
// LDS, the wave evaluates the list 64 at a time (a lane loads the sample of whichever pixel its item belongs to: 224 bytes from
// L2 instead of registers kept across a loop) and leaves each term where the item was; then every lane sums ITS pixel's terms in
// tap order, constants included -- the float sum of gris_pixel<1>, term for term (gris_first_term is the loop's body).
#ifndef VRT_GRIS_FIRST_ROUNDS
#define VRT_GRIS_FIRST_ROUNDS 1
#endif
template <int G, bool INSTR>
__global__ __launch_bounds__(256, VRT_GRIS_MIN_WAVES_A) void k_gris_first(FrameParams fp, SceneData sc, GrisBuffers gb, int r0, int r_first, int r1, int tiles_x, int band_w) {
    __shared__ float s_mats[128 * 14];
    __shared__ float s_mats_x[128 * 8];
    __shared__ float s_cs[4][64];          // per wave (= 8x8 pixel tile): cos / sin of the 32 tap angles
    __shared__ uint32_t s_item[4][64 * 32 / VRT_GRIS_FIRST_ROUNDS];  // per wave: (lane << 8 | tap) of every tap to evaluate, then the tap's term in its place
    __shared__ float s_radius[4][64];      // per wave: each pixel's radius_shift
    for (int i = threadIdx.x; i < 128 * 14; i += blockDim.x) s_mats[i] = sc.mats[i];
    for (int i = threadIdx.x; i < 128 * 8; i += blockDim.x) s_mats_x[i] = gb.mats_x[i];
    SceneData scl = sc;
    scl.mats = s_mats;
    GrisBuffers gbl = gb;
    gbl.mats_x = s_mats_x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = xcd * band_w + slot % band_w, by = slot / band_w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u0 = bx * 16 + (wave & 1) * 8, v0 = r0 + by * 16 + (wave >> 1) * 8;   // the wave's 8x8 tile (one tile of the tap-angle hash)
    const int u = u0 + (lane & 7), v = v0 + (lane >> 3);
    if (lane < 32) gris_tap_cs(u, v, 0, lane, s_cs[wave]);
    __syncthreads();
    GrisTaps taps;
    taps.cs = s_cs[wave]; taps.off = nullptr; taps.off_stride = 0;
    TraceStats ts;
    stats_zero(ts);
    const int max_taps = 32;
    const float max_radius = 24.0f;
    uint32_t* const item = s_item[wave];

    // this lane's own pixel: gris_pixel<1>'s early outs, its masks and the constant term
    const int idx = (v - fp.row0) * fp.W + u;
    bool mine = bx < tiles_x && u < fp.W && v >= r_first && v < r1 && !outside_render_area(fp, (float)u, (float)v);
    unsigned accepted = 0u, walk = 0u;
    float const_term = 0.0f;
    if (mine) {
        const GrisGeo* cg = &gb.geo[idx];
        if (near_zero3(cg->x1)) mine = false;
        else {
            dm_rng rng = dm_rng_init(fp.seed, fp.frame, (uint32_t)(v * fp.W + u), 1u);
            (void)dm_rng_f32(&rng);  // start_index draw (:827), value unused
            s_radius[wave][lane] = dm_rng_f32(&rng);
            accepted = cg->pad;
            walk = accepted & gb.src[idx].pad2;
            Reservoir c;
            c.z.F = gb.src[idx].F; c.M = gb.src[idx].M;
            const_term = gris_first_const_term(c, max_taps);
        }
    }
    // Two rounds, the pixels of lanes 0-31 and then of lanes 32-63 (VRT_GRIS_FIRST_ROUNDS = 2: a list of 32 x 32 entries per wave
    // -- 16 KB per workgroup instead of 32 -- so that LDS leaves room for more waves per SIMD; 1: the whole wave's pixels at once).
    float canonical_mis = 1.0f;
    constexpr int ROUNDS = VRT_GRIS_FIRST_ROUNDS, PER = 64 / ROUNDS;
    for (int round = 0; round < ROUNDS; round++) {
        const bool in_round = lane / PER == round;
        // where this lane's items start in the wave's list: exclusive prefix sum of the counts
        const int cnt = in_round ? __builtin_popcount(walk) : 0;
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        const int base = incl - cnt;
        const int total = __shfl(incl, 63, 64);
        if (in_round) {
            int k = base;
            for (unsigned m = walk; m != 0u; m &= m - 1u) item[k++] = ((uint32_t)lane << 8) | (uint32_t)__builtin_ctz(m);
        }
        wave_lds_sync();
        for (int k0 = 0; k0 < total; k0 += 64) {
            const int k = k0 + lane;
            if (k < total) {
                const uint32_t e = item[k];
                const int q = (int)(e >> 8), i = (int)(e & 31u);
                const int uq = u0 + (q & 7), vq = v0 + (q >> 3);
                int tx, ty;
                (void)gris_tap(fp, taps, uq, vq, i, s_radius[wave][q], max_radius, max_taps, tx, ty);
                Reservoir center;
                f3 center_rc_ty, center_sky_t;
                RcPre center_pre;
                gris_load_src(center, center_rc_ty, center_sky_t, center_pre, gb.src[(vq - fp.row0) * fp.W + uq]);
                item[k] = dm_f2u(gris_first_term(fp, scl, gbl, center, center_rc_ty, center_sky_t, center_pre, tx, ty, max_taps, ts));
            }
        }
        wave_lds_sync();
        if (mine && in_round) {
            int k = base;
            for (unsigned m = accepted; m != 0u; m &= m - 1u) {
                const int i = __builtin_ctz(m);
                canonical_mis += ((walk >> i) & 1u) ? dm_u2f(item[k++]) : const_term;
            }
            gb.geo[idx].pad3 = dm_f2u(canonical_mis);
        }
        wave_lds_sync();   // (the list is written again by the next round)
    }
    if (INSTR) flush_stats(ts, sc.counters);
}

// split pass, before its two kernels: masks of accepted and of live taps per pixel (gris_classify_pixel).  Same tiling as k_gris:
// a wave is one 8x8 tile of the tap-angle hash.
#ifndef VRT_CLASSIFY_MIN_WAVES
#define VRT_CLASSIFY_MIN_WAVES 4
#endif
template <bool INSTR>
__global__ __launch_bounds__(256, VRT_CLASSIFY_MIN_WAVES) void k_gris_classify(FrameParams fp, SceneData sc, GrisBuffers gb, int r0, int r_first, int r1, int tiles_x, int band_w) {
    __shared__ float s_cs[4][64];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int bx = xcd * band_w + slot % band_w, by = slot / band_w;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u = bx * 16 + (wave & 1) * 8 + (lane & 7);
    const int v = r0 + by * 16 + (wave >> 1) * 8 + (lane >> 3);
    if (lane < 32) gris_tap_cs(u, v, 0, lane, s_cs[wave]);
    __syncthreads();
    GrisTaps taps;
    taps.cs = s_cs[wave]; taps.off = nullptr; taps.off_stride = 0;
    TraceStats ts;
    stats_zero(ts);
    if (bx < tiles_x && u < fp.W && v >= r_first && v < r1) gris_classify_pixel(fp, gb, taps, u, v, 24.0f, 32, ts);
    if (INSTR) flush_stats(ts, sc.counters);
}

// once per pixel of every row the launch holds: the records k_gris reads ~32 times per pixel (vrt_restir.h)
__global__ __launch_bounds__(256) void k_gris_prepare(FrameParams fp, SceneData sc, GrisBuffers gb) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = fp.row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u < fp.W && v < fp.row1) gris_prepare_pixel(fp, sc, gb, u, v);
}
__global__ void k_mat_derived(const float* mats, float* mats_x) {
    const int id = threadIdx.x;
    if (id < 128) { const Material m = load_material(mats, id); store_mat_derived(mats_x, id, mat_derive(m), material_unit_range(m)); }
}

// ---- temporal accumulation + presentation ------------------------------------------------------
// Register budgets chosen together (overlapped launches, vrt_api.hip): two pooled render waves at 208 registers leave
// 96 of a SIMD's 512, which is what this kernel is held to, so that a temporal pass runs on the same CUs BESIDE the next
// render launch instead of waiting for its persistent workgroups to retire.  (A 48-register build beside 232-register
// render waves was the first version: its spills made it 1.2 ms long beside a launch; this one measured 4 % faster.
// Raising its waves' priority with s_setprio measured 2.5 % slower.)
#ifndef VRT_TEMPORAL_HALF_VGPRS
#define VRT_TEMPORAL_HALF_VGPRS 48   // the attribute counts half of the unified register file
#endif
// Workgroups of VRT_TEMPORAL_ROWS waves (one image row each).  One: beside a render launch a SIMD has room for one wave of this
// kernel, and a workgroup of four waits until all four SIMDs of a CU have theirs free at once (one-wave workgroups: the frame's
// upper rows as a rank's tile +4 %, one-sample calls +2 %, the whole frame unchanged; profiles/r04_zj_*).
#ifndef VRT_TEMPORAL_ROWS
#define VRT_TEMPORAL_ROWS 1
#endif
__global__ __launch_bounds__(64 * VRT_TEMPORAL_ROWS) __attribute__((amdgpu_num_vgpr(VRT_TEMPORAL_HALF_VGPRS))) void k_temporal(FrameParams fp, TemporalBuffers tb, int r0, int r1, int n_samples) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = r0 + blockIdx.y * VRT_TEMPORAL_ROWS + (threadIdx.x >> 6);
    if (u < fp.W && v < r1) temporal_pixel(fp, tb, u, v, n_samples);
}
// the same pass over a striped context's OWN rows (vrt_set_row_stripes): n_own of them, stripe after stripe, in ONE launch (a launch
// per stripe put 17 dispatches on the context's stream per step of an eighth of 1080p in 8-row stripes: 0.66 ms a step for 0.19)
__global__ __launch_bounds__(64 * VRT_TEMPORAL_ROWS) __attribute__((amdgpu_num_vgpr(VRT_TEMPORAL_HALF_VGPRS))) void k_temporal_stripes(FrameParams fp, TemporalBuffers tb, int n_own, int n_samples) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int k = blockIdx.y * VRT_TEMPORAL_ROWS + (threadIdx.x >> 6);
    if (k >= n_own) return;
    const int v = (k / fp.stripe_rows) * fp.stripe_period + fp.stripe_first + k % fp.stripe_rows;
    if (u < fp.W && v < fp.H) temporal_pixel<true>(fp, tb, u, v, n_samples);
}
__global__ __launch_bounds__(256) void k_tonemap(FrameParams fp, const f3* hdr, f4* ldr, int r0, int r1) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = r0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u < fp.W && v < r1) ldr[(v - fp.row0) * fp.W + u] = tonemap_pixel(fp, hdr, u, v);
}
// the same image as 8-bit rgba for a display or a PNG: u8(clamp(c, 0, 1) * 255 + 0.5), the conversion scene.py's image writer does
__global__ __launch_bounds__(256) void k_tonemap8(FrameParams fp, const f3* hdr, uint32_t* ldr8, int r0, int r1) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = r0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u < fp.W && v < r1) {
        const f4 c = tonemap_pixel(fp, hdr, u, v);
        ldr8[(v - fp.row0) * fp.W + u] = dm_f2u32(dm_saturate(c.x) * 255.0f + 0.5f) | (dm_f2u32(dm_saturate(c.y) * 255.0f + 0.5f) << 8) |
                                         (dm_f2u32(dm_saturate(c.z) * 255.0f + 0.5f) << 16) | (dm_f2u32(dm_saturate(c.w) * 255.0f + 0.5f) << 24);
    }
}
#if defined(VRT_DIAG_REGIONS)
__global__ void k_diag_read(unsigned long long* out, int reset) {
    int i = threadIdx.x;
    if (i < 64) { out[i] = g_vrt_region[i]; if (reset) g_vrt_region[i] = 0ULL; }
}
#endif
hipError_t launch_diag_read(hipStream_t st, unsigned long long* out, int reset) {
#if defined(VRT_DIAG_REGIONS)
    hipLaunchKernelGGL(k_diag_read, dim3(1), dim3(64), 0, st, out, reset);
    return hipGetLastError();
#else
    (void)st; (void)out; (void)reset;
    return hipErrorNotSupported;
#endif
}
__global__ void k_detmath_probe(int op, int n, const float* a, const float* b, float* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r = 0.0f;
    switch (op) {
        case 0: r = dm_sin(a[i]); break;
        case 1: r = dm_cos(a[i]); break;
        case 2: r = dm_exp(a[i]); break;
        case 3: r = dm_log(a[i]); break;
        case 4: r = dm_pow(a[i], b[i]); break;
        case 5: r = dm_acos(a[i]); break;
        case 6: r = dm_atan2(a[i], b[i]); break;
        case 7: r = dm_min(a[i], b[i]); break;
        case 8: r = dm_max(a[i], b[i]); break;
        case 9: r = dm_round_f16(a[i]); break;
        case 10: r = a[i] / b[i]; break;
        case 11: r = dm_sqrt(a[i]); break;
        case 12: r = a[i] * b[i] + a[i]; break;  // must NOT contract
        case 13: r = (float)dm_f2i(a[i]); break;
        case 14: r = dm_u2f((uint32_t)dm_f32_to_f16(a[i])); break;             // the half code, as the low bits of the output word
        case 15: r = dm_f16_to_f32((uint16_t)(dm_f2u(a[i]) & 0xffffu)); break;  // input word's low bits = half code
    }
    out[i] = r;
}

// ---- host-side launchers -----------------------------------------------------------------------
#define VRT_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_prepare(hipStream_t st, int G, const int8_t* mat, const uint8_t* rgb, uint32_t* grid, unsigned long long* l0,
                          unsigned long long* l1, unsigned long long* l2, unsigned long long* l3, unsigned long long* l0c, uint32_t* l0c_base,
                          float* cull) {
    const int n = G * G * G, n0 = G / 4, n1 = G / 16, n2 = G / 64;
    if (G == 256) hipLaunchKernelGGL(k_pack_grid<256>, dim3((n + 255) / 256), dim3(256), 0, st, mat, rgb, grid);
    else hipLaunchKernelGGL(k_pack_grid<128>, dim3((n + 255) / 256), dim3(256), 0, st, mat, rgb, grid);
    VRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_build_l0, dim3((n0 * n0 * n0 + 255) / 256), dim3(256), 0, st, mat, l0, G);
    VRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cull_box, dim3(1), dim3(256), 0, st, (const unsigned long long*)l0, n0, cull);
    VRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_build_coarse, dim3((n1 * n1 * n1 + 255) / 256), dim3(256), 0, st, (const unsigned long long*)l0, l1, n1);
    VRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_build_coarse, dim3(1), dim3(64), 0, st, (const unsigned long long*)l1, l2, n2);
    VRT_LAUNCH_CHECK();
    if (G == 256) {  // the top word: bit = l2 word non-zero
        hipLaunchKernelGGL(k_build_coarse, dim3(1), dim3(64), 0, st, (const unsigned long long*)l2, l3, 1);
    } else {         // the compacted fine level of the pooled kernel's LDS copy (128^3 only)
        hipLaunchKernelGGL(k_build_l0c, dim3(1), dim3(512), 0, st, (const unsigned long long*)l0, (const unsigned long long*)l1, l0c, l0c_base);
    }
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}

// Kernel variants are picked by four switches; these macros spell the dispatch once.
#define VRT_BY_GRID(G_, CALL) do { if ((G_) == 256) { constexpr int G = 256; CALL; } else { constexpr int G = 128; CALL; } } while (0)
#define VRT_BY_2(A_, B_, CALL) do { if (A_) { constexpr bool A = true; if (B_) { constexpr bool B = true; CALL; } else { constexpr bool B = false; CALL; } } \
                                    else { constexpr bool A = false; if (B_) { constexpr bool B = true; CALL; } else { constexpr bool B = false; CALL; } } } while (0)

hipError_t query_render_residency(int grid_res, bool restir, bool instr, int* blocks_per_cu) {
    hipError_t e = hipSuccess;
    VRT_BY_GRID(grid_res, VRT_BY_2(restir, instr, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_render<G, A, B>, VRT_RENDER_THREADS, 0)));
    return e;
}

hipError_t launch_render(hipStream_t st, int grid_res, bool restir, bool instr, int n_blocks, const FrameParams& fp, const SceneData& sc,
                         const PixelBuffers& out, unsigned* work_counters, unsigned launch_seq, int n_samples, int chunk_override) {
    // sixteen sets of heads rotate: launch k counts on set k % 16 and zeroes set (k + 8) % 16 -- up to four launches are in
    // flight together (vrt_accumulate) and none of their sets may be touched; launch k + 8 runs on launch k's stream (the
    // pipeline is 2 or 4 streams deep), so its set is clean before it starts whatever the other streams do
    unsigned* work_counter = work_counters + (launch_seq & 15u) * (VRT_WORK_HEADS * VRT_WORK_HEAD_STRIDE);  // the fused kernel uses head 0 only
    unsigned* next_counter = work_counters + ((launch_seq + 8u) & 15u) * (VRT_WORK_HEADS * VRT_WORK_HEAD_STRIDE);
    dim3 g(n_blocks), b(VRT_RENDER_THREADS);
    // pixels a wave reserves per atomic: whole 8x8 tiles.  One tile keeps the tail short (measured: 192-pixel chunks
    // cost 13 % at 1080p on the sparse scene) and still cuts the atomic rate ~3x against per-refill atomics, which
    // is what the dense 4K frame needed (67 -> 22 dequeues/us, 5.9 -> 4.8 ms).  chunk_override: development builds (VRT_CHUNK).
    const unsigned chunk = chunk_override > 0 ? (unsigned)chunk_override : 64u;
    VRT_BY_GRID(grid_res, VRT_BY_2(restir, instr, hipLaunchKernelGGL((k_render<G, A, B>), g, b, 0, st, fp, sc, out, work_counter, next_counter, chunk, n_samples)));
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
int pool_waves_per_block(int grid_res) { return grid_res == 256 ? PoolGeom<256>::waves : PoolGeom<128>::waves; }
hipError_t query_render_pool_dense12_residency(int grid_res, bool instr, int* blocks_per_cu) {
    hipError_t e = hipSuccess;
    VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_render_pool_dense12<G, A, true>, 64 * VRT_D12_WAVES, 0)));
    return e;
}
hipError_t query_render_pool_residency(int grid_res, bool restir, bool instr, int* blocks_per_cu) {
    hipError_t e = hipSuccess;
    if (restir) VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_render_pool_restir<G, A, true>, 64 * PoolGeom<G>::waves, 0)));
    else VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, e = hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, k_render_pool<G, A, B, true>, 64 * PoolGeom<G>::waves, 0)));
    return e;
}
size_t pool_scratch_bytes(int grid_res, bool restir, int n_blocks, int n_blocks_dense12) {   // room for either geometry
    const size_t line = (size_t)(restir ? ColdLine<true>::count : ColdLine<false>::count) * sizeof(uint32_t);
    const size_t a = (size_t)n_blocks * pool_waves_per_block(grid_res) * (grid_res == 256 ? PoolSlots<256>::value : PoolSlots<128>::value) * line;
    const size_t b = (size_t)n_blocks_dense12 * VRT_D12_WAVES * VRT_D12_SLOTS * line;
    return a > b ? a : b;
}
hipError_t launch_render_pool(hipStream_t st, int grid_res, bool restir, bool instr, int n_blocks, const FrameParams& fp, const SceneData& sc,
                              const PixelBuffers& out, unsigned* work_counters, unsigned launch_seq, int n_samples, uint32_t* cold,
                              uint32_t* drain_signal, PrimaryRecord* prim_cache, bool cull, bool dense, bool dense12) {
    unsigned* work_counter = work_counters + (launch_seq & 15u) * (VRT_WORK_HEADS * VRT_WORK_HEAD_STRIDE);
    unsigned* next_counter = work_counters + ((launch_seq + 8u) & 15u) * (VRT_WORK_HEADS * VRT_WORK_HEAD_STRIDE);
    dim3 g(n_blocks), b(64 * (dense12 ? VRT_D12_WAVES : pool_waves_per_block(grid_res)));
    // the signal carries launch_seq + 1 of the latest launch that has begun to drain
    // the black-sun variant (scene.py's default light) compiles the light sample out of the shading stage
    const bool black_sun = !((fp.light_color.x != 0.0f || fp.light_color.y != 0.0f || fp.light_color.z != 0.0f) && fp.light_weight != 0.0f);
#define VRT_POOL_ARGS g, b, 0, st, fp, sc, out, work_counter, next_counter, n_samples, cold, drain_signal, launch_seq + 1u, prim_cache
    if (restir) {  // the reservoir needs the light sample whatever the sun's colour
        if (cull) VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, hipLaunchKernelGGL((k_render_pool_restir<G, A, true>), VRT_POOL_ARGS)));
        else VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, hipLaunchKernelGGL((k_render_pool_restir<G, A, false>), VRT_POOL_ARGS)));
    } else if (dense12) {               // (the caller's choice: pool_uses_dense12)
        VRT_BY_GRID(grid_res, VRT_BY_2(instr, cull, hipLaunchKernelGGL((k_render_pool_dense12<G, A, B>), VRT_POOL_ARGS)));
    } else if (dense && !black_sun) {   // (with a black sun SHADE walks next to no shadow rays)
        VRT_BY_GRID(grid_res, VRT_BY_2(instr, cull, hipLaunchKernelGGL((k_render_pool_dense<G, A, B>), VRT_POOL_ARGS)));
    } else {
        if (cull) VRT_BY_GRID(grid_res, VRT_BY_2(instr, black_sun, hipLaunchKernelGGL((k_render_pool<G, A, B, true>), VRT_POOL_ARGS)));
        else VRT_BY_GRID(grid_res, VRT_BY_2(instr, black_sun, hipLaunchKernelGGL((k_render_pool<G, A, B, false>), VRT_POOL_ARGS)));
    }
#undef VRT_POOL_ARGS
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
// the launches that take the twelve-wave geometry: the ones the dense variant takes at 128^3
bool pool_uses_dense12(int grid_res, bool restir, bool dense, const FrameParams& fp) {
    const bool black_sun = !((fp.light_color.x != 0.0f || fp.light_color.y != 0.0f || fp.light_color.z != 0.0f) && fp.light_weight != 0.0f);
    (void)grid_res;
    return !restir && dense && !black_sun;
}
hipError_t launch_mat_derived(hipStream_t st, const float* mats, float* mats_x) {
    hipLaunchKernelGGL(k_mat_derived, dim3(1), dim3(128), 0, st, mats, mats_x);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_gris(hipStream_t st, int grid_res, bool instr, const FrameParams& fp, const SceneData& sc, const GrisBuffers& gb, int r0, int r1) {
    hipLaunchKernelGGL(k_gris_prepare, dim3((fp.W + 63) / 64, (fp.row1 - fp.row0 + 3) / 4), dim3(256), 0, st, fp, sc, gb);
    // the tap angles of a pixel are hashed from its 8x8 tile in FRAME coordinates (pathtracer.py:834-836) and worked out once
    // per wave: the wave tiles have to sit on that grid, so the launch starts at the multiple of 8 at or below r0
    const int ra = r0 & ~7;
    const int tiles_x = (fp.W + 15) / 16, tiles_y = (r1 - ra + 15) / 16, band_w = (tiles_x + 7) / 8;
    dim3 g(8 * band_w * tiles_y), b(256);
#if VRT_GRIS_SPLIT
    if (instr) hipLaunchKernelGGL((k_gris_classify<true>), g, b, 0, st, fp, sc, gb, ra, r0, r1, tiles_x, band_w);
    else hipLaunchKernelGGL((k_gris_classify<false>), g, b, 0, st, fp, sc, gb, ra, r0, r1, tiles_x, band_w);
    VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, hipLaunchKernelGGL((k_gris_first<G, A>), g, b, 0, st, fp, sc, gb, ra, r0, r1, tiles_x, band_w)));
    VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, hipLaunchKernelGGL((k_gris<G, A, 2>), g, b, 0, st, fp, sc, gb, ra, r0, r1, tiles_x, band_w)));
#else
    VRT_BY_GRID(grid_res, VRT_BY_2(instr, false, hipLaunchKernelGGL((k_gris<G, A>), g, b, 0, st, fp, sc, gb, ra, r0, r1, tiles_x, band_w)));
#endif
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_temporal(hipStream_t st, const FrameParams& fp, const TemporalBuffers& tb, int r0, int r1, int n_samples) {
    dim3 g((fp.W + 63) / 64, (r1 - r0 + VRT_TEMPORAL_ROWS - 1) / VRT_TEMPORAL_ROWS), b(64 * VRT_TEMPORAL_ROWS);
    if (fp.stripe_period) hipLaunchKernelGGL(k_temporal_stripes, g, b, 0, st, fp, tb, r1 - r0, n_samples);   // (r0 = 0, r1 = the context's own rows)
    else hipLaunchKernelGGL(k_temporal, g, b, 0, st, fp, tb, r0, r1, n_samples);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_tonemap(hipStream_t st, const FrameParams& fp, const f3* hdr, f4* ldr, int r0, int r1) {
    dim3 g((fp.W + 63) / 64, (r1 - r0 + 3) / 4), b(256);
    hipLaunchKernelGGL(k_tonemap, g, b, 0, st, fp, hdr, ldr, r0, r1);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_tonemap8(hipStream_t st, const FrameParams& fp, const f3* hdr, uint32_t* ldr8, int r0, int r1) {
    dim3 g((fp.W + 63) / 64, (r1 - r0 + 3) / 4), b(256);
    hipLaunchKernelGGL(k_tonemap8, g, b, 0, st, fp, hdr, ldr8, r0, r1);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_detmath_probe(hipStream_t st, int op, int n, const float* a, const float* b, float* out) {
    hipLaunchKernelGGL(k_detmath_probe, dim3((n + 255) / 256), dim3(256), 0, st, op, n, a, b, out);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace vrt
