// walk_alone.cpp -- what a WALK-only kernel could do: the DDA stage of the pooled render kernel (vrt_pool.h / vrt_trace.h:
// walk_trip + descend_flat, the same code) by itself, fed from and draining into global memory, at 2, 3 and 4 waves per SIMD.
//
// The pooled kernel runs WALK inside a 208-register kernel whose budget is set by SHADE, two waves per SIMD, and its WALK
// passes run at 28-42 of 64 lanes because a wave only has the rays of its own 128-slot pool.  The round-2 review asked what a
// separate persistent WALK kernel (about 100 registers, four waves per SIMD, rays exchanged through rings in L2) would gain.
// This probe measures the UPPER BOUND of that design without building the rings: prepared walks (RayWalk records, 64 bytes) of
// real ray populations lie in a global array, ONE resident grid of 1..8 waves per SIMD walks them, every wave owning a
// contiguous stretch (hundreds of rays) and refilling idle lanes in the loop exactly as the WALK stage does (so the loop runs
// full until the stretch is used up); results go back as 16-byte records.
// No SHADE beside it, no LDS pools, no ring synchronisation: what it reaches is what the split could reach at best for the
// WALK share of a launch.  Ray populations: camera rays of the bench camera, and the first and second bounce rays from their
// hit points (cosine-distributed about the face normal) -- on the sparse S1-style scene and on the dense random fill.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt
//        -fno-gpu-flush-denormals-to-zero -Xclang -target-feature -Xclang -packed-fp32-ops -o walk_alone walk_alone.cpp
// run:   walk_alone <mat.bin: int8[128^3], index (x*128+y)*128+z>     (tools/probes/run_walk_alone.py writes the two scenes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../voxel_rt2_amd/csrc/vrt_types.h"
#include "../../voxel_rt2_amd/csrc/vrt_trace.h"
using namespace vrt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// the pooled kernel's LDS view of the 128^3 pyramid (LdsPyramid2<128, false>, vrt_kernels.hip)
struct LdsPyr {
    static constexpr int G = 128;
    static constexpr bool cull = false;
    static constexpr bool flat_descend = true;
    const unsigned long long* l0;
    const ulonglong2* l12;
    const unsigned long long* l2;
    const uint32_t* fine_base;
    const unsigned long long* fine;
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
struct WalkRec { float ox, oy, oz, dx, dy, dz, ix, iy, iz, t, far; int cx, cy, cz, alive, pad; };   // 64 bytes
struct WalkOut { float t; int cx, cy, czn; };                                                       // 16 bytes

__device__ inline uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }
__device__ inline float u01(uint32_t& s) { s = pcg(s); return (float)(s >> 8) * (1.0f / 16777216.0f); }

struct Tables { const unsigned long long *l0, *l1, *l2, *l0c; const uint32_t* l0c_base; uint32_t n_fine; };

template <int MINW>
__global__ __launch_bounds__(256, MINW) void k_walk(Tables tb, const WalkRec* __restrict__ recs, WalkOut* __restrict__ out, int n, int per_wave,
                                                    unsigned long long* counters) {
    __shared__ ulonglong2 s_l12[512];
    __shared__ unsigned long long s_l2[8];
    __shared__ unsigned long long s_fine[1024];
    __shared__ uint32_t s_base[512];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) {
        ulonglong2 v;
        v.x = tb.l1[i];
        v.y = tb.l2[(((i >> 8) & 1) << 2) | (((i >> 5) & 1) << 1) | ((i >> 2) & 1)];
        s_l12[i] = v;
        s_base[i] = tb.l0c_base[i];
    }
    if (threadIdx.x < 8) s_l2[threadIdx.x] = tb.l2[threadIdx.x];
    const uint32_t n_fine = tb.n_fine > 1024u ? 1024u : tb.n_fine;
    for (uint32_t i = threadIdx.x; i < n_fine; i += blockDim.x) s_fine[i] = tb.l0c[i];
    __syncthreads();
    LdsPyr P;
    P.l0 = tb.l0; P.l12 = s_l12; P.l2 = s_l2; P.fine_base = s_base; P.fine = s_fine; P.n_fine = n_fine;
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int head = wave * per_wave;
    const int end = head + per_wave < n ? head + per_wave : n;
    bool active = false, ended = false;
    RayWalk w;
    BrickCache bc; bc.key = -1; bc.word = 0ULL;
    CoarseWords cw; cw.w1 = 0ULL; cw.w2 = 0ULL; cw.fine_base = 0u;
    int mine = 0;
    unsigned long long lane_steps = 0ULL, trips = 0ULL;
    if (head >= end) return;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        const int n_idle = __popcll(idle);
        if (ended) {
            WalkOut o;
            o.t = w.t; o.cx = w.ix; o.cy = w.iy; o.czn = w.iz | (w.iters << 16);
            out[mine] = o;
            ended = false;
        }
        if (head >= end) {
            if (n_idle == 64) break;
        } else {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0u));
            const int idx = head + rank;
            if (!active && idx < end) {
                const WalkRec r = recs[idx];
                mine = idx;
                if (r.alive) {
                    w.o = mk3(r.ox, r.oy, r.oz); w.d = mk3(r.dx, r.dy, r.dz);
                    w.sd = mk3(sgn(w.d.x), sgn(w.d.y), sgn(w.d.z));
                    w.inv_dir = mk3(r.ix, r.iy, r.iz);
                    w.t = r.t; w.far = r.far; w.ix = r.cx; w.iy = r.cy; w.iz = r.cz; w.lod = 0; w.iters = 0; w.hn = mk3(0.0f);
                    coarse_fetch(P, w.ix, w.iy, w.iz, cw);
                    bc.key = -1;
                    active = true;
                } else {
                    WalkOut o; o.t = DM_INF; o.cx = -1; o.cy = -1; o.czn = -1;
                    out[idx] = o;
                }
            }
            head += (n_idle < end - head) ? n_idle : end - head;
        }
        const int target = (head < end) ? 16 : 64;
        do {
            if (active) {
                int nq;
                if (walk_trip(P, w, bc, cw, nq)) { active = false; ended = true; }
                lane_steps += 1ULL;
            }
            trips += 1ULL;
        } while (__popcll(__ballot(!active)) < target);
    }
    for (int off = 32; off > 0; off >>= 1) lane_steps += __shfl_down(lane_steps, off, 64);
    if (lane == 0) { atomicAdd(&counters[0], lane_steps); atomicAdd(&counters[1], trips); }
}

// ray set-up: camera rays, and bounce rays from the results of the previous set
__global__ void k_camera(WalkRec* recs, int W, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * H) return;
    const int u = i % W, v = i / W;
    // pinhole at (0.4, 0.5, 2.0) looking at the origin, vertical field of view 50 degrees (scene.py:28-29, pathtracer.py:89)
    const f3 pos = mk3(0.4f, 0.5f, 2.0f);
    const f3 fw = norm3(mk3(0.0f, 0.0f, 0.0f) - pos), rt = norm3(cross3(fw, mk3(0.0f, 1.0f, 0.0f))), up = cross3(rt, fw);
    const float th = 0.46630766f;  // tan(25 deg)
    const float sx = (2.0f * ((float)u + 0.5f) / (float)W - 1.0f) * th * (float)W / (float)H, sy = (2.0f * ((float)v + 0.5f) / (float)H - 1.0f) * th;
    const f3 d = norm3(fw + sx * rt + sy * up);
    const f3 o = 64.0f * pos + 64.0f;
    RayWalk w;
    const bool alive = walk_prepare<128, false>(o, d, w, nullptr);
    WalkRec r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z; r.dx = d.x; r.dy = d.y; r.dz = d.z; r.ix = w.inv_dir.x; r.iy = w.inv_dir.y; r.iz = w.inv_dir.z;
    r.t = w.t; r.far = w.far; r.cx = w.ix; r.cy = w.iy; r.cz = w.iz; r.alive = alive ? 1 : 0; r.pad = 0;
    recs[i] = r;
}
__global__ void k_bounce(const WalkRec* prev, const WalkOut* res, WalkRec* recs, int n, uint32_t seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    WalkRec r;
    r.alive = 0; r.pad = 0;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = r.ix = r.iy = r.iz = r.t = r.far = 0.0f; r.cx = r.cy = r.cz = -1;
    const WalkOut h = res[i];
    if (prev[i].alive && h.t < DM_INF) {
        const f3 po = mk3(prev[i].ox, prev[i].oy, prev[i].oz), pd = mk3(prev[i].dx, prev[i].dy, prev[i].dz);
        const f3 hp = po + pd * h.t;
        // face normal: the axis along which the hit point sits on a cell boundary, against the ray
        const f3 c = mk3((float)h.cx + 0.5f, (float)h.cy + 0.5f, (float)(h.czn & 0xffff) + 0.5f);
        const f3 q = abs3(hp - c);
        f3 nrm = (q.x >= q.y && q.x >= q.z) ? mk3(-sgn(pd.x), 0.0f, 0.0f) : (q.y >= q.z ? mk3(0.0f, -sgn(pd.y), 0.0f) : mk3(0.0f, 0.0f, -sgn(pd.z)));
        uint32_t s = seed + (uint32_t)i * 0x9E3779B9u;
        f3 d;
        for (;;) {  // cosine-distributed: normal + point in the unit sphere (math_utils.py:22-30)
            const f3 p = mk3(u01(s) * 2.0f - 1.0f, u01(s) * 2.0f - 1.0f, u01(s) * 2.0f - 1.0f);
            if (dot3(p, p) < 1.0f) { d = norm3(nrm + norm3(p) * 0.999f); break; }
        }
        const f3 o = hp + nrm * 1e-3f;
        RayWalk w;
        const bool alive = walk_prepare<128, false>(o, d, w, nullptr);
        r.ox = o.x; r.oy = o.y; r.oz = o.z; r.dx = d.x; r.dy = d.y; r.dz = d.z; r.ix = w.inv_dir.x; r.iy = w.inv_dir.y; r.iz = w.inv_dir.z;
        r.t = w.t; r.far = w.far; r.cx = w.ix; r.cy = w.iy; r.cz = w.iz; r.alive = alive ? 1 : 0;
    }
    recs[i] = r;
}

// One resident grid: n_cu x `occ` workgroups of four waves = `occ` waves per SIMD, every wave a stretch of n / waves rays (the
// kernel needs 55 registers, so up to eight waves per SIMD fit: occupancy is set by the grid, not by the register budget).
static int run(const char* label, Tables tb, const WalkRec* recs, WalkOut* out, int n, unsigned long long* d_cnt, int n_cu, int occ) {
    constexpr int MINW = 2;
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_walk<MINW>, 256, 0));
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)k_walk<MINW>));
    if (occ > per_cu) occ = per_cu;
    const int blocks = n_cu * occ, waves = blocks * 4, per_wave = (n + waves - 1) / waves;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    unsigned long long cnt[2] = {0, 0};
    for (int rep = 0; rep < 4; rep++) {
        CK(hipMemset(d_cnt, 0, 16));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_walk<MINW>, dim3(blocks), dim3(256), 0, 0, tb, recs, out, n, per_wave, d_cnt);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        CK(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    }
    printf("  %-8s %d waves/SIMD (%3d VGPR), %5d rays per wave: %7.3f ms  %6.1f G lane-steps/s  %5.2f G wave-trips/s  %5.1f of 64 lanes per trip  "
           "%5.2f steps per ray\n", label, occ, fa.numRegs, per_wave, best, (double)cnt[0] / best * 1e-6, (double)cnt[1] / best * 1e-6,
           (double)cnt[0] / (double)cnt[1], (double)cnt[0] / n);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("usage: walk_alone mat.bin\n"); return 1; }
    const int G = 128;
    std::vector<int8_t> mat((size_t)G * G * G);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(mat.data(), 1, mat.size(), f) != mat.size()) { printf("cannot read %s\n", argv[1]); return 1; }
    fclose(f);
    // the bit-brick pyramid (vrt_kernels.hip: k_build_l0 / k_build_coarse / k_build_l0c)
    std::vector<unsigned long long> l0(32 * 32 * 32, 0ULL), l1(512, 0ULL), l2(8, 0ULL), l0c;
    std::vector<uint32_t> base(513, 0u);
    size_t solid = 0;
    for (int x = 0; x < G; x++) for (int y = 0; y < G; y++) for (int z = 0; z < G; z++)
        if (mat[((size_t)x * G + y) * G + z] > 0) { l0[((z >> 2) * 32 + (y >> 2)) * 32 + (x >> 2)] |= 1ULL << ((z & 3) * 16 + (y & 3) * 4 + (x & 3)); solid++; }
    for (int b = 0; b < 32768; b++) if (l0[b]) { const int bx = b & 31, by = (b >> 5) & 31, bz = b >> 10; l1[((bz >> 2) * 8 + (by >> 2)) * 8 + (bx >> 2)] |= 1ULL << ((bz & 3) * 16 + (by & 3) * 4 + (bx & 3)); }
    for (int b = 0; b < 512; b++) if (l1[b]) { const int bx = b & 7, by = (b >> 3) & 7, bz = b >> 6; l2[((bz >> 2) * 2 + (by >> 2)) * 2 + (bx >> 2)] |= 1ULL << ((bz & 3) * 16 + (by & 3) * 4 + (bx & 3)); }
    for (int i = 0; i < 512; i++) {
        base[i] = (uint32_t)l0c.size();
        const int bx1 = i & 7, by1 = (i >> 3) & 7, bz1 = i >> 6;
        for (int b = 0; b < 64; b++) if ((l1[i] >> b) & 1ULL) l0c.push_back(l0[(((bz1 * 4 + (b >> 4)) * 32) + by1 * 4 + ((b >> 2) & 3)) * 32 + bx1 * 4 + (b & 3)]);
    }
    base[512] = (uint32_t)l0c.size();
    if (l0c.empty()) l0c.push_back(0ULL);
    printf("%s: %zu solid voxels (%.2f %%), %u non-empty fine words\n", argv[1], solid, 100.0 * solid / mat.size(), base[512]);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned long long *d_l0, *d_l1, *d_l2, *d_l0c, *d_cnt;
    uint32_t* d_base;
    CK(hipMalloc(&d_l0, l0.size() * 8)); CK(hipMalloc(&d_l1, 4096)); CK(hipMalloc(&d_l2, 64)); CK(hipMalloc(&d_l0c, l0c.size() * 8)); CK(hipMalloc(&d_base, 513 * 4)); CK(hipMalloc(&d_cnt, 16));
    CK(hipMemcpy(d_l0, l0.data(), l0.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_l1, l1.data(), 4096, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_l2, l2.data(), 64, hipMemcpyHostToDevice)); CK(hipMemcpy(d_l0c, l0c.data(), l0c.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_base, base.data(), 513 * 4, hipMemcpyHostToDevice));
    Tables tb{d_l0, d_l1, d_l2, d_l0c, d_base, base[512]};
    const int W = 1920, H = 1080, n = W * H;
    WalkRec* recs[3];
    WalkOut* outs[3];
    for (int k = 0; k < 3; k++) { CK(hipMalloc(&recs[k], (size_t)n * sizeof(WalkRec))); CK(hipMalloc(&outs[k], (size_t)n * sizeof(WalkOut))); }
    hipLaunchKernelGGL(k_camera, dim3((n + 255) / 256), dim3(256), 0, 0, recs[0], W, H);
    const char* names[3] = {"camera", "bounce 1", "bounce 2"};
    for (int k = 0; k < 3; k++) {
        if (k > 0) hipLaunchKernelGGL(k_bounce, dim3((n + 255) / 256), dim3(256), 0, 0, recs[k - 1], outs[k - 1], recs[k], n, 1234u * (uint32_t)k);
        CK(hipDeviceSynchronize());
        printf("%s rays:\n", names[k]);
        for (int occ : {1, 2, 3, 4, 6, 8})
            if (run(names[k], tb, recs[k], outs[k], n, d_cnt, prop.multiProcessorCount, occ)) return 1;
    }
    return 0;
}
