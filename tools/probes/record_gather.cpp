// record_gather.cpp -- what a per-lane record read costs on gfx950, and what a wave-cooperative read of the same records
// costs.  k_gris (vrt_restir.h) reads one 128-byte and one 224-byte record per lane and tap: 22 dwordx4 loads whose 64 lanes
// touch 64 different cache lines each.  Three ways to bring REC_BYTES per lane into registers, same records, same arithmetic
// between the reads (WORK dependent fma per 16 bytes read, standing in for the shift):
//   lane   : every lane reads its own record, 16 bytes at a time (k_gris today)
//   coop   : 8 lanes read one 128-byte line together (one load instruction = 8 whole lines), through LDS to the owner
//   direct : as coop, the load writing LDS itself (global_load_lds_dwordx4), no registers in between
// Records are laid out like the frame's (1920 x 1080) and a wave's lanes are an 8x8 tile reading taps within +-24 pixels,
// as k_gris does.  Two workgroups of 256 per CU (70 KB of LDS each), k_gris' occupancy.
// build: hipcc --offload-arch=gfx950 -O3 -o record_gather record_gather.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ inline uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }
constexpr int W = 1920, H = 1080;

__device__ inline int tap_record(int u, int v, uint32_t seed, int t) {
    const uint32_t h = pcg(seed + (uint32_t)t * 0x9E3779B9u);
    int x = u + (int)(h % 49u) - 24, y = v + (int)((h >> 8) % 49u) - 24;
    x = x < 0 ? 0 : (x >= W ? W - 1 : x);
    y = y < 0 ? 0 : (y >= H ? H - 1 : y);
    return y * W + x;
}
__device__ inline float work(float acc, uint4 d, int n) {
    float a = __uint_as_float((d.x & 0x007FFFFFu) | 0x3F000000u), b = __uint_as_float((d.y & 0x007FFFFFu) | 0x3F000000u);
    a += __uint_as_float((d.z & 0x007FFFFFu) | 0x3F000000u) * __uint_as_float((d.w & 0x007FFFFFu) | 0x3F000000u);
    for (int i = 0; i < n; i++) acc = __builtin_fmaf(acc, a, b);
    return acc;
}

template <int CHUNKS, int MODE>   // CHUNKS x 16 bytes per record; record stride CHUNKS * 16
__global__ __launch_bounds__(256) void k_gather(const uint4* __restrict__ tab, float* out, int taps, int n_work) {
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4* my = lds + wave * (64 * CHUNKS);
    const int tiles_x = W / 8;
    float acc = 1.0f;
    for (int tile = blockIdx.x * 4 + wave; tile < tiles_x * (H / 8); tile += gridDim.x * 4) {
        const int u = (tile % tiles_x) * 8 + (lane & 7), v = (tile / tiles_x) * 8 + (lane >> 3);
        const uint32_t seed = (uint32_t)(v * W + u);
        for (int t = 0; t < taps; t++) {
            const int rec = tap_record(u, v, seed, t);
            uint4 d[CHUNKS];
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < CHUNKS; c++) d[c] = tab[(size_t)rec * CHUNKS + c];
            } else {
                // instruction k brings the 16-byte chunks [64k, 64k + 64) of the wave's 64 x CHUNKS chunks; chunk g belongs to
                // record g / CHUNKS (lane g / CHUNKS), position g % CHUNKS; consecutive lanes read consecutive addresses
#pragma unroll
                for (int k = 0; k < CHUNKS; k++) {
                    const int g = 64 * k + lane;
                    const int owner = g / CHUNKS, c = g % CHUNKS;
                    const int orec = __shfl(rec, owner);
                    const uint4* src = tab + (size_t)orec * CHUNKS + c;
                    if (MODE == 1) my[g] = *src;
                    else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(my + 64 * k), 16, 0, 0);
                }
                if (MODE == 2) __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                // the owner reads its record back: chunk c of lane l at l * CHUNKS + c; rotate the start by the lane so that the
                // lanes of one access fall into different banks
#pragma unroll
                for (int c = 0; c < CHUNKS; c++) d[c] = my[lane * CHUNKS + ((c + lane) % CHUNKS)];
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int c = 0; c < CHUNKS; c++) acc = work(acc, d[c], n_work);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int CHUNKS, int MODE>
static int run(const char* name, const uint4* tab, float* out, int taps, int n_work) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = 512;
    const size_t lds = 70 * 1024;
    CK(hipFuncSetAttribute((const void*)k_gather<CHUNKS, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((k_gather<CHUNKS, MODE>), dim3(blocks), dim3(256), lds, 0, tab, out, taps, n_work);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double recs = (double)W * H * taps;
    printf("%-7s %3d B/record  work %4d fma/16B  %8.3f ms  %7.2f G records/s  %7.1f GB/s\n", name, CHUNKS * 16, n_work, best, recs / best / 1e6, recs * CHUNKS * 16 / best / 1e6);
    return 0;
}

int main() {
    const size_t n = (size_t)W * H * 14;
    uint4* tab; float* out;
    CK(hipMalloc(&tab, n * 16)); CK(hipMalloc(&out, 512 * 256 * 4));
    CK(hipMemset(tab, 0x5a, n * 16));
    const int taps = 24;
    for (int n_work : {0, 16, 64, 128}) {
        if (run<8, 0>("lane", tab, out, taps, n_work)) return 1;
        if (run<8, 1>("coop", tab, out, taps, n_work)) return 1;
        if (run<8, 2>("direct", tab, out, taps, n_work)) return 1;
        if (run<14, 0>("lane", tab, out, taps, n_work)) return 1;
        if (run<14, 1>("coop", tab, out, taps, n_work)) return 1;
        if (run<14, 2>("direct", tab, out, taps, n_work)) return 1;
    }
    return 0;
}
