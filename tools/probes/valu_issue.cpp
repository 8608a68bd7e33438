// valu_issue.cpp -- measured VALU issue cost on MI355X: cycles per wave64 vector instruction per SIMD, at 1 / 2 / 4 waves
// per SIMD, for v_fma_f32, v_rcp_f32, v_sqrt_f32 and the correctly rounded f32 division / square root the numeric contract
// compiles to (-fhip-fp32-correctly-rounded-divide-sqrt).  Cycles are shader cycles (s_memtime), so DVFS drops out.
// Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES ...` it also calibrates what
// the SQ counters report for a stream whose pipe occupancy is known.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o valu_issue valu_issue.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int OP>
__global__ void k_stream(float* out, unsigned long long* cyc, int iters, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = seed + (float)(threadIdx.x + i);
    const float m = 0.999f + seed * 1e-9f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int k = 0; k < iters; k++) {
        if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 1) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 2) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            REP16(X) REP16(X)
#undef X
        } else if (OP == 3) {   // 32 correctly rounded divisions
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = m / a[i];
        } else {                // 32 correctly rounded square roots
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = __builtin_sqrtf(a[i] + m);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int OP>
static int run(const char* name, int waves_per_simd, int iters, float* out, unsigned long long* cyc, int n_cu) {
    const int threads = 256 * waves_per_simd, blocks = n_cu;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_stream<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 16, 1.0f);  // warm
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_stream<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int n_waves = blocks * threads / 64;
    std::vector<unsigned long long> h(n_waves);
    CK(hipMemcpy(h.data(), cyc, n_waves * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[n_waves / 2], per_wave_ops = 32.0 * iters;
    // a SIMD hosts waves_per_simd waves, each issuing per_wave_ops ops in `med` cycles
    printf("%-22s waves/SIMD %d: %8.2f cycles per op per wave, %6.2f cycles per op per SIMD, kernel %.3f ms -> %.1f G wave-ops/s chip\n", name,
           waves_per_simd, med / per_wave_ops, med / (per_wave_ops * waves_per_simd), ms, n_waves * per_wave_ops / (ms * 1e-3) / 1e9);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, n_cu, p.clockRate);
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, (size_t)n_cu * 1024 * 4)); CK(hipMalloc(&cyc, (size_t)n_cu * 16 * 8));
    for (int w : {1, 2, 4}) {
        if (run<0>("v_fma_f32", w, 4096, out, cyc, n_cu)) return 1;
        if (run<1>("v_rcp_f32", w, 2048, out, cyc, n_cu)) return 1;
        if (run<2>("v_sqrt_f32", w, 2048, out, cyc, n_cu)) return 1;
        if (run<3>("a / b (IEEE)", w, 512, out, cyc, n_cu)) return 1;
        if (run<4>("sqrtf (IEEE)", w, 512, out, cyc, n_cu)) return 1;
    }
    return 0;
}
