// kernel_boundary.cpp -- what a kernel boundary costs on MI355X while ANOTHER kernel is running: N tiny kernels (one wave, one store)
// queued back to back on a stream of their own, timed (a) on an idle chip, (b) beside a long kernel that only computes, (c) beside a
// long kernel that keeps writing a 64 MB buffer (dirty lines in every XCD's L2), (d) beside one that keeps reading it.
// Why: the accumulation pass of the renderer is one kernel per step beside the persistent render launches, and a step pays about
// 0.07 ms for it whatever it does (profiles/r04_zm_*).  The chip's eight L2s are written back / invalidated at kernel boundaries.
// build: hipcc --offload-arch=gfx950 -O3 -o kernel_boundary kernel_boundary.cpp        run: ./kernel_boundary
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_tiny(float* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.0f; }

// mode 0: arithmetic only; 1: writes its slice of buf over and over; 2: reads it over and over.  Ends after `rounds` rounds or when
// *stop (host memory) is set, whichever comes first: every wave gets there.
__global__ __launch_bounds__(256) void k_background(float* buf, size_t per_block, int mode, int rounds, volatile int* stop, float* sink) {
    float* mine = buf + (size_t)blockIdx.x * per_block;
    float acc = (float)threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        if (*stop) break;
        if (mode == 1) {
            for (size_t i = threadIdx.x; i < per_block; i += 256) mine[i] = acc + (float)r;
        } else if (mode == 2) {
            for (size_t i = threadIdx.x; i < per_block; i += 256) acc += mine[i];
        } else {
            for (int i = 0; i < 4096; i++) acc = acc * 1.0000001f + 0.5f;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;   // (keeps the loops)
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    const int blocks = 512;
    const size_t per_block = (64u << 20) / sizeof(float) / blocks;   // 64 MB in all
    float *buf, *out, *sink;
    int* stop;
    CK(hipMalloc(&buf, (size_t)blocks * per_block * sizeof(float)));
    CK(hipMemset(buf, 0, (size_t)blocks * per_block * sizeof(float)));
    CK(hipMalloc(&out, 4096 * sizeof(float)));
    CK(hipMalloc(&sink, sizeof(float)));
    CK(hipHostMalloc(&stop, sizeof(int)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 400;
    const char* names[4] = {"idle chip", "beside a kernel that computes", "beside a kernel that writes 64 MB over and over", "beside a kernel that reads 64 MB over and over"};
    for (int warm = 0; warm < 50; warm++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, out);
    CK(hipStreamSynchronize(sb));
    for (int c = 0; c < 4; c++) {
        *stop = 0;
        if (c > 0) {
            const int mode = c == 1 ? 0 : c == 2 ? 1 : 2;
            hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, mode, mode == 0 ? 200000 : 40000, (volatile int*)stop, sink);
            CK(hipGetLastError());
            std::this_thread::sleep_for(std::chrono::milliseconds(5));   // it is running
        }
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, out);
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const bool still = c > 0 && hipStreamQuery(sa) == hipErrorNotReady;
        (void)hipGetLastError();
        *stop = 1;
        CK(hipStreamSynchronize(sa));
        printf("%-52s %7.2f us per tiny kernel (%d back to back)%s\n", names[c], 1e3 * ms / N, N, c > 0 ? (still ? "" : "   [the background kernel had ended: too short]") : "");
    }
    // the same with 256 CUs' worth of work per small kernel (one wave per CU x 4): a kernel that spans every XCD
    for (int c = 0; c < 3; c += 2) {
        *stop = 0;
        if (c > 0) { hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, 1, 40000, (volatile int*)stop, sink); std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(1024), dim3(64), 0, sb, out);
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        *stop = 1;
        CK(hipStreamSynchronize(sa));
        printf("1024-wave kernels, %-34s %7.2f us per kernel\n", c == 0 ? "idle chip" : "beside the writing kernel", 1e3 * ms / N);
    }
    // how it grows with the share of the chip the other kernel holds (arithmetic only)
    for (int nb = 32; nb <= 512; nb *= 4) {   // (2048 workgroups are more waves than the chip holds: the tiny kernels then wait for workgroups of the other kernel to END, 94 s)
        *stop = 0;
        hipLaunchKernelGGL(k_background, dim3(nb), dim3(256), 0, sa, buf, per_block, 0, 200000, (volatile int*)stop, sink);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, out);
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        *stop = 1;
        CK(hipStreamSynchronize(sa));
        printf("beside %4d computing workgroups of 4 waves          %7.2f us per tiny kernel\n", nb, 1e3 * ms / 100);
    }
    return 0;
}
