// kernel_boundary.cpp -- what a kernel boundary costs on MI355X while ANOTHER kernel is running: N tiny kernels (one wave, one store)
// queued back to back on a stream of their own, timed (a) on an idle chip, (b) beside a long kernel that only computes, (c) beside a
// long kernel that keeps writing a 64 MB buffer (dirty lines in every XCD's L2), (d) beside one that keeps reading it.
// Why: the accumulation pass of the renderer is one kernel per step beside the persistent render launches, and a step pays about
// 0.07 ms for it whatever it does (profiles/r04_zm_*).  The chip's eight L2s are written back / invalidated at kernel boundaries.
// build: hipcc --offload-arch=gfx950 -O3 -o kernel_boundary kernel_boundary.cpp        run: ./kernel_boundary
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_tiny(float* out) { if (threadIdx.x == 0) out[blockIdx.x] = 1.0f; }
__global__ void k_tiny_prio(float* out) { __builtin_amdgcn_s_setprio(3); if (threadIdx.x == 0) out[blockIdx.x] = 1.0f; }

// mode 0: arithmetic only; 1: writes its slice of buf over and over; 2: reads it over and over.  Ends after `rounds` rounds or when
// *stop (host memory) is set, whichever comes first: every wave gets there.
__global__ __launch_bounds__(256) void k_background(float* buf, size_t per_block, int mode, int rounds, volatile int* stop, float* sink) {
    float* mine = buf + (size_t)blockIdx.x * per_block;
    float acc = (float)threadIdx.x;
    for (int r = 0; r < rounds; r++) {
        if ((r & 255) == 0 && threadIdx.x == 0 && *stop) mine = nullptr;   // (host memory, looked at rarely and by one lane: polling it from every wave is PCIe traffic that the small kernels' completion signals then queue behind)
        if (__syncthreads_or(mine == nullptr)) break;
        if (mode == 1) {
            for (size_t i = threadIdx.x; i < per_block; i += 256) mine[i] = acc + (float)r;
        } else if (mode == 2) {
            for (size_t i = threadIdx.x; i < per_block; i += 256) acc += mine[i];
        } else if (mode == 3) {   // arithmetic that gives the SIMD away every 64 instructions
            for (int i = 0; i < 4096; i++) { acc = acc * 1.0000001f + 0.5f; if ((i & 63) == 63) __builtin_amdgcn_s_sleep(1); }
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
    // the same small kernels on a stream of the HIGHEST priority the device offers
    {
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        hipStream_t sp;
        CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi));
        for (int warm = 0; warm < 50; warm++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sp, out);
        CK(hipStreamSynchronize(sp));
        for (int c = 1; c < 3; c++) {
            *stop = 0;
            hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, c == 1 ? 0 : 1, c == 1 ? 200000 : 40000, (volatile int*)stop, sink);
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
            CK(hipEventRecord(e0, sp));
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sp, out);
            CK(hipEventRecord(e1, sp));
            CK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            *stop = 1;
            CK(hipStreamSynchronize(sa));
            printf("priority %d stream (range %d..%d), %-28s %7.2f us per tiny kernel\n", hi, lo, hi, c == 1 ? "beside the computing kernel" : "beside the writing kernel", 1e3 * ms / N);
        }
    }
    // the small kernel's wave asking for priority (s_setprio 3); and a computing kernel that sleeps a little every 64 instructions
    for (int c = 0; c < 2; c++) {
        *stop = 0;
        hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, c == 0 ? 0 : 3, 200000, (volatile int*)stop, sink);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < N; i++) { if (c == 0) hipLaunchKernelGGL(k_tiny_prio, dim3(1), dim3(64), 0, sb, out); else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, out); }
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        *stop = 1;
        CK(hipStreamSynchronize(sa));
        printf("%-52s %7.2f us per tiny kernel\n", c == 0 ? "s_setprio 3 in the tiny kernel, beside computing" : "beside a computing kernel that sleeps every 64 instr", 1e3 * ms / N);
    }
    // the renderer's pattern: the small kernel of step i on stream B waits for an event of a small kernel on stream C and is followed
    // by an event a later kernel on C waits for (cross-queue dependencies either side of it), N steps queued at once
    {
        hipStream_t sc;
        CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
        const int M = 200;
        static hipEvent_t ec[200], eb[200];
        for (int flavour = 0; flavour < 3; flavour++) {
        const unsigned ef = hipEventDisableTiming | (flavour == 1 ? hipEventReleaseToDevice : 0u) | (flavour == 2 ? hipEventDisableSystemFence : 0u);
        printf("events created with %s\n", flavour == 0 ? "hipEventDisableTiming" : flavour == 1 ? "hipEventDisableTiming | hipEventReleaseToDevice" : "hipEventDisableTiming | hipEventDisableSystemFence  (no fence at all: ordering only -- NOT a way to hand data over)");
        for (int i = 0; i < M; i++) { CK(hipEventCreateWithFlags(&ec[i], ef)); CK(hipEventCreateWithFlags(&eb[i], ef)); }
        for (int c = 0; c < 3; c++) {
            *stop = 0;
            if (c > 0) { hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, c == 1 ? 0 : 1, c == 1 ? 200000 : 40000, (volatile int*)stop, sink); std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
            CK(hipEventRecord(e0, sb));
            for (int i = 0; i < M; i++) {
                if (i >= 4) CK(hipStreamWaitEvent(sc, eb[i - 4], 0));          // ("the copy this launch writes is free again")
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sc, out + 1);
                CK(hipEventRecord(ec[i], sc));
                CK(hipStreamWaitEvent(sb, ec[i], 0));
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, out);
                CK(hipEventRecord(eb[i], sb));
            }
            CK(hipEventRecord(e1, sb));
            CK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            *stop = 1;
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sc));
            printf("kernel on C -> event -> kernel on B -> event, %-26s %7.2f us per step\n", c == 0 ? "idle chip" : c == 1 ? "beside the computing kernel" : "beside the writing kernel", 1e3 * ms / M);
        }
        for (int i = 0; i < M; i++) { CK(hipEventDestroy(ec[i])); CK(hipEventDestroy(eb[i])); }
        }
    }
    // the cross-queue chain again, the events recorded BY the kernels' own completion (hipExtLaunchKernelGGL's stopEvent: no marker packet)
    {
        hipStream_t sc;
        CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
        const int M = 200;
        static hipEvent_t ec[200], eb[200];
        for (int i = 0; i < M; i++) { CK(hipEventCreateWithFlags(&ec[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eb[i], hipEventDisableTiming)); }
        for (int c = 0; c < 3; c++) {
            *stop = 0;
            if (c > 0) { hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, c == 1 ? 0 : 1, c == 1 ? 200000 : 40000, (volatile int*)stop, sink); std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
            CK(hipEventRecord(e0, sb));
            for (int i = 0; i < M; i++) {
                if (i >= 4) CK(hipStreamWaitEvent(sc, eb[i - 4], 0));
                hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sc, nullptr, ec[i], 0, out + 1);
                CK(hipStreamWaitEvent(sb, ec[i], 0));
                hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, nullptr, eb[i], 0, out);
            }
            CK(hipEventRecord(e1, sb));
            CK(hipEventSynchronize(e1));
            float ms = 0.0f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            *stop = 1;
            CK(hipStreamSynchronize(sa));
            CK(hipStreamSynchronize(sc));
            printf("the chain with stopEvent instead of hipEventRecord, %-26s %7.2f us per step\n", c == 0 ? "idle chip" : c == 1 ? "beside the computing kernel" : "beside the writing kernel", 1e3 * ms / M);
        }
    }
    // one queue only: per step [wait for an event that completed long ago] [small kernel] [record an event nobody waits for yet]
    {
        const int M = 200;
        static hipEvent_t ed[200];
        hipEvent_t done;
        CK(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        CK(hipEventRecord(done, sb));
        CK(hipStreamSynchronize(sb));
        for (int i = 0; i < M; i++) CK(hipEventCreateWithFlags(&ed[i], hipEventDisableTiming));
        for (int what = 0; what < 3; what++)
            for (int c = 0; c < 3; c += 2) {
                *stop = 0;
                if (c > 0) { hipLaunchKernelGGL(k_background, dim3(blocks), dim3(256), 0, sa, buf, per_block, 1, 40000, (volatile int*)stop, sink); std::this_thread::sleep_for(std::chrono::milliseconds(5)); }
                CK(hipEventRecord(e0, sb));
                for (int i = 0; i < M; i++) {
                    if (what >= 1) CK(hipStreamWaitEvent(sb, done, 0));
                    hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sb, out);
                    if (what >= 2 || what == 0) CK(hipEventRecord(ed[i], sb));
                }
                CK(hipEventRecord(e1, sb));
                CK(hipEventSynchronize(e1));
                float ms = 0.0f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                *stop = 1;
                CK(hipStreamSynchronize(sa));
                printf("one queue, per step %-44s %-26s %7.2f us\n", what == 0 ? "[kernel][record]" : what == 1 ? "[wait on a completed event][kernel]" : "[wait on a completed event][kernel][record]", c == 0 ? "idle chip" : "beside the writing kernel", 1e3 * ms / M);
            }
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
