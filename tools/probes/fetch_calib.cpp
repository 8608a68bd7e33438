// fetch_calib.cpp -- what rocprofv3's FETCH_SIZE reports for the access patterns of the render kernels: N random 4-byte and
// 8-byte reads from a table far larger than the caches (every read a different 128-byte line), and a 16-byte-per-lane
// streaming read of the same table.  Run under `rocprofv3 --pmc FETCH_SIZE` / `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`.
// build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ inline uint32_t pcg(uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (w >> 22) ^ w; }
template <class T>
__global__ void k_gather(const T* __restrict__ tab, uint32_t lines, T* out, int per_thread) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    T acc = 0;
    for (int k = 0; k < per_thread; k++) {
        const uint32_t line = pcg(tid * 977u + (uint32_t)k * 0x9E3779B9u) % lines;   // a random 128-byte line
        acc += tab[(size_t)line * (128 / sizeof(T)) + (pcg(line) % (128 / sizeof(T)))];
    }
    out[tid] = acc;
}
__global__ void k_stream16(const uint4* __restrict__ tab, size_t n16, uint4* out) {
    uint4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = tab[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    const size_t bytes = (size_t)2 << 30;   // 2 GiB: 8x the Infinity Cache
    void *tab, *out;
    CK(hipMalloc(&tab, bytes)); CK(hipMalloc(&out, 64 << 20));
    CK(hipMemset(tab, 1, bytes));
    const int blocks = 4096, threads = 256, per = 16;
    const uint32_t lines = (uint32_t)(bytes / 128);
    hipLaunchKernelGGL(k_gather<uint32_t>, dim3(blocks), dim3(threads), 0, 0, (const uint32_t*)tab, lines, (uint32_t*)out, per);
    hipLaunchKernelGGL(k_gather<unsigned long long>, dim3(blocks), dim3(threads), 0, 0, (const unsigned long long*)tab, lines, (unsigned long long*)out, per);
    hipLaunchKernelGGL(k_stream16, dim3(2048), dim3(256), 0, 0, (const uint4*)tab, bytes / 16, (uint4*)out);
    CK(hipDeviceSynchronize());
    printf("random reads per gather kernel: %d (4-byte, then 8-byte); streamed bytes: %zu\n", blocks * threads * per, bytes);
    return 0;
}
