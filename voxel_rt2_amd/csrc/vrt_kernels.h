// vrt_kernels.h -- launcher declarations shared by vrt_kernels.hip, vrt_sky_kernels.hip and vrt_api.hip.
#ifndef VRT_KERNELS_H
#define VRT_KERNELS_H

#include <hip/hip_runtime.h>
#include "vrt_types.h"
#include "vrt_trace.h"
#include "vrt_bsdf.h"
#include "vrt_sky.h"
#include "vrt_path.h"
#include "vrt_pool.h"
#include "vrt_restir.h"
#include "vrt_temporal.h"

#define VRT_RENDER_THREADS 256
#ifndef VRT_RENDER_MIN_WAVES
#define VRT_RENDER_MIN_WAVES 2   // waves per SIMD the register allocator must leave room for (tuned on MI355X)
#endif

// Work distribution: the render kernels pull (tile, sample, pixel) items from head words with returning atomics.
// One word saturates near 88 pulls/us chip-wide and answers in ~3 us with 256 CUs pulling (MI355X_MICROARCH.md, row
// `dequeue`), so the pooled kernel splits the items into VRT_WORK_HEADS contiguous ranges, one head per XCD on a
// 128-byte line of its own; a wave pulls from the head of its XCD and moves on to the next head when that range is
// used up.  Sixteen sets of heads rotate with the launch number (a launch zeroes the set eight launches ahead: vrt_kernels.hip).
#define VRT_WORK_HEADS 8
#define VRT_WORK_HEAD_STRIDE 32    // uints between heads
#ifndef VRT_POOL_WAVES
#define VRT_POOL_WAVES 4           // waves (= path pools) per workgroup of the pooled render kernel
#endif
#ifndef VRT_POOL_MIN_WAVES
#define VRT_POOL_MIN_WAVES 2
#endif

namespace vrt {

// grid_res (128 or 256) selects the kernel instantiation everywhere below (GridDim, vrt_types.h)
hipError_t launch_prepare(hipStream_t st, int grid_res, const int8_t* mat, const uint8_t* rgb, uint32_t* grid, unsigned long long* l0,
                          unsigned long long* l1, unsigned long long* l2, unsigned long long* l3, unsigned long long* l0c, uint32_t* l0c_base,
                          float* cull /*[6]: cull_ray()'s box, vrt_trace.h*/);
hipError_t query_render_residency(int grid_res, bool restir, bool instr, int* blocks_per_cu);
hipError_t launch_render(hipStream_t st, int grid_res, bool restir, bool instr, int n_blocks, const FrameParams& fp, const SceneData& sc,
                         const PixelBuffers& out, unsigned* work_counters, unsigned launch_seq, int n_samples, int chunk_override);
// pooled schedule (vrt_pool.h).  `cold` holds pool_scratch_bytes(grid_res, restir, n_blocks) bytes.
hipError_t query_render_pool_residency(int grid_res, bool restir, bool instr, int* blocks_per_cu);
int pool_waves_per_block(int grid_res);
size_t pool_scratch_bytes(int grid_res, bool restir, int n_blocks, int n_blocks_dense12);
hipError_t query_render_pool_dense12_residency(int grid_res, bool instr, int* blocks_per_cu);
bool pool_uses_dense12(int grid_res, bool restir, bool dense, const FrameParams& fp);
hipError_t launch_render_pool(hipStream_t st, int grid_res, bool restir, bool instr, int n_blocks, const FrameParams& fp, const SceneData& sc,
                              const PixelBuffers& out, unsigned* work_counters, unsigned launch_seq, int n_samples, uint32_t* cold,
                              uint32_t* drain_signal,   // signal memory (or null): receives launch_seq + 1 when the launch starts to drain
                              PrimaryRecord* prim_cache,        // per-pixel camera-ray records shared by the fused samples (or null), npix entries
                              bool cull,                        // the instantiation that tests rays against sc.cull (cull_ray, vrt_trace.h)
                              bool dense,                       // ... whose SHADE walks its shadow rays with the branchy descent (dense grids)
                              bool dense12);                    // ... on the twelve-wave geometry (pool_uses_dense12; n_blocks counts ITS workgroups)
hipError_t launch_mat_derived(hipStream_t st, const float* mats, float* mats_x /*[128][8]*/);  // after every material upload
// spatial reuse over rows [r0, r1); first a per-pixel prepare pass over all rows the launch holds (fp.row0..fp.row1) into gb.geo / gb.src
hipError_t launch_gris(hipStream_t st, int grid_res, bool instr, const FrameParams& fp, const SceneData& sc, const GrisBuffers& gb, int r0, int r1);
hipError_t launch_temporal(hipStream_t st, const FrameParams& fp, const TemporalBuffers& tb, int r0, int r1, int n_samples);
hipError_t launch_tonemap(hipStream_t st, const FrameParams& fp, const f3* hdr, f4* ldr, int r0, int r1);
hipError_t launch_tonemap8(hipStream_t st, const FrameParams& fp, const f3* hdr, uint32_t* ldr8 /* rgba, 8 bits each */, int r0, int r1);
hipError_t launch_diag_read(hipStream_t st, unsigned long long* out, int reset);  // -DVRT_DIAG_REGIONS builds only
hipError_t launch_detmath_probe(hipStream_t st, int op, int n, const float* a, const float* b, float* out);

// sky precompute (vrt_sky_kernels.hip)
struct SkyPrecompute {
    float* scattering;        // [res][res][3]
    float* transmittance;     // [res][res][3]
    uint16_t* trans_lut;      // [256][128][3] binary16
    const uint8_t* cloud_tex; // [256][256][3]
    float* cloud_ambient;     // [3]
    int res;
    float fres;
    int use_clouds;
    uint32_t seed;
};
hipError_t launch_sky_probe(hipStream_t st, const SkyPrecompute& sp, int op, int n, const float* in, int in_stride, float* out, int out_stride, f3 ambient);
hipError_t launch_sky_prepare(hipStream_t st, const SkyPrecompute& sp, f3 sun_dir, f3 sun_col, float sun_cos);
hipError_t launch_sky_clouds(hipStream_t st, const SkyPrecompute& sp, f3 sun_dir, f3 sun_col, float sun_cos, int max_samples,
                             uint32_t pass, int u0, int u1);   // table columns [u0, u1)
hipError_t launch_sky_slice(hipStream_t st, const SkyPrecompute& sp, f3 sun_dir, f3 sun_col, float sun_cos, int u0, int u1);

}  // namespace vrt
#endif
