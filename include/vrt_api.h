/* vrt_api.h -- C ABI of libvrt_hip.so, the MI355X (gfx950) replacement for the reference's
 * renderer/{pathtracer,raytracer,bsdf,atmos,reservoir,voxel_world}.py.
 *
 * The reference (taichi-dev/voxel-rt2) has no FFI: its seam is the Python object
 * `Renderer` that scene.py drives.  Every entry point below names the
 * reference interface it replaces as /root/reference-relative file:line.  The Python facade
 * voxel_rt2_amd/renderer.py binds these with ctypes and re-exposes the reference's method names.
 *
 * Conventions: single host thread per context; every int-returning function gives 0 or a
 * negative VRT_E_* code and vrt_last_error() a thread-local message; the library owns all
 * device memory; host pointers are borrowed for the duration of the call; work is queued on the
 * context's own HIP stream and is asynchronous until a fetch / sync / get_stats call.
 * There is NO CPU backend: vrt_create fails if no gfx950 device is usable.
 *
 * Layouts: images are row-major [row v][column u], v = 0 at the bottom like the reference's
 * (u, v) field indices.  Voxel arrays are [x+G/2][y+G/2][z+G/2] (C order; G = grid_res), the index space of
 * voxel_world.py:14-18.  Matrices are row-major 4x4 in mathematical convention (element
 * [row*4+col]); the facade has already transposed the glm-ordered arrays the reference receives
 * (pathtracer.py:266-268, 278-280) and supplies the inverses.
 */
#ifndef VRT_API_H
#define VRT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vrt_ctx vrt_ctx;

enum {
    VRT_OK = 0,
    VRT_E_INVALID = -1,   /* bad argument / unsupported configuration */
    VRT_E_DEVICE = -2,    /* HIP error or no gfx950 device */
    VRT_E_STATE = -3      /* call out of order (e.g. accumulate before prepare) */
};

/* Renderer.__init__(dx, image_res, up, voxel_edges, exposure): pathtracer.py:28-136, plus the
 * module constants USE_RESTIR_PT / MAX_RAY_DEPTH (pathtracer.py:15-17) and the skybox size
 * (atmos.py:66-67) made into parameters. */
typedef struct vrt_config {
    int32_t width, height;        /* image_res */
    int32_t grid_res;             /* voxel_grid_res: 128 (pathtracer.py:83) or 256 (BASELINE config 5; VoxelWorld and
                                     VoxelOctreeRaytracer take it as a parameter: voxel_world.py:6, raytracer.py:7-9) */
    float dx;                     /* voxel size in world units, must be 2 / grid_res: the grid spans the world box
                                     [-1,1]^3 (scene.py:11: 1/64 at 128) */
    float voxel_edges;            /* voxel_world.py:25 */
    float exposure;               /* pathtracer.py:58 */
    int32_t max_depth;            /* MAX_RAY_DEPTH, 1..64 */
    int32_t use_restir;           /* USE_RESTIR_PT */
    uint32_t seed;                /* seed of the per-pixel random streams (replaces ti.random) */
    int32_t sky_res;              /* skybox edge length, 3840 in the reference; 0 = no sky tables */
    int32_t device;               /* HIP device ordinal */
    int32_t row_begin, row_end;   /* rows [row_begin,row_end) this context produces; 0,0 = all.
                                     Row-tile sharding across GPUs (one context per process). */
} vrt_config;

/* the scalar fields scene.py pokes with [None] (scene.py:148-169) and
 * Renderer.set_directional_light (pathtracer.py:139-144) */
typedef struct vrt_scene_params {
    float floor_height;
    float floor_color[3];
    int32_t floor_material;
    float background_color[3];
    float light_direction[3];     /* normalised */
    float light_cos_theta_max;    /* cos(cone_angle / 2) */
    float light_color[3];
    float light_weight;           /* 3.0 once set_directional_light ran (pathtracer.py:144) */
    int32_t use_physical_sky;
    int32_t use_clouds;
} vrt_scene_params;

/* set_camera_pos / set_proj_mat / set_view_mat / set_camera_is_moving / set_render_scale /
 * set_max_samples: pathtracer.py:146-150, 246-281, 1306-1307 */
typedef struct vrt_camera {
    float view[16], proj[16], view_inv[16], proj_inv[16];
    float pos[3];
    uint32_t jitter_index;        /* how many times set_proj_mat ran: selects the TAA jitter draw
                                     (pathtracer.py:264-265) from random stream 3 */
    int32_t camera_is_moving;
    float render_scale;
    float max_accum_frames;
} vrt_camera;

typedef struct vrt_stats {
    uint64_t path_samples;        /* pixels x accumulate passes since creation */
    uint64_t rays, dda_iters, occupancy_queries, closest_hits, sky_lookups; /* instrumented build only, else 0 */
    double render_ms, temporal_ms, gris_ms;  /* summed device time of the kernels (hipEvent).  Launches of up to 12 M work items (ReSTIR off)
                                                carry their timers one time in eight (the events cost a step of 0.15-0.3 ms 3-25 % of its
                                                rate); the sum is then the timed launches' mean times the number of all launches */
    uint32_t render_launches, temporal_launches, gris_launches;
    uint32_t pipeline_flags;      /* bit 0: launches of consecutive vrt_accumulate calls overlap; bit 1: the next launch's dispatch
                                     is held until the running one starts to drain (stream wait on a kernel-raised word; left out
                                     where a self-test finds that queue operations are serialised, e.g. under rocprofv3 --pmc);
                                     bits 2..4: HALF the render launches the pipeline keeps in flight (1, 2, 4 for 2, 4, 8; 0 while not overlapped);
                                     bits 5..7: a launch takes 1 / this many of the workgroup slots (1, 2 or 4);
                                     bits 8..23: times the host released that wait (error paths, synchronisation watchdog);
                                     bits 24..31: times the pipeline was drained to change its depth (a caller that changes the
                                     sample count of its calls: the depth follows the launch size) */
} vrt_stats;

/* buffers readable through vrt_fetch_buffer (tests and the multi-GPU gather) */
enum {
    VRT_BUF_GBUF_DEPTH = 1,       /* f32  [H][W]      gbuff_depth            pathtracer.py:114 */
    VRT_BUF_GBUF_NORMAL = 2,      /* f16x2[H][W]      gbuff_normals          :113 */
    VRT_BUF_GBUF_POSITION = 3,    /* f32x3[H][W]      gbuff_position         :116 */
    VRT_BUF_GBUF_MAT = 4,         /* u32  [H][W]      gbuff_mat_id           :112 */
    VRT_BUF_GBUF_REFL_DEPTH = 5,  /* f32  [H][W]      gbuff_depth_reflection :115 */
    VRT_BUF_HISTORY_DIFFUSE = 6,  /* f32x4[H][W]      history_buffer[...,0]  :43 */
    VRT_BUF_HISTORY_SPECULAR = 7, /* f32x4[H][W]      history_buffer_specular[...,0] :44 */
    VRT_BUF_SKY_SCATTERING = 8,   /* f32x3[R][R]      atmos.py:68 */
    VRT_BUF_SKY_TRANSMITTANCE = 9,/* f32x3[R][R]      atmos.py:69 */
    VRT_BUF_TRANS_LUT = 10        /* f16x3[256][128]  atmos.py:64 */
};

/* Renderer.__init__ (pathtracer.py:28). NULL on failure; see vrt_last_error(). */
vrt_ctx* vrt_create(const vrt_config* cfg);
void vrt_destroy(vrt_ctx* ctx);

/* Renderer.set_voxel / get_voxel storage (pathtracer.py:1325-1334, voxel_world.py:7-18):
 * mat int8[G^3], rgb uint8[G^3][3] with G = grid_res, index [x+G/2][y+G/2][z+G/2]. */
int vrt_upload_voxels(vrt_ctx* ctx, const int8_t* mat, const uint8_t* rgb);
/* MaterialList (materials.py:48-112): 128 rows of 14 f32 in bsdf.py:26-37 field order. */
int vrt_upload_materials(vrt_ctx* ctx, const float* table);
/* Atmos.load_textures (atmos.py:85-87): cloud tile uint8[256][256][3] indexed [x][y]. */
int vrt_upload_cloud_texture(vrt_ctx* ctx, const uint8_t* rgb);
int vrt_set_scene(vrt_ctx* ctx, const vrt_scene_params* scene);
int vrt_set_camera(vrt_ctx* ctx, const vrt_camera* cam);
/* Renderer.prepare_data (pathtracer.py:314-323): packed voxel grid, occupancy pyramid,
 * and with use_physical_sky the transmittance LUT + cloud ambient + cleared sky tables. */
int vrt_prepare(vrt_ctx* ctx);
/* Renderer.accumulate_clouds / compute_atmosphere (pathtracer.py:325-329) */
int vrt_sky_accumulate_clouds(vrt_ctx* ctx, int max_samples);
int vrt_sky_compute_slice(vrt_ctx* ctx, int slice_idx, int max_slices);
/* The same precompute split across GPUs (the reference already slices the atmosphere pass by table columns,
 * atmos.py:159-164, scene.py:243-253): one cloud pass (atmos.py:140-157) over the columns of ONE slice, and the copy of table
 * columns between the library's tables and caller-owned device memory that an all-gather needs.  A texel depends on no other
 * texel in either pass, so a rank that owns slice r runs the cloud passes and the atmosphere pass on slice r only and the
 * ranks exchange columns: voxel_rt2_amd/parallel.py, precompute_sky_sharded(). */
int vrt_sky_accumulate_clouds_slice(vrt_ctx* ctx, int max_samples, int slice_idx, int max_slices);
int vrt_sky_table_io(vrt_ctx* ctx, int which /* VRT_BUF_SKY_SCATTERING | VRT_BUF_SKY_TRANSMITTANCE */, int u0, int u1,
                     void* device_ptr /* f32[u1-u0][sky_res][3] */, int to_library);
/* Renderer.accumulate (pathtracer.py:1310-1319), n_samples times */
int vrt_accumulate(vrt_ctx* ctx, int n_samples);
/* Renderer.reset_framebuffer (pathtracer.py:664-668) / copy_prev_matrices (283-287) */
int vrt_reset(vrt_ctx* ctx);
int vrt_end_frame(vrt_ctx* ctx);
/* color_buffer after accumulate() = the HDR frame: f32[H][W][3].  Rows outside
 * [row_begin,row_end) are zero. */
int vrt_fetch_hdr(vrt_ctx* ctx, float* out);
/* same, this context's rows only, copied device-to-device into caller-owned device memory
 * (f32[row_end-row_begin][W][3]); the call returns after the copy completed. */
int vrt_fetch_hdr_device(vrt_ctx* ctx, void* device_ptr);
/* same copy, only queued on the context's stream (no host synchronisation) */
int vrt_fetch_hdr_device_async(vrt_ctx* ctx, void* device_ptr);
/* Queue all further work of this context on the caller's HIP stream (e.g. torch.cuda.current_stream().cuda_stream),
 * so that rendering, the tile copy and a following RCCL collective are ordered on the device without host
 * round trips; NULL returns to a private stream.  The stream must outlive the context or be reset first.
 * Everything a caller can observe (results, fetches, stats) is ordered on this stream.  Render launches of
 * consecutive vrt_accumulate calls may run on two to eight internal streams of the context so that one starts while the
 * previous one drains; each is followed, on THIS stream, by the temporal pass that waits for it. */
int vrt_set_stream(vrt_ctx* ctx, void* hip_stream);
/* Renderer.fetch_image (pathtracer.py:1321-1323, 634-662): LDR rgba f32[H][W][4] */
int vrt_fetch_ldr(vrt_ctx* ctx, float* out);
/* The reference presents EVERY frame (scene.py:255-262: accumulate, fetch_image, copy_prev_matrices).  These forms of
 * vrt_fetch_hdr / vrt_fetch_ldr queue the tonemap and the copy behind the passes queued so far, on a stream of the
 * library's own, and return: the caller goes on queueing frames and collects the image with vrt_fetch_wait(slot),
 * slot = 0..3 chosen by the caller, one fetch per slot at a time.  `out` should be page-locked memory (vrt_host_alloc) so
 * that the copy runs beside the following launches; for a shard only its rows of `out` are written. */
int vrt_fetch_hdr_async(vrt_ctx* ctx, float* out, int slot);
int vrt_fetch_ldr_async(vrt_ctx* ctx, float* out, int slot);
/* the LDR image as rgba8[H][W][4] = u8(clamp(c, 0, 1) * 255 + 0.5): what a display or a PNG takes, a quarter of the bytes
 * (at 1080p the f32 image is 33 MB a frame, more than the host link carries at a frame per millisecond) */
int vrt_fetch_ldr8_async(vrt_ctx* ctx, uint8_t* out, int slot);
int vrt_fetch_wait(vrt_ctx* ctx, int slot);
/* page-locked host memory for the asynchronous fetches (hipHostMalloc / hipHostFree) */
int vrt_host_alloc(vrt_ctx* ctx, uint64_t bytes, void** out);
int vrt_host_free(vrt_ctx* ctx, void* ptr);
/* Multi-GPU hand-over without a copy (the RCCL gather of the row tiles): the last temporal pass of every
 * vrt_accumulate call also writes this context's HDR rows (f32[row_end-row_begin][W][3]) to device_ptrs[k % n], k = tiles
 * written so far (vrt_hdr_targets_written); n = 0 ends it.  The pass is queued on the context's stream by the call, so an
 * event the caller records on that stream afterwards orders the gather behind the tile; n must exceed the gathers the
 * caller keeps in flight. */
int vrt_set_hdr_targets(vrt_ctx* ctx, void* const* device_ptrs, int n);
int vrt_hdr_targets_written(vrt_ctx* ctx, uint64_t* count);
/* Multi-GPU partition by INTERLEAVED ROW STRIPES instead of contiguous row tiles ("prefer interleaved 8-row stripes
 * (row_tile % 8 == rank) if contiguous tiles miss the target"): a whole-frame context (row_begin = row_end = 0) produces, of every
 * stripe_rows * n_parts rows, the stripe_rows rows starting at part * stripe_rows -- every rank then carries the frame's average
 * cost by construction, without a balancing pass.  stripe_rows is a multiple of 8 (the pixel tile, pathtracer.py:74); each stripe
 * is rendered with two more rows either side (what the accumulation pass reads of its neighbours: (stripe_rows + 4) / stripe_rows
 * of the work -- 32 rows: +12.5 %).  Buffers stay frame-sized; vrt_fetch_hdr returns the frame with the other rows zero;
 * vrt_fetch_hdr_device / the tiles of vrt_set_hdr_targets hold the context's rows stripe after stripe.  Static camera, ReSTIR off.
 * Call before the first vrt_accumulate; stripe_rows = 0 turns it off.  No counterpart in the reference (one GPU). */
int vrt_set_row_stripes(vrt_ctx* ctx, int stripe_rows, int n_parts, int part);
int vrt_fetch_buffer(vrt_ctx* ctx, int which, void* out);
int vrt_sync(vrt_ctx* ctx);
int vrt_get_stats(vrt_ctx* ctx, vrt_stats* out);
int vrt_reset_stats(vrt_ctx* ctx);
const char* vrt_last_error(void);
/* The render kernels are persistent: their grid fills every CU (two workgroups of 79 KB LDS each, or one of 148 KB at
 * grid_res 256), so another kernel -- an RCCL collective of a multi-GPU run -- finds no workgroup slot until a launch
 * drains.  This leaves n_cus CUs' worth of slots out of the grid (0 = none, the single-GPU default).  No counterpart in the
 * reference (one GPU, pathtracer.py); voxel_rt2_amd/parallel.py calls it when the process group has more than one rank. */
int vrt_reserve_cus(vrt_ctx* ctx, int n_cus);
/* Select the kernel variants that count rays / DDA iterations / occupancy queries / hits
 * (vrt_stats.rays etc.); off by default -- the counters are what the reference's disabled
 * iteration heat-map (pathtracer.py:419-425) would have shown.  on = 1 counts the reference algorithm's work (every
 * camera ray walked, as the reference and the oracle do); on = 2 counts what the timed schedule does (the samples fused
 * into one launch share their camera rays, so a pixel's camera ray is walked -- and counted -- once). */
int vrt_set_instrumented(vrt_ctx* ctx, int on);
/* Occupancy queries OUTSIDE the grid.  The reference's walk can take one more step after its ray has left the grid
 * (hit_distance a rounding error short of `far`, raytracer.py:104) and then calls query_occupancy with a cell coordinate of -1
 * or grid_res; linearize_index (raytracer.py:17-38) does not check, so the bit it reads belongs to ANOTHER cell (x = 128 is
 * x = 0 of the next row; z = 128 at LOD 0 is the start of the LOD-1 region) and a set bit is reported as a hit on a voxel
 * outside the grid, which voxel_surface_color paints black (voxel_world.py:27-32, 46): black specks on the far faces of dense
 * grids.  on = 0 (default): such a query reads "empty" -- the walk's documented meaning.  on = 1: it reads the bit the
 * reference's index arithmetic addresses (bits before the array or behind its 2 * grid_res^3 read 0), so the frame equals
 * what the reference's source computes, specks included; every ray is walked (no culling) and the launches use the
 * instrumented kernel instantiations, which carry that code.  Takes effect at the next vrt_accumulate. */
int vrt_set_reference_indexing(vrt_ctx* ctx, int on);
/* Hash of the sources this library was built from (voxel_rt2_amd/build.py): ties measured counters
 * (profiles/traffic.json) to the build they were measured on. */
const char* vrt_build_id(void);
/* Evaluate one vrt_detmath.h operation on the device (numeric-contract test hook):
 * op 0 sin 1 cos 2 exp 3 log 4 pow 5 acos 6 atan2 7 min 8 max 9 f16 round trip 10 a/b 11 sqrt
 * 12 a*b+a (uncontracted) 13 float->int. */
int vrt_detmath_probe(int device, int op, int n, const float* a, const float* b, float* out);
/* Evaluate single functions of the sky / cloud precompute on the device (test hook; the rows the reference's own atmos.py was
 * executed on: tests/golden/reference/functions_sky.npz).  op: 0 rsi (atmos.py:9-15) | 1 get_ozone_density (:500-518) | 2 get_density
 * (:520-523) | 3 cloud_phase (:262-267) | 4 sample_cloud_density (:195-224) | 5 clouds_shadow_od (:231-260) | 6 get_ray_transmittance
 * (:475-498) | 7 clouds_scattering (:269-349) | 8 / 9 atmospheric_scattering at template depth 0 / 1 (:355-425).  `in` holds n rows of
 * in_stride floats: position, direction, then the function's other arguments (functions 7-9: sun direction, sun colour, cone cosine,
 * dither or step count, index of the row's random stream); `out` n rows of out_stride floats.  trans_lut (f16[256][128][3]) and
 * cloud_ambient (f32[3]), when not NULL, replace what vrt_prepare computed.  Needs sky_res > 0. */
int vrt_sky_probe(vrt_ctx* ctx, int op, int n, const float* in, int in_stride, float* out, int out_stride, const uint16_t* trans_lut,
                  const float* cloud_ambient);
/* Diagnostic builds only (library compiled with -DVRT_DIAG_REGIONS, see tools/diag_regions.py): copy out the
 * 32 x {wave entries, active lanes} counters of the instrumented regions of the render kernel and optionally
 * zero them.  The shipped library returns VRT_E_STATE -- it carries no region counters. */
int vrt_diag_regions(vrt_ctx* ctx, unsigned long long* out64, int reset);

#ifdef __cplusplus
}
#endif
#endif /* VRT_API_H */
