// vrt_types.h -- device-side vector types and the kernel parameter block of libvrt_hip.so.
//
// Everything under csrc/ is written for gfx950 (MI355X, wave64).  The headers also compile as
// plain C++ (VRT_DEV becomes `inline`) so that the CPU sanitizer / emulation build used by the
// test-suite can step the same per-path functions without a GPU; that build is test tooling,
// the shipped library has no CPU path.
#ifndef VRT_TYPES_H
#define VRT_TYPES_H

#include <stdint.h>
#include "../../include/vrt_detmath.h"

#if defined(__HIPCC__)
#define VRT_DEV __host__ __device__ __forceinline__
#define VRT_DEV_NOINLINE __device__ __noinline__
#else
#define VRT_DEV inline
#define VRT_DEV_NOINLINE inline
#endif

namespace vrt {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct i3 { int x, y, z; };

VRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
VRT_DEV f3 mk3(float s) { return mk3(s, s, s); }
VRT_DEV f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
VRT_DEV f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

VRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VRT_DEV f3 operator/(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
VRT_DEV f3 operator+(f3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
VRT_DEV f3 operator-(f3 a, float s) { return mk3(a.x - s, a.y - s, a.z - s); }
VRT_DEV f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VRT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
VRT_DEV f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
VRT_DEV f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
VRT_DEV float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

// sums are left to right; nothing here may be contracted into an fma (-ffp-contract=off)
VRT_DEV float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VRT_DEV f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
VRT_DEV float len3(f3 a) { return dm_sqrt(dot3(a, a)); }
VRT_DEV f3 norm3(f3 a) { float inv = 1.0f / len3(a); return inv * a; }
VRT_DEV f3 abs3(f3 a) { return mk3(dm_abs(a.x), dm_abs(a.y), dm_abs(a.z)); }
VRT_DEV f3 floor3(f3 a) { return mk3(dm_floor(a.x), dm_floor(a.y), dm_floor(a.z)); }
VRT_DEV f3 clamp3(f3 a, float lo, float hi) { return mk3(dm_clamp(a.x, lo, hi), dm_clamp(a.y, lo, hi), dm_clamp(a.z, lo, hi)); }
VRT_DEV f3 max3s(f3 a, float s) { return mk3(dm_max(a.x, s), dm_max(a.y, s), dm_max(a.z, s)); }
VRT_DEV f3 sat3(f3 a) { return mk3(dm_saturate(a.x), dm_saturate(a.y), dm_saturate(a.z)); }
VRT_DEV float sgn(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
VRT_DEV float lerp1(float x, float y, float a) { return x * (1.0f - a) + y * a; }
VRT_DEV f3 lerp3(f3 x, f3 y, float a) { return x * (1.0f - a) + y * a; }
VRT_DEV float frac1(float x) { return x - dm_floor(x); }
VRT_DEV float sq(float x) { return x * x; }
VRT_DEV bool near_zero3(f3 v) { return dot3(v, v) < 1e-7f; }
VRT_DEV float lum(f3 c) { return dot3(mk3(0.2125f, 0.7154f, 0.0721f), c); }
VRT_DEV f3 firefly(f3 v) { return clamp3(v, 0.0f, 300.0f); }

// Per-pixel results the render kernel writes once and never reads back: streamed (non-temporal) on the device so
// that they do not push the pooled kernel's scratch lines and the brick words out of L2.
template <class T>
VRT_DEV void stream_store(T* p, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
VRT_DEV void stream_store3(f3* p, f3 v) { stream_store(&p->x, v.x); stream_store(&p->y, v.y); stream_store(&p->z, v.z); }

struct mat4 { float m[16]; };  // row-major, m[row*4+col]
VRT_DEV f4 mul4(const mat4& M, f4 v) {
    return mk4(M.m[0] * v.x + M.m[1] * v.y + M.m[2] * v.z + M.m[3] * v.w,
               M.m[4] * v.x + M.m[5] * v.y + M.m[6] * v.z + M.m[7] * v.w,
               M.m[8] * v.x + M.m[9] * v.y + M.m[10] * v.z + M.m[11] * v.w,
               M.m[12] * v.x + M.m[13] * v.y + M.m[14] * v.z + M.m[15] * v.w);
}

#define VRT_EPS 1e-6f

// Diagnostic build only (-DVRT_DIAG_REGIONS): per code region, how many times a WAVE entered it and with how many
// active lanes -- where the issue slots of the divergent render kernel go.  Never defined in the shipped library.
#if defined(VRT_DIAG_REGIONS) && defined(__HIPCC__)
static __device__ unsigned long long g_vrt_region[64];
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VRT_DIAG_CLOCKS_ONLY)  // CLOCKS_ONLY: just the pooled kernel's stage clocks
#define VRT_REGION_COUNT(id)                                                                                        \
    do {                                                                                                            \
        unsigned long long m_ = __ballot(1);                                                                        \
        if ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == (int)__ffsll((long long)m_) - 1) { \
            atomicAdd(&g_vrt_region[2 * (id)], 1ULL);                                                               \
            atomicAdd(&g_vrt_region[2 * (id) + 1], (unsigned long long)__popcll(m_));                               \
        }                                                                                                           \
    } while (0)
#else
#define VRT_REGION_COUNT(id) ((void)0)
#endif
#else
#define VRT_REGION_COUNT(id) ((void)0)
#endif
// -DVRT_DIAG_GRIS (with -DVRT_DIAG_REGIONS): the same 32 counters count the regions of the spatial-reuse kernels instead
// (VRT_GREGION, vrt_restir.h; tools/diag_regions.py gris)
#if defined(VRT_DIAG_GRIS)
#define VRT_REGION(id) ((void)0)
#define VRT_GREGION(id) VRT_REGION_COUNT(id)
#else
#define VRT_REGION(id) VRT_REGION_COUNT(id)
#define VRT_GREGION(id) ((void)0)
#endif

// The voxel grid is G^3 cells, G = 128 (the reference's voxel_grid_res, pathtracer.py:83) or 256 (BASELINE config 5:
// VoxelWorld / VoxelOctreeRaytracer are parametric in it -- n_lods = log2 G, raytracer.py:9; offset -G/2,
// voxel_world.py:14).  G is a template parameter of everything that walks the grid, so the 128 kernels keep their
// literal constants.  The world box stays [-1, 1]^3: voxel size dx = 2 / G (scene.py:11's 1/64 at G = 128).
template <int G>
struct GridDim {
    static_assert(G == 128 || G == 256, "grid_res is 128 or 256");
    static constexpr int lods = (G == 256) ? 8 : 7;      // raytracer.py:9
    static constexpr int max_lod = lods - 1;             // raytracer.py:147
    static constexpr int s0 = (G == 256) ? 6 : 5;        // log2 of the words per axis of brick level 0, 1, 2
    static constexpr int s1 = s0 - 2;
    static constexpr int s2 = s0 - 4;
    static constexpr int n0 = 1 << s0, n1 = 1 << s1, n2 = 1 << s2;
    static constexpr float half = (float)(G / 2);        // voxel_inv_size (= 1 / dx) and -voxel_grid_offset
    static constexpr float voxel_size = 2.0f / (float)G; // dx
};

// One Disney material row (14 f32, the order of the host table).
struct Material {
    f3 base;
    float subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint, clearcoat,
        clearcoat_gloss, ior_minus_one;
};

// Traversal counters of the instrumented build (one slot per counter, atomically summed).
struct Counters { unsigned long long rays, iters, queries, closest_hits, sky_lookups; };

// Occupancy pyramid as levels of 4x4x4 bit bricks (64-bit words), n_k = G / 4^(k+1) words per axis:
//   l0[n0^3]: bit = voxel solid          -> LOD 0 (bit), LOD 1 (2x2x2 sub-mask), LOD 2 (word != 0)
//   l1[n1^3]: bit = l0 word non-zero     -> LOD 2 (bit), LOD 3 (sub-mask),       LOD 4 (word != 0)
//   l2[n2^3]: bit = l1 word non-zero     -> LOD 4 (bit), LOD 5 (sub-mask),       LOD 6 (word != 0)
//   l3[1]   : bit = l2 word non-zero     -> LOD 6 (bit), LOD 7 (sub-mask)        (G = 256 only: 32^3 / 8^3 / 2^3 words
//                                                                                 at 128, 64^3 / 16^3 / 4^3 / 1 at 256)
// Word index inside a level: (bz * n + by) * n + bx; bit inside a word: (z&3)*16 + (y&3)*4 + (x&3).
// l0c (G = 128) is l0 without its empty words: the non-zero fine words in (l1 brick, bit) order, i.e. the word of the l0
// brick behind set bit b of l1 word i sits at l0c[l0c_base[i] + popcount(l1[i] & ((1 << b) - 1))].  A sparse scene's
// whole fine level is then a few KB and the pooled render kernel keeps (the head of) it in LDS.
struct Pyramid {
    const unsigned long long* l0;
    const unsigned long long* l1;
    const unsigned long long* l2;
    const unsigned long long* l3;   // [1], G = 256
    const unsigned long long* l0c;
    const uint32_t* l0c_base;   // [512]
    const uint32_t* l0c_count;  // [1]: non-empty l0 words
    int ref_oob;                // queries outside the grid follow the reference's index arithmetic (vrt_set_reference_indexing; ref_bit, vrt_trace.h)
};

struct SkyTables {
    const float* scattering;     // [res][res][3]
    const float* transmittance;  // [res][res][3]
    int res;
    float fres;
};

// Everything the per-pixel kernels read that is not a per-pixel buffer.  Passed by value.
struct FrameParams {
    mat4 view, proj, view_inv, proj_inv;
    f3 camera_pos;
    f2 taa_jitter;
    f2 inv_res;
    int W, H;
    int row0, row1;             // rows rendered by this launch (shard + halo)
    // Interleaved row stripes (vrt_set_row_stripes; 0 = off): of every stripe_period rows this launch renders the stripe_rows rows
    // from stripe_first on, plus two rows either side (what the temporal pass reads of its neighbours).  stripe_tile_rows: 8-row
    // tile rows the launch enumerates -- per stripe its own stripe_rows / 8 and one above and below for the halo rows.
    int stripe_rows, stripe_period, stripe_first, stripe_tile_rows;
    int camera_is_moving;
    float render_scale, max_accum_frames;
    f3 light_dir, light_color;
    float light_cos_max, light_weight;
    float floor_height;
    f3 floor_color;
    int floor_material;
    f3 background;
    int use_sky;
    float voxel_edges, exposure;
    int max_depth;
    uint32_t seed, frame;
};

// First image row of the launch's t-th 8-row tile row, and whether row v is one the launch renders (always, without stripes).
VRT_DEV int launch_tile_row(const FrameParams& fp, int t) {
    if (fp.stripe_period == 0) return fp.row0 + t * 8;
    const int per = fp.stripe_rows / 8 + 2, k = t / per, j = t - k * per;
    return k * fp.stripe_period + fp.stripe_first - 8 + j * 8;
}
VRT_DEV int launch_tile_rows(const FrameParams& fp) { return fp.stripe_period == 0 ? (fp.row1 - fp.row0 + 7) >> 3 : fp.stripe_tile_rows; }
VRT_DEV bool launch_renders_row(const FrameParams& fp, int v) {
    if (fp.stripe_period == 0) return true;
    if (v < fp.row0) return false;
    const int m = (v - fp.stripe_first + 2 + fp.stripe_period) % fp.stripe_period;   // (v >= 0 > stripe_first - 2 - period)
    return m < fp.stripe_rows + 4;
}

}  // namespace vrt
#endif
