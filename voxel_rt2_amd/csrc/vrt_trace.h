// vrt_trace.h -- hierarchical DDA over the 128^3 voxel grid and the closest-hit query.
//
// Replaces VoxelOctreeRaytracer.raytrace / query_occupancy (reference renderer/raytracer.py:40-44,
// 72-155), VoxelWorld.voxel_surface_color (voxel_world.py:34-56) and Renderer._trace_sdf /
// _trace_voxel / next_hit (pathtracer.py:173-244).
//
// Same walk, different machine mapping.  The reference keeps one bit per cell per LOD in a flat
// i32 array and issues one dependent 4-byte load per LOD it descends (up to 7 per DDA step).
// Here occupancy lives in 4x4x4 bit bricks (Pyramid, vrt_types.h): ONE 64-bit word answers three
// LODs at once, the two coarse brick levels (4 KiB + 64 B) sit in LDS, and the fine brick word of
// the cell a ray is in stays in registers across DDA steps, so a step that remains inside its
// brick issues no memory instruction at all.  The sequence of (cell, LOD) states a ray visits --
// and with it every float the reference computes -- is unchanged: descend() returns the state the
// reference's inner `while True` loop (raytracer.py:110-118) ends in.
#ifndef VRT_TRACE_H
#define VRT_TRACE_H

#include "vrt_types.h"

namespace vrt {

struct SceneData {
    Pyramid pyr;
    const uint32_t* grid;   // rgba8 texel per voxel, [x][y][z] (voxel_world.py:76-87 holds the same bytes)
    const float* mats;      // [128][14]
    SkyTables sky;
    Counters* counters;     // instrumented build only
};

struct BrickCache { int key; unsigned long long word; };

VRT_DEV int brick_bit(int cx, int cy, int cz) { return ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3); }
VRT_DEV unsigned long long brick_sub(int cx, int cy, int cz) {  // the 2x2x2 block containing (cx,cy,cz)
    return 0x0000000000330033ULL << (((cz & 2) << 4) | ((cy & 2) << 2) | (cx & 2));
}

// Level the reference's descent ends at when it starts at `lod` on LOD-0 cell (x,y,z), and whether
// it ended on a solid voxel.  `nq` receives the number of query_occupancy calls the reference
// would have made (for the algorithmic-bytes counters).
template <class PyrT>
VRT_DEV int descend(const PyrT& P, int x, int y, int z, int lod, bool& solid, BrickCache& bc, int& nq) {
    solid = false;
    nq = 1;
    if ((x | y | z) & ~(VRT_GRID - 1)) return lod;  // outside the grid: empty (UB in the reference)
    const int start = lod;
    if (lod >= 4) {
        unsigned long long w2 = P.load_l2((((z >> 6) << 1) | (y >> 6)) << 1 | (x >> 6));
        if (lod == 6) { if (w2 == 0ULL) return 6; lod = 5; }
        if (lod == 5) { if ((w2 & brick_sub(x >> 4, y >> 4, z >> 4)) == 0ULL) { nq = start - 5 + 1; return 5; } lod = 4; }
        if (((w2 >> brick_bit(x >> 4, y >> 4, z >> 4)) & 1ULL) == 0ULL) { nq = start - 4 + 1; return 4; }
        lod = 3;
    }
    if (lod >= 2) {
        unsigned long long w1 = P.load_l1((((z >> 4) << 3) | (y >> 4)) << 3 | (x >> 4));
        if (lod == 3) { if ((w1 & brick_sub(x >> 2, y >> 2, z >> 2)) == 0ULL) { nq = start - 3 + 1; return 3; } lod = 2; }
        if (((w1 >> brick_bit(x >> 2, y >> 2, z >> 2)) & 1ULL) == 0ULL) { nq = start - 2 + 1; return 2; }
        lod = 1;
    }
    const int key = (((z >> 2) << 5) | (y >> 2)) << 5 | (x >> 2);
    if (key != bc.key) { bc.key = key; bc.word = P.load_l0(key); }
    const unsigned long long w0 = bc.word;
    if (lod == 1) { if ((w0 & brick_sub(x, y, z)) == 0ULL) { nq = start - 1 + 1; return 1; } }
    nq = start + 1;
    solid = ((w0 >> brick_bit(x, y, z)) & 1ULL) != 0ULL;
    return 0;
}

struct GlobalPyramid {  // all three brick levels read from global memory
    Pyramid p;
    VRT_DEV unsigned long long load_l0(int i) const { return p.l0[i]; }
    VRT_DEV unsigned long long load_l1(int i) const { return p.l1[i]; }
    VRT_DEV unsigned long long load_l2(int i) const { return p.l2[i]; }
};

struct TraceOut { float dist; int ix, iy, iz; f3 normal; int iters; };

// raytracer.py:72-155 with ray_min_t = eps, ray_max_t = inf (the only call site, pathtracer.py:201-202).
template <class PyrT>
VRT_DEV void raytrace(const PyrT& P, f3 o, f3 d, TraceOut& r, int& queries) {
    float hit_distance = DM_INF;
    int ix = -1, iy = -1, iz = -1;
    f3 hn = mk3(0.0f);
    int iters = 0;
    queries = 0;
    const float res = (float)VRT_GRID;

    // math_utils.py:103-123 against the box [0,128]^3
    float near_t = -DM_INF, far_t = DM_INF;
#define VRT_SLAB(oc, dc)                                                     \
    if (dc != 0.0f) {                                                        \
        float i1 = (0.0f - oc) / dc, i2 = (res - oc) / dc;                   \
        far_t = dm_min(dm_max(i1, i2), far_t);                               \
        near_t = dm_max(dm_min(i1, i2), near_t);                             \
    }
    VRT_SLAB(o.x, d.x) VRT_SLAB(o.y, d.y) VRT_SLAB(o.z, d.z)
#undef VRT_SLAB

    if (near_t <= far_t && VRT_EPS < far_t && DM_INF > near_t) {
        hit_distance = dm_max(near_t, VRT_EPS);
        f3 p0 = o + d * (hit_distance + VRT_EPS);
        f3 c = clamp3(floor3(p0), 0.0f, res - 1.0f);
        ix = (int)c.x; iy = (int)c.y; iz = (int)c.z;
        const f3 inv_dir = mk3(1.0f / dm_abs(d.x), 1.0f / dm_abs(d.y), 1.0f / dm_abs(d.z));
        const f3 sd = mk3(sgn(d.x), sgn(d.y), sgn(d.z));
        int lod = 0;
        const float far = dm_min(DM_INF, far_t) - VRT_EPS;

        f3 id = abs3(p0 - res * 0.5f);
        float md = dm_max(dm_max(id.x, id.y), id.z);
        hn = mk3(md == id.x ? 1.0f : 0.0f, md == id.y ? 1.0f : 0.0f, md == id.z ? 1.0f : 0.0f);

        BrickCache bc;
        bc.key = -1;
        bc.word = 0ULL;
        while (iters < 512) {
            if (hit_distance > far) { hit_distance = DM_INF; break; }
            bool solid;
            int nq;
            VRT_REGION(1);
            lod = descend(P, ix, iy, iz, lod, solid, bc, nq);
            queries += nq;
            if (solid) break;

            const float cell_size = (float)(1 << lod);
            const f3 cell_base = mk3((float)(ix >> lod), (float)(iy >> lod), (float)(iz >> lod)) * cell_size;
            const f3 fp = (o + d * hit_distance) - cell_base;
            f3 dist;
            dist.x = (d.x > 0.0f) ? cell_size - fp.x : fp.x;
            dist.y = (d.y > 0.0f) ? cell_size - fp.y : fp.y;
            dist.z = (d.z > 0.0f) ? cell_size - fp.z : fp.z;
            const f3 t = dist * inv_dir;
            const float min_t = dm_min(dm_min(t.x, t.y), t.z);
            const f3 edge = clamp3(floor3(fp + min_t * d), 0.0f, cell_size - 1.0f);
            hit_distance += min_t;
            hn = mk3(t.x == min_t ? 1.0f : 0.0f, t.y == min_t ? 1.0f : 0.0f, t.z == min_t ? 1.0f : 0.0f) * sd;
            const f3 nxt = cell_base + edge + hn;
            ix = (int)nxt.x; iy = (int)nxt.y; iz = (int)nxt.z;
            lod = (lod + 1 > 6) ? 6 : lod + 1;
            iters += 1;
        }
    }
    if (dot3(d, hn) > 0.0f) hn = -hn;
    r.dist = hit_distance;
    r.ix = ix; r.iy = iy; r.iz = iz;
    r.normal = hn;
    r.iters = iters;
}

struct Hit { float closest; f3 normal; f3 albedo; int hit_light; int mat_id; };

struct TraceStats { unsigned rays, iters, queries, closest_hits, sky_lookups; };
VRT_DEV void stats_zero(TraceStats& s) { s.rays = s.iters = s.queries = s.closest_hits = s.sky_lookups = 0u; }

// pathtracer.py:218-244 (floor plane 173-190, voxel grid 192-216).  SHADOW: no surface lookup.
template <bool SHADOW, class PyrT>
VRT_DEV void next_hit(const FrameParams& fp, const SceneData& sc, const PyrT& P, f3 pos, f3 d, Hit& h, TraceStats& ts) {
    h.closest = DM_INF;
    h.normal = mk3(0.0f);
    h.albedo = mk3(1.0f);
    h.hit_light = 0;
    h.mat_id = 0;
    // infinite floor y = floor_height, accepted inside the radius-10 "disc" of pathtracer.py:183:
    // the scalar dot(hit, up) = hit.y is subtracted from all three components
    {
        float t = (fp.floor_height - pos.y) / d.y;
        if (t > VRT_EPS && t < h.closest) {
            f3 hp = pos + d * t;
            float s = hp.x * 0.0f + hp.y * 1.0f + hp.z * 0.0f;
            if (len3(hp - s) < 10.0f) {
                h.closest = t;
                h.normal = mk3(0.0f, 1.0f, 0.0f);
                if (dot3(h.normal, d) > 0.0f) h.normal = -h.normal;
                h.albedo = fp.floor_color;
                h.hit_light = (fp.floor_material == 2) ? 1 : 0;
                h.mat_id = fp.floor_material;
            }
        }
    }
    const float voxel_size = 1.0f / 64.0f, voxel_inv_size = 64.0f;
    f3 eye = voxel_inv_size * pos - (-64.0f);
    TraceOut tr;
    int nq;
    raytrace(P, eye, d, tr, nq);
    ts.rays += 1u; ts.iters += (unsigned)tr.iters; ts.queries += (unsigned)nq;
    VRT_REGION(SHADOW ? 3 : 9);  // ray set-up + result (one entry per ray)
    if (tr.dist * voxel_size < h.closest) {
        h.closest = tr.dist * voxel_size;
        if (!SHADOW) {
            ts.closest_hits += 1u;
            f3 uv = clamp3(eye + tr.dist * d - mk3((float)tr.ix, (float)tr.iy, (float)tr.iz), 0.0f, 1.0f);
            int cnt = 0;
            const float b = fp.voxel_edges;
            if (uv.x < b || uv.x > 1.0f - b) cnt++;
            if (uv.y < b || uv.y > 1.0f - b) cnt++;
            if (uv.z < b || uv.z > 1.0f - b) cnt++;
            const float f = (cnt >= 2) ? 1.0f : 0.0f;
            f3 col = mk3(0.0f);
            int m = 0, light = 0;
            if (((tr.ix | tr.iy | tr.iz) & ~(VRT_GRID - 1)) == 0) {
                uint32_t texel = sc.grid[((tr.ix << 7) | tr.iy) << 7 | tr.iz];
                col = mk3((float)(texel & 255u) / 255.0f, (float)((texel >> 8) & 255u) / 255.0f,
                          (float)((texel >> 16) & 255u) / 255.0f);
                float a = (float)(texel >> 24) / 255.0f;
                m = (int)(a * 255.0f);
                light = (m == 2) ? 1 : 0;
            }
            h.albedo = col * (1.0f - 0.9f * f);
            h.hit_light = light;
            h.mat_id = m;
            h.normal = tr.normal;
        }
    }
}

}  // namespace vrt
#endif
