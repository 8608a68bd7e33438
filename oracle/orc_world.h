/* orc_world.h -- ORACLE (test infrastructure, not product code).
 *
 * Literal CPU restatement of /root/reference/renderer/raytracer.py (occupancy bit pyramid +
 * hierarchical DDA) and renderer/voxel_world.py (voxel store, surface colour).
 *
 * Deviations from the reference, all on undefined behaviour:
 *  - LOD bases: the reference formula (raytracer.py:32) addresses past the end of its own
 *    allocation for LOD >= 2 (it needs 2n - 2n/2^lods bits, n = res^3, and allocates 8n/7).  The
 *    oracle uses the reference's bases and row-major order inside a LOD (raytracer.py:32-37) over an
 *    array long enough to hold what they address: 2n bits.
 *  - a query outside the grid (the DDA steps to x = 128 or -1 when a ray leaves the volume and
 *    float rounding keeps hit_distance <= far for one more iteration) reads a neighbouring bit
 *    or out of bounds in the reference; here it is EMPTY by default.  With `reference_indexing`
 *    (orc_set_reference_indexing; the product's vrt_set_reference_indexing) such a query reads the bit
 *    the reference's own index arithmetic addresses -- x = res is x = 0 of the next row, z = res at
 *    LOD 0 is the start of the LOD-1 region, ... -- and memory before or behind the 2n bits reads 0:
 *    what the reference's source computes when executed over a zero-initialised array of that
 *    length (tests/golden/make_reference_vectors.py), black specks on the far faces of dense grids
 *    included.
 */
#ifndef ORC_WORLD_H
#define ORC_WORLD_H

#include <vector>
#include <cstring>
#include "orc_math.h"

namespace orc {

struct Stats { /* per-call counters for the algorithmic-bytes figure */
    uint64_t rays = 0, iters = 0, queries = 0, closest_hits = 0, sky_lookups = 0;
};

struct VoxelOctreeRaytracer {
    int voxel_grid_res = 128;
    int n_lods = 7; /* raytracer.py:9 */
    uint32_t lod_base[16];
    std::vector<uint32_t> occupancy;
    bool reference_indexing = false; /* queries outside the grid: false = empty, true = the reference's index arithmetic */

    void init(int res) {
        voxel_grid_res = res;
        n_lods = 0;
        while ((1 << n_lods) < res) n_lods++;
        const uint32_t n = (uint32_t)res * (uint32_t)res * (uint32_t)res;
        for (int i = 0; i < n_lods; i++) lod_base[i] = i == 0 ? 0u : (n << 1) - ((n << 1) >> i); /* raytracer.py:20-32 */
        /* raytracer.py:10-15 allocates sum(r^3) / 32 + 1 words: LODs >= 2 lie behind that (header).  2n bits hold them all */
        occupancy.assign((size_t)(n << 1) / 32, 0u);
    }
    /* raytracer.py:17-38 (signed, like the reference's i32 arithmetic: a coordinate may be -1) */
    int32_t linearize_index(I3 ipos, int lod) const {
        int r = voxel_grid_res >> lod;
        return (int32_t)lod_base[lod] + (ipos.z * (r * r) + ipos.y * r + ipos.x);
    }
    /* raytracer.py:40-44 */
    bool query_occupancy(I3 ipos, int lod, Stats* st = nullptr) const {
        if (st) st->queries++;
        int r = voxel_grid_res >> lod;
        if (!reference_indexing && (ipos.x < 0 || ipos.y < 0 || ipos.z < 0 || ipos.x >= r || ipos.y >= r || ipos.z >= r)) return false;
        int32_t idx = linearize_index(ipos, lod);
        if (idx < 0 || (size_t)(idx >> 5) >= occupancy.size()) return false; /* memory nobody wrote */
        return (occupancy[idx >> 5] & (1u << (idx & 31))) != 0;
    }
    /* raytracer.py:46-70.  voxels: int8 [res][res][res] indexed [x][y][z] (offset already applied) */
    void update_lods(const int8_t* voxels) {
        std::fill(occupancy.begin(), occupancy.end(), 0u);
        int n = voxel_grid_res;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++)
                for (int k = 0; k < n; k++)
                    if (voxels[((size_t)i * n + j) * n + k] > 0) {
                        int32_t idx = linearize_index(I3{i, j, k}, 0);
                        occupancy[idx >> 5] |= 1u << (idx & 31);
                    }
        for (int lod = 1; lod < n_lods; lod++) {
            int s = n >> lod;
            for (int i = 0; i < s; i++)
                for (int j = 0; j < s; j++)
                    for (int k = 0; k < s; k++) {
                        bool empty = true;
                        for (int a = 0; a < 2; a++)
                            for (int b = 0; b < 2; b++)
                                for (int c = 0; c < 2; c++)
                                    empty = empty && !query_occupancy(I3{i * 2 + a, j * 2 + b, k * 2 + c}, lod - 1);
                        if (!empty) {
                            int32_t idx = linearize_index(I3{i, j, k}, lod);
                            occupancy[idx >> 5] |= 1u << (idx & 31);
                        }
                    }
        }
    }

    struct Hit {
        float distance;
        I3 ipos;
        V3 normal;
        int iters;
    };
    /* raytracer.py:72-155 */
    Hit raytrace(V3 origin, V3 direction, float ray_min_t, float ray_max_t, Stats* st = nullptr) const {
        float hit_distance = INF;
        I3 ipos_lod0 = I3{-1, -1, -1};
        V3 hit_normal = v3(0.0f);
        int iters = 0;
        const float res = (float)voxel_grid_res;

        float bbox_near, bbox_far;
        bool bbox_intersect = ray_aabb_intersection(v3(0.0f), v3(res), origin, direction, &bbox_near, &bbox_far);

        if (bbox_intersect && ray_min_t < bbox_far && ray_max_t > bbox_near) {
            hit_distance = dm_max(bbox_near, ray_min_t);
            V3 initial_p = origin + direction * (hit_distance + EPS);
            V3 c = vclamp(vfloor(initial_p), 0.0f, res - 1.0f);
            ipos_lod0 = I3{(int)c.x, (int)c.y, (int)c.z};
            V3 inv_dir = v3(1.0f / dm_abs(direction.x), 1.0f / dm_abs(direction.y), 1.0f / dm_abs(direction.z));
            int current_lod = 0;
            float far = dm_min(ray_max_t, bbox_far) - EPS;

            V3 initial_dist = vabs(initial_p - res * 0.5f);
            float max_dist = max3(initial_dist.x, initial_dist.y, initial_dist.z);
            hit_normal = v3(max_dist == initial_dist.x ? 1.0f : 0.0f, max_dist == initial_dist.y ? 1.0f : 0.0f,
                            max_dist == initial_dist.z ? 1.0f : 0.0f);

            while (iters < 512) {
                if (hit_distance > far) {
                    hit_distance = INF;
                    break;
                }
                I3 ipos = I3{0, 0, 0};
                bool sample = false;
                while (true) {
                    ipos = I3{ipos_lod0.x >> current_lod, ipos_lod0.y >> current_lod, ipos_lod0.z >> current_lod};
                    sample = query_occupancy(ipos, current_lod, st);
                    if (sample && current_lod > 0) current_lod -= 1;
                    else break;
                }
                if (sample) break;

                float cell_size = (float)(1 << current_lod);
                V3 cell_base = v3((float)ipos.x, (float)ipos.y, (float)ipos.z) * cell_size;
                V3 voxel_pos = origin + direction * hit_distance;
                V3 frac_pos = voxel_pos - cell_base;
                V3 dist = frac_pos;
                for (int i = 0; i < 3; i++)
                    if (direction[i] > 0.0f) dist[i] = cell_size - frac_pos[i];
                V3 t = dist * inv_dir;
                float min_t = min3(t.x, t.y, t.z);
                V3 edge_frac_pos = vclamp(vfloor(frac_pos + min_t * direction), 0.0f, cell_size - 1.0f);
                hit_distance += min_t;
                hit_normal = v3(t.x == min_t ? 1.0f : 0.0f, t.y == min_t ? 1.0f : 0.0f, t.z == min_t ? 1.0f : 0.0f) *
                             v3(sign(direction.x), sign(direction.y), sign(direction.z));
                V3 nxt = cell_base + edge_frac_pos + hit_normal;
                ipos_lod0 = I3{(int)nxt.x, (int)nxt.y, (int)nxt.z};
                current_lod = (n_lods - 1 < current_lod + 1) ? n_lods - 1 : current_lod + 1;
                iters += 1;
            }
        }
        if (dot(direction, hit_normal) > 0.0f) hit_normal = -hit_normal;
        if (st) { st->rays++; st->iters += (uint64_t)iters; }
        return Hit{hit_distance, ipos_lod0, hit_normal, iters};
    }
};

/* voxel_world.py:6-25: AoS {u8x3 colour, i8 material}, dense res^3, index offset -res/2.
 * The rgba8 3-D texture of :20-21, 69-87 holds the same bytes (unorm8 store of b/255 is b);
 * a negative material byte clamps to 0 in the unorm store. */
struct VoxelWorld {
    int voxel_grid_res = 128;
    float voxel_size = 1.0f / 64.0f, voxel_inv_size = 64.0f;
    int voxel_grid_offset = -64;
    float voxel_edges = 0.06f;
    std::vector<int8_t> voxel_material;  /* [x][y][z] */
    std::vector<uint8_t> voxel_color;    /* [x][y][z][3] */
    std::vector<uint32_t> texture;       /* rgba8 texel per voxel, same [x][y][z] order */

    void init(float dx, int res, float edges) {
        voxel_size = dx;
        voxel_inv_size = 1.0f / dx;
        voxel_grid_res = res;
        voxel_grid_offset = -(res / 2);
        voxel_edges = edges;
        voxel_material.assign((size_t)res * res * res, 0);
        voxel_color.assign((size_t)res * res * res * 3, 0);
        texture.assign((size_t)res * res * res, 0u);
    }
    /* voxel_world.py:69-87 */
    void make_texture() {
        size_t n = (size_t)voxel_grid_res * voxel_grid_res * voxel_grid_res;
        for (size_t i = 0; i < n; i++) {
            int m = voxel_material[i];
            uint32_t a = (m < 0) ? 0u : (uint32_t)m;
            texture[i] = (uint32_t)voxel_color[3 * i] | ((uint32_t)voxel_color[3 * i + 1] << 8) |
                         ((uint32_t)voxel_color[3 * i + 2] << 16) | (a << 24);
        }
    }
    /* voxel_world.py:27-32 */
    bool inside_grid(I3 ipos) const {
        int mn = ipos.x < ipos.y ? (ipos.x < ipos.z ? ipos.x : ipos.z) : (ipos.y < ipos.z ? ipos.y : ipos.z);
        int mx = ipos.x > ipos.y ? (ipos.x > ipos.z ? ipos.x : ipos.z) : (ipos.y > ipos.z ? ipos.y : ipos.z);
        return mn >= -(voxel_grid_res / 2) && mx < voxel_grid_res / 2;
    }
    /* voxel_world.py:34-56 */
    void voxel_surface_color(I3 voxel_index, V3 voxel_uv, V3* color, int* is_light, int* material) const {
        float boundary = voxel_edges;
        int count = 0;
        for (int i = 0; i < 3; i++)
            if (voxel_uv[i] < boundary || voxel_uv[i] > 1.0f - boundary) count += 1;
        float f = 0.0f;
        if (count >= 2) f = 1.0f;
        V3 voxel_color_v = v3(0.0f);
        int voxel_material_v = 0;
        int light = 0;
        if (inside_grid(voxel_index)) {
            int n = voxel_grid_res;
            size_t idx = ((size_t)(voxel_index.x - voxel_grid_offset) * n + (voxel_index.y - voxel_grid_offset)) * n +
                         (voxel_index.z - voxel_grid_offset);
            uint32_t texel = texture[idx];
            voxel_color_v = v3((float)(texel & 255u) / 255.0f, (float)((texel >> 8) & 255u) / 255.0f,
                               (float)((texel >> 16) & 255u) / 255.0f);
            float a = (float)(texel >> 24) / 255.0f;
            voxel_material_v = (int)(a * 255.0f);
            if (voxel_material_v == 2) light = 1;
        }
        *color = voxel_color_v * (1.0f - 0.9f * f);
        *is_light = light;
        *material = voxel_material_v;
    }
};

} /* namespace orc */
#endif
