// vrt_trace.h -- hierarchical DDA over the G^3 voxel grid (G = 128 or 256, vrt_types.h) and the closest-hit query.
//
// Replaces VoxelOctreeRaytracer.raytrace / query_occupancy (reference renderer/raytracer.py:40-44,
// 72-155), VoxelWorld.voxel_surface_color (voxel_world.py:34-56) and Renderer._trace_sdf /
// _trace_voxel / next_hit (pathtracer.py:173-244).
//
// Same walk, different machine mapping.  The reference keeps one bit per cell per LOD in a flat
// i32 array and issues one dependent 4-byte load per LOD it descends (up to 7 per DDA step).
// Here occupancy lives in 4x4x4 bit bricks (Pyramid, vrt_types.h): ONE 64-bit word answers three
// LODs at once, the coarse brick levels (4 KiB + 64 B at 128^3, 32 KiB + 512 B + one word at 256^3) sit in LDS, and the fine brick word of
// the cell a ray is in stays in registers across DDA steps, so a step that remains inside its
// brick issues no memory instruction at all.  The sequence of (cell, LOD) states a ray visits --
// and with it every float the reference computes -- is unchanged: descend() returns the state the
// reference's inner `while True` loop (raytracer.py:110-118) ends in.
#ifndef VRT_TRACE_H
#define VRT_TRACE_H

#include <type_traits>
#include "vrt_types.h"

namespace vrt {

struct SceneData {
    Pyramid pyr;
    const uint32_t* grid;   // rgba8 texel per voxel (voxel_world.py:76-87 holds the same bytes), order: texel_index()
    const float* mats;      // [128][14]
    SkyTables sky;
    Counters* counters;     // instrumented build only
    const float* cull;      // [8] the solid voxels' bounding box grown by VRT_CULL_MARGIN, voxel units: lo xyz, hi xyz (cull_ray);
                            // [6] != 0: the box leaves part of the grid out, i.e. there are rays to cull (a dense scene: 0)
};

struct BrickCache { int key; unsigned long long word; };

VRT_DEV int brick_bit(int cx, int cy, int cz) { return ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3); }
VRT_DEV unsigned long long brick_sub(int cx, int cy, int cz) {  // the 2x2x2 block containing (cx,cy,cz)
    return 0x0000000000330033ULL << (((cz & 2) << 4) | ((cy & 2) << 2) | (cx & 2));
}

// Where the texel of voxel (x,y,z) sits.  128^3 (8 MiB, cache resident): [x][y][z], the reference's storage order.
// 256^3 (64 MiB: beyond the L2s): brick-tiled -- the 64 texels of a 4x4x4 brick are contiguous (two 128-byte lines), bricks
// in l0 word order -- so that the hits of neighbouring pixels, which land on neighbouring voxels of a surface, share lines.
template <int G>
VRT_DEV int texel_index(int x, int y, int z) {
    if (G == 128) return ((x << 7) | y) << 7 | z;   // (brick-tiled at 128^3 as well: measured no difference, the 8 MiB stay cached)
    return (((((z >> 2) << 6) | (y >> 2)) << 6 | (x >> 2)) << 6) | brick_bit(x, y, z);
}

// ---- queries outside the grid, read the reference's way (vrt_set_reference_indexing) --------------------------------------
// The reference's walk can take one more step after its ray has left the grid (hit_distance a rounding error short of `far`,
// raytracer.py:104) and then queries a cell with a coordinate of -1 or G.  linearize_index (raytracer.py:17-38) does not
// check: bit index = base(lod) + z r^2 + y r + x with r = G >> lod and base(lod) = 2n - (2n >> lod), n = G^3 -- so x = r
// is x = 0 of the next row, z = r at LOD 0 the start of the LOD-1 region, and so on; ANOTHER cell's bit, or the unused gap
// behind a level (LOD l holds n / 8^l bits of the n / 2^l its base reserves).  By default such a query reads "empty" here
// (DESIGN.md section 5).  A pyramid type that is `oob_capable` and whose oob_ref() is true follows the reference instead:
// ref_bit() finds the level and cell a bit index belongs to and reads THAT cell's occupancy from the bit bricks; bits
// before the array or behind its 2n read 0 (memory nobody wrote).  Cold code: only cells outside the grid come here, and
// only instrumented kernel instantiations compile it in (the timed ones keep "empty").
template <class T, class = void> struct oob_capable_of { static constexpr bool value = false; };
template <class T> struct oob_capable_of<T, std::void_t<decltype(T::oob_capable)>> { static constexpr bool value = T::oob_capable; };

// occupancy of cell (cx, cy, cz) of LOD l (inside the grid), from the brick words: what descend() tests level by level
template <class PyrT>
VRT_DEV bool cell_occupied(const PyrT& P, int l, int cx, int cy, int cz) {
    constexpr int G = PyrT::G;
    typedef GridDim<G> D;
    const int x = cx << l, y = cy << l, z = cz << l;   // a LOD-0 cell inside it
    if (l >= 6) {
        if (G == 256) {
            const unsigned long long w3 = P.load_l3();
            return l == 7 ? (w3 & brick_sub(x >> 6, y >> 6, z >> 6)) != 0ULL : ((w3 >> brick_bit(x >> 6, y >> 6, z >> 6)) & 1ULL) != 0ULL;
        }
        return P.load_l2((((z >> 6) << D::s2) | (y >> 6)) << D::s2 | (x >> 6)) != 0ULL;
    }
    if (l >= 4) {
        const unsigned long long w2 = P.load_l2((((z >> 6) << D::s2) | (y >> 6)) << D::s2 | (x >> 6));
        return l == 5 ? (w2 & brick_sub(x >> 4, y >> 4, z >> 4)) != 0ULL : ((w2 >> brick_bit(x >> 4, y >> 4, z >> 4)) & 1ULL) != 0ULL;
    }
    if (l >= 2) {
        const unsigned long long w1 = P.load_l1((((z >> 4) << D::s1) | (y >> 4)) << D::s1 | (x >> 4));
        return l == 3 ? (w1 & brick_sub(x >> 2, y >> 2, z >> 2)) != 0ULL : ((w1 >> brick_bit(x >> 2, y >> 2, z >> 2)) & 1ULL) != 0ULL;
    }
    const unsigned long long w0 = P.load_l0((((z >> 2) << D::s0) | (y >> 2)) << D::s0 | (x >> 2));
    return l == 1 ? (w0 & brick_sub(x, y, z)) != 0ULL : ((w0 >> brick_bit(x, y, z)) & 1ULL) != 0ULL;
}
// bit `idx` of the reference's occupancy array (raytracer.py:40-44 over the layout of :17-38)
template <class PyrT>
VRT_DEV bool ref_bit(const PyrT& P, int idx) {
    constexpr int G = PyrT::G, LG = (G == 256) ? 8 : 7, n2 = 2 * G * G * G;
    if (idx < 0) return false;
    int l = 0;
    for (int k = 1; k < GridDim<G>::lods; k++) if (idx >= n2 - (n2 >> k)) l = k;
    const int off = idx - (l ? n2 - (n2 >> l) : 0), rb = LG - l;
    if (off >= (1 << (3 * rb))) return false;   // the gap behind LOD l, or past the last level
    return cell_occupied(P, l, off & ((1 << rb) - 1), (off >> rb) & ((1 << rb) - 1), off >> (2 * rb));
}
// the reference's inner loop (raytracer.py:110-118) for a LOD-0 cell (x, y, z) with a coordinate outside the grid
template <class PyrT>
VRT_DEV int descend_outside(const PyrT& P, int x, int y, int z, int lod, bool& solid, int& nq) {
    constexpr int G = PyrT::G, LG = (G == 256) ? 8 : 7, n2 = 2 * G * G * G;
    nq = 0;
    bool sample;
    for (;;) {
        const int r = 1 << (LG - lod);
        sample = ref_bit(P, (lod ? n2 - (n2 >> lod) : 0) + (z >> lod) * (r * r) + (y >> lod) * r + (x >> lod));
        nq += 1;
        if (sample && lod > 0) lod -= 1; else break;
    }
    solid = sample;
    return lod;
}

// Level the reference's descent ends at when it starts at `lod` on LOD-0 cell (x,y,z), and whether
// it ended on a solid voxel.  `nq` receives the number of query_occupancy calls the reference
// would have made (for the algorithmic-bytes counters).
template <class PyrT>
VRT_DEV int descend(const PyrT& P, int x, int y, int z, int lod, bool& solid, BrickCache& bc, int& nq) {
    constexpr int G = PyrT::G;
    typedef GridDim<G> D;
    solid = false;
    nq = 1;
    if ((x | y | z) & ~(G - 1)) {  // outside the grid: empty (undefined in the reference) ...
        if constexpr (oob_capable_of<PyrT>::value) { if (P.oob_ref()) return descend_outside(P, x, y, z, lod, solid, nq); }  // ... or the reference's own reading
        return lod;
    }
    const int start = lod;
    if (G == 256 && lod == 7) {  // the 2x2x2 block of 64^3 cells around the cell, from the top word
        if ((P.load_l3() & brick_sub(x >> 6, y >> 6, z >> 6)) == 0ULL) return 7;
        lod = 6;
    }
    if (lod >= 4) {
        unsigned long long w2 = P.load_l2((((z >> 6) << D::s2) | (y >> 6)) << D::s2 | (x >> 6));
        if (lod == 6) { if (w2 == 0ULL) { nq = start - 6 + 1; return 6; } lod = 5; }
        if (lod == 5) { if ((w2 & brick_sub(x >> 4, y >> 4, z >> 4)) == 0ULL) { nq = start - 5 + 1; return 5; } lod = 4; }
        if (((w2 >> brick_bit(x >> 4, y >> 4, z >> 4)) & 1ULL) == 0ULL) { nq = start - 4 + 1; return 4; }
        lod = 3;
    }
    if (lod >= 2) {
        unsigned long long w1 = P.load_l1((((z >> 4) << D::s1) | (y >> 4)) << D::s1 | (x >> 4));
        if (lod == 3) { if ((w1 & brick_sub(x >> 2, y >> 2, z >> 2)) == 0ULL) { nq = start - 3 + 1; return 3; } lod = 2; }
        if (((w1 >> brick_bit(x >> 2, y >> 2, z >> 2)) & 1ULL) == 0ULL) { nq = start - 2 + 1; return 2; }
        lod = 1;
    }
    const int key = (((z >> 2) << D::s0) | (y >> 2)) << D::s0 | (x >> 2);
    if (key != bc.key) { bc.key = key; bc.word = P.load_l0(key); }
    const unsigned long long w0 = bc.word;
    if (lod == 1) { if ((w0 & brick_sub(x, y, z)) == 0ULL) { nq = start - 1 + 1; return 1; } }
    nq = start + 1;
    solid = ((w0 >> brick_bit(x, y, z)) & 1ULL) != 0ULL;
    return 0;
}

// One 4x4x4 brick word answers two LODs for the cell (cx,cy,cz) of its finer level: bit 0 of the result = that
// cell is occupied, bit 1 = the 2x2x2 block around it holds an occupied cell.  One 64-bit shift brings the block to
// the bottom of the word; what is left is 32-bit work.
VRT_DEV unsigned brick_two_lods(unsigned long long w, int cx, int cy, int cz) {
    const int bit = ((cz & 3) << 4) | ((cy & 3) << 2) | (cx & 3);   // the cell's bit
    const unsigned u = (unsigned)(w >> (bit & 0x2a));               // word shifted down to the cell's 2x2x2 block: its 8 bits are 0x00330033
    const unsigned cell = (u >> (bit & 0x15)) & 1u;
    const unsigned block = (u & 0x00330033u) != 0u ? 2u : 0u;
    return block | cell;
}

// The coarse context of a LOD-0 cell: the l1 and l2 brick words it falls in and where the fine words behind that
// l1 word start in the compacted fine level (Pyramid::l0c).
struct CoarseWords { unsigned long long w1, w2; uint32_t fine_base; };

// All of it in one request (one 16-byte and one 4-byte LDS read in the kernels).  It depends on the cell only, so
// the WALK loop asks for the next cell's context at the end of a step and meets it a step later.
template <class PyrT>
VRT_DEV void coarse_fetch(const PyrT& P, int x, int y, int z, CoarseWords& c) {
    constexpr int G = PyrT::G;
    typedef GridDim<G> D;
    const int xm = x & (G - 1), ym = y & (G - 1), zm = z & (G - 1);  // keeps the table index in range
    P.load_coarse((((zm >> 4) << D::s1) | (ym >> 4)) << D::s1 | (xm >> 4), (((zm >> 6) << D::s2) | (ym >> 6)) << D::s2 | (xm >> 6),
                  c.w1, c.w2, c.fine_base);
}

// descend() without the walk: same result, fixed cost.  descend() is a seven-way nest of branches; a wave whose
// lanes sit at mixed LODs executes most of it (about 190 of the 256 instructions of a DDA step), which is what a
// FULL wave of rays does -- the pooled schedule's WALK stage (vrt_pool.h).  Here the occupancy of all seven levels
// of the cell is gathered into one 7-bit mask -- the coarse context and the cached fine brick word -- and the level
// the walk stops at is the highest empty level at or below the starting one: a complement, a mask, a count-leading-
// zeros.  (A wave with few active lanes, whose rays mostly stop at their first query, is better off with descend():
// measured 3-10 % on the fused kernel.)  The fine word is asked for by its rank among the set bits of the l1 word,
// which is where the compacted fine level keeps it (load_fine; a pyramid without one just indexes l0 by `key`).
template <class PyrT>
VRT_DEV int descend_flat(const PyrT& P, const CoarseWords& c, int x, int y, int z, int lod, bool& solid, BrickCache& bc, int& nq) {
    constexpr int G = PyrT::G;
    typedef GridDim<G> D;
    const bool inside = ((x | y | z) & ~(G - 1)) == 0;  // outside the grid: empty, as in descend()
    if constexpr (oob_capable_of<PyrT>::value) { if (!inside && P.oob_ref()) return descend_outside(P, x, y, z, lod, solid, nq); }
    const int xm = x & (G - 1), ym = y & (G - 1), zm = z & (G - 1);
    unsigned occ = (c.w2 != 0ULL ? 64u : 0u) | (brick_two_lods(c.w2, xm >> 4, ym >> 4, zm >> 4) << 4) |
                   (brick_two_lods(c.w1, xm >> 2, ym >> 2, zm >> 2) << 2);
    if (G == 256) occ |= (P.load_l3() & brick_sub(xm >> 6, ym >> 6, zm >> 6)) != 0ULL ? 128u : 0u;  // LOD 7 (LOD 6 = its bit = w2 != 0)
    const unsigned upto = (2u << lod) - 1u;  // levels 0..lod
    // the fine brick word matters only when every level from `lod` down to 2 is occupied
    if (inside && (~occ & upto & ~3u) == 0u) {
        const int key = (((zm >> 2) << D::s0) | (ym >> 2)) << D::s0 | (xm >> 2);
        VRT_REGION(19);
        if (key != bc.key) {
            VRT_REGION(18);
            const int bit = brick_bit(xm >> 2, ym >> 2, zm >> 2);  // of this fine brick in the l1 word
            const unsigned long long below = c.w1 & ((1ULL << bit) - 1ULL);
            bc.key = key;
            bc.word = 0ULL;  // a walk that starts below LOD 2 can stand in an empty brick: nothing to load
            if ((c.w1 >> bit) & 1ULL) bc.word = P.load_fine(key, c.fine_base + (uint32_t)__builtin_popcountll(below));
        }
        occ |= brick_two_lods(bc.word, xm, ym, zm);
    }
    const unsigned empty = inside ? (~occ & upto) : upto;
    solid = empty == 0u;
    const int level = solid ? 0 : 31 - __builtin_clz(empty);
    nq = lod - level + 1;
    return level;
}

template <int G_>
struct GlobalPyramid {  // all brick levels read from global memory
    static constexpr int G = G_;
    static constexpr bool flat_descend = false;
    static constexpr bool cull = true;   // rays that cannot hit a voxel are not walked (cull_ray); false compiles the test out
    static constexpr bool oob_capable = true;
    Pyramid p;
    VRT_DEV bool oob_ref() const { return p.ref_oob != 0; }
    VRT_DEV unsigned long long load_l0(int i) const { return p.l0[i]; }
    VRT_DEV unsigned long long load_l1(int i) const { return p.l1[i]; }
    VRT_DEV unsigned long long load_l2(int i) const { return p.l2[i]; }
    VRT_DEV unsigned long long load_l3() const { return p.l3[0]; }
    VRT_DEV unsigned long long load_fine(int key, uint32_t idx) const { return G == 128 ? p.l0c[idx] : p.l0[key]; }
    // i1 / i2: the l1 word of the cell and the l2 word holding it
    VRT_DEV void load_coarse(int i1, int i2, unsigned long long& w1, unsigned long long& w2, uint32_t& fine_base) const {
        w1 = p.l1[i1];
        w2 = p.l2[i2];
        fine_base = G == 128 ? p.l0c_base[i1] : 0u;
    }
};

struct TraceOut { float dist; int ix, iy, iz; f3 normal; int iters; };

// ---- rays that cannot hit anything -----------------------------------------------------------------------------------
// The walk's only observable results are "first solid voxel on the ray" or "none" (distance inf: the caller then ignores
// cell, normal and step count, pathtracer.py:203-205).  A ray that stays clear of every solid voxel is therefore a miss
// whatever the walk does on the way, and the part of a ray behind its last chance of a hit need not be walked.  `cull` is
// the bounding box of the solid voxels grown by VRT_CULL_MARGIN on every side (built by prepare, brick granular):
//   * a ray that misses the grown box misses every voxel -- no walk at all (and none of the nine divisions of its set-up);
//   * inside the grid the walk may stop where the ray leaves the grown box (G = 128, see below).
// Why a margin, and why 8: (1) the slab test below is approximate (v_rcp_f32, no correctly rounded division): its error
// moves the box faces by < 1e-4 voxel; (2) the reference's loop also ends after 512 steps with a FINITE distance -- "a hit
// on an empty cell" (raytracer.py:103) -- and a culled ray must not be one of those: a point 8 or more
// voxels from every solid lies in empty cells at LODs 0-3 (an aligned cube of edge <= 8 around it holds no solid), so the
// unwalked part advances a LOD-3 cell (8 voxels) or more per step: <= 3 * 256 / 8 = 96 steps, and the walked part of a 128
// grid at most 3 * 128 = 384 cell-boundary crossings: 512 is out of reach.  (At 256 a walked part could in principle
// use more than 416 steps, so the walk is not cut short there -- only whole rays are culled.)
// Rays with a zero or non-finite direction component are left to the walk (their slab arithmetic is the reference's NaN
// business).  Instrumented launches that count the reference's work get a box that holds everything: nothing is culled.
#ifndef VRT_CULL_MARGIN
#define VRT_CULL_MARGIN 8.0f
#endif
VRT_DEV float fast_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
// false: the ray (origin o, direction d, voxel units) cannot hit a solid voxel.  true: it may; t_exit = where it leaves
// the grown box (inf when no statement is made).
VRT_DEV bool cull_ray(const float* cull, f3 o, f3 d, float& t_exit) {
    t_exit = DM_INF;
    if (cull[6] == 0.0f) return true;   // the box is the grid (or culling is off): the same for every ray of the launch
    const f3 ad = abs3(d);
    if (!(ad.x > 0.0f && ad.y > 0.0f && ad.z > 0.0f && ad.x < DM_INF && ad.y < DM_INF && ad.z < DM_INF)) return true;
    const f3 inv = mk3(fast_rcp(d.x), fast_rcp(d.y), fast_rcp(d.z));
    const f3 a = (mk3(cull[0], cull[1], cull[2]) - o) * inv, b = (mk3(cull[3], cull[4], cull[5]) - o) * inv;
    const float tn = dm_max(dm_max(dm_min(a.x, b.x), dm_min(a.y, b.y)), dm_min(a.z, b.z));
    const float tf = dm_min(dm_min(dm_max(a.x, b.x), dm_max(a.y, b.y)), dm_max(a.z, b.z));
    if (!(tf >= dm_max(tn, 0.0f))) return false;   // also an empty box (lo > hi: no solids at all) and a NaN
    t_exit = tf;
    return true;
}

// raytracer.py:72-155 with ray_min_t = eps, ray_max_t = inf (the only call site, pathtracer.py:201-202).
// FLAT: which descent (descend_flat / descend) -- by default what the pyramid type says suits its kernel.  A pyramid type may
// also say `shadow_branchy`: the shadow rays the pooled kernel's SHADE stage walks inline run at ~15 of 64 lanes, where the
// branchy descent's early outs win IN A DENSE GRID (a ray ends after 1.2 steps: dense 128^3 / 256^3 at 4K +3.8 %), while the
// long shadow rays of a sparse scene lose (S6 -5.6 %): the launcher picks the instantiation by the share of non-empty bricks.
template <class T, class = void> struct shadow_branchy_of { static constexpr bool value = false; };
template <class T> struct shadow_branchy_of<T, std::void_t<decltype(T::shadow_branchy)>> { static constexpr bool value = T::shadow_branchy; };
template <class PyrT, bool FLAT = PyrT::flat_descend>
VRT_DEV void raytrace(const PyrT& P, f3 o, f3 d, TraceOut& r, int& queries, const float* cull) {
    float hit_distance = DM_INF;
    int ix = -1, iy = -1, iz = -1;
    f3 hn = mk3(0.0f);
    int iters = 0;
    queries = 0;
    constexpr int G = PyrT::G;
    const float res = (float)G;
    float t_exit = DM_INF;
    if constexpr (PyrT::cull) {
        if (!cull_ray(cull, o, d, t_exit)) {  // clear of every solid voxel: a miss
            r.dist = DM_INF; r.ix = -1; r.iy = -1; r.iz = -1; r.normal = hn; r.iters = 0;
            return;
        }
    }

    // math_utils.py:103-123 against the box [0,G]^3
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
        float far = dm_min(DM_INF, far_t) - VRT_EPS;
        if (G == 128) far = dm_min(far, t_exit);   // behind the grown box nothing can be hit (cull_ray)

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
            VRT_REGION(7);
            if constexpr (FLAT) {  // the pyramid type says which descent suits its kernel (see descend_flat)
                CoarseWords c;
                coarse_fetch(P, ix, iy, iz, c);
                lod = descend_flat(P, c, ix, iy, iz, lod, solid, bc, nq);
            } else {
                lod = descend(P, ix, iy, iz, lod, solid, bc, nq);
            }
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
            lod = (lod + 1 > GridDim<G>::max_lod) ? GridDim<G>::max_lod : lod + 1;
            iters += 1;
        }
    }
    if (dot3(d, hn) > 0.0f) hn = -hn;
    r.dist = hit_distance;
    r.ix = ix; r.iy = iy; r.iz = iz;
    r.normal = hn;
    r.iters = iters;
}

// ---- the same walk as a resumable record ---------------------------------------------------------
// RayWalk holds exactly the loop-carried state of raytrace() above, so a walk can be set up by one piece of code,
// advanced a step at a time by another and suspended in between (vrt_pool.h keeps suspended walks in LDS).
// walk_prepare + walk_trip until it returns true + walk_result compute what raytrace() computes, bit for bit.
struct RayWalk {
    f3 o, d, inv_dir, sd;  // sd = sign(d), a function of d kept beside it
    float t, far;          // hit_distance, exit distance of the grid box
    int ix, iy, iz, lod, iters;
    f3 hn;
};

// raytracer.py:81-101: clip against the grid box, first cell, entry-face normal.  False = the box is missed.
template <int G, bool CULL = true>
VRT_DEV bool walk_prepare(f3 o, f3 d, RayWalk& w, const float* cull) {
    w.o = o; w.d = d;
    w.sd = mk3(sgn(d.x), sgn(d.y), sgn(d.z));
    w.t = DM_INF; w.far = 0.0f;
    w.ix = -1; w.iy = -1; w.iz = -1; w.lod = 0; w.iters = 0;
    w.hn = mk3(0.0f);
    w.inv_dir = mk3(0.0f);
    float t_exit = DM_INF;
    if constexpr (CULL) { if (!cull_ray(cull, o, d, t_exit)) return false; }   // clear of every solid voxel: nothing to walk, the ray is a miss
    const float res = (float)G;
    float near_t = -DM_INF, far_t = DM_INF;
#define VRT_SLAB(oc, dc)                                                     \
    if (dc != 0.0f) {                                                        \
        float i1 = (0.0f - oc) / dc, i2 = (res - oc) / dc;                   \
        far_t = dm_min(dm_max(i1, i2), far_t);                               \
        near_t = dm_max(dm_min(i1, i2), near_t);                             \
    }
    VRT_SLAB(o.x, d.x) VRT_SLAB(o.y, d.y) VRT_SLAB(o.z, d.z)
#undef VRT_SLAB
    if (!(near_t <= far_t && VRT_EPS < far_t && DM_INF > near_t)) return false;
    w.t = dm_max(near_t, VRT_EPS);
    const f3 p0 = o + d * (w.t + VRT_EPS);
    const f3 c = clamp3(floor3(p0), 0.0f, res - 1.0f);
    w.ix = (int)c.x; w.iy = (int)c.y; w.iz = (int)c.z;
    w.inv_dir = mk3(1.0f / dm_abs(d.x), 1.0f / dm_abs(d.y), 1.0f / dm_abs(d.z));
    w.far = dm_min(DM_INF, far_t) - VRT_EPS;
    if (G == 128) w.far = dm_min(w.far, t_exit);
    const f3 id = abs3(p0 - res * 0.5f);
    const float md = dm_max(dm_max(id.x, id.y), id.z);
    w.hn = mk3(md == id.x ? 1.0f : 0.0f, md == id.y ? 1.0f : 0.0f, md == id.z ? 1.0f : 0.0f);
    return true;
}

// One pass of the loop raytracer.py:103-147.  True = the walk is over (hit, miss or 512 steps).  `c` holds the
// coarse words of the walk's current cell (coarse_fetch) on entry and of its next cell on return.
template <class PyrT>
VRT_DEV bool walk_trip(const PyrT& P, RayWalk& w, BrickCache& bc, CoarseWords& c, int& nq) {
    nq = 0;
    if (w.iters >= 512) return true;
    if (w.t > w.far) { w.t = DM_INF; return true; }
    bool solid;
    VRT_REGION(1);
    // (the branchy descent here too, in the dense-grid variant: -0.8 % / 0 on the dense 4K frames, profiles/README.md)
    w.lod = descend_flat(P, c, w.ix, w.iy, w.iz, w.lod, solid, bc, nq);
    if (solid) return true;
    // The step of raytrace() with three of its values produced by cheaper, bit-identical means: 2^lod assembled from
    // its exponent, the cell base as float(ix with its low lod bits cleared) (= float(ix >> lod) * 2^lod: both exact),
    // and the step normal selected between sd and sd's signed zero (= {1, 0} * sd).
    const int lod = w.lod;
    const f3 d = w.d;
    const float cell_size = dm_u2f(0x3f800000u + ((uint32_t)lod << 23));
    const int keep = ~((1 << lod) - 1);
    const f3 cell_base = mk3((float)(w.ix & keep), (float)(w.iy & keep), (float)(w.iz & keep));
    const f3 fp = (w.o + d * w.t) - cell_base;
    f3 dist;
    dist.x = (d.x > 0.0f) ? cell_size - fp.x : fp.x;
    dist.y = (d.y > 0.0f) ? cell_size - fp.y : fp.y;
    dist.z = (d.z > 0.0f) ? cell_size - fp.z : fp.z;
    const f3 t = dist * w.inv_dir;
    const float min_t = dm_min(dm_min(t.x, t.y), t.z);
    const f3 edge = clamp3(floor3(fp + min_t * d), 0.0f, cell_size - 1.0f);
    w.t += min_t;
    w.hn = mk3(t.x == min_t ? w.sd.x : dm_u2f(dm_f2u(w.sd.x) & 0x80000000u), t.y == min_t ? w.sd.y : dm_u2f(dm_f2u(w.sd.y) & 0x80000000u),
               t.z == min_t ? w.sd.z : dm_u2f(dm_f2u(w.sd.z) & 0x80000000u));
    const f3 nxt = cell_base + edge + w.hn;
    w.ix = (int)nxt.x; w.iy = (int)nxt.y; w.iz = (int)nxt.z;
    coarse_fetch(P, w.ix, w.iy, w.iz, c);
    w.lod = (lod + 1 > GridDim<PyrT::G>::max_lod) ? GridDim<PyrT::G>::max_lod : lod + 1;
    w.iters += 1;
    return false;
}

// raytracer.py:152-155
VRT_DEV void walk_result(f3 d, float t, int ix, int iy, int iz, f3 hn, int iters, TraceOut& r) {
    if (dot3(d, hn) > 0.0f) hn = -hn;
    r.dist = t;
    r.ix = ix; r.iy = iy; r.iz = iz;
    r.normal = hn;
    r.iters = iters;
}

struct Hit { float closest; f3 normal; f3 albedo; int hit_light; int mat_id; };

struct TraceStats { unsigned rays, iters, queries, closest_hits, sky_lookups; };
VRT_DEV void stats_zero(TraceStats& s) { s.rays = s.iters = s.queries = s.closest_hits = s.sky_lookups = 0u; }

VRT_DEV void hit_init(Hit& h) {
    h.closest = DM_INF;
    h.normal = mk3(0.0f);
    h.albedo = mk3(1.0f);
    h.hit_light = 0;
    h.mat_id = 0;
}
// pathtracer.py:173-190: infinite floor y = floor_height, accepted inside the radius-10 "disc" of :183 -- the scalar
// dot(hit, up) = hit.y is subtracted from all three components.  Returns the hit distance, inf for no hit.
VRT_DEV float floor_probe(const FrameParams& fp, f3 pos, f3 d) {
    const float t = (fp.floor_height - pos.y) / d.y;
    if (t > VRT_EPS && t < DM_INF) {
        const f3 hp = pos + d * t;
        const float s = hp.x * 0.0f + hp.y * 1.0f + hp.z * 0.0f;
        if (len3(hp - s) < 10.0f) return t;
    }
    return DM_INF;
}
VRT_DEV void hit_floor(const FrameParams& fp, f3 d, float t, Hit& h) {
    h.closest = t;
    h.normal = mk3(0.0f, 1.0f, 0.0f);
    if (dot3(h.normal, d) > 0.0f) h.normal = -h.normal;
    h.albedo = fp.floor_color;
    h.hit_light = (fp.floor_material == 2) ? 1 : 0;
    h.mat_id = fp.floor_material;
}
template <int G>
VRT_DEV f3 world_to_voxel(f3 pos) { return GridDim<G>::half * pos - (-GridDim<G>::half); }  // pathtracer.py:165-171: voxel_inv_size * pos - voxel_grid_offset
// pathtracer.py:203-216 + voxel_world.py:34-56: the walk's result against what is closest so far.  SHADOW: no surface lookup.
template <bool SHADOW, int G>
VRT_DEV void hit_voxel(const FrameParams& fp, const SceneData& sc, f3 eye, f3 d, const TraceOut& tr, Hit& h, TraceStats& ts) {
    const float voxel_size = GridDim<G>::voxel_size;
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
            if (((tr.ix | tr.iy | tr.iz) & ~(G - 1)) == 0) {
                uint32_t texel = sc.grid[texel_index<G>(tr.ix, tr.iy, tr.iz)];
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

// pathtracer.py:218-244 (floor plane 173-190, voxel grid 192-216).
template <bool SHADOW, class PyrT>
VRT_DEV void next_hit(const FrameParams& fp, const SceneData& sc, const PyrT& P, f3 pos, f3 d, Hit& h, TraceStats& ts) {
    hit_init(h);
    const float ft = floor_probe(fp, pos, d);
    if (ft < DM_INF) hit_floor(fp, d, ft, h);
    const f3 eye = world_to_voxel<PyrT::G>(pos);
    TraceOut tr;
    int nq;
    raytrace<PyrT, PyrT::flat_descend && !(SHADOW && shadow_branchy_of<PyrT>::value)>(P, eye, d, tr, nq, sc.cull);
    ts.rays += 1u; ts.iters += (unsigned)tr.iters; ts.queries += (unsigned)nq;
    VRT_REGION(SHADOW ? 3 : 9);  // ray set-up + result (one entry per ray)
    hit_voxel<SHADOW, PyrT::G>(fp, sc, eye, d, tr, h, ts);
}

}  // namespace vrt
#endif
