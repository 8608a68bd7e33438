// vrt_pool.h -- camera paths as records in a per-wave LDS pool, advanced stage by stage.
//
// Same arithmetic as vrt_path.h (Renderer.render, reference renderer/pathtracer.py:355-632), different schedule.
// In the fused kernel (k_render) a lane owns one path and walks it segment by segment; a wave of 64 rays then
// runs its DDA loop until the LONGEST of them is done (measured: 25 of 64 lanes busy in that loop on the sparse
// scene, 16 on the dense one) and shades hits, misses and fresh pixels side by side in one divergent pass.
//
// Here a wave owns VRT_POOL_SLOTS (> 64) paths.  Their state lives in LDS, not in lanes, so that the wave can pick
// what to do next by what there is most of:
//
//   WALK    the DDA loop over every pending ray.  A lane whose ray ends hands the result to the ray's slot and
//           takes the next pending ray IN the loop, so the loop runs full until the list is empty; the few rays
//           still going when too few lanes are left are suspended back into their slots (RayWalk is resumable).
//   SHADE   64 paths whose ray found a surface: path_shade<HIT_SOMETHING>, then the set-up of the bounce ray.
//   ESCAPE  64 paths whose ray left the scene: sky / background, path_finish.
//   BEGIN   64 empty slots: fresh (tile, sample, pixel) work items, path_begin, set-up of the camera ray.
//
// Each stage therefore runs one piece of code on (nearly) 64 lanes.  Results do not depend on the schedule: a
// path owns its random stream and its output pixel, and every stage function is the fused path's code
// (walk_prepare/walk_trip = raytrace(); path_shade = the rest of path_segment()).
//
// The record is split by how often it is touched: 25 dwords that every stage needs stay in LDS (SoA, one
// column per slot); 14 dwords written at the primary vertex and read back only by path_finish (light-sample sums,
// primary vertex data) go to a per-wave scratch line in global memory, which stays in L2.
#ifndef VRT_POOL_H
#define VRT_POOL_H

#include "vrt_path.h"

#ifndef VRT_POOL_SLOTS
#define VRT_POOL_SLOTS 128
#endif

namespace vrt {

enum {  // slot states
    SLOT_EMPTY = 0,   // no path: BEGIN may use it
    SLOT_RAY = 1,     // closest-hit ray pending or suspended
    SLOT_SHADE = 2,   // ray ended on the floor or a voxel
    SLOT_ESCAPE = 3   // ray ended on nothing
};

enum {  // LDS columns of a slot (dwords)
    PF_POS = 0, PF_DIR = 3, PF_THR = 6, PF_CONTRIB = 9, PF_RNG = 12,
    PF_IDS = 13,      // pix_u | pix_v << 12 | depth << 24 | sample << 28
    PF_FLAGS = 14,    // first_lobe | sky_primary << 2
    PF_REFL = 15,
    PF_FLOOR_T = 16,  // floor_probe() of the pending ray
    PF_T = 17,        // RayWalk.t: distance so far / final distance
    PF_FAR = 18,
    PF_INV = 19,      // RayWalk.inv_dir
    PF_CELL_XY = 22,  // ix | iy << 16 (two's complement halves)
    PF_CELL_Z = 23,   // iz | lod << 16 | normal code << 20
    PF_ITERS = 24,
    PF_COUNT = 25
};
enum { PC_NEE_D = 0, PC_NEE_S = 3, PC_ALBEDO = 6, PC_PPOS = 9, PC_INVPDF = 12, PC_MAT = 13, PC_COUNT = 16,   // scratch line
       // ReSTIR: the reconnection state of PathRestir (vrt_path.h) follows.  Each of its fields is produced at ONE depth (the
       // first light sample at the primary vertex, the reconnection vertex at depth 1, its outgoing direction at depth 2) except
       // the two running values behind the reconnection vertex: a SHADE / ESCAPE therefore reads and writes only what its depth
       // touches (restir_cold_*), and the whole line is read once, when the path ends
       PC_RS = 16, PC_COUNT_RESTIR = 48,
       // offsets behind PC_RS
       RS_THR = 0, RS_FIRST_DIR = 3, RS_FIRST_LIGHT_DIR = 6, RS_RC_POS = 9, RS_RC_NORMAL = 12, RS_RC_INC_DIR = 15, RS_RC_INC_L = 18,
       RS_RC_NEE_DIR = 21, RS_RC_MAT = 24, RS_FIRST_LIGHT_PDF = 25, RS_RC_LOBE = 26 };
template <bool RESTIR> struct ColdLine { static constexpr int count = RESTIR ? (int)PC_COUNT_RESTIR : (int)PC_COUNT; };

// A slot's column in the pool: field f lives at base[f * stride].
struct SlotRef {
    uint32_t* base;
    int stride;
    VRT_DEV uint32_t u(int f) const { return base[f * stride]; }
    VRT_DEV float f(int f_) const { return dm_u2f(base[f_ * stride]); }
    VRT_DEV f3 v(int f_) const { return mk3(f(f_), f(f_ + 1), f(f_ + 2)); }
    VRT_DEV void su(int f_, uint32_t x) const { base[f_ * stride] = x; }
    VRT_DEV void sf(int f_, float x) const { base[f_ * stride] = dm_f2u(x); }
    VRT_DEV void sv(int f_, f3 x) const { sf(f_, x.x); sf(f_ + 1, x.y); sf(f_ + 2, x.z); }
};

// the step normal has components in {+0, -0, +1, -1}: two bits each (sign, magnitude)
VRT_DEV uint32_t normal_code(f3 n) {
    const uint32_t cx = (dm_f2u(n.x) >> 31) | (n.x != 0.0f ? 2u : 0u);
    const uint32_t cy = (dm_f2u(n.y) >> 31) | (n.y != 0.0f ? 2u : 0u);
    const uint32_t cz = (dm_f2u(n.z) >> 31) | (n.z != 0.0f ? 2u : 0u);
    return cx | (cy << 2) | (cz << 4);
}
VRT_DEV float normal_comp(uint32_t c) { return dm_u2f(((c & 1u) << 31) | ((c & 2u) ? 0x3f800000u : 0u)); }
VRT_DEV f3 normal_decode(uint32_t c) { return mk3(normal_comp(c), normal_comp(c >> 2), normal_comp(c >> 4)); }

VRT_DEV void walk_store(const SlotRef& s, const RayWalk& w) {
    s.sf(PF_T, w.t);
    s.su(PF_CELL_XY, ((uint32_t)w.ix & 0xffffu) | ((uint32_t)w.iy << 16));
    s.su(PF_CELL_Z, ((uint32_t)w.iz & 0xffffu) | ((uint32_t)w.lod << 16) | (normal_code(w.hn) << 20));
    s.su(PF_ITERS, (uint32_t)w.iters);
}
VRT_DEV void walk_store_constants(const SlotRef& s, const RayWalk& w) {
    s.sf(PF_FAR, w.far);
    s.sv(PF_INV, w.inv_dir);
}
template <int G>
VRT_DEV void walk_load(const SlotRef& s, RayWalk& w) {
    w.o = world_to_voxel<G>(s.v(PF_POS));
    w.d = s.v(PF_DIR);
    w.sd = mk3(sgn(w.d.x), sgn(w.d.y), sgn(w.d.z));
    w.inv_dir = s.v(PF_INV);
    w.t = s.f(PF_T);
    w.far = s.f(PF_FAR);
    const uint32_t a = s.u(PF_CELL_XY), b = s.u(PF_CELL_Z);
    w.ix = (int)(int16_t)(a & 0xffffu); w.iy = (int)(int16_t)(a >> 16); w.iz = (int)(int16_t)(b & 0xffffu);
    w.lod = (int)((b >> 16) & 7u);   // LOD 0..7
    w.hn = normal_decode(b >> 20);
    w.iters = (int)s.u(PF_ITERS);
}

VRT_DEV uint32_t pack_ids(int u, int v, int depth, int sample) {
    return (uint32_t)u | ((uint32_t)v << 12) | ((uint32_t)depth << 24) | ((uint32_t)sample << 28);
}

template <bool RESTIR>
VRT_DEV void path_store_hot(const SlotRef& s, const Path<RESTIR>& p) {
    s.sv(PF_POS, p.pos); s.sv(PF_DIR, p.d); s.sv(PF_THR, p.thr); s.sv(PF_CONTRIB, p.contrib);
    s.su(PF_RNG, p.rng.s);
    s.su(PF_IDS, pack_ids(p.pix_u, p.pix_v, p.depth, p.sample));
    s.su(PF_FLAGS, (uint32_t)p.first_lobe | ((uint32_t)p.sky_primary << 2));
    s.sf(PF_REFL, p.refl_dist);
}
// the fields a path leaves its primary vertex with (path_begin's defaults until then)
template <bool RESTIR>
VRT_DEV void path_cold_defaults(Path<RESTIR>& p) {
    p.nee_d = mk3(0.0f); p.nee_s = mk3(0.0f);
    p.primary_albedo = mk3(1.0f); p.primary_pos = mk3(0.0f);
    p.first_invpdf = 1.0f;
    p.primary_mat_info = 0u;
    if constexpr (RESTIR) {  // as path_begin leaves them
        p.rs.thr_after_rc = mk3(1.0f);
        p.rs.first_dir = mk3(0.0f); p.rs.first_light_dir = mk3(0.0f);
        p.rs.rc_pos = mk3(0.0f); p.rs.rc_normal = mk3(0.0f); p.rs.rc_incident_dir = mk3(0.0f);
        p.rs.rc_incident_L = mk3(0.0f); p.rs.rc_nee_dir = mk3(0.0f);
        p.rs.rc_mat_info = 0u;
        p.rs.first_light_bsdf_pdf = 1.0f;
        p.rs.rc_lobe = 0;
    }
}
template <bool RESTIR>
VRT_DEV void path_load_hot(const SlotRef& s, Path<RESTIR>& p) {
    p.pos = s.v(PF_POS); p.d = s.v(PF_DIR); p.thr = s.v(PF_THR); p.contrib = s.v(PF_CONTRIB);
    p.rng.s = s.u(PF_RNG);
    const uint32_t ids = s.u(PF_IDS), fl = s.u(PF_FLAGS);
    p.pix_u = (int)(ids & 0xfffu); p.pix_v = (int)((ids >> 12) & 0xfffu);
    p.depth = (int)((ids >> 24) & 15u); p.sample = (int)(ids >> 28);
    p.first_lobe = (int)(fl & 3u); p.sky_primary = (int)((fl >> 2) & 1u);
    p.refl_dist = s.f(PF_REFL);
    path_cold_defaults(p);
}
// (paths without ReSTIR: written once, at the primary vertex, read once, when the path ends; with ReSTIR: restir_cold_* below)
template <bool RESTIR>
VRT_DEV void path_store_cold(uint32_t* line, const Path<RESTIR>& p) {
    line[PC_NEE_D] = dm_f2u(p.nee_d.x); line[PC_NEE_D + 1] = dm_f2u(p.nee_d.y); line[PC_NEE_D + 2] = dm_f2u(p.nee_d.z);
    line[PC_NEE_S] = dm_f2u(p.nee_s.x); line[PC_NEE_S + 1] = dm_f2u(p.nee_s.y); line[PC_NEE_S + 2] = dm_f2u(p.nee_s.z);
    line[PC_ALBEDO] = dm_f2u(p.primary_albedo.x); line[PC_ALBEDO + 1] = dm_f2u(p.primary_albedo.y); line[PC_ALBEDO + 2] = dm_f2u(p.primary_albedo.z);
    line[PC_PPOS] = dm_f2u(p.primary_pos.x); line[PC_PPOS + 1] = dm_f2u(p.primary_pos.y); line[PC_PPOS + 2] = dm_f2u(p.primary_pos.z);
    line[PC_INVPDF] = dm_f2u(p.first_invpdf);
    line[PC_MAT] = p.primary_mat_info;
}
template <bool RESTIR>
VRT_DEV void path_load_cold(const uint32_t* line, Path<RESTIR>& p) {
    p.nee_d = mk3(dm_u2f(line[PC_NEE_D]), dm_u2f(line[PC_NEE_D + 1]), dm_u2f(line[PC_NEE_D + 2]));
    p.nee_s = mk3(dm_u2f(line[PC_NEE_S]), dm_u2f(line[PC_NEE_S + 1]), dm_u2f(line[PC_NEE_S + 2]));
    p.primary_albedo = mk3(dm_u2f(line[PC_ALBEDO]), dm_u2f(line[PC_ALBEDO + 1]), dm_u2f(line[PC_ALBEDO + 2]));
    p.primary_pos = mk3(dm_u2f(line[PC_PPOS]), dm_u2f(line[PC_PPOS + 1]), dm_u2f(line[PC_PPOS + 2]));
    p.first_invpdf = dm_u2f(line[PC_INVPDF]);
    p.primary_mat_info = line[PC_MAT];
}

// ---- the ReSTIR part of the scratch line, by depth ------------------------------------------------------------------------
// Which PathRestir field a vertex at depth d produces (path_shade, vrt_path.h): d = 0: first_light_bsdf_pdf, first_light_dir
// (and the primary-vertex words PC_NEE_D .. PC_MAT); d = 1: first_dir, rc_pos, rc_normal, rc_mat_info, rc_nee_dir, rc_lobe
// (an escape at depth 1: rc_pos, rc_incident_L -- and the path ends); d = 2: rc_incident_dir; d >= 2: thr_after_rc and
// rc_incident_L, the only fields a vertex READS.  A path that goes on from depth 1 found a surface there, so it arrives at
// depth 2 with thr_after_rc = 1 and rc_incident_L = 0 -- path_begin's values, which path_load_hot() has already put in `p`.
VRT_DEV void cold_st3(uint32_t* at, f3 v) { at[0] = dm_f2u(v.x); at[1] = dm_f2u(v.y); at[2] = dm_f2u(v.z); }
VRT_DEV f3 cold_ld3(const uint32_t* at) { return mk3(dm_u2f(at[0]), dm_u2f(at[1]), dm_u2f(at[2])); }
// before the vertex at `depth` is shaded: what it reads
VRT_DEV void restir_cold_load_running(const uint32_t* line, Path<true>& p, int depth) {
    if (depth >= 3) { p.rs.thr_after_rc = cold_ld3(line + PC_RS + RS_THR); p.rs.rc_incident_L = cold_ld3(line + PC_RS + RS_RC_INC_L); }
}
// the path goes on: what the vertex at `depth` produced
VRT_DEV void restir_cold_store(uint32_t* line, const Path<true>& p, int depth) {
    uint32_t* r = line + PC_RS;
    if (depth == 0) {
        cold_st3(line + PC_NEE_D, p.nee_d); cold_st3(line + PC_NEE_S, p.nee_s); cold_st3(line + PC_ALBEDO, p.primary_albedo);
        cold_st3(line + PC_PPOS, p.primary_pos);
        line[PC_INVPDF] = dm_f2u(p.first_invpdf); line[PC_MAT] = p.primary_mat_info;
        cold_st3(r + RS_FIRST_LIGHT_DIR, p.rs.first_light_dir);
        r[RS_FIRST_LIGHT_PDF] = dm_f2u(p.rs.first_light_bsdf_pdf);
    } else if (depth == 1) {
        cold_st3(r + RS_FIRST_DIR, p.rs.first_dir); cold_st3(r + RS_RC_POS, p.rs.rc_pos); cold_st3(r + RS_RC_NORMAL, p.rs.rc_normal);
        cold_st3(r + RS_RC_NEE_DIR, p.rs.rc_nee_dir);
        r[RS_RC_MAT] = p.rs.rc_mat_info; r[RS_RC_LOBE] = (uint32_t)p.rs.rc_lobe;
    } else {
        cold_st3(r + RS_THR, p.rs.thr_after_rc); cold_st3(r + RS_RC_INC_L, p.rs.rc_incident_L);
        if (depth == 2) cold_st3(r + RS_RC_INC_DIR, p.rs.rc_incident_dir);
    }
}
// the path ends at `depth` (> 0): everything earlier vertices produced, for path_finish / restir_finish
VRT_DEV void restir_cold_load_rest(const uint32_t* line, Path<true>& p, int depth) {
    const uint32_t* r = line + PC_RS;
    p.nee_d = cold_ld3(line + PC_NEE_D); p.nee_s = cold_ld3(line + PC_NEE_S);
    p.primary_albedo = cold_ld3(line + PC_ALBEDO); p.primary_pos = cold_ld3(line + PC_PPOS);
    p.first_invpdf = dm_u2f(line[PC_INVPDF]);
    p.primary_mat_info = line[PC_MAT];
    p.rs.first_light_dir = cold_ld3(r + RS_FIRST_LIGHT_DIR);
    p.rs.first_light_bsdf_pdf = dm_u2f(r[RS_FIRST_LIGHT_PDF]);
    if (depth >= 2) {
        p.rs.first_dir = cold_ld3(r + RS_FIRST_DIR); p.rs.rc_pos = cold_ld3(r + RS_RC_POS); p.rs.rc_normal = cold_ld3(r + RS_RC_NORMAL);
        p.rs.rc_nee_dir = cold_ld3(r + RS_RC_NEE_DIR);
        p.rs.rc_mat_info = r[RS_RC_MAT]; p.rs.rc_lobe = (int)r[RS_RC_LOBE];
        if (depth >= 3) p.rs.rc_incident_dir = cold_ld3(r + RS_RC_INC_DIR);
    }
}

// What becomes of a closest-hit ray that ended at distance t (voxel units) given the floor distance of its path
// (the comparison of hit_voxel(), pathtracer.py:203-204).
template <int G>
VRT_DEV int slot_state_after_walk(float t, float floor_t) {
    return (t * GridDim<G>::voxel_size < floor_t || floor_t < DM_INF) ? SLOT_SHADE : SLOT_ESCAPE;
}

// Set up the closest-hit ray of the path in `s` (its pos and dir are stored already): floor distance and the
// prepared walk.  A ray that misses the grid box has nothing to walk and goes straight to SHADE / ESCAPE.
template <int G, bool CULL = true>
VRT_DEV int pool_launch_ray(const FrameParams& fp, const float* cull, const SlotRef& s, f3 pos, f3 d, TraceStats& ts) {
    const float ft = floor_probe(fp, pos, d);
    s.sf(PF_FLOOR_T, ft);
    RayWalk w;
    const bool alive = walk_prepare<G, CULL>(world_to_voxel<G>(pos), d, w, cull);
    ts.rays += 1u;
    walk_store(s, w);
    walk_store_constants(s, w);
    return alive ? SLOT_RAY : slot_state_after_walk<G>(w.t, ft);
}

// The samples fused into one launch share camera and jitter, so a pixel's camera ray -- three quarters of all DDA
// steps on the sparse scene -- and its whole closest-hit record are the same for all of them.  The path of sample 0
// leaves the record (distance, cell, normal code; tagged with the launch) in a per-pixel table when its walk ends -- or in
// BEGIN, when there was nothing to walk; work items are ordered sample-major within a work range, so by the time a pixel's
// later samples begin, the record is there and they start at SHADE / ESCAPE without setting up or walking a ray.  The
// record also carries what every sample of the pixel would work out again before it could use the hit: the ray's
// direction (camera_ray_dir: a matrix, a perspective divide and a normalisation -- 15 correctly rounded divisions and a
// square root) and its floor distance (floor_probe), as the bits sample 0 computed.  A missing or stale record just
// means the ray is set up and walked as usual.  Writer and reader are different CUs with no fence between them (a release /
// acquire pair costs microseconds, MI355X_MICROARCH.md: this is per ray), so nothing is ASSUMED about how the 32 bytes
// (two 16-byte stores, two 16-byte loads) arrive: the last word is the launch tag mixed with a hash of the seven payload
// words, and a reader accepts a record only if that word matches the payload it read and the launch it runs in.  A stale
// record (other launch), a missing one and a torn one -- words of two different stores, or halves of them -- all fail the
// check (a tear passes with probability 2^-32) and the ray is walked.
struct alignas(16) PrimaryRecord { uint32_t x, y, z, dx, dy, dz, ft, w; };  // PF_T, PF_CELL_XY, PF_CELL_Z, PF_DIR[3], PF_FLOOR_T, check word
VRT_DEV uint32_t primary_check(const PrimaryRecord& r, uint32_t tag) {
    uint32_t h = r.x * 0x9E3779B1u;
    h = (h ^ (h >> 15)) + r.y * 0x85EBCA77u;
    h = (h ^ (h >> 13)) + r.z * 0xC2B2AE3Du;
    h = (h ^ (h >> 16)) + r.dx * 0x27D4EB2Fu;
    h = (h ^ (h >> 15)) + r.dy * 0x165667B1u;
    h = (h ^ (h >> 13)) + r.dz * 0x9E3779B1u;
    h = (h ^ (h >> 16)) + r.ft * 0x85EBCA77u;
    h ^= h >> 15;
    return h ^ tag;
}
// the record of the camera ray whose walk (or set-up, when there was nothing to walk) has just been stored in `s`
VRT_DEV PrimaryRecord primary_record(const SlotRef& s, uint32_t tag) {
    PrimaryRecord r;
    r.x = s.u(PF_T); r.y = s.u(PF_CELL_XY); r.z = s.u(PF_CELL_Z);
    r.dx = s.u(PF_DIR); r.dy = s.u(PF_DIR + 1); r.dz = s.u(PF_DIR + 2); r.ft = s.u(PF_FLOOR_T);
    r.w = primary_check(r, tag);
    return r;
}
VRT_DEV bool primary_record_valid(const PrimaryRecord& r, uint32_t tag) { return r.w == primary_check(r, tag); }
#if defined(__HIP_DEVICE_COMPILE__)
typedef unsigned int vrt_u32x4 __attribute__((ext_vector_type(4)));
VRT_DEV void primary_record_store(PrimaryRecord* at, const PrimaryRecord& r) {  // two 16-byte stores
    const vrt_u32x4 a = {r.x, r.y, r.z, r.dx}, b = {r.dy, r.dz, r.ft, r.w};
    ((vrt_u32x4*)at)[0] = a; ((vrt_u32x4*)at)[1] = b;
}
VRT_DEV PrimaryRecord primary_record_load(const PrimaryRecord* at) {  // two 16-byte loads past L1: another CU wrote it
    const vrt_u32x4 a = __builtin_nontemporal_load((const vrt_u32x4*)at), b = __builtin_nontemporal_load((const vrt_u32x4*)at + 1);
    PrimaryRecord r;
    r.x = a.x; r.y = a.y; r.z = a.z; r.dx = a.w; r.dy = b.x; r.dz = b.y; r.ft = b.z; r.w = b.w;
    return r;
}
#else   // host builds of this header (tests/emul)
VRT_DEV void primary_record_store(PrimaryRecord* at, const PrimaryRecord& r) { *at = r; }
VRT_DEV PrimaryRecord primary_record_load(const PrimaryRecord* at) { return *at; }
#endif
template <int G>
VRT_DEV int pool_begin_known(const FrameParams& fp, const SlotRef& s, int u, int v, int sample, const PrimaryRecord& rec) {
    Path<false> p;
    path_begin_along(fp, p, u, v, sample, mk3(dm_u2f(rec.dx), dm_u2f(rec.dy), dm_u2f(rec.dz)));
    path_store_hot(s, p);
    s.su(PF_FLOOR_T, rec.ft);
    s.su(PF_T, rec.x); s.su(PF_CELL_XY, rec.y); s.su(PF_CELL_Z, rec.z); s.su(PF_ITERS, 0u);
    return slot_state_after_walk<G>(dm_u2f(rec.x), dm_u2f(rec.ft));
}

// BEGIN: work item (u, v, sample) -> camera ray pending.
template <int G, bool CULL = true>
VRT_DEV int pool_begin(const FrameParams& fp, const float* cull, const SlotRef& s, int u, int v, int sample, TraceStats& ts) {
    Path<false> p;
    path_begin(fp, p, u, v, sample);
    path_store_hot(s, p);
    return pool_launch_ray<G, CULL>(fp, cull, s, p.pos, p.d, ts);
}

// SHADE (KIND = HIT_SOMETHING) and ESCAPE (KIND = HIT_NOTHING): rebuild the closest hit from the slot, run the
// segment, then either set up the bounce ray or finish the path.  Returns the slot's next state.
template <int KIND, bool BLACK_SUN, bool RESTIR = false, class PyrT>
VRT_DEV int pool_shade(const FrameParams& fp, const SceneData& sc, const PyrT& P, const PixelBuffers& out, const SlotRef& s,
                       uint32_t* cold_line, TraceStats& ts) {
    Path<RESTIR> p;
    path_load_hot(s, p);
    const int local_idx = (p.pix_v - fp.row0) * fp.W + p.pix_u;
    const int depth = p.depth;
    Hit h;
    hit_init(h);
    if (KIND == HIT_SOMETHING) {
        const float ft = s.f(PF_FLOOR_T);
        if (ft < DM_INF) hit_floor(fp, p.d, ft, h);
        const uint32_t a = s.u(PF_CELL_XY), b = s.u(PF_CELL_Z);
        TraceOut tr;
        walk_result(p.d, s.f(PF_T), (int)(int16_t)(a & 0xffffu), (int)(int16_t)(a >> 16), (int)(int16_t)(b & 0xffffu),
                    normal_decode(b >> 20), (int)s.u(PF_ITERS), tr);
        hit_voxel<false, PyrT::G>(fp, sc, world_to_voxel<PyrT::G>(p.pos), p.d, tr, h, ts);
    }
    if constexpr (RESTIR) restir_cold_load_running(cold_line, p, depth);
    const bool done = path_shade<RESTIR, KIND, BLACK_SUN>(fp, sc, P, out, local_idx, p, h, ts);
    if (done) {
        if (depth > 0) {
            if constexpr (RESTIR) restir_cold_load_rest(cold_line, p, depth);
            else path_load_cold(cold_line, p);
        }
        path_finish<RESTIR>(fp, sc, out, local_idx, p, ts);
        return SLOT_EMPTY;
    }
    if constexpr (RESTIR) restir_cold_store(cold_line, p, depth);
    else { if (depth == 0) path_store_cold(cold_line, p); }
    path_store_hot(s, p);
    return pool_launch_ray<PyrT::G, PyrT::cull>(fp, sc.cull, s, p.pos, p.d, ts);
}

}  // namespace vrt
#endif
