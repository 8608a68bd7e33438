// vrt_restir.h -- ReSTIR-PT: input reservoir construction, reconnection shift, spatial reuse.
//
// Replaces Sample / Reservoir / StorageReservoir (reference renderer/reservoir.py:8-141), the
// reservoir part of Renderer.render (pathtracer.py:549-607, 620-626), Renderer.shift (672-812)
// and Renderer.spatial_GRIS (815-989).  Only compiled into the RESTIR instantiation of the render
// kernel and into the spatial-reuse kernel; with ReSTIR off (the reference default,
// pathtracer.py:15) no reservoir is stored at all -- nothing reads it.
//
// Reference quirks kept: self.current_frame inside spatial_GRIS is the
// value baked at first compile (0); angle_shift value-casts the u32 bit pattern; the jacobian
// rejection test sits in the `jacobian = 0` branch.  Taps outside the image are skipped (the
// reference reads out of bounds there).
#ifndef VRT_RESTIR_H
#define VRT_RESTIR_H

#include "vrt_path.h"

namespace vrt {

struct RSample {
    f3 F, rc_pos, rc_normal, rc_incident_dir, rc_incident_L, rc_nee_dir;
    uint32_t rc_mat_info;
    float jac;
    int lobes;
};
struct Reservoir {
    RSample z;
    float M, weight;
};

VRT_DEV void reservoir_init(Reservoir& r) {  // reservoir.py:46-57
    r.z.F = mk3(0.0f); r.z.rc_pos = mk3(0.0f); r.z.rc_normal = mk3(0.0f); r.z.rc_incident_dir = mk3(0.0f);
    r.z.rc_incident_L = mk3(0.0f); r.z.rc_nee_dir = mk3(0.0f);
    r.z.rc_mat_info = 0u; r.z.jac = 1.0f; r.z.lobes = 0;
    r.M = 0.0f; r.weight = 0.0f;
}
VRT_DEV void reservoir_update_jacobian(Reservoir& r, f3 x1) {  // :59-62
    f3 d = r.z.rc_pos - x1;
    r.z.jac = dot3(d, d) / dm_abs(dot3(norm3(d), r.z.rc_normal));
}
// :64-74 (input_sample, in_M = 1) and :76-86 (merge, in_M = in_r.M)
VRT_DEV bool reservoir_add(Reservoir& r, const RSample& z, float in_M, float in_w, dm_rng& rng, bool force_add) {
    r.M += in_M;
    bool selected = false;
    if (in_w > 0.0f) {
        r.weight += in_w;
        bool lt = dm_rng_f32(&rng) * r.weight <= in_w;
        selected = lt || force_add;
        if (selected) r.z = z;
    }
    return selected;
}
VRT_DEV void reservoir_finalize_without_M(Reservoir& r) {  // :96-102
    float p_hat = lum(r.z.F);
    r.weight = (p_hat < 1e-6f) ? 0.0f : r.weight / p_hat;
}
VRT_DEV uint32_t pack4x8(float a, float b, float c, float d) {  // math_utils.py:250-255, sizes (8,8,8,8)
    return dm_f2u32(a * 255.0f + 0.5f) | (dm_f2u32(b * 255.0f + 0.5f) << 8) | (dm_f2u32(c * 255.0f + 0.5f) << 16) |
           (dm_f2u32(d * 255.0f + 0.5f) << 24);
}
VRT_DEV ReservoirRec reservoir_encode(const Reservoir& r) {  // reservoir.py:104-124
    ReservoirRec e;
    e.F = r.z.F;
    e.rc_pos = r.z.rc_pos;
    e.rc_incident_L = r.z.rc_incident_L;
    uint32_t on = oct_encode(r.z.rc_normal), od = oct_encode(r.z.rc_nee_dir);
    e.rc_normal_and_nee = pack4x8(dm_f16_to_f32((uint16_t)(on & 0xffffu)), dm_f16_to_f32((uint16_t)(on >> 16)),
                                  dm_f16_to_f32((uint16_t)(od & 0xffffu)), dm_f16_to_f32((uint16_t)(od >> 16)));
    e.rc_incident_dir = oct_encode(r.z.rc_incident_dir);
    e.rc_mat_info = r.z.rc_mat_info;
    e.M_W = (uint32_t)dm_f32_to_f16(r.M) | ((uint32_t)dm_f32_to_f16(r.weight) << 16);
    e.jac_lobes = (uint32_t)dm_f32_to_f16(r.z.jac) | (((uint32_t)(int8_t)r.z.lobes & 0xffu) << 16);
    e.pad[0] = 0u; e.pad[1] = 0u;
    return e;
}
VRT_DEV void reservoir_decode(Reservoir& r, const ReservoirRec& e) {  // reservoir.py:126-141
    r.M = dm_f16_to_f32((uint16_t)(e.M_W & 0xffffu));
    r.weight = dm_f16_to_f32((uint16_t)(e.M_W >> 16));
    r.z.F = e.F;
    r.z.rc_pos = e.rc_pos;
    uint32_t q = e.rc_normal_and_nee;
    r.z.rc_normal = oct_decode_f((float)(q & 255u) / 255.0f, (float)((q >> 8) & 255u) / 255.0f);
    r.z.rc_nee_dir = oct_decode_f((float)((q >> 16) & 255u) / 255.0f, (float)((q >> 24) & 255u) / 255.0f);
    r.z.rc_incident_dir = oct_decode(e.rc_incident_dir);
    r.z.rc_incident_L = e.rc_incident_L;
    r.z.rc_mat_info = e.rc_mat_info;
    r.z.jac = dm_f16_to_f32((uint16_t)(e.jac_lobes & 0xffffu));
    r.z.lobes = (int)(int8_t)((e.jac_lobes >> 16) & 0xffu);
}

// pathtracer.py:549-607 and 620-626 (the USE_RESTIR_PT = True arm)
template <class PathT>
VRT_DEV void restir_finish(const FrameParams& fp, const SceneData& sc, const PixelBuffers& out, int local_idx, PathT& p,
                           f3 primary_pos, f3& diffuse, f3& specular, TraceStats& ts) {
    Reservoir r;
    reservoir_init(r);
    r.z.rc_pos = p.rs.rc_pos; r.z.rc_normal = p.rs.rc_normal; r.z.rc_incident_dir = p.rs.rc_incident_dir;
    r.z.rc_incident_L = p.rs.rc_incident_L; r.z.rc_nee_dir = p.rs.rc_nee_dir; r.z.rc_mat_info = p.rs.rc_mat_info;
    r.z.F = p.contrib;
    r.z.lobes = p.rs.rc_lobe * 10 + p.first_lobe;
    r.M = 1.0f;
    reservoir_update_jacobian(r, primary_pos);
    bool chose_nee = false;
    if (!p.sky_primary) {
        float bsdf_pdf = 1.0f / p.first_invpdf;
        float light_pdf = cone_pdf(fp.light_cos_max, dot3(fp.light_dir, p.rs.first_dir));
        if (near_zero3(p.nee_d + p.nee_s)) light_pdf = 0.0f;
        float bsdf_mis = power_heuristic(bsdf_pdf, light_pdf);
        float light_mis = power_heuristic(cone_pdf(fp.light_cos_max, 1.0f), p.rs.first_light_bsdf_pdf);
        float p_hat = lum(r.z.F);
        r.weight = bsdf_mis * p_hat * p.first_invpdf;
        float light_w = light_mis * lum(p.nee_d + p.nee_s);
        f3 sky_t = mk3(1.0f);
        if (fp.use_sky == 1) { sky_t = sky_transmittance(sc.sky, p.rs.first_light_dir); ts.sky_lookups += 1u; }
        RSample ls;
        ls.F = p.nee_d + p.nee_s;
        ls.rc_pos = p.rs.first_light_dir;
        ls.rc_normal = mk3(0.0f); ls.rc_incident_dir = mk3(0.0f);
        ls.rc_incident_L = sky_t * fp.light_weight * fp.light_color;
        ls.rc_nee_dir = mk3(0.0f);
        ls.rc_mat_info = 0u; ls.jac = 1.0f; ls.lobes = LOBE_ALL * 10 + LOBE_ALL;
        chose_nee = reservoir_add(r, ls, 1.0f, light_w, p.rng, false);
        reservoir_finalize_without_M(r);
    } else {
        r.weight = 1.0f;
    }
    out.reservoir[local_idx + p.sample * out.sample_stride] = reservoir_encode(r);   // (one plane per fused sample, like the colours)
    if (!chose_nee) {
        diffuse = diffuse + ((p.first_lobe == LOBE_DIFFUSE) ? r.z.F : mk3(0.0f));
        specular = specular + ((p.first_lobe == LOBE_SPEC) ? r.z.F : mk3(0.0f));
    } else {
        diffuse = diffuse + p.nee_d;
        specular = specular + p.nee_s;
    }
}

VRT_DEV Material material_from_bits(const float* mats, uint32_t enc, int& id) {  // math_utils.py:238-247
    id = (int)(enc & 255u);
    Material m = load_material(mats, id);
    m.base = unpack_albedo(enc);
    return m;
}

VRT_DEV MatDerived load_mat_derived(const float* mats_x, int id) {
    const float* p = mats_x + 8 * (id & 127);
    MatDerived x;
    x.ax = p[0]; x.ay = p[1]; x.cc_alpha = p[2]; x.w_d = p[3]; x.w_s = p[4]; x.w_c = p[5];
    return x;
}
VRT_DEV void store_mat_derived(float* mats_x, int id, const MatDerived& x, float unit_range = 0.0f) {
    float* p = mats_x + 8 * id;
    p[0] = x.ax; p[1] = x.ay; p[2] = x.cc_alpha; p[3] = x.w_d; p[4] = x.w_s; p[5] = x.w_c; p[6] = unit_range; p[7] = 0.0f;
}
// 1: every parameter of the material row that the BSDF evaluation reads lies in [0, 1] (true of the whole default CSV).  What
// shift_is_constant() below needs to BOUND a BSDF value without evaluating it; rows outside are simply always evaluated.
VRT_DEV float material_unit_range(const Material& m) {
    const float q[10] = {m.subsurface, m.metallic, m.specular, m.specular_tint, m.roughness, m.anisotropic, m.sheen, m.sheen_tint, m.clearcoat,
                         m.clearcoat_gloss};
    bool ok = true;
    for (int k = 0; k < 10; k++) ok = ok && q[k] >= 0.0f && q[k] <= 1.0f;
    return ok ? 1.0f : 0.0f;
}

// pathtracer.py:672-812: shift `src`'s sample to the primary vertex at dst_pos whose shading frame is `ds`
// (normal, material, view direction toward the camera -- built by the caller so the centre pixel's frame, which is
// the destination of every neighbour's shift, is set up once per pixel instead of once per tap)
// `rc_ty` is the bitangent of the sample's reconnection vertex (ortho_basis of rc_normal, which does not depend on the
// destination), `src_sky_t` the sky transmittance toward its sun sample (GrisSrc), `mats_x` the per-material-id table of
// mat_derive().
// `dsc` = surf_shared(ds, ...) with at least the groups of lobe src.z.lobes % 10.
// `pre` = what the sample's reconnection vertex contributes whatever the destination (RcPre, from the prepare pass).
struct RcPre {
    f3 base;             // unpack_albedo(rc_mat_info)
    MatColours col;      // mat_colours() of the reconnection vertex' material
    DirTerms inc, nee;   // dir_terms() of rc_incident_dir and rc_nee_dir at the reconnection vertex
};
// The Jacobian shift_sample() returns (out_jac), alone: :681-688 and :789-803 need the two primary vertices and three fields of the sample.
VRT_DEV float shift_jacobian(f3 dst_pos, f3 dst_normal, f3 rc_pos, f3 rc_normal, float cached_jac, float* dst_nl = nullptr) {
    const bool escape = near_zero3(rc_normal);
    const f3 to_rc = escape ? rc_pos : norm3(rc_pos - dst_pos);
    float passed = 1.0f;
    if (dst_nl) *dst_nl = dot3(dst_normal, to_rc);   // the destination's n.l as bsdf_eval_pdf() will compute it
    if (dot3(dst_normal, to_rc) < 1e-5f || (!escape && dot3(rc_normal, -to_rc) < 1e-5f)) passed = 0.0f;
    float jac = 1.0f;
    if (!escape) {
        f3 dv = rc_pos - dst_pos;
        jac = cached_jac;
        jac *= dm_abs(dot3(norm3(dv), rc_normal)) / dot3(dv, dv);
    }
    if (jac < 0.0f || dm_isnan(jac) || dm_isinf(jac)) jac = 0.0f;
    return jac * passed;
}
// Is the CENTRE -> NEIGHBOUR shift of the first tap loop (:917-931) a known constant, so that it need not be evaluated?
// Its term is 1 - cw, cw = (L j M_n) / (L j M_n + X), with j the shift's Jacobian, L = lum(BSDF_dst cos contrib), M_n the
// neighbour's M and X = lum(F_c) M_c / max_taps a value of the centre alone.  With j = 0 and L and M_n FINITE, L j M_n is a
// zero and the term is 1 - 0 / X whatever the shift computes: a constant of the pixel.  j is shift_jacobian() (the same
// expressions as in shift_sample, so the same bits); M_n is in the record; L is finite when
//   * the destination's n.l is not > 0 -- or its n.v is not: eval_lobes' `front` test fails and the BSDF is an exact zero --, or
//   * n.l >= 1e-5 and n.v >= 1e-5 and the destination's material row lies in [0, 1] (material_unit_range): then every factor of
//     the evaluation is bounded -- |l + v| >= n.l + n.v >= 2e-5; the anisotropic GGX term <= 1 / (pi 1e-6 (1 / 10.3)^2) < 4e7
//     (alphas in [1e-3, 3.2]); each Smith term <= 1 / (2 * 1e-5); gtr1 < 4e4 (its t >= alpha^2 - 3e-7 > 0 for alpha >= 1e-3);
//     Fresnel and colour factors <= 1; 1 / (n.l + n.v) <= 5e4 -- so |BSDF| < 1e18, times cos <= 1, times contrib, a sum of
//     firefly-clamped terms and an albedo within [0, 601]: finite.
// Everything else (a grazing 0 < n.l < 1e-5, a material row outside [0, 1], a non-finite M) is evaluated as before.  `dst_ok` is
// the destination's share of the test, worked out once per pixel by the prepare pass: !(n.v > 0) || (n.v >= 1e-5 && unit range).
VRT_DEV bool shift_is_constant(f3 dst_pos, f3 dst_normal, bool dst_ok, float dst_M, f3 rc_pos, f3 rc_normal, float cached_jac) {
    float nl;
    if (shift_jacobian(dst_pos, dst_normal, rc_pos, rc_normal, cached_jac, &nl) != 0.0f) return false;
    if (!(dst_M >= 0.0f && dst_M < DM_INF)) return false;
    return !(nl > 0.0f) || (nl >= 1e-5f && dst_ok);
}
VRT_DEV void shift_sample(const FrameParams& fp, const SceneData& sc, const float* mats_x, f3 dst_pos, const Surf& ds, const SurfShared& dsc,
                          const Reservoir& src, f3 rc_ty, f3 src_sky_t, const RcPre& pre, f3& out_d, f3& out_s, float& out_jac, TraceStats& ts,
                          int dg = 0 /* first diagnostic region of the caller (VRT_GREGION; nothing in the shipped library) */) {
    const bool escape = near_zero3(src.z.rc_normal);
    const bool last = near_zero3(src.z.rc_incident_dir);
    const bool nee_vis = !near_zero3(src.z.rc_nee_dir);
    const f3 to_rc = escape ? src.z.rc_pos : norm3(src.z.rc_pos - dst_pos);
    float passed = 1.0f;
    const f3 dst_normal = ds.n;
    if (dot3(dst_normal, to_rc) < 1e-5f || (!escape && dot3(src.z.rc_normal, -to_rc) < 1e-5f)) passed = 0.0f;

    const int rc_id = (int)(src.z.rc_mat_info & 255u);
    Material rc_mat = load_material(sc.mats, rc_id);   // material_from_bits() with the albedo unpacked once per sample
    rc_mat.base = pre.base;
    f3 contrib = mk3(0.0f);
    if (!escape) {
        VRT_GREGION(dg + 1);
        Surf rc;
        surf_set(rc, rc_mat, load_mat_derived(mats_x, rc_id), src.z.rc_normal, -to_rc, cross3(src.z.rc_normal, rc_ty), rc_ty);
        // two directions (the path's continuation and its sun sample) with their pdfs at one vertex: bsdf_eval_pdf
        const int rl = src.z.lobes / 10;
        const SurfShared rcc = surf_shared_view(rc, pre.col, nee_vis || (!last && lobe_has(rl, LOBE_DIFFUSE)), nee_vis || (!last && lobe_has(rl, LOBE_SPEC)),
                                                nee_vis || (!last && lobe_has(rl, LOBE_CLEARCOAT)));
        if (!last) {
            VRT_GREGION(dg + 2);
            f3 bd, bs;
            float dst_rc_pdf;
            bsdf_eval_pdf_pre(rc, rcc, src.z.rc_incident_dir, pre.inc, rl, PDF_LOBE, bd, bs, dst_rc_pdf);
            f3 rc_brdf = (bd + bs) * dm_saturate(dot3(src.z.rc_normal, src.z.rc_incident_dir));
            float lp = cone_pdf(fp.light_cos_max, dot3(fp.light_dir, src.z.rc_incident_dir));
            float w = power_heuristic(dst_rc_pdf, lp * (nee_vis ? 1.0f : 0.0f));
            contrib = contrib + firefly(w * rc_brdf / dst_rc_pdf * src.z.rc_incident_L);
        }
        if (nee_vis) {
            VRT_GREGION(dg + 3);
            f3 bd, bs;
            float pdf_nee;
            bsdf_eval_pdf_pre(rc, rcc, src.z.rc_nee_dir, pre.nee, LOBE_ALL, PDF_ALL, bd, bs, pdf_nee);
            f3 nee_brdf = (bd + bs) * dm_saturate(dot3(src.z.rc_normal, src.z.rc_nee_dir));
            float w = power_heuristic(cone_pdf(fp.light_cos_max, 1.0f), pdf_nee);
            f3 sky_t = mk3(1.0f);
            if (fp.use_sky == 1) { sky_t = src_sky_t; ts.sky_lookups += 1u; }  // counted where the reference looks it up
            contrib = contrib + firefly(w * nee_brdf * sky_t * fp.light_weight * fp.light_color);
        }
    } else {
        contrib = contrib + firefly(src.z.rc_incident_L);
    }
    contrib = contrib + ((rc_id != 2) ? mk3(0.0f) : rc_mat.base);

    f3 pd, ps;
    float no_pdf;
    if (dot3(ds.n, to_rc) > 0.0f && ds.n_v > 0.0f) {
        VRT_GREGION(dg + 4);                                            // (diagnostic only: the destination's BSDF is not an exact zero ...
        if (lobe_has(src.z.lobes % 10, LOBE_DIFFUSE)) { VRT_GREGION(dg + 5); }   // ... and which of its lobes are evaluated)
        if (lobe_has(src.z.lobes % 10, LOBE_SPEC)) { VRT_GREGION(dg + 6); }
    }
    bsdf_eval_pdf(ds, dsc, to_rc, src.z.lobes % 10, PDF_NONE, pd, ps, no_pdf);
    const float c = dm_saturate(dot3(dst_normal, to_rc));
    pd = pd * c;
    ps = ps * c;
    out_d = pd * contrib;
    out_s = ps * contrib;

    float jac = 1.0f;
    if (!escape) {
        f3 dv = src.z.rc_pos - dst_pos;
        jac = src.z.jac;
        jac *= dm_abs(dot3(norm3(dv), src.z.rc_normal)) / dot3(dv, dv);
    }
    if (jac < 0.0f || dm_isnan(jac) || dm_isinf(jac)) {
        jac = 0.0f;
        if (dm_max(jac, 1.0f / jac) > 11.0f) { out_d = mk3(0.0f); out_s = mk3(0.0f); }
    }
    out_jac = jac * passed;
}

VRT_DEV uint32_t hash3(uint32_t x, uint32_t y, uint32_t z) {  // math_utils.py:217-229
    x += x >> 11; x ^= x << 7; x += y; x ^= x << 3; x += z ^ (x >> 14);
    x ^= x << 6; x += x >> 15; x ^= x << 5; x += x >> 12; x ^= x << 9;
    return x;
}

// What the spatial-reuse kernel needs of a pixel each of the ~32 times it is somebody's neighbour, worked out once per
// pixel by gris_prepare_pixel() instead (same expressions, so the same bits): the primary vertex as the destination of a
// shift (GrisGeo, pathtracer.py:896-912 + 731-732) and the decoded reservoir as the source of one (GrisSrc,
// reservoir.py:126-141).  16-byte aligned so that a tap is four / eight 128-bit loads.
struct alignas(16) GrisGeo {
    f3 n; float dist;        // g-buffer normal; distance of x1 from the camera  (the two the acceptance test reads)
    f3 x1; uint32_t mat;     // primary vertex from the g-buffer depth; packed material
    f3 v; float M;           // unit vector toward the camera; the pixel's reservoir M
    f3 ty; uint32_t pad;     // bitangent of ortho_basis(n); the tangent is cross(n, ty)
    // the pixel as the DESTINATION of a shift: its shading point never changes (own normal, material, view vector), so the
    // whole SurfShared of it and the unpacked albedo are worked out here
    f3 base; float fv;
    f3 lambert; float g_v;
    f3 sheen_col; float gc_v;
    f3 spec_col; uint32_t pad3;
};
struct alignas(16) GrisSrc {
    f3 F; float M;
    f3 rc_pos; float weight;
    f3 rc_normal; float jac;
    f3 rc_incident_dir; int lobes;
    f3 rc_incident_L; uint32_t rc_mat_info;
    f3 rc_nee_dir; uint32_t pad0;
    f3 rc_ty; uint32_t pad1;  // bitangent of ortho_basis(rc_normal)
    f3 sky_t; uint32_t pad2;  // sky transmittance toward rc_nee_dir (atmos.py:117-131): depends on the sample alone
    // RcPre: the reconnection vertex' material colours and the terms of its two fixed light directions
    f3 base; float inc_nl;
    f3 lambert; float inc_fl;
    f3 sheen_col; float inc_g;
    f3 spec_col; float inc_gc;
    float nee_nl, nee_fl, nee_g, nee_gc;
    float inc_pd, nee_pd; uint32_t pad4, pad5;
};

// What the classify kernel reads of a pixel each time it is somebody's tap: 48 bytes of its own, because read out of the two records
// above (16 of 128 bytes, 32 of 224) every tap drags two whole cache lines through an L2 that the tap windows of the waves in flight
// then do not fit (measured: 2.5 ms for the kernel; with this record the windows are a few MB per XCD).
struct alignas(16) GrisTest {
    f3 n; float dist;        // = GrisGeo: the geometric test (:912)
    f3 rc_pos; float jac;    // = GrisSrc: what shift_jacobian() needs of the sample
    f3 rc_normal; float M;
    f3 x1; uint32_t dst_ok;  // = GrisGeo: the pixel as the destination of the centre's sample (shift_is_constant)
};
struct GrisBuffers {
    GrisGeo* geo;            // [rows of the launch][W], written by the prepare pass
    GrisSrc* src;
    GrisTest* tst;
    const float* mats_x;     // [128][8]: mat_derive() of every material id
    const f3* color_d_in;
    const f3* color_s_in;
    f3* color_d_out;
    f3* color_s_out;
    const uint32_t* gb_normal;
    const float* gb_depth;
    const uint32_t* gb_mat;
    const ReservoirRec* res_in;
    ReservoirRec* res_out;
};

// once per pixel of every row the launch holds (the rows it renders and their halo)
VRT_DEV void gris_prepare_pixel(const FrameParams& fp, const SceneData& sc, const GrisBuffers& gb, int u, int v) {
    const int idx = (v - fp.row0) * fp.W + u;
    GrisGeo g;
    g.n = oct_decode(gb.gb_normal[idx]);
    g.x1 = xform(fp.view_inv, screen_to_view(pixel_texcoord(fp, (float)u, (float)v), gb.gb_depth[idx], fp.proj_inv), 1.0f);
    g.dist = len3(g.x1 - fp.camera_pos);
    g.mat = gb.gb_mat[idx];
    g.v = norm3(fp.camera_pos - g.x1);
    Reservoir r;
    reservoir_init(r);
    reservoir_decode(r, gb.res_in[idx]);
    g.M = r.M;
    f3 tx;
    ortho_basis(g.n, tx, g.ty);
    g.pad = 0u;
    bool dst_ok;
    {   // the pixel's own shading point as every neighbour's shift will set it up (gris_pixel, first tap loop)
        int id;
        const Material m = material_from_bits(sc.mats, g.mat, id);
        Surf ds;
        surf_set(ds, m, load_mat_derived(gb.mats_x, id), g.n, g.v, cross3(g.n, g.ty), g.ty);
        dst_ok = !(ds.n_v > 0.0f) || (ds.n_v >= 1e-5f && gb.mats_x[8 * (id & 127) + 6] != 0.0f);   // shift_is_constant()
        const SurfShared c = surf_shared(ds, true, true, true);
        g.base = m.base; g.fv = c.fv; g.lambert = c.lambert; g.g_v = c.g_v; g.sheen_col = c.sheen_col; g.gc_v = c.gc_v;
        g.spec_col = c.spec_col; g.pad3 = 0u;
    }
    gb.geo[idx] = g;
    GrisSrc s;
    s.F = r.z.F; s.M = r.M; s.rc_pos = r.z.rc_pos; s.weight = r.weight; s.rc_normal = r.z.rc_normal; s.jac = r.z.jac;
    s.rc_incident_dir = r.z.rc_incident_dir; s.lobes = r.z.lobes; s.rc_incident_L = r.z.rc_incident_L;
    s.rc_mat_info = r.z.rc_mat_info; s.rc_nee_dir = r.z.rc_nee_dir; s.pad0 = 0u; s.pad1 = 0u;
    ortho_basis(r.z.rc_normal, tx, s.rc_ty);
    // every shift of this sample that sees its sun sample weighs it by the sky's transmittance toward the sun sample
    // (pathtracer.py:770-772): one table lookup here instead of one per shift (~60 per pixel)
    s.sky_t = (fp.use_sky == 1 && !near_zero3(r.z.rc_nee_dir)) ? sky_transmittance(sc.sky, r.z.rc_nee_dir) : mk3(1.0f);
    s.pad2 = 0u;
    {   // the reconnection vertex as shift_sample() sets it up, minus the view vector
        int id;
        const Material m = material_from_bits(sc.mats, r.z.rc_mat_info, id);
        const MatDerived x = load_mat_derived(gb.mats_x, id);
        const MatColours mc = mat_colours(m);
        const f3 rtx = cross3(r.z.rc_normal, s.rc_ty);
        const DirTerms a = dir_terms(r.z.rc_normal, rtx, s.rc_ty, x.ax, x.ay, r.z.rc_incident_dir);
        const DirTerms b = dir_terms(r.z.rc_normal, rtx, s.rc_ty, x.ax, x.ay, r.z.rc_nee_dir);
        s.base = m.base; s.lambert = mc.lambert; s.sheen_col = mc.sheen_col; s.spec_col = mc.spec_col;
        s.inc_nl = a.nl; s.inc_fl = a.fl; s.inc_g = a.g_l; s.inc_gc = a.gc_l; s.inc_pd = a.pd;
        s.nee_nl = b.nl; s.nee_fl = b.fl; s.nee_g = b.g_l; s.nee_gc = b.gc_l; s.nee_pd = b.pd;
        s.pad4 = 0u; s.pad5 = 0u;
    }
    gb.src[idx] = s;
    GrisTest t;
    t.n = g.n; t.dist = g.dist; t.rc_pos = s.rc_pos; t.jac = s.jac; t.rc_normal = s.rc_normal; t.M = g.M;
    t.x1 = g.x1; t.dst_ok = dst_ok ? 1u : 0u;
    gb.tst[idx] = t;
}
VRT_DEV void gris_load_src(Reservoir& r, f3& rc_ty, f3& sky_t, RcPre& pre, const GrisSrc& s) {
    r.z.F = s.F; r.M = s.M; r.z.rc_pos = s.rc_pos; r.weight = s.weight; r.z.rc_normal = s.rc_normal; r.z.jac = s.jac;
    r.z.rc_incident_dir = s.rc_incident_dir; r.z.lobes = s.lobes; r.z.rc_incident_L = s.rc_incident_L;
    r.z.rc_mat_info = s.rc_mat_info; r.z.rc_nee_dir = s.rc_nee_dir;
    rc_ty = s.rc_ty;
    sky_t = s.sky_t;
    pre.base = s.base; pre.col.lambert = s.lambert; pre.col.sheen_col = s.sheen_col; pre.col.spec_col = s.spec_col;
    pre.inc.nl = s.inc_nl; pre.inc.fl = s.inc_fl; pre.inc.g_l = s.inc_g; pre.inc.gc_l = s.inc_gc; pre.inc.pd = s.inc_pd;
    pre.nee.nl = s.nee_nl; pre.nee.fl = s.nee_fl; pre.nee.g_l = s.nee_g; pre.nee.gc_l = s.nee_gc; pre.nee.pd = s.nee_pd;
}

// pathtracer.py:876-891: the taps lie on a golden-angle spiral whose phase is hashed from the pixel's 8x8 tile (pass 0),
// so the angle, sine and cosine of tap i are the same for all 64 pixels of a tile: worked out once per tile into
// cs[2i] = cos, cs[2i+1] = sin.  Only the radius (one draw per pixel) differs inside a tile.
VRT_DEV float gris_angle_shift(int u, int v, int pass_id) {
    const uint32_t sx = (pass_id == 0) ? ((uint32_t)u >> 3) : 2u, sy = (pass_id == 0) ? ((uint32_t)v >> 3) : 2u;
    const uint32_t hs = hash3(sx, sy, 0u + (uint32_t)pass_id);
    return (float)((hs & 0x007FFFFFu) | 0x3F800000u) / 4294967295.0f * DM_PI;
}
VRT_DEV void gris_tap_cs(int u, int v, int pass_id, int i, float* cs) {
    const float golden = 2.399963229728f;
    float angle = ((float)i + gris_angle_shift(u, v, pass_id)) * golden;
    dm_sincos(angle, &cs[2 * i + 1], &cs[2 * i]);
}
// Scratch of one pixel's tap loop: the tile's table above, and one 16-bit slot per tap where the first loop leaves the
// tap's pixel offset for the two expensive loops (slot of tap i at off[i * off_stride]).
struct GrisTaps {
    const float* cs;
    uint16_t* off;
    int off_stride;
};
// pathtracer.py:883-906: offset of tap i; false for the centre itself and for taps outside the image (the reference
// reads out of bounds there)
VRT_DEV bool gris_tap(const FrameParams& fp, const GrisTaps& taps, int u, int v, int i, float radius_shift, float max_radius,
                      int max_taps, int& tx, int& ty) {
    float rad = dm_sqrt(((float)i + radius_shift) / (float)max_taps) * max_radius;
    int ox = dm_f2i(taps.cs[2 * i] * rad), oy = dm_f2i(taps.cs[2 * i + 1] * rad);
    tx = u + ox; ty = v + oy;
    if (ox == 0 && oy == 0) return false;
    return !(tx < 0 || ty < 0 || tx >= fp.W || ty >= fp.H);
}

#ifndef VRT_CLASSIFY_BATCH
#define VRT_CLASSIFY_BATCH 4
#endif
// Split pass, before its two kernels: which taps pass the geometric test (:883-912, the mask both kernels walk), and which of
// those carry a sample whose shift into this pixel's domain has a Jacobian that is not zero (see the resampling loop of
// gris_pixel).  Needs one 48-byte record per tap and no BSDF: a kernel of 6 waves per SIMD instead of 3.
VRT_DEV void gris_classify_pixel(const FrameParams& fp, const GrisBuffers& gb, const GrisTaps& taps, int u, int v, float max_radius, int max_taps,
                                 TraceStats& ts) {
    const int idx = (v - fp.row0) * fp.W + u;
    if (outside_render_area(fp, (float)u, (float)v)) return;
    dm_rng rng = dm_rng_init(fp.seed, fp.frame, (uint32_t)(v * fp.W + u), 1u);
    (void)dm_rng_f32(&rng);  // start_index draw (:827), value unused
    const float radius_shift = dm_rng_f32(&rng);
    const GrisGeo* cg = &gb.geo[idx];
    const f3 cx1 = cg->x1, cn1 = cg->n;
    const float cdist = cg->dist;
    if (near_zero3(cx1)) return;
    unsigned accepted = 0u, live = 0u, live_first = 0u;
    float dead_M = 0.0f;
    // the centre's own sample, for its shifts into the neighbours' domains (first kernel)
    const GrisTest ct = gb.tst[idx];
    const bool c_counts_sky = fp.use_sky == 1 && !near_zero3(ct.rc_normal) && !near_zero3(gb.src[idx].rc_nee_dir);
    // VRT_CLASSIFY_BATCH taps at a time, their records fetched before any is looked at: the loop is a chain of dependent cache misses otherwise
    // (a tap outside the image fetches the pixel's own record, which is not used)
    for (int i0 = 0; i0 < max_taps; i0 += VRT_CLASSIFY_BATCH) {
        GrisTest rec[VRT_CLASSIFY_BATCH];
        int at[VRT_CLASSIFY_BATCH];
#pragma unroll
        for (int k = 0; k < VRT_CLASSIFY_BATCH; k++) {
            int tx, ty;
            const bool in = i0 + k < max_taps && gris_tap(fp, taps, u, v, i0 + k, radius_shift, max_radius, max_taps, tx, ty);
            at[k] = in ? (ty - fp.row0) * fp.W + tx : -1;
            rec[k] = gb.tst[in ? at[k] : idx];
        }
#pragma unroll
        for (int k = 0; k < VRT_CLASSIFY_BATCH; k++) {
            if (at[k] < 0) continue;
            const GrisTest& nt = rec[k];
            if (dm_abs(nt.dist - cdist) > 0.1f * cdist || dot3(cn1, nt.n) < 0.5f) continue;  // :912
            accepted |= 1u << (i0 + k);
            if (!shift_is_constant(nt.x1, nt.n, nt.dst_ok != 0u, nt.M, ct.rc_pos, ct.rc_normal, ct.jac)) live_first |= 1u << (i0 + k);
            else if (c_counts_sky) ts.sky_lookups += 1u;   // (the lookup the reference makes inside the shift that is not evaluated)
            if (shift_jacobian(cx1, cn1, nt.rc_pos, nt.rc_normal, nt.jac) != 0.0f) live |= 1u << (i0 + k);
            else {
                dead_M += nt.M;
                // (the reference still looks the sky's transmittance up for such a shift, :770-772: counted for instrumented launches)
                if (fp.use_sky == 1 && !near_zero3(nt.rc_normal) && !near_zero3(gb.src[at[k]].rc_nee_dir)) ts.sky_lookups += 1u;
            }
        }
    }
    gb.geo[idx].pad = accepted;
    gb.src[idx].pad0 = live;
    gb.src[idx].pad1 = dm_f2u(dead_M);
    gb.src[idx].pad2 = live_first;
}

// One trip of the first tap loop (:917-931): the CENTRE's sample shifted into the domain of the neighbour at pixel (tx, ty); returns its
// term 1 - cw of the canonical MIS weight.  Shared by the per-pixel loop below and by the first kernel's wave-level schedule
// (k_gris<., ., 1>, vrt_kernels.hip), where a lane works on whichever pixel's tap it is dealt.
VRT_DEV float gris_first_term(const FrameParams& fp, const SceneData& sc, const GrisBuffers& gb, const Reservoir& center, f3 center_rc_ty,
                              f3 center_sky_t, const RcPre& center_pre, int tx, int ty, int max_taps, TraceStats& ts) {
    const GrisGeo ng = gb.geo[(ty - fp.row0) * fp.W + tx];
    const float nb_M = ng.M;
    const int nmat_id = (int)(ng.mat & 255u);
    Material nmat = load_material(sc.mats, nmat_id);
    nmat.base = ng.base;
    f3 cd, cs;
    float cjac;
    Surf nds;
    surf_set(nds, nmat, load_mat_derived(gb.mats_x, nmat_id), ng.n, ng.v, cross3(ng.n, ng.ty), ng.ty);
    SurfShared ndsc;   // the neighbour's own shading point: all of it from the prepare pass
    ndsc.lambert = ng.lambert; ndsc.sheen_col = ng.sheen_col; ndsc.spec_col = ng.spec_col; ndsc.fv = ng.fv; ndsc.g_v = ng.g_v; ndsc.gc_v = ng.gc_v;
    VRT_GREGION(0);
    shift_sample(fp, sc, gb.mats_x, ng.x1, nds, ndsc, center, center_rc_ty, center_sky_t, center_pre, cd, cs, cjac, ts, 0);
    float c_p_hat = lum(cd + cs) * cjac;
    float cw = c_p_hat * nb_M;
    cw /= c_p_hat * nb_M + lum(center.z.F) * center.M / (float)max_taps;
    return 1.0f - cw;
}
// The term of a tap whose centre -> neighbour shift is a known constant (shift_is_constant): c_p_hat = L * 0.
VRT_DEV float gris_first_const_term(const Reservoir& center, int max_taps) {
    float cw = 0.0f;
    cw /= 0.0f + lum(center.z.F) * center.M / (float)max_taps;
    return 1.0f - cw;
}

// pathtracer.py:815-989, called as spatial_GRIS(0, 24.0, 32, 1) (:1313)
// PHASE 0: the whole pass.  PHASE 1 / 2: the pass as two kernels behind gris_classify_pixel, which leaves them the mask of
// accepted taps in the pixel's own GrisGeo record (pad) -- the canonical MIS weight (second loop), left in pad3 (nobody else reads
// those words); then the resampling loop and everything after it.  Each half carries only its own loop invariants (the
// centre's sample in the first, its shading point and the output reservoir in the second), which is what lets each fit the
// 168 registers of a third wave per SIMD (the whole pass needs about 250).  Same arithmetic in the same order.
template <int PHASE = 0, class PyrT>
VRT_DEV void gris_pixel(const FrameParams& fp, const SceneData& sc, const PyrT& P, const GrisBuffers& gb, const GrisTaps& taps, int u, int v,
                        int pass_id, float max_radius, int max_taps, int pass_total, TraceStats& ts) {
    const int idx = (v - fp.row0) * fp.W + u;
    if (outside_render_area(fp, (float)u, (float)v)) {
        // Untouched by the reference, whose colour buffers are updated in place.  Diffuse: the input buffer holds the
        // last HDR value there (it swaps roles with the HDR target every pass): carry it across.  Specular: what the
        // reference still has there is the LAST PASS'S RESULT OF THIS KERNEL, which is exactly what the output buffer
        // still holds -- leave it (copying the input would bring back the last render output from before its reuse).
        if (PHASE != 1) gb.color_d_out[idx] = gb.color_d_in[idx];
        return;
    }
    dm_rng rng = dm_rng_init(fp.seed, fp.frame, (uint32_t)(v * fp.W + u), 1u);
    (void)dm_rng_f32(&rng);  // start_index draw (:827), value unused
    const float radius_shift = dm_rng_f32(&rng);

    Reservoir center, outr;
    f3 center_rc_ty, center_sky_t;
    RcPre center_pre;
    gris_load_src(center, center_rc_ty, center_sky_t, center_pre, gb.src[idx]);
    reservoir_init(outr);

    const GrisGeo cg = gb.geo[idx];
    const f3 cx1 = cg.x1;
    const float cdist = cg.dist;
    const f3 cn1 = cg.n;
    if (near_zero3(cx1)) {
        if (PHASE == 1) return;
        gb.color_d_out[idx] = center.z.F;
        gb.color_s_out[idx] = gb.color_s_in[idx];
        gb.res_out[idx] = gb.res_in[idx];
        return;
    }
    int valid = 0;
    float canonical_mis = 1.0f;
    f3 chosen_d = mk3(0.0f), chosen_s = mk3(0.0f);
    int chosen_tap = -1;   // (second kernel of the split pass: the tap whose sample the output reservoir holds)

    // The reference's tap loop does two independent things per accepted tap: (1) shift the CENTRE sample into the
    // neighbour's domain to grow the canonical MIS weight (:917-931), (2) shift the NEIGHBOUR's sample into the centre's
    // domain and stream it through the output reservoir (:922-956).  (1) only sums into canonical_mis, (2) only draws
    // from the random stream; run as separate loops over the same taps, in tap order, every sum and draw keeps its place
    // while each loop carries one shift's worth of live state instead of two (k_gris is register bound).
    // A first cheap loop only decides which taps pass the geometric test (:883-912); the two expensive loops then walk
    // each pixel's OWN accepted taps (ascending, so the order of sums and draws is the reference's): a wave runs as many
    // trips as its busiest pixel accepted taps instead of all 32 with the rejected pixels' lanes idle.
    unsigned accepted = 0u;  // max_taps <= 32
    if (PHASE == 1) accepted = cg.pad;    // (split pass: from gris_classify_pixel)
    if (PHASE == 2) { accepted = cg.pad; canonical_mis = dm_u2f(cg.pad3); }
    for (int i = 0; PHASE == 0 && i < max_taps; i++) {
        int tx, ty;
        if (!gris_tap(fp, taps, u, v, i, radius_shift, max_radius, max_taps, tx, ty)) continue;
        taps.off[i * taps.off_stride] = (uint16_t)((tx - u + 128) | ((ty - v + 128) << 8));  // |offset| <= max_radius < 128
        const GrisGeo* ng = &gb.geo[(ty - fp.row0) * fp.W + tx];
        const f3 nn1 = ng->n;
        const float ndist = ng->dist;
        if (dm_abs(ndist - cdist) > 0.1f * cdist || dot3(cn1, nn1) < 0.5f) continue;  // :912
        accepted |= 1u << i;
    }
    // Split pass: the taps whose centre -> neighbour shift is a known constant (shift_is_constant(): a zero Jacobian, two in three
    // in a scene open to the sky) are not evaluated.  Their term 1 - 0 / X is summed where the tap stands in the order -- the
    // sum is a float sum, its order is the reference's -- so a lane's trips are its taps that NEED a shift, and a wave's its
    // busiest pixel's.
    unsigned walk = (PHASE == 2) ? 0u : accepted, pending = 0u;
    float const_term = 0.0f;
    if (PHASE == 1) {
        walk = accepted & gb.src[idx].pad2;
        pending = accepted;
        const_term = gris_first_const_term(center, max_taps);
    }
    for (unsigned m = walk; m != 0u; m &= m - 1u) {
        const int i = __builtin_ctz(m);
        if (PHASE == 1) {
            for (int k = __builtin_popcount(pending & ((1u << i) - 1u)); k > 0; k--) canonical_mis += const_term;
            pending &= ~(((1u << i) - 1u) | (1u << i));
        }
        int tx, ty;
        if (PHASE == 1) (void)gris_tap(fp, taps, u, v, i, radius_shift, max_radius, max_taps, tx, ty);
        else {
            const int packed = taps.off[i * taps.off_stride];
            tx = u + (packed & 255) - 128; ty = v + (packed >> 8) - 128;
        }
        canonical_mis += gris_first_term(fp, sc, gb, center, center_rc_ty, center_sky_t, center_pre, tx, ty, max_taps, ts);
    }
    // Register budget (the kernel is held to 256 and waits on spill reloads): the first tap loop carries the centre's sample,
    // the second the centre's shading point and the output reservoir -- neither needs the other's, so the shading point is
    // built only now and the centre's sample is read again after the second loop instead of being kept across it.
    if (PHASE == 1) {   // what the second kernel needs of the first
        for (int k = __builtin_popcount(pending); k > 0; k--) canonical_mis += const_term;
        gb.geo[idx].pad3 = dm_f2u(canonical_mis);
        return;
    }
    const float center_M = center.M;
    int cmat_id;
    const Material cmat = material_from_bits(sc.mats, cg.mat, cmat_id);
    Surf cds;  // the centre pixel's shading frame (pathtracer.py:731-732 with dst = centre)
    surf_set(cds, cmat, load_mat_derived(gb.mats_x, cmat_id), cn1, cg.v, cross3(cn1, cg.ty), cg.ty);
    SurfShared cdsc;   // destination of every neighbour's sample, whatever its lobe: from the prepare pass (= surf_shared(cds, 1, 1, 1))
    cdsc.lambert = cg.lambert; cdsc.sheen_col = cg.sheen_col; cdsc.spec_col = cg.spec_col; cdsc.fv = cg.fv; cdsc.g_v = cg.g_v; cdsc.gc_v = cg.gc_v;
    // A shift whose Jacobian comes out zero -- the sample's reconnection vertex lies behind the destination's horizon or faces away
    // from it (:684-688), the usual fate of a stored escape sample, whose zero normal comes back from its 8-bit code as (0, 0, -1)
    // -- leaves no trace beyond the two counters: p_hat / 0 is inf or NaN, so the MIS weight is inf / inf = NaN -> 0 (:941-943) and
    // the merge weight w * p_hat * 0 * 0 is 0 or NaN, never > 0: no draw, no selection, M += and valid += only (reservoir.py:78-85).
    // Two shifts in three (nine in ten at small frame sizes) end that way in a scene open to the sky.  The Jacobian needs 32 bytes
    // of the sample and ~100 instructions (shift_jacobian: the same expressions as shift_sample, so the same bits):
    //  * split pass: a small kernel of its own (gris_classify_pixel, many waves per SIMD) works it out for every tap and leaves the
    //    mask of live taps and the summed M of the dead ones in the pixel's GrisSrc record (pad0, pad1); this kernel then runs the
    //    BSDF work for live taps only -- a wave's trips are its busiest pixel's LIVE taps.  (M counts merged samples: 1 or 2 out
    //    of the render pass, whole numbers far below 2^24, so their sum does not depend on the order.  Worked out inside the first
    //    kernel's tap loop instead, the same test cost that kernel 0.78 ms for the 0.9 ms it saves here.)
    //  * whole pass: each lane runs through its dead taps before the wave meets at the BSDF work.
    // (The first loop cannot drop its own dead shifts: its weight is 0 / X only if the integrand it multiplies by 0 is finite.)
    if (PHASE == 2) {
        const unsigned live = gb.src[idx].pad0;
        outr.M = dm_u2f(gb.src[idx].pad1);
        valid = __builtin_popcount(accepted & ~live);
        accepted = live;
    }
    for (unsigned m = accepted; m != 0u;) {
        int i, tx, ty;
        if (PHASE == 2) {
            i = __builtin_ctz(m);
            m &= m - 1u;
            (void)gris_tap(fp, taps, u, v, i, radius_shift, max_radius, max_taps, tx, ty);   // (the first kernel's table of offsets is gone)
        } else {
            bool live;
            do {
                i = __builtin_ctz(m);
                m &= m - 1u;
                const int packed = taps.off[i * taps.off_stride];
                tx = u + (packed & 255) - 128; ty = v + (packed >> 8) - 128;
                const GrisSrc* sp = &gb.src[(ty - fp.row0) * fp.W + tx];
                live = shift_jacobian(cx1, cn1, sp->rc_pos, sp->rc_normal, sp->jac) != 0.0f;
                if (!live) {
                    outr.M += sp->M;
                    valid += 1;
                    // (the reference still looks the sky's transmittance up for such a shift, :770-772: counted for instrumented launches)
                    if (fp.use_sky == 1 && !near_zero3(sp->rc_normal) && !near_zero3(sp->rc_nee_dir)) ts.sky_lookups += 1u;
                }
            } while (!live && m != 0u);
            if (!live) break;
        }
        Reservoir nb;
        f3 nb_rc_ty, nb_sky_t;
        RcPre nb_pre;
        gris_load_src(nb, nb_rc_ty, nb_sky_t, nb_pre, gb.src[(ty - fp.row0) * fp.W + tx]);
        f3 sd, ss;
        float jac;
        VRT_GREGION(8);
        shift_sample(fp, sc, gb.mats_x, cx1, cds, cdsc, nb, nb_rc_ty, nb_sky_t, nb_pre, sd, ss, jac, ts, 8);

        float p_hat = lum(sd + ss);
        float p_hat_n = p_hat / jac;
        float nw = p_hat_n * nb.M;
        nw /= p_hat_n * nb.M + p_hat * center_M / (float)max_taps;
        if (dm_isinf(nw) || dm_isnan(nw)) nw = 0.0f;

        nb.z.F = sd + ss;
        bool sel;
        if (PHASE == 2) {
            // reservoir_add() without the copy of the sample: the second kernel remembers WHICH tap the reservoir holds and
            // reads that sample again after the loop (its record is one load away; carried through the loop it is 17 registers)
            const float in_w = nb.weight * p_hat * jac * nw;
            outr.M += nb.M;
            sel = false;
            if (in_w > 0.0f) {
                outr.weight += in_w;
                sel = dm_rng_f32(&rng) * outr.weight <= in_w;
                if (sel) chosen_tap = i;
            }
        } else sel = reservoir_add(outr, nb.z, nb.M, nb.weight * p_hat * jac * nw, rng, false);
        if (sel) { chosen_d = sd; chosen_s = ss; }
        valid += 1;
    }
    if (PHASE == 2 && chosen_tap >= 0) {
        int tx, ty;
        (void)gris_tap(fp, taps, u, v, chosen_tap, radius_shift, max_radius, max_taps, tx, ty);
        Reservoir nb;
        f3 nb_rc_ty, nb_sky_t;
        RcPre nb_pre;
        gris_load_src(nb, nb_rc_ty, nb_sky_t, nb_pre, gb.src[(ty - fp.row0) * fp.W + tx]);
        outr.z = nb.z;
        outr.z.F = chosen_d + chosen_s;
    }

#if defined(__HIP_DEVICE_COMPILE__)
    __asm__ volatile("" ::: "memory");   // (so that the values below are loaded here, not carried in registers through the loop above)
#endif
    gris_load_src(center, center_rc_ty, center_sky_t, center_pre, gb.src[idx]);
    // visibility of the chosen sample's reconnection (:959-967)
    bool force_canonical = false;
    const bool out_escape = near_zero3(outr.z.rc_normal);
    // For an escape vertex the reference compares against actual_dist = inf: |dist - inf| > 0.1 * inf is never true,
    // so that ray's result cannot be observed -- and an EMPTY output reservoir (rc_pos = 0) would make it a
    // zero-direction ray that spins the DDA for all 512 iterations.  Only the reconnection case is traced.
    if (!out_escape) {
        VRT_GREGION(15);
        const f3 to_rc = norm3(outr.z.rc_pos - cx1);
        Hit sh;
        next_hit<true>(fp, sc, P, cx1 + cn1 * 0.003f * cdist, to_rc, sh, ts);
        const float actual = len3(cx1 - outr.z.rc_pos);
        if (sh.closest < DM_INF && dm_abs(sh.closest - actual) > 0.1f * actual) { outr.weight = 0.0f; force_canonical = true; }
    }

    if (reservoir_add(outr, center.z, center.M, center.weight * lum(center.z.F) * canonical_mis, rng, force_canonical)) {
        chosen_d = gb.color_d_in[idx];
        chosen_s = gb.color_s_in[idx];
    }
    reservoir_finalize_without_M(outr);
    outr.weight /= (float)(valid + 1);

    f3 od = gb.color_d_in[idx], os = gb.color_s_in[idx];
    if (pass_id == pass_total - 1) {
        f3 emission = (cmat_id == 2) ? cmat.base : mk3(0.0f);
        if (fp.camera_is_moving == 1) chosen_d = chosen_d / max3s(cmat.base, 1e-2f);
        float wc = dm_clamp(outr.weight, 0.0f, 50.0f);
        od = chosen_d * wc + emission;
        os = chosen_s * wc;
    }
    gb.color_d_out[idx] = od;
    gb.color_s_out[idx] = os;
    reservoir_update_jacobian(outr, cx1);
    gb.res_out[idx] = reservoir_encode(outr);
}

}  // namespace vrt
#endif
