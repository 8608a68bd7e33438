// vrt_bsdf.h -- Disney BSDF evaluation, pdfs and samplers on the device.
//
// Replaces DisneyBSDF (reference renderer/bsdf.py:39-458; 107-110 and 460-659 are dead there) and
// the sampling helpers of renderer/math_utils.py:21-64.  A shading point's geometry (frame, view
// vector, roughness-derived alphas, lobe weights) is gathered once in `Surf` and shared by the
// pdf / eval / sample calls of one path vertex instead of being re-derived in each, which is where
// a per-thread GPU implementation wins its registers back; the value of every expression is the
// reference's (no reassociation, no contraction).
#ifndef VRT_BSDF_H
#define VRT_BSDF_H

#include "vrt_types.h"

namespace vrt {

enum { LOBE_DIFFUSE = 0, LOBE_SPEC = 1, LOBE_CLEARCOAT = 2, LOBE_ALL = 9 };  // bsdf.py:15-20

#define VRT_INV_PI ((float)(1.0 / 3.141592653589793))

// ---- math_utils.py sampling helpers ---------------------------------------------------------
VRT_DEV void ortho_basis(f3 n, f3& x, f3& y) {  // :32-37
    f3 h = (dm_abs(n.y) > 0.9f) ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 1.0f, 0.0f);
    y = norm3(cross3(n, h));
    x = cross3(n, y);
}
VRT_DEV f3 cone_dir_from(float cos_max, f3 n, f3 bx, f3 by, float u0, float u1) {  // :44-59 after its two draws, basis hoisted
    float ct = (1.0f - u0) + u0 * cos_max;
    float st = dm_sqrt(1.0f - ct * ct);
    float s, c;
    dm_sincos(DM_TWO_PI * u1, &s, &c);
    f3 v = mk3(st * c, st * s, ct);
    return mk3(bx.x * v.x + by.x * v.y + n.x * v.z, bx.y * v.x + by.y * v.y + n.y * v.z, bx.z * v.x + by.z * v.y + n.z * v.z);
}
VRT_DEV f3 cone_dir(float cos_max, f3 n, f3 bx, f3 by, dm_rng& rng) {
    const float u0 = dm_rng_f32(&rng), u1 = dm_rng_f32(&rng);
    return cone_dir_from(cos_max, n, bx, by, u0, u1);
}
VRT_DEV float cone_pdf(float cos_max, float cos_theta) {  // :61-63
    return (cos_theta >= cos_max) ? 1.0f / (DM_TWO_PI * (1.0f - cos_max)) : 0.0f;
}
VRT_DEV f3 cosine_dir(f3 n, dm_rng& rng) {  // :21-30
    float u0 = dm_rng_f32(&rng), u1 = dm_rng_f32(&rng);
    float a = 1.0f - 2.0f * u0;
    float b = dm_sqrt(1.0f - a * a);
    a *= (float)(1.0 - 1e-5);
    b *= (float)(1.0 - 1e-5);
    float s, c;
    dm_sincos(DM_TWO_PI * u1, &s, &c);
    return norm3(mk3(n.x + b * c, n.y + b * s, n.z + a));
}

// ---- one shading point ----------------------------------------------------------------------
struct Surf {
    Material m;
    f3 n, tx, ty, v;       // normal, tangent, bitangent, view (= -incoming)
    float ax, ay;          // anisotropic GGX alphas (bsdf.py:95-98)
    float cc_alpha;        // clearcoat alpha (bsdf.py:131)
    float w_d, w_s, w_c;   // lobe selection probabilities (bsdf.py:351-363)
    float n_v, v_x, v_y;   // view projections
};

// The part of a shading point that depends on the material row alone (not on its base colour): one table entry per
// material id for callers that set up many surfaces (spatial reuse), computed by the same expressions.
struct MatDerived { float ax, ay, cc_alpha, w_d, w_s, w_c; };
VRT_DEV MatDerived mat_derive(const Material& m) {
    MatDerived x;
    float aspect = dm_sqrt(1.0f - 0.9f * m.anisotropic);
    x.ax = dm_max(sq(m.roughness) / aspect, 1e-3f);
    x.ay = dm_max(sq(m.roughness) * aspect, 1e-3f);
    x.cc_alpha = lerp1(0.1f, 0.001f, m.clearcoat_gloss);
    float dw = (1.0f - m.metallic) * dm_clamp(1.0f - m.specular, 0.4f, 0.9f);
    float sw = 1.0f - dw;
    float cw = m.clearcoat * 0.7f;
    float sum = dw + sw + cw;
    x.w_d = dw / sum; x.w_s = sw / sum; x.w_c = cw / sum;
    return x;
}
VRT_DEV void surf_set(Surf& s, const Material& m, const MatDerived& x, f3 n, f3 v, f3 tx, f3 ty) {
    s.m = m; s.n = n; s.v = v; s.tx = tx; s.ty = ty;
    s.ax = x.ax; s.ay = x.ay; s.cc_alpha = x.cc_alpha; s.w_d = x.w_d; s.w_s = x.w_s; s.w_c = x.w_c;
    s.n_v = dot3(n, v); s.v_x = dot3(v, s.tx); s.v_y = dot3(v, s.ty);
}
VRT_DEV void surf_init(Surf& s, const Material& m, f3 n, f3 v) {
    f3 tx, ty;
    ortho_basis(n, tx, ty);
    surf_set(s, m, mat_derive(m), n, v, tx, ty);
}

VRT_DEV float gtr2_aniso(float nh, float hx, float hy, float ax, float ay) {  // :69-71
    return 1.0f / (DM_PI * ax * ay * sq(sq(hx / ax) + sq(hy / ay) + sq(nh)));
}
VRT_DEV float smith_aniso(float nv, float vx, float vy, float ax, float ay) {  // :73-75
    return 1.0f / (nv + dm_sqrt(sq(vx * ax) + sq(vy * ay) + sq(nv)));
}
VRT_DEV float gtr1(float nh, float alpha) {  // :112-121
    float a2 = alpha * alpha;
    float t = 1.0f + (a2 - 1.0f) * nh * nh;
    float D = (a2 - 1.0f) / (DM_PI * dm_log(a2) * t);
    if (alpha >= 1.0f) D = 1.0f / DM_PI;
    return D;
}
VRT_DEV float smith_iso(float nv, float alpha) {  // :123-127
    float a2 = alpha * alpha, b = nv * nv;
    return 1.0f / (nv + dm_sqrt(a2 + b - a2 * b));
}
VRT_DEV f3 tint_of(f3 base) {  // base / luminance, or white (bsdf.py:60-61, 79-80)
    float l = dot3(base, mk3(0.2125f, 0.7154f, 0.0721f));
    return (l > 0.0f) ? base / l : mk3(1.0f);
}

// bsdf.py:48-67 (+ 39-46): diffuse, retro-reflection, subsurface and sheen
VRT_DEV f3 eval_diffuse(const Material& m, float nl, float nv, float lh) {
    float rr = 2.0f * m.roughness * sq(lh);
    float fl = dm_pow5(1.0f - nl), fv = dm_pow5(1.0f - nv);
    f3 lambert = m.base / DM_PI;
    f3 retro = lambert * rr * (fl + fv + fl * fv * (rr - 1.0f));
    f3 fd = lambert * (1.0f - 0.5f * fl) * (1.0f - 0.5f * fv) + retro;
    f3 sheen = m.sheen * lerp3(mk3(1.0f), tint_of(m.base), m.sheen_tint) * dm_pow5(1.0f - lh);
    float fss90 = lh * lh * m.roughness;
    float fss = lerp1(1.0f, fss90, fl) * lerp1(1.0f, fss90, fv);
    float ss = 1.25f * (fss * (1.0f / (nl + nv) - 0.5f) + 0.5f);
    f3 sub = VRT_INV_PI * ss * m.base;
    return lerp3(fd, sub, m.subsurface) + sheen;
}
// bsdf.py:77-105
VRT_DEV f3 eval_specular(const Surf& s, float nl, float lh, float nh, float hx, float hy, float lx, float ly) {
    const Material& m = s.m;
    float D = gtr2_aniso(nh, hx, hy, s.ax, s.ay);
    float G = smith_aniso(nl, lx, ly, s.ax, s.ay) * smith_aniso(s.n_v, s.v_x, s.v_y, s.ax, s.ay);
    f3 spec_col = lerp3(m.specular * 0.08f * lerp3(mk3(1.0f), tint_of(m.base), m.specular_tint), m.base, m.metallic);
    f3 F = lerp3(spec_col, mk3(1.0f), dm_pow5(1.0f - lh));
    return D * G * F;
}
// bsdf.py:129-135
VRT_DEV float eval_clearcoat(const Surf& s, float nl, float nh, float lh) {
    float D = gtr1(dm_abs(nh), s.cc_alpha);
    float F = lerp1(0.04f, 1.0f, dm_pow5(1.0f - lh));
    float G = smith_iso(nl, 0.25f) * smith_iso(s.n_v, 0.25f);
    return s.m.clearcoat * D * F * G;
}

// bsdf.py:306-344; lobe = LOBE_ALL is disney_evaluate_split (138-172)
VRT_DEV void eval_lobes(const Surf& s, f3 l, int lobe, f3& out_d, f3& out_s) {
    out_d = mk3(0.0f);
    out_s = mk3(0.0f);
    float nl = dot3(s.n, l);
    if (nl > 0.0f && s.n_v > 0.0f) {
        f3 h = norm3(l + s.v);
        float lh = dot3(l, h), nh = dot3(s.n, h);
        if (lobe == LOBE_DIFFUSE || lobe == LOBE_ALL) out_d = out_d + eval_diffuse(s.m, nl, s.n_v, lh) * (1.0f - s.m.metallic);
        if (lobe == LOBE_SPEC || lobe == LOBE_ALL)
            out_s = out_s + eval_specular(s, nl, lh, nh, dot3(h, s.tx), dot3(h, s.ty), dot3(l, s.tx), dot3(l, s.ty));
        if (lobe == LOBE_CLEARCOAT || lobe == LOBE_ALL) out_s = out_s + mk3(eval_clearcoat(s, nl, nh, lh));
    }
}

// bsdf.py:179-182, 254-277, 191-199
VRT_DEV float pdf_diffuse_lobe(const Surf& s, f3 l) { return dm_saturate(dot3(l, s.n)) / DM_PI; }
VRT_DEV float pdf_specular_lobe(const Surf& s, f3 l) {
    f3 h = norm3(s.v + l);
    float nl = dm_abs(dot3(s.n, l));
    float lh = dm_abs(dot3(l, h));
    float D = gtr2_aniso(dot3(s.n, h), dot3(h, s.tx), dot3(h, s.ty), s.ax, s.ay);
    float G = smith_aniso(s.n_v, s.v_x, s.v_y, s.ax, s.ay);
    return G * lh * D / nl;
}
VRT_DEV float pdf_clearcoat_lobe(const Surf& s, f3 l) {
    f3 h = norm3(s.v + l);
    float nh = dm_abs(dot3(s.n, h));
    float vh = dot3(s.v, h);
    return gtr1(nh, s.cc_alpha) * nh / (4.0f * vh);
}
// bsdf.py:383-393
VRT_DEV float pdf_all(const Surf& s, f3 l) {
    float pdf = 0.0f;
    pdf += pdf_diffuse_lobe(s, l) * s.w_d;
    pdf += pdf_specular_lobe(s, l) * s.w_s;
    pdf += pdf_clearcoat_lobe(s, l) * s.w_c;
    return pdf;
}
// bsdf.py:365-381
VRT_DEV float pdf_lobe(const Surf& s, f3 l, int lobe) {
    float pdf = 1.0f;
    if (lobe == LOBE_DIFFUSE) pdf *= pdf_diffuse_lobe(s, l) * s.w_d;
    else if (lobe == LOBE_SPEC) pdf *= pdf_specular_lobe(s, l) * s.w_s;
    else pdf *= pdf_clearcoat_lobe(s, l) * s.w_c;
    if (dm_isinf(pdf) || dm_isnan(pdf)) pdf = 1.0f;
    return pdf;
}

// ---- several directions at one shading point (spatial reuse) ---------------------------------------------------
// eval_lobes() and the pdfs above re-derive, per call, terms that depend on the material and the view vector only, and
// an eval + pdf pair for the same direction derives the half vector and the two distributions twice.  The reconnection
// shift (vrt_restir.h) evaluates two directions with their pdfs at one vertex ~64 times per pixel: SurfShared holds the
// per-vertex terms, bsdf_eval_pdf() is eval_lobes() + pdf_lobe() / pdf_all() for one direction with the shared terms
// worked out once.  Every expression is the one above (same operands, same order), so the values are the same bits.
struct SurfShared {
    f3 lambert, sheen_col, spec_col;   // eval_diffuse's base / pi and sheen * mix(1, tint, sheen_tint); eval_specular's spec_col
    float fv, g_v, gc_v;               // pow5(1 - n.v); smith_aniso and smith_iso(0.25) of the view vector
};
// groups: d = what the diffuse lobe reads, sp = the specular lobe's, cc = the clearcoat lobe's (unrequested fields stay 0)
VRT_DEV SurfShared surf_shared(const Surf& s, bool d, bool sp, bool cc) {
    SurfShared c;
    c.lambert = mk3(0.0f); c.sheen_col = mk3(0.0f); c.spec_col = mk3(0.0f); c.fv = 0.0f; c.g_v = 0.0f; c.gc_v = 0.0f;
    const Material& m = s.m;
    if (d || sp) {
        const f3 tint = tint_of(m.base);
        if (d) {
            c.lambert = m.base / DM_PI;
            c.sheen_col = m.sheen * lerp3(mk3(1.0f), tint, m.sheen_tint);
            c.fv = dm_pow5(1.0f - s.n_v);
        }
        if (sp) {
            c.spec_col = lerp3(m.specular * 0.08f * lerp3(mk3(1.0f), tint, m.specular_tint), m.base, m.metallic);
            c.g_v = smith_aniso(s.n_v, s.v_x, s.v_y, s.ax, s.ay);
        }
    }
    if (cc) c.gc_v = smith_iso(s.n_v, 0.25f);
    return c;
}
VRT_DEV bool lobe_has(int lobe, int which) { return lobe == which || lobe == LOBE_ALL; }
enum { PDF_NONE = 0, PDF_LOBE = 1, PDF_ALL = 2 };
// out_d / out_s = eval_lobes(s, l, lobe);  pdf = pdf_lobe(s, l, lobe) (PDF_LOBE) or pdf_all(s, l) (PDF_ALL).
// `c` must hold the groups of every lobe that is evaluated or whose pdf is taken.
VRT_DEV void bsdf_eval_pdf(const Surf& s, const SurfShared& c, f3 l, int lobe, int pdf_mode, f3& out_d, f3& out_s, float& pdf) {
    const Material& m = s.m;
    out_d = mk3(0.0f);
    out_s = mk3(0.0f);
    const float nl = dot3(s.n, l);
    const bool front = nl > 0.0f && s.n_v > 0.0f;
    const bool e_d = front && lobe_has(lobe, LOBE_DIFFUSE), e_s = front && lobe_has(lobe, LOBE_SPEC), e_c = front && lobe_has(lobe, LOBE_CLEARCOAT);
    const bool p_d = pdf_mode == PDF_ALL || (pdf_mode == PDF_LOBE && lobe == LOBE_DIFFUSE);
    const bool p_s = pdf_mode == PDF_ALL || (pdf_mode == PDF_LOBE && lobe == LOBE_SPEC);
    const bool p_c = pdf_mode == PDF_ALL || (pdf_mode == PDF_LOBE && lobe != LOBE_DIFFUSE && lobe != LOBE_SPEC);
    const f3 h = norm3(l + s.v);
    const float lh = dot3(l, h), nh = dot3(s.n, h);
    float D = 0.0f, Dc = 0.0f;
    if (e_s || p_s) D = gtr2_aniso(nh, dot3(h, s.tx), dot3(h, s.ty), s.ax, s.ay);
    if (e_c || p_c) Dc = gtr1(dm_abs(nh), s.cc_alpha);
    if (e_d) {  // eval_diffuse
        float rr = 2.0f * m.roughness * sq(lh);
        float fl = dm_pow5(1.0f - nl), fv = c.fv;
        f3 retro = c.lambert * rr * (fl + fv + fl * fv * (rr - 1.0f));
        f3 fd = c.lambert * (1.0f - 0.5f * fl) * (1.0f - 0.5f * fv) + retro;
        f3 sheen = c.sheen_col * dm_pow5(1.0f - lh);
        float fss90 = lh * lh * m.roughness;
        float fss = lerp1(1.0f, fss90, fl) * lerp1(1.0f, fss90, fv);
        float ss = 1.25f * (fss * (1.0f / (nl + s.n_v) - 0.5f) + 0.5f);
        f3 sub = VRT_INV_PI * ss * m.base;
        out_d = out_d + (lerp3(fd, sub, m.subsurface) + sheen) * (1.0f - m.metallic);
    }
    if (e_s) {  // eval_specular
        float G = smith_aniso(nl, dot3(l, s.tx), dot3(l, s.ty), s.ax, s.ay) * c.g_v;
        f3 F = lerp3(c.spec_col, mk3(1.0f), dm_pow5(1.0f - lh));
        out_s = out_s + D * G * F;
    }
    if (e_c) {  // eval_clearcoat
        float F = lerp1(0.04f, 1.0f, dm_pow5(1.0f - lh));
        float G = smith_iso(nl, 0.25f) * c.gc_v;
        out_s = out_s + mk3(m.clearcoat * Dc * F * G);
    }
    float pd = 0.0f, ps = 0.0f, pc = 0.0f;
    if (p_d) pd = dm_saturate(nl) / DM_PI;
    if (p_s) ps = c.g_v * dm_abs(lh) * D / dm_abs(nl);
    if (p_c) { float anh = dm_abs(nh); pc = Dc * anh / (4.0f * dot3(s.v, h)); }
    if (pdf_mode == PDF_ALL) {
        pdf = 0.0f;
        pdf += pd * s.w_d;
        pdf += ps * s.w_s;
        pdf += pc * s.w_c;
    } else if (pdf_mode == PDF_LOBE) {
        pdf = p_d ? pd * s.w_d : (p_s ? ps * s.w_s : pc * s.w_c);
        if (dm_isinf(pdf) || dm_isnan(pdf)) pdf = 1.0f;
    } else {
        pdf = 0.0f;
    }
}

// A reconnection vertex is evaluated ~30 times per pixel with the SAME normal, material and light directions (the sample's
// continuation and its sun sample) and a different view vector each time.  What bsdf_eval_pdf() derives from (surface, light
// direction) alone and from the material alone is worked out once per sample (vrt_restir.h, gris_prepare_pixel) by these
// two helpers -- the same expressions with the same operands, so the same bits:
struct DirTerms { float nl, fl, g_l, gc_l, pd; };   // n.l; pow5(1 - n.l); smith_aniso and smith_iso(0.25) of the light vector; saturate(n.l) / pi
VRT_DEV DirTerms dir_terms(f3 n, f3 tx, f3 ty, float ax, float ay, f3 l) {
    DirTerms t;
    t.nl = dot3(n, l);
    t.fl = dm_pow5(1.0f - t.nl);
    t.g_l = smith_aniso(t.nl, dot3(l, tx), dot3(l, ty), ax, ay);
    t.gc_l = smith_iso(t.nl, 0.25f);
    t.pd = dm_saturate(t.nl) / DM_PI;
    return t;
}
struct MatColours { f3 lambert, sheen_col, spec_col; };   // the material-only fields of SurfShared
VRT_DEV MatColours mat_colours(const Material& m) {
    MatColours c;
    const f3 tint = tint_of(m.base);
    c.lambert = m.base / DM_PI;
    c.sheen_col = m.sheen * lerp3(mk3(1.0f), tint, m.sheen_tint);
    c.spec_col = lerp3(m.specular * 0.08f * lerp3(mk3(1.0f), tint, m.specular_tint), m.base, m.metallic);
    return c;
}
// surf_shared() with the material-only fields taken from `mc`: only the view-dependent ones are computed
VRT_DEV SurfShared surf_shared_view(const Surf& s, const MatColours& mc, bool d, bool sp, bool cc) {
    SurfShared c;
    c.lambert = mc.lambert; c.sheen_col = mc.sheen_col; c.spec_col = mc.spec_col;
    c.fv = d ? dm_pow5(1.0f - s.n_v) : 0.0f;
    c.g_v = sp ? smith_aniso(s.n_v, s.v_x, s.v_y, s.ax, s.ay) : 0.0f;
    c.gc_v = cc ? smith_iso(s.n_v, 0.25f) : 0.0f;
    return c;
}
// bsdf_eval_pdf() for a direction whose DirTerms are at hand
VRT_DEV void bsdf_eval_pdf_pre(const Surf& s, const SurfShared& c, f3 l, const DirTerms& lt, int lobe, int pdf_mode, f3& out_d, f3& out_s, float& pdf) {
    const Material& m = s.m;
    out_d = mk3(0.0f);
    out_s = mk3(0.0f);
    const float nl = lt.nl;
    const bool front = nl > 0.0f && s.n_v > 0.0f;
    const bool e_d = front && lobe_has(lobe, LOBE_DIFFUSE), e_s = front && lobe_has(lobe, LOBE_SPEC), e_c = front && lobe_has(lobe, LOBE_CLEARCOAT);
    const bool p_d = pdf_mode == PDF_ALL || (pdf_mode == PDF_LOBE && lobe == LOBE_DIFFUSE);
    const bool p_s = pdf_mode == PDF_ALL || (pdf_mode == PDF_LOBE && lobe == LOBE_SPEC);
    const bool p_c = pdf_mode == PDF_ALL || (pdf_mode == PDF_LOBE && lobe != LOBE_DIFFUSE && lobe != LOBE_SPEC);
    const f3 h = norm3(l + s.v);
    const float lh = dot3(l, h), nh = dot3(s.n, h);
    float D = 0.0f, Dc = 0.0f;
    if (e_s || p_s) D = gtr2_aniso(nh, dot3(h, s.tx), dot3(h, s.ty), s.ax, s.ay);
    if (e_c || p_c) Dc = gtr1(dm_abs(nh), s.cc_alpha);
    if (e_d) {
        float rr = 2.0f * m.roughness * sq(lh);
        float fl = lt.fl, fv = c.fv;
        f3 retro = c.lambert * rr * (fl + fv + fl * fv * (rr - 1.0f));
        f3 fd = c.lambert * (1.0f - 0.5f * fl) * (1.0f - 0.5f * fv) + retro;
        f3 sheen = c.sheen_col * dm_pow5(1.0f - lh);
        float fss90 = lh * lh * m.roughness;
        float fss = lerp1(1.0f, fss90, fl) * lerp1(1.0f, fss90, fv);
        float ss = 1.25f * (fss * (1.0f / (nl + s.n_v) - 0.5f) + 0.5f);
        f3 sub = VRT_INV_PI * ss * m.base;
        out_d = out_d + (lerp3(fd, sub, m.subsurface) + sheen) * (1.0f - m.metallic);
    }
    if (e_s) {
        float G = lt.g_l * c.g_v;
        f3 F = lerp3(c.spec_col, mk3(1.0f), dm_pow5(1.0f - lh));
        out_s = out_s + D * G * F;
    }
    if (e_c) {
        float F = lerp1(0.04f, 1.0f, dm_pow5(1.0f - lh));
        float G = lt.gc_l * c.gc_v;
        out_s = out_s + mk3(m.clearcoat * Dc * F * G);
    }
    float pd = 0.0f, ps = 0.0f, pc = 0.0f;
    if (p_d) pd = lt.pd;
    if (p_s) ps = c.g_v * dm_abs(lh) * D / dm_abs(nl);
    if (p_c) { float anh = dm_abs(nh); pc = Dc * anh / (4.0f * dot3(s.v, h)); }
    if (pdf_mode == PDF_ALL) {
        pdf = 0.0f;
        pdf += pd * s.w_d;
        pdf += ps * s.w_s;
        pdf += pc * s.w_c;
    } else if (pdf_mode == PDF_LOBE) {
        pdf = p_d ? pd * s.w_d : (p_s ? ps * s.w_s : pc * s.w_c);
        if (dm_isinf(pdf) || dm_isnan(pdf)) pdf = 1.0f;
    } else {
        pdf = 0.0f;
    }
}

VRT_DEV f3 reflect3(f3 i, f3 n) { return i - 2.0f * dot3(n, i) * n; }
VRT_DEV f3 to_world(const Surf& s, f3 m) { return m.x * s.tx + m.z * s.ty + m.y * s.n; }  // (tangent, normal, bitangent) frame

// bsdf.py:395-458 with the three samplers (184-189, 279-304 + 226-252, 201-224) inlined
VRT_DEV f3 sample_bsdf(const Surf& s, dm_rng& rng, f3& brdf, float& pdf, int& lobe) {
    f3 dir;
    float r = dm_rng_f32(&rng);
    if (r <= s.w_d) {
        dir = cosine_dir(s.n, rng);
        pdf = dm_saturate(dot3(dir, s.n)) / DM_PI;
        lobe = LOBE_DIFFUSE;
    } else if (r <= s.w_d + s.w_s) {
        // GGX visible-normal sampling in the stretched (tangent, normal, bitangent) frame
        f3 vt = mk3(dot3(s.tx, s.v), dot3(s.n, s.v), dot3(s.ty, s.v));
        float ux = dm_rng_f32(&rng), uy = dm_rng_f32(&rng);
        f3 V = norm3(mk3(vt.x * s.ax, vt.y, vt.z * s.ay));
        f3 t1 = (V.y < 0.9999f) ? norm3(cross3(V, mk3(0.0f, 1.0f, 0.0f))) : mk3(1.0f, 0.0f, 0.0f);
        f3 t2 = cross3(t1, V);
        float a = 1.0f / (1.0f + V.y);
        float rad = dm_sqrt(ux);
        float phi = (uy < a) ? (uy / a) * DM_PI : DM_PI + (uy - a) / (1.0f - a) * DM_PI;
        float sp, cp;
        dm_sincos(phi, &sp, &cp);
        float p1 = rad * cp;
        float p2 = rad * sp * ((uy < a) ? 1.0f : V.y);
        f3 mm = p1 * t1 + p2 * t2 + dm_sqrt(dm_max(0.0f, 1.0f - p1 * p1 - p2 * p2)) * V;
        mm = norm3(mk3(s.ax * mm.x, mm.y, s.ay * mm.z));
        mm = to_world(s, mm);
        if (dot3(mm, s.v) < 0.0f) mm = mm * -1.0f;
        dir = reflect3(-s.v, mm);
        float nl = dm_abs(dot3(s.n, dir));
        float lh = dm_abs(dot3(dir, mm));
        float D = gtr2_aniso(dot3(s.n, mm), dot3(mm, s.tx), dot3(mm, s.ty), s.ax, s.ay);
        float G = smith_aniso(s.n_v, s.v_x, s.v_y, s.ax, s.ay);
        pdf = G * lh * D / nl;
        lobe = LOBE_SPEC;
    } else {
        float ux = dm_rng_f32(&rng), uy = dm_rng_f32(&rng);
        float a2 = sq(s.cc_alpha);
        float ct = dm_sqrt(dm_max(1e-4f, (1.0f - dm_pow(a2, 1.0f - ux)) / (1.0f - a2)));
        float st = dm_sqrt(dm_max(1e-4f, 1.0f - ct * ct));
        float sp, cp;
        dm_sincos(DM_TWO_PI * uy, &sp, &cp);
        f3 mm = to_world(s, mk3(st * cp, ct, st * sp));
        if (dot3(mm, s.v) < 0.0f) mm = mm * -1.0f;
        dir = reflect3(-s.v, mm);
        float nh = dm_abs(dot3(s.n, mm));
        float vh = dot3(s.v, mm);
        pdf = gtr1(nh, s.cc_alpha) * nh / (4.0f * vh);
        lobe = LOBE_CLEARCOAT;
    }
    // value of the chosen lobe only (bsdf.py:439-453)
    float nl = dot3(s.n, dir);
    f3 h = norm3(dir + s.v);
    float lh = dot3(dir, h), nh = dot3(s.n, h);
    brdf = mk3(0.0f);
    if (lobe == LOBE_DIFFUSE) {
        brdf = brdf + eval_diffuse(s.m, nl, s.n_v, lh) * (1.0f - s.m.metallic);
        pdf *= s.w_d;
    } else if (lobe == LOBE_SPEC) {
        brdf = brdf + eval_specular(s, nl, lh, nh, dot3(h, s.tx), dot3(h, s.ty), dot3(dir, s.tx), dot3(dir, s.ty));
        pdf *= s.w_s;
    } else {
        brdf = brdf + mk3(eval_clearcoat(s, nl, nh, lh));
        pdf *= s.w_c;
    }
    if (dm_isinf(pdf) || dm_isnan(pdf)) pdf = 1.0f;
    return dir;
}

}  // namespace vrt
#endif
