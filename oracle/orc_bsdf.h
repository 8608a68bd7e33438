/* orc_bsdf.h -- ORACLE (test infrastructure, not product code).
 *
 * Literal CPU restatement of /root/reference/renderer/bsdf.py:15-458 (Disney BSDF eval / pdf /
 * sample; lines 107-110 and 460-659 are dead code in the reference and are not restated) and of
 * the material table of renderer/materials.py:50-63.
 * pow(x, 5.0) of the Schlick terms is the multiply chain dm_pow5 (include/vrt_detmath.h).
 */
#ifndef ORC_BSDF_H
#define ORC_BSDF_H

#include "orc_math.h"

namespace orc {

enum { LOBE_DIFFUSE = 0, LOBE_SPEC_REFL = 1, LOBE_CLEARC = 2, LOBE_ALL = 9 }; /* bsdf.py:15-20 */

/* bsdf.py:26-37 -- 14 floats, the layout of one row of the material table */
struct DisneyMaterial {
    V3 base_col;
    float subsurface, metallic, specular, specular_tint, roughness, anisotropic, sheen, sheen_tint, clearcoat,
        clearcoat_gloss, ior_minus_one;
};

static const float INV_PI = (float)(1.0 / 3.141592653589793);

/* bsdf.py:39-46 */
inline V3 disneySubsurface(const DisneyMaterial& mat, float n_dot_l, float n_dot_v, float l_dot_h, float F_L, float F_V) {
    float Fss90 = l_dot_h * l_dot_h * mat.roughness;
    float Fss = mix(1.0f, Fss90, F_L) * mix(1.0f, Fss90, F_V);
    float ss = 1.25f * (Fss * (1.0f / (n_dot_l + n_dot_v) - 0.5f) + 0.5f);
    return INV_PI * ss * mat.base_col;
}
/* bsdf.py:48-67 */
inline V3 disney_diffuse(const DisneyMaterial& mat, float n_dot_l, float n_dot_v, float l_dot_h) {
    float R_R = 2.0f * mat.roughness * sqr(l_dot_h);
    float F_L = dm_pow5(1.0f - n_dot_l);
    float F_V = dm_pow5(1.0f - n_dot_v);
    V3 f_lambert = mat.base_col / PI;
    V3 f_retro = f_lambert * R_R * (F_L + F_V + F_L * F_V * (R_R - 1.0f));
    V3 f_d = f_lambert * (1.0f - 0.5f * F_L) * (1.0f - 0.5f * F_V) + f_retro;
    float albedo_lum = dot(mat.base_col, v3(0.2125f, 0.7154f, 0.0721f));
    V3 sheen_col = (albedo_lum > 0.0f) ? mat.base_col / albedo_lum : v3(1.0f);
    float sheen_schlick = dm_pow5(1.0f - l_dot_h);
    V3 sheen = mat.sheen * mix(v3(1.0f), sheen_col, mat.sheen_tint) * sheen_schlick;
    V3 ss = disneySubsurface(mat, n_dot_l, n_dot_v, l_dot_h, F_L, F_V);
    return mix(f_d, ss, mat.subsurface) + sheen;
}
/* bsdf.py:69-71 */
inline float GTR2_anisotropic(float n_dot_h, float h_dot_x, float h_dot_y, float ax, float ay) {
    return 1.0f / (PI * ax * ay * sqr(sqr(h_dot_x / ax) + sqr(h_dot_y / ay) + sqr(n_dot_h)));
}
/* bsdf.py:73-75 */
inline float smithG_GGX_aniso(float n_dot_v, float v_dot_x, float v_dot_y, float ax, float ay) {
    return 1.0f / (n_dot_v + dm_sqrt(sqr(v_dot_x * ax) + sqr(v_dot_y * ay) + sqr(n_dot_v)));
}
/* bsdf.py:77-83 */
inline V3 disney_fresnel(const DisneyMaterial& mat, float l_dot_h) {
    float albedo_lum = dot(mat.base_col, v3(0.2125f, 0.7154f, 0.0721f));
    V3 spec_tint = (albedo_lum > 0.0f) ? mat.base_col / albedo_lum : v3(1.0f);
    V3 spec_col = mix(mat.specular * 0.08f * mix(v3(1.0f), spec_tint, mat.specular_tint), mat.base_col, mat.metallic);
    float F_L = dm_pow5(1.0f - l_dot_h);
    return mix(spec_col, v3(1.0f), F_L);
}
inline void aniso_alphas(const DisneyMaterial& mat, float* ax, float* ay) { /* bsdf.py:95-98 */
    float aspect = dm_sqrt(1.0f - 0.9f * mat.anisotropic);
    *ax = dm_max(sqr(mat.roughness) / aspect, 1e-3f);
    *ay = dm_max(sqr(mat.roughness) * aspect, 1e-3f);
}
/* bsdf.py:86-105 (the 1/(4 n.l n.v) factor is folded into the Smith terms) */
inline V3 disney_specular(const DisneyMaterial& mat, float n_dot_l, float n_dot_v, float l_dot_h, float n_dot_h,
                          float h_dot_x, float h_dot_y, float l_dot_x, float l_dot_y, float v_dot_x, float v_dot_y) {
    float ax, ay;
    aniso_alphas(mat, &ax, &ay);
    float D = GTR2_anisotropic(n_dot_h, h_dot_x, h_dot_y, ax, ay);
    float G = smithG_GGX_aniso(n_dot_l, l_dot_x, l_dot_y, ax, ay) * smithG_GGX_aniso(n_dot_v, v_dot_x, v_dot_y, ax, ay);
    V3 F = disney_fresnel(mat, l_dot_h);
    return D * G * F;
}
/* bsdf.py:112-121 */
inline float GTR1(float n_dot_h, float alpha) {
    float a2 = alpha * alpha;
    float t = 1.0f + (a2 - 1.0f) * n_dot_h * n_dot_h;
    float D = (a2 - 1.0f) / (PI * dm_log(a2) * t);
    if (alpha >= 1.0f) D = 1.0f / PI;
    return D;
}
/* bsdf.py:123-127 */
inline float smithG_GGX(float n_dot_v, float alpha) {
    float a2 = alpha * alpha;
    float b = n_dot_v * n_dot_v;
    return 1.0f / (n_dot_v + dm_sqrt(a2 + b - a2 * b));
}
/* bsdf.py:129-135 */
inline float disney_clearcoat(const DisneyMaterial& mat, float n_dot_l, float n_dot_v, float n_dot_h, float l_dot_h) {
    float alpha = mix(0.1f, 0.001f, mat.clearcoat_gloss);
    float D = GTR1(dm_abs(n_dot_h), alpha);
    float F = mix(0.04f, 1.0f, dm_pow5(1.0f - l_dot_h));
    float G = smithG_GGX(n_dot_l, 0.25f) * smithG_GGX(n_dot_v, 0.25f);
    return mat.clearcoat * D * F * G;
}

/* bsdf.py:306-344 (lobe_id = LOBE_ALL gives disney_evaluate_split, bsdf.py:138-172) */
inline void disney_evaluate_lobewise_split(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang, int lobe_id,
                                           V3* bsdf_d, V3* bsdf_s, float specular_mult = 1.0f) {
    float n_dot_l = dot(n, l);
    float n_dot_v = dot(n, v);
    *bsdf_d = v3(0.0f);
    *bsdf_s = v3(0.0f);
    if (n_dot_l > 0.0f && n_dot_v > 0.0f) {
        V3 h = normalized(l + v);
        float l_dot_h = dot(l, h), n_dot_h = dot(n, h);
        float h_dot_x = dot(h, tang), h_dot_y = dot(h, bitang);
        float l_dot_x = dot(l, tang), l_dot_y = dot(l, bitang);
        float v_dot_x = dot(v, tang), v_dot_y = dot(v, bitang);
        if (lobe_id == LOBE_DIFFUSE || lobe_id == LOBE_ALL)
            *bsdf_d += disney_diffuse(mat, n_dot_l, n_dot_v, l_dot_h) * (1.0f - mat.metallic);
        if (lobe_id == LOBE_SPEC_REFL || lobe_id == LOBE_ALL)
            *bsdf_s += disney_specular(mat, n_dot_l, n_dot_v, l_dot_h, n_dot_h, h_dot_x, h_dot_y, l_dot_x, l_dot_y,
                                       v_dot_x, v_dot_y) * specular_mult;
        if (lobe_id == LOBE_CLEARC || lobe_id == LOBE_ALL)
            *bsdf_s += v3(disney_clearcoat(mat, n_dot_l, n_dot_v, n_dot_h, l_dot_h) * specular_mult);
    }
}
/* bsdf.py:138-172.  Written out separately because the reference adds the clearcoat scalar to
 * the specular vector without the specular_mult multiply of the lobewise variant. */
inline void disney_evaluate_split(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang, V3* bsdf_diffuse,
                                  V3* bsdf_spec) {
    float n_dot_l = dot(n, l);
    float n_dot_v = dot(n, v);
    *bsdf_diffuse = v3(0.0f);
    *bsdf_spec = v3(0.0f);
    if (n_dot_l > 0.0f && n_dot_v > 0.0f) {
        V3 h = normalized(l + v);
        float l_dot_h = dot(l, h), n_dot_h = dot(n, h);
        float h_dot_x = dot(h, tang), h_dot_y = dot(h, bitang);
        float l_dot_x = dot(l, tang), l_dot_y = dot(l, bitang);
        float v_dot_x = dot(v, tang), v_dot_y = dot(v, bitang);
        *bsdf_diffuse += disney_diffuse(mat, n_dot_l, n_dot_v, l_dot_h) * (1.0f - mat.metallic);
        *bsdf_spec += disney_specular(mat, n_dot_l, n_dot_v, l_dot_h, n_dot_h, h_dot_x, h_dot_y, l_dot_x, l_dot_y,
                                      v_dot_x, v_dot_y);
        *bsdf_spec += v3(disney_clearcoat(mat, n_dot_l, n_dot_v, n_dot_h, l_dot_h));
    }
}
inline V3 disney_evaluate(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang) { /* bsdf.py:174-177 */
    V3 d, s;
    disney_evaluate_split(mat, v, n, l, tang, bitang, &d, &s);
    return d + s;
}
inline V3 disney_evaluate_lobewise(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang, int lobe_id) {
    V3 d, s; /* bsdf.py:346-349 */
    disney_evaluate_lobewise_split(mat, v, n, l, tang, bitang, lobe_id, &d, &s);
    return d + s;
}

/* bsdf.py:179-182 */
inline float pdf_diffuse(V3 n, V3 l) { return saturate(dot(l, n)) / PI; }
/* bsdf.py:184-189 */
inline V3 sample_diffuse(V3 n, float* pdf, dm_rng* rng) {
    V3 dir = sample_cosine_weighted_hemisphere(n, rng);
    *pdf = saturate(dot(dir, n)) / PI;
    return dir;
}
/* bsdf.py:191-199 */
inline float pdf_clearcoat(const DisneyMaterial& mat, V3 v, V3 n, V3 l) {
    float alpha = mix(0.1f, 0.001f, mat.clearcoat_gloss);
    V3 h = normalized(v + l);
    float n_dot_h = dm_abs(dot(n, h));
    float v_dot_h = dot(v, h);
    float D = GTR1(n_dot_h, alpha);
    return D * n_dot_h / (4.0f * v_dot_h);
}
/* bsdf.py:201-224 */
inline V3 sample_clearcoat(const DisneyMaterial& mat, V3 v, V3 n, V3 tang, V3 bitang, float* pdf, dm_rng* rng) {
    float ux = dm_rng_f32(rng);
    float uy = dm_rng_f32(rng);
    float alpha = mix(0.1f, 0.001f, mat.clearcoat_gloss);
    float a2 = sqr(alpha);
    float cosTheta = dm_sqrt(dm_max(1e-4f, (1.0f - dm_pow(a2, 1.0f - ux)) / (1.0f - a2)));
    float sinTheta = dm_sqrt(dm_max(1e-4f, 1.0f - cosTheta * cosTheta));
    float phi = DM_TWO_PI * uy;
    V3 m = v3(sinTheta * dm_cos(phi), cosTheta, sinTheta * dm_sin(phi));
    m = m.x * tang + m.z * bitang + m.y * n;
    if (dot(m, v) < 0.0f) m *= -1.0f;
    V3 sampled_dir = reflect(-v, m);
    float n_dot_h = dm_abs(dot(n, m));
    float v_dot_h = dot(v, m);
    float D = GTR1(n_dot_h, alpha);
    *pdf = D * n_dot_h / (4.0f * v_dot_h);
    return sampled_dir;
}
/* bsdf.py:226-252 */
inline V3 GGX_VNDF_aniso(V3 v, V3 n, V3 tang, V3 bitang, float ax, float ay, dm_rng* rng) {
    V3 v_t = v3(dot(tang, v), dot(n, v), dot(bitang, v)); /* mat3(tang, n, bitang) @ v, rows */
    float ux = dm_rng_f32(rng);
    float uy = dm_rng_f32(rng);
    V3 V = normalized(v3(v_t.x * ax, v_t.y, v_t.z * ay));
    V3 t1 = (V.y < 0.9999f) ? normalized(cross(V, v3(0.0f, 1.0f, 0.0f))) : v3(1.0f, 0.0f, 0.0f);
    V3 t2 = cross(t1, V);
    float a = 1.0f / (1.0f + V.y);
    float r = dm_sqrt(ux);
    float phi = (uy < a) ? (uy / a) * PI : PI + (uy - a) / (1.0f - a) * PI;
    float p1 = r * dm_cos(phi);
    float p2 = r * dm_sin(phi) * ((uy < a) ? 1.0f : V.y);
    V3 m = p1 * t1 + p2 * t2 + dm_sqrt(dm_max(0.0f, 1.0f - p1 * p1 - p2 * p2)) * V;
    m = normalized(v3(ax * m.x, m.y, ay * m.z));
    m = m.x * tang + m.z * bitang + m.y * n;
    if (dot(m, v) < 0.0f) m *= -1.0f;
    return m;
}
/* bsdf.py:254-277 */
inline float pdf_specular(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang) {
    float ax, ay;
    aniso_alphas(mat, &ax, &ay);
    V3 h = normalized(v + l);
    float n_dot_l = dm_abs(dot(n, l));
    float n_dot_v = dot(n, v);
    float l_dot_h = dm_abs(dot(l, h));
    float n_dot_h = dot(n, h);
    float h_dot_x = dot(h, tang), h_dot_y = dot(h, bitang);
    float v_dot_x = dot(v, tang), v_dot_y = dot(v, bitang);
    float D = GTR2_anisotropic(n_dot_h, h_dot_x, h_dot_y, ax, ay);
    float G = smithG_GGX_aniso(n_dot_v, v_dot_x, v_dot_y, ax, ay);
    return G * l_dot_h * D / n_dot_l;
}
/* bsdf.py:279-304 */
inline V3 sample_specular(const DisneyMaterial& mat, V3 v, V3 n, V3 tang, V3 bitang, float* pdf, dm_rng* rng) {
    float ax, ay;
    aniso_alphas(mat, &ax, &ay);
    V3 m = GGX_VNDF_aniso(v, n, tang, bitang, ax, ay, rng);
    V3 sampled_dir = reflect(-v, m);
    float n_dot_l = dm_abs(dot(n, sampled_dir));
    float n_dot_v = dot(n, v);
    float l_dot_h = dm_abs(dot(sampled_dir, m));
    float n_dot_h = dot(n, m);
    float h_dot_x = dot(m, tang), h_dot_y = dot(m, bitang);
    float v_dot_x = dot(v, tang), v_dot_y = dot(v, bitang);
    float D = GTR2_anisotropic(n_dot_h, h_dot_x, h_dot_y, ax, ay);
    float G = smithG_GGX_aniso(n_dot_v, v_dot_x, v_dot_y, ax, ay);
    *pdf = G * l_dot_h * D / n_dot_l;
    return sampled_dir;
}
/* bsdf.py:351-363 */
inline void disney_get_lobe_probabilities(const DisneyMaterial& mat, float* diffuse_w, float* specular_w, float* clearcoat_w) {
    float dw = (1.0f - mat.metallic) * dm_clamp(1.0f - mat.specular, 0.4f, 0.9f);
    float sw = 1.0f - dw;
    float cw = mat.clearcoat * 0.7f;
    float w_sum = dw + sw + cw;
    *diffuse_w = dw / w_sum;
    *specular_w = sw / w_sum;
    *clearcoat_w = cw / w_sum;
}
/* bsdf.py:365-381 */
inline float pdf_disney_lobewise(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang, int lobe_id) {
    float dw, sw, cw;
    disney_get_lobe_probabilities(mat, &dw, &sw, &cw);
    float pdf = 1.0f;
    if (lobe_id == LOBE_DIFFUSE) pdf *= pdf_diffuse(n, l) * dw;
    else if (lobe_id == LOBE_SPEC_REFL) pdf *= pdf_specular(mat, v, n, l, tang, bitang) * sw;
    else pdf *= pdf_clearcoat(mat, v, n, l) * cw;
    if (dm_isinf(pdf) || dm_isnan(pdf)) pdf = 1.0f;
    return pdf;
}
/* bsdf.py:383-393 */
inline float pdf_disney(const DisneyMaterial& mat, V3 v, V3 n, V3 l, V3 tang, V3 bitang) {
    float dw, sw, cw;
    disney_get_lobe_probabilities(mat, &dw, &sw, &cw);
    float pdf = 0.0f;
    pdf += pdf_diffuse(n, l) * dw;
    pdf += pdf_specular(mat, v, n, l, tang, bitang) * sw;
    pdf += pdf_clearcoat(mat, v, n, l) * cw;
    return pdf;
}
/* bsdf.py:395-458 */
inline V3 sample_disney(const DisneyMaterial& mat, V3 v, V3 n, V3 tang, V3 bitang, V3* brdf_out, float* pdf_out,
                        int* lobe_out, dm_rng* rng) {
    float dw, sw, cw;
    disney_get_lobe_probabilities(mat, &dw, &sw, &cw);
    V3 sample_dir = v3(1.0f);
    V3 brdf = v3(0.0f);
    float pdf = 1.0f;
    float rand = dm_rng_f32(rng);
    int chosen_lobe = -1;
    if (rand <= dw) {
        sample_dir = sample_diffuse(n, &pdf, rng);
        chosen_lobe = LOBE_DIFFUSE;
    } else if (rand <= dw + sw) {
        sample_dir = sample_specular(mat, v, n, tang, bitang, &pdf, rng);
        chosen_lobe = LOBE_SPEC_REFL;
    } else {
        sample_dir = sample_clearcoat(mat, v, n, tang, bitang, &pdf, rng);
        chosen_lobe = LOBE_CLEARC;
    }
    float n_dot_l = dot(n, sample_dir);
    float n_dot_v = dot(n, v);
    V3 h = normalized(sample_dir + v);
    float l_dot_h = dot(sample_dir, h), n_dot_h = dot(n, h);
    float h_dot_x = dot(h, tang), h_dot_y = dot(h, bitang);
    float l_dot_x = dot(sample_dir, tang), l_dot_y = dot(sample_dir, bitang);
    float v_dot_x = dot(v, tang), v_dot_y = dot(v, bitang);
    if (chosen_lobe == LOBE_DIFFUSE) {
        brdf += disney_diffuse(mat, n_dot_l, n_dot_v, l_dot_h) * (1.0f - mat.metallic);
        pdf *= dw;
    } else if (chosen_lobe == LOBE_SPEC_REFL) {
        brdf += disney_specular(mat, n_dot_l, n_dot_v, l_dot_h, n_dot_h, h_dot_x, h_dot_y, l_dot_x, l_dot_y, v_dot_x, v_dot_y);
        pdf *= sw;
    } else {
        brdf += v3(disney_clearcoat(mat, n_dot_l, n_dot_v, n_dot_h, l_dot_h));
        pdf *= cw;
    }
    if (dm_isinf(pdf) || dm_isnan(pdf)) pdf = 1.0f;
    *brdf_out = brdf;
    *pdf_out = pdf;
    *lobe_out = chosen_lobe;
    return sample_dir;
}

/* materials.py:50-63 defaults */
inline DisneyMaterial default_material() {
    DisneyMaterial m;
    m.base_col = v3(1.0f);
    m.subsurface = 0.0f; m.metallic = 0.0f; m.specular = 0.04f; m.specular_tint = 0.0f; m.roughness = 0.9f;
    m.anisotropic = 0.0f; m.sheen = 0.0f; m.sheen_tint = 0.0f; m.clearcoat = 0.0f; m.clearcoat_gloss = 0.0f;
    m.ior_minus_one = 0.0f;
    return m;
}

} /* namespace orc */
#endif
