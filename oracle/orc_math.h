/* orc_math.h -- ORACLE (test infrastructure, not product code).
 *
 * Literal CPU restatement of /root/reference/renderer/math_utils.py and
 * renderer/space_transformations.py (sampling, packing, hash, tonemap, projection helpers).
 * Each function cites the reference lines it follows.  ti.random() is replaced by the
 * per-pixel dm_rng stream (include/vrt_detmath.h); draw ORDER follows the reference.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include "orc_vec.h"

namespace orc {

/* math_utils.py:21-30  (Shirley et al. cosine-weighted hemisphere, grazing-angle fix) */
inline V3 sample_cosine_weighted_hemisphere(V3 n, dm_rng* rng) {
    float u0 = dm_rng_f32(rng);
    float u1 = dm_rng_f32(rng);
    float a = 1.0f - 2.0f * u0;
    float b = dm_sqrt(1.0f - a * a);
    a *= (float)(1.0 - 1e-5);
    b *= (float)(1.0 - 1e-5);
    float phi = DM_TWO_PI * u1;
    return normalized(v3(n.x + b * dm_cos(phi), n.y + b * dm_sin(phi), n.z + a));
}

/* math_utils.py:32-37 */
inline void make_orthonormal_basis(V3 n, V3* x, V3* y) {
    V3 h = (dm_abs(n.y) > 0.9f) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f);
    *y = normalized(cross(n, h));
    *x = cross(n, *y);
}

/* math_utils.py:44-54 */
inline V3 sample_cone(float cos_theta_max, dm_rng* rng) {
    float u0 = dm_rng_f32(rng);
    float u1 = dm_rng_f32(rng);
    float cos_theta = (1.0f - u0) + u0 * cos_theta_max;
    float sin_theta = dm_sqrt(1.0f - cos_theta * cos_theta);
    float phi = DM_TWO_PI * u1;
    return v3(sin_theta * dm_cos(phi), sin_theta * dm_sin(phi), cos_theta);
}

/* math_utils.py:39-42, 56-59: mat3(x, y, n).transpose() @ s  -> columns x, y, n */
inline V3 sample_cone_oriented(float cos_theta_max, V3 n, dm_rng* rng) {
    V3 x, y;
    make_orthonormal_basis(n, &x, &y);
    V3 s = sample_cone(cos_theta_max, rng);
    return v3(x.x * s.x + y.x * s.y + n.x * s.z,
              x.y * s.x + y.y * s.y + n.y * s.z,
              x.z * s.x + y.z * s.y + n.z * s.z);
}

/* math_utils.py:61-63 */
inline float cone_sample_pdf(float cos_theta_max, float cos_theta) {
    return (cos_theta >= cos_theta_max) ? 1.0f / (DM_TWO_PI * (1.0f - cos_theta_max)) : 0.0f;
}

/* math_utils.py:103-123.  The d[i]==0 branch only sets a flag that line 122 overwrites. */
inline bool ray_aabb_intersection(V3 box_min, V3 box_max, V3 o, V3 d, float* near_out, float* far_out) {
    float near_int = -INF, far_int = INF;
    for (int i = 0; i < 3; i++) {
        if (d[i] == 0.0f) {
            /* reference: intersect = 0 if outside the slab -- dead store */
        } else {
            float i1 = (box_min[i] - o[i]) / d[i];
            float i2 = (box_max[i] - o[i]) / d[i];
            float new_far = dm_max(i1, i2);
            float new_near = dm_min(i1, i2);
            far_int = dm_min(new_far, far_int);
            near_int = dm_max(new_near, near_int);
        }
    }
    *near_out = near_int;
    *far_out = far_int;
    return near_int <= far_int;
}

/* math_utils.py:151-153 */
inline float luminance(V3 x) { return dot(v3(0.2125f, 0.7154f, 0.0721f), x); }

/* ti.math.smoothstep / step (Appendix A-4) */
inline float smoothstep(float e0, float e1, float x) {
    float t = dm_clamp((x - e0) / (e1 - e0), 0.0f, 1.0f);
    return t * t * (3.0f - 2.0f * t);
}
inline float step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }

/* math_utils.py:163-186 (Uchimura 2017).  Python-level constants are folded in double. */
inline float uchimura1(float x) {
    const double P = 1.0, a = 1.0, m = 0.22, l = 0.4, c = 1.33, b = 0.0;
    const double l0 = ((P - m) * l) / a, S0 = m + l0, S1 = m + a * l0, C2 = (a * P) / (P - S1), CP = -C2 / P;
    float w0 = 1.0f - smoothstep(0.0f, (float)m, x);
    float w2 = step((float)(m + l0), x);
    float w1 = 1.0f - w0 - w2;
    float T = (float)m * dm_pow(x / (float)m, (float)c) + (float)b;
    float S = (float)P - (float)(P - S1) * dm_exp((float)CP * (x - (float)S0));
    float L = (float)m + (float)a * (x - (float)m);
    return T * w0 + L * w1 + S * w2;
}
inline V3 uchimura(V3 x) { return v3(uchimura1(x.x), uchimura1(x.y), uchimura1(x.z)); }

/* math_utils.py:202-207.  Returns the two binary16 codes. */
inline void encode_unit_vector_3x16(V3 vec, uint16_t out[2]) {
    float s = dm_abs(vec.x) + dm_abs(vec.y) + dm_abs(vec.z);
    vec.x /= s;
    vec.y /= s;
    float ex, ey;
    if (vec.z <= 0.0f) {
        ex = (1.0f - dm_abs(vec.y)) * ((vec.x >= 0.0f) ? 1.0f : -1.0f);
        ey = (1.0f - dm_abs(vec.x)) * ((vec.y >= 0.0f) ? 1.0f : -1.0f);
    } else {
        ex = vec.x;
        ey = vec.y;
    }
    out[0] = dm_f32_to_f16(ex * 0.5f + 0.5f);
    out[1] = dm_f32_to_f16(ey * 0.5f + 0.5f);
}
/* math_utils.py:209-215 (argument already widened to f32) */
inline V3 decode_unit_vector_3x16(float ax, float ay) {
    float ex = ax * 2.0f - 1.0f, ey = ay * 2.0f - 1.0f;
    V3 vec = v3(ex, ey, 1.0f - dm_abs(ex) - dm_abs(ey));
    float t = dm_max(-vec.z, 0.0f);
    float dx = (vec.x >= 0.0f) ? -t : t;
    float dy = (vec.y >= 0.0f) ? -t : t;
    vec.x += dx;
    vec.y += dy;
    return normalized(vec);
}
inline V3 decode_unit_vector_3x16(const uint16_t h[2]) {
    return decode_unit_vector_3x16(dm_f16_to_f32(h[0]), dm_f16_to_f32(h[1]));
}

/* math_utils.py:217-229 */
inline uint32_t hash3(uint32_t x, uint32_t y, uint32_t z) {
    x += x >> 11; x ^= x << 7; x += y; x ^= x << 3; x += z ^ (x >> 14);
    x ^= x << 6; x += x >> 15; x ^= x << 5; x += x >> 12; x ^= x << 9;
    return x;
}

/* math_utils.py:231-236: truncating albedo*255 */
inline uint32_t encode_material(int mat_id, V3 albedo) {
    uint32_t d0 = dm_f2u32((float)mat_id);
    uint32_t d1 = dm_f2u32(albedo.x * 255.0f);
    uint32_t d2 = dm_f2u32(albedo.y * 255.0f);
    uint32_t d3 = dm_f2u32(albedo.z * 255.0f);
    return (d0 << 0) | (d1 << 8) | (d2 << 16) | (d3 << 24);
}
/* math_utils.py:238-247: returns id, albedo (caller looks the material up and overrides base_col) */
inline void decode_material_bits(uint32_t enc, int* mat_id, V3* albedo) {
    uint32_t u0 = (enc >> 0) & 255u, u1 = (enc >> 8) & 255u, u2 = (enc >> 16) & 255u, u3 = (enc >> 24) & 255u;
    *albedo = v3((float)u1 / 255.0f, (float)u2 / 255.0f, (float)u3 / 255.0f);
    *mat_id = (int)u0;
}

/* math_utils.py:250-263 with size = (8,8,8,8): mult = 2^8 - 1 */
inline uint32_t encode_u32_arb8(float d0, float d1, float d2, float d3) {
    const float mult = 255.0f;
    uint32_t s0 = dm_f2u32(d0 * mult + 0.5f), s1 = dm_f2u32(d1 * mult + 0.5f);
    uint32_t s2 = dm_f2u32(d2 * mult + 0.5f), s3 = dm_f2u32(d3 * mult + 0.5f);
    return (s0 << 0) | (s1 << 8) | (s2 << 16) | (s3 << 24);
}
inline V4 decode_u32_arb8(uint32_t enc) {
    return v4((float)((enc >> 0) & 255u) / 255.0f, (float)((enc >> 8) & 255u) / 255.0f,
              (float)((enc >> 16) & 255u) / 255.0f, (float)((enc >> 24) & 255u) / 255.0f);
}

/* ---- space_transformations.py ---------------------------------------------------------- */
inline float linearize_depth(float depth, const M4& inv_proj) { /* :6-8 */
    return 1.0f / ((depth * 2.0f - 1.0f) * inv_proj.m[3][2] + inv_proj.m[3][3]);
}
inline float delinearize_depth(float lindepth, const M4& proj) { /* :10-12 */
    return ((-lindepth * proj.m[2][2] + proj.m[2][3]) / -lindepth) * -0.5f + 0.5f;
}
inline V3 screen_to_view(V2 uv, float depth, const M4& inv_proj) { /* :14-20 */
    V4 pos = v4(uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f, depth * 2.0f - 1.0f, 1.0f);
    pos = mul(inv_proj, pos);
    return v3(pos.x / pos.w, pos.y / pos.w, pos.z / pos.w);
}
inline V3 view_to_screen(V3 view_pos, const M4& proj) { /* :22-26 */
    V4 pos = mul(proj, v4(view_pos.x, view_pos.y, view_pos.z, 1.0f));
    V3 p = v3(pos.x / pos.w, pos.y / pos.w, pos.z / pos.w);
    return p * 0.5f + 0.5f;
}
inline V3 view_to_world(V3 pos, const M4& inv_view, float is_position = 1.0f) { /* :28-30 */
    return xyz(mul(inv_view, v4(pos.x, pos.y, pos.z, is_position)));
}
inline V3 world_to_view(V3 pos, const M4& view, float is_position = 1.0f) { /* :32-34 */
    return xyz(mul(view, v4(pos.x, pos.y, pos.z, is_position)));
}

} /* namespace orc */
#endif
