/* vrt_detmath.h -- the NUMERIC CONTRACT shared by the HIP product and the CPU oracle.
 *
 * The reference (taichi-dev/voxel-rt2) gets its transcendentals, its f16 conversions and its
 * ti.random() stream from the Taichi runtime (requirements.txt:2), which is neither vendored
 * nor reproducible across backends.  To make "same seed ->
 * same HDR buffer" a testable statement, both sides of the parity test build every
 * non-IEEE-basic operation from this header:
 *
 *   - only + - * / sqrt fma on binary32 (all correctly rounded on x86-64 and on gfx950 with
 *     hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), compiled -ffp-contract=off;
 *   - dm_sin/cos/exp/log/pow/acos/atan2: Cody-Waite reduction + Cephes-style minimax
 *     polynomials, every fused multiply-add written explicitly (dm_fma);
 *   - dm_f32_to_f16 / dm_f16_to_f32: software round-to-nearest-even;
 *   - dm_rng: per-pixel counter-based PCG stream (replaces ti.random(), 24-bit floats in [0,1)).
 *
 * This file contains no rendering algorithm.  It is plain C++ and is compiled unchanged by
 * g++ (oracle, tests) and hipcc (device code).
 */
#ifndef VRT_DETMATH_H
#define VRT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DM_HD __host__ __device__ __forceinline__
#else
#define DM_HD inline
#endif

#define DM_INF (__builtin_inff())
#define DM_NAN (__builtin_nanf(""))
#define DM_PI 3.14159274101257324f      /* f32(pi)  */
#define DM_TWO_PI 6.28318548202514648f  /* f32(2pi) */

DM_HD uint32_t dm_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
DM_HD float dm_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
DM_HD float dm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DM_HD float dm_abs(float x) { return __builtin_fabsf(x); }
DM_HD float dm_floor(float x) { return __builtin_floorf(x); }
DM_HD float dm_sqrt(float x) { return __builtin_sqrtf(x); }
DM_HD float dm_rint(float x) { return __builtin_rintf(x); }
/* min/max with the semantics of llvm.minnum/maxnum as gfx950 executes them (v_min_f32 /
 * v_max_f32, IEEE mode): a NaN operand is ignored, and -0 orders below +0.  On the device this
 * is the single hardware instruction; on the host the same result is spelled out, so both sides
 * agree on every input including NaN and signed zeros (tests/test_detmath_gpu.py checks the
 * instruction against this definition on the special values). */
#if defined(__HIP_DEVICE_COMPILE__)
DM_HD float dm_min(float a, float b) { return __builtin_fminf(a, b); }
DM_HD float dm_max(float a, float b) { return __builtin_fmaxf(a, b); }
#else
DM_HD float dm_min(float a, float b) {
    if (b != b) return a;
    if (a != a) return b;
    if (a < b) return a;
    if (b < a) return b;
    return (__builtin_bit_cast(uint32_t, a) >> 31) ? a : b; /* equal: prefer -0 */
}
DM_HD float dm_max(float a, float b) {
    if (b != b) return a;
    if (a != a) return b;
    if (a > b) return a;
    if (b > a) return b;
    return (__builtin_bit_cast(uint32_t, a) >> 31) ? b : a; /* equal: prefer +0 */
}
#endif
DM_HD float dm_clamp(float x, float lo, float hi) { return dm_min(dm_max(x, lo), hi); }
DM_HD float dm_saturate(float x) { return dm_min(dm_max(x, 0.0f), 1.0f); }
DM_HD bool dm_isnan(float x) { return x != x; }
/* float -> integer casts with the saturating semantics of v_cvt_i32_f32 / v_cvt_u32_f32
 * (NaN -> 0, out of range clamps); plain C casts are undefined there. */
DM_HD int32_t dm_f2i(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)x;
}
DM_HD uint32_t dm_f2u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 4294967295u;
    return (uint32_t)x;
}
DM_HD bool dm_isinf(float x) { return dm_abs(x) == DM_INF; }

/* 2^k * x for integer k, exact unless the result is subnormal/overflows (two-step scaling). */
DM_HD float dm_ldexp(float x, int k) {
    if (k > 127) { x *= dm_u2f(254u << 23); k -= 127; if (k > 127) k = 127; }
    if (k < -126) {
        x *= dm_u2f(1u << 23); k += 126;
        if (k < -126) { x *= dm_u2f(1u << 23); k += 126; if (k < -126) k = -126; }
    }
    return x * dm_u2f((uint32_t)(127 + k) << 23);
}

/* ---- sin / cos ------------------------------------------------------------------------- */
/* r = x - k*pi/2 with a three-term pi/2; |r| <= pi/4 (+ rounding).  Accurate for |x| < ~1e5;
 * deterministic for all finite x below 2^30, NaN otherwise. */
DM_HD float dm_reduce_pio2(float x, int* quadrant) {
    float kf = dm_rint(x * 6.36619746685028076e-01f);
    float r = dm_fma(-kf, 1.57079637050628662e+00f, x);
    r = dm_fma(-kf, -4.37113882867379289e-08f, r);
    r = dm_fma(-kf, -1.71512451000588187e-15f, r);
    *quadrant = (int)kf & 3;
    return r;
}
DM_HD float dm_sin_poly(float r) {
    float z = r * r;
    float p = dm_fma(-1.9515295891e-4f, z, 8.3321608736e-3f);
    p = dm_fma(p, z, -1.6666654611e-1f);
    return dm_fma(p * z, r, r);
}
DM_HD float dm_cos_poly(float r) {
    float z = r * r;
    float p = dm_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    p = dm_fma(p, z, 4.166664568298827e-2f);
    return dm_fma(p * z, z, dm_fma(-0.5f, z, 1.0f));
}
DM_HD float dm_sin(float x) {
    if (!(dm_abs(x) < 1073741824.0f)) return DM_NAN;
    int q;
    float r = dm_reduce_pio2(x, &q);
    float v = (q & 1) ? dm_cos_poly(r) : dm_sin_poly(r);
    return (q & 2) ? -v : v;
}
DM_HD float dm_cos(float x) {
    if (!(dm_abs(x) < 1073741824.0f)) return DM_NAN;
    int q;
    float r = dm_reduce_pio2(x, &q);
    float v = (q & 1) ? dm_sin_poly(r) : dm_cos_poly(r);
    return ((q + 1) & 2) ? -v : v;
}
DM_HD void dm_sincos(float x, float* s, float* c) {
    if (!(dm_abs(x) < 1073741824.0f)) { *s = DM_NAN; *c = DM_NAN; return; }
    int q;
    float r = dm_reduce_pio2(x, &q);
    float sv = dm_sin_poly(r), cv = dm_cos_poly(r);
    float a = (q & 1) ? cv : sv;
    float b = (q & 1) ? sv : cv;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}

/* ---- exp / log / pow ------------------------------------------------------------------- */
DM_HD float dm_exp(float x) {
    if (dm_isnan(x)) return x;
    if (x > 88.7228394f) return DM_INF;
    if (x < -103.972084f) return 0.0f;
    float kf = dm_rint(x * 1.44269502162933350e+00f);
    float r = dm_fma(-kf, 6.93147182464599609e-01f, x);
    r = dm_fma(-kf, -1.90465421212593355e-09f, r);
    float p = dm_fma(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = dm_fma(p, r, 8.3334519073e-3f);
    p = dm_fma(p, r, 4.1665795894e-2f);
    p = dm_fma(p, r, 1.6666665459e-1f);
    p = dm_fma(p, r, 5.0000001201e-1f);
    float y = dm_fma(p * r, r, r) + 1.0f;
    return dm_ldexp(y, (int)kf);
}
DM_HD float dm_log(float x) {
    if (dm_isnan(x) || x < 0.0f) return DM_NAN;
    if (x == 0.0f) return -DM_INF;
    if (x == DM_INF) return x;
    int e = 0;
    uint32_t u = dm_f2u(x);
    if (u < 0x00800000u) { x *= 8388608.0f; u = dm_f2u(x); e = -23; } /* subnormal */
    e += (int)(u >> 23) - 126;
    float m = dm_u2f((u & 0x007fffffu) | 0x3f000000u); /* [0.5, 1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m; }
    float f = m - 1.0f;
    float z = f * f;
    float p = dm_fma(7.0376836292e-2f, f, -1.1514610310e-1f);
    p = dm_fma(p, f, 1.1676998740e-1f);
    p = dm_fma(p, f, -1.2420140846e-1f);
    p = dm_fma(p, f, 1.4249322787e-1f);
    p = dm_fma(p, f, -1.6668057665e-1f);
    p = dm_fma(p, f, 2.0000714765e-1f);
    p = dm_fma(p, f, -2.4999993993e-1f);
    p = dm_fma(p, f, 3.3333331174e-1f);
    float fe = (float)e;
    float y = (p * f) * z;
    y = dm_fma(-2.12194440e-4f, fe, y);
    y = dm_fma(-0.5f, z, y);
    return dm_fma(0.693359375f, fe, f + y);
}
/* x^y for x >= 0 (the only general-exponent uses on this path have non-negative bases:
 * bsdf.py:206, math_utils.py:182, pathtracer.py:660).  Negative x -> NaN. */
DM_HD float dm_pow(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (dm_isnan(x) || dm_isnan(y) || x < 0.0f) return DM_NAN;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : DM_INF;
    if (x == 1.0f) return 1.0f;
    return dm_exp(y * dm_log(x));
}
/* integer-exponent powers written as multiply chains (Schlick x^5, atmos.py:514 x^3) */
DM_HD float dm_pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
DM_HD float dm_pow3(float x) { return (x * x) * x; }

/* ---- acos / atan2 ---------------------------------------------------------------------- */
DM_HD float dm_asin_core(float a) { /* 0 <= a <= 0.5 */
    float z = a * a;
    float p = dm_fma(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = dm_fma(p, z, 4.5470025998e-2f);
    p = dm_fma(p, z, 7.4953002686e-2f);
    p = dm_fma(p, z, 1.6666752422e-1f);
    return dm_fma(p * z, a, a);
}
DM_HD float dm_acos(float x) {
    if (dm_isnan(x) || dm_abs(x) > 1.0f) return DM_NAN;
    float a = dm_abs(x);
    if (a > 0.5f) {
        float s = dm_sqrt(0.5f * (1.0f - a));
        float t = 2.0f * dm_asin_core(s);
        return (x < 0.0f) ? (DM_PI - t) : t;
    }
    float t = dm_asin_core(a);
    t = (x < 0.0f) ? -t : t;
    return 1.57079637050628662e+00f - t;
}
DM_HD float dm_atan_pos(float x) { /* x >= 0 */
    float y0 = 0.0f;
    if (x > 2.414213562373095f) { y0 = 1.57079637050628662e+00f; x = -1.0f / x; }
    else if (x > 0.4142135623730950f) { y0 = 0.785398185253143311f; x = (x - 1.0f) / (x + 1.0f); }
    float z = x * x;
    float p = dm_fma(8.05374449538e-2f, z, -1.38776856032e-1f);
    p = dm_fma(p, z, 1.99777106478e-1f);
    p = dm_fma(p, z, -3.33329491539e-1f);
    return y0 + dm_fma(p * z, x, x);
}
DM_HD float dm_atan2(float y, float x) {
    if (dm_isnan(x) || dm_isnan(y)) return DM_NAN;
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return (y > 0.0f) ? 1.57079637050628662e+00f : -1.57079637050628662e+00f;
    }
    float a = dm_atan_pos(dm_abs(y) / dm_abs(x));
    if (x < 0.0f) a = DM_PI - a;
    return (y < 0.0f) ? -a : a;
}

/* ---- binary16 <-> binary32, round to nearest even ---------------------------------------- */
/* On the device both conversions are the hardware instruction (v_cvt_f16_f32 / v_cvt_f32_f16: round to nearest even,
 * subnormal halves kept -- a HIP kernel runs with f16 denormals on), with the two cases where the instruction's NaN
 * handling differs from the definition below spelled out; tests/test_gpu_parity.py checks every half code and the
 * neighbourhood of every rounding boundary against the host definition. */
DM_HD uint16_t dm_f32_to_f16(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (f != f) return (uint16_t)0x7e00u;
    return __builtin_bit_cast(uint16_t, (_Float16)f);
#else
    uint32_t u = dm_f2u(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)0x7e00u;   /* NaN: one canonical code (0/0 is -NaN on x86, +NaN on gfx950) */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);           /* >= 65520 -> inf */
    if (a < 0x33000001u) return (uint16_t)sign;                        /* <= 2^-25 -> 0 */
    if (a < 0x38800000u) {                                             /* subnormal half */
        uint32_t m = (a & 0x007fffffu) | 0x00800000u;
        int shift = 126 - (int)(a >> 23);                              /* 14..24 */
        uint32_t h = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((a >> 13) - (112u << 10));
    uint32_t rem = a & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
#endif
}
DM_HD float dm_f16_to_f32(uint16_t h) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (((uint32_t)h & 0x7c00u) == 0x7c00u)   /* inf / NaN: payload carried over as below (the instruction would quiet a signalling NaN) */
        return dm_u2f((((uint32_t)h & 0x8000u) << 16) | 0x7f800000u | (((uint32_t)h & 0x3ffu) << 13));
    return (float)__builtin_bit_cast(_Float16, h);
#else
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return dm_u2f(sign);
        float v = (float)m * 5.9604644775390625e-08f;                  /* m * 2^-24, exact */
        return (sign ? -v : v);
    }
    if (e == 31) return dm_u2f(sign | 0x7f800000u | (m << 13));
    return dm_u2f(sign | ((e + 112u) << 23) | (m << 13));
#endif
}
/* value of x after a round trip through binary16 (ti.cast(x, ti.f16) then back) */
DM_HD float dm_round_f16(float x) { return dm_f16_to_f32(dm_f32_to_f16(x)); }

/* ---- counter-based per-pixel random stream ----------------------------------------------- */
typedef struct dm_rng { uint32_t s; } dm_rng;
DM_HD uint32_t dm_pcg_hash(uint32_t v) {
    uint32_t s = v * 747796405u + 2891336453u;
    uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
    return (w >> 22u) ^ w;
}
/* stream = which kernel draws (0 render, 1 spatial reuse, 2 sky precompute, 3 host jitter);
 * index = global pixel / texel index, so results do not depend on how rows are sharded. */
DM_HD dm_rng dm_rng_init(uint32_t seed, uint32_t frame, uint32_t index, uint32_t stream) {
    dm_rng r;
    uint32_t h = dm_pcg_hash(seed ^ (stream * 0x9E3779B9u));
    h = dm_pcg_hash(h + frame);
    r.s = dm_pcg_hash(h + index);
    return r;
}
DM_HD uint32_t dm_rng_u32(dm_rng* r) {
    r->s = r->s * 747796405u + 2891336453u;
    uint32_t w = ((r->s >> ((r->s >> 28u) + 4u)) ^ r->s) * 277803737u;
    return (w >> 22u) ^ w;
}
/* uniform in [0,1) with 24-bit resolution, like ti.random(ti.f32) (Appendix A-3) */
DM_HD float dm_rng_f32(dm_rng* r) { return (float)(dm_rng_u32(r) >> 8) * 5.9604644775390625e-08f; }

#endif /* VRT_DETMATH_H */
