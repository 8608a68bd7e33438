// vrt_sky_kernels.hip -- one-time sky / cloud precompute on the device.
//
// Replaces Atmos.generate_transmittance_lut, get_ray_transmittance, compute_cloud_ambient,
// accumulate_clouds, clouds_scattering, clouds_shadow_od, sample_cloud_density, cloud_phase,
// compute_skybox, atmospheric_scattering, get_density, get_ozone_density, rsi and the phase
// functions (reference renderer/atmos.py:9-31, 134-189, 195-425, 457-528).
//
// One thread per skybox texel; a texel's march is ~80 k inner steps for the sky pass, so the
// kernels are VALU-bound and the only shared data (transmittance LUT 196 KiB, cloud tile 192 KiB)
// is L2-resident.  Python-level constant expressions of the reference are folded in double and
// rounded once, pow(x,1.5) = x*sqrt(x), pow(x,3) = x*x*x (numeric contract).
#include <hip/hip_runtime.h>
#include "vrt_kernels.h"

namespace vrt {

#define PLANET_R 6371e3f
#define ATMOS_H 110e3f
#define CLOUD_H 2000.0f
#define CLOUD_T 340.0f
#define CLOUD_DENSITY 0.27f
#define CLOUD_EXT 0.075f
#define MIE_G 0.75f

struct AtmosConst {
    f3 rayleigh, ozone;
    float mie, mie_ext;
};
__device__ __forceinline__ AtmosConst atmos_const() {
    const double ozone_num = 2.5035422e25 * 0.012588 * 8e-6;
    AtmosConst c;
    c.rayleigh = mk3(0.00000519673f, 0.0000121427f, 0.0000296453f);
    c.mie = 8.6e-6f;
    c.mie_ext = (float)(8.6e-6 * 1.11);
    c.ozone = mk3((float)(4.51103766177301e-21 * 0.0001 * ozone_num), (float)(3.2854797958699e-21 * 0.0001 * ozone_num),
                  (float)(1.96774621921165e-22 * 0.0001 * ozone_num));
    return c;
}
__device__ __forceinline__ f3 extinc_mul(const AtmosConst& c, f3 d) {
    return mk3(c.rayleigh.x * d.x + c.mie_ext * d.y + c.ozone.x * d.z, c.rayleigh.y * d.x + c.mie_ext * d.y + c.ozone.y * d.z,
               c.rayleigh.z * d.x + c.mie_ext * d.y + c.ozone.z * d.z);
}
__device__ __forceinline__ f2 rsi(f3 pos, f3 dir, float r) {  // atmos.py:9-15
    float b = dot3(pos, dir);
    float discr = dm_sqrt(b * b - dot3(pos, pos) + r * r);
    if (discr < 0.0f) return mk2(-1.0f, -1.0f);
    return mk2(-b + -discr, -b + discr);
}
__device__ __forceinline__ float rayleigh_phase(float c) { return (float)(3.0 / (16.0 * 3.141592653589793)) * (1.0f + c * c); }
__device__ __forceinline__ float mie_phase(float c, float g) {
    float x = 1.0f + g * g - 2.0f * g * c;
    return (1.0f - g * g) / ((float)(4.0 * 3.141592653589793) * (x * dm_sqrt(x)));
}
__device__ __forceinline__ f3 unit_vec(f2 r) {  // atmos.py:27-31
    r.x *= DM_TWO_PI;
    r.y = r.y * 2.0f - 1.0f;
    float s = dm_sqrt(1.0f - r.y * r.y);
    float sn, cs;
    dm_sincos(r.x, &sn, &cs);
    return norm3(mk3(sn * s, cs * s, r.y));
}
__device__ __forceinline__ float elevation_of(f3 p) { return dm_sqrt(p.x * p.x + p.y * p.y + p.z * p.z) - PLANET_R; }
__device__ __forceinline__ float ozone_density(float h) {  // atmos.py:500-518
    float h_km = h * 0.001f;
    float rel = h_km - 25.0f;
    rel = rel * rel;
    float d = 0.625f * dm_exp(-rel / 49.0f);
    d += 0.375f * dm_exp(-rel / 256.0f);
    d += dm_max(0.0f, -0.000015f * dm_pow3(h_km - 15.0f));
    return d * 4.0f;
}
__device__ __forceinline__ f3 density_at(float h) {  // atmos.py:520-523
    h = dm_max(h, 0.0f);
    return mk3(dm_exp(-h / 8500.0f), dm_exp(-h / 1200.0f), ozone_density(h));
}
__device__ __forceinline__ f3 read_trans_lut(const uint16_t* lut, float cos_theta, float h) {  // atmos.py:457-460
    int sx = dm_f2i(dm_clamp((cos_theta * 0.5f + 0.5f) * 256.0f, 0.0f, 255.0f));
    int sy = dm_f2i(dm_clamp((h / ATMOS_H) * 128.0f, 0.0f, 127.0f));
    const uint16_t* p = lut + (sx * 128 + sy) * 3;
    return mk3(dm_f16_to_f32(p[0]), dm_f16_to_f32(p[1]), dm_f16_to_f32(p[2]));
}

// atmos.py:475-498
__device__ f3 ray_transmittance(const AtmosConst& c, f3 pos, f3 dir) {
    float step_delta = rsi(pos, dir, (float)(6371e3 + 110e3)).y * (1.0f / 128.0f);
    f3 step = dir * step_delta;
    pos = pos + step * (0.5f * (dm_max(dir.y, 0.0f) * 0.5f + 0.5f));
    f3 od = mk3(0.0f);
    for (int s = 0; s < 128; s++) {
        od = od + density_at(elevation_of(pos)) * step_delta;
        pos = pos + step;
    }
    od = extinc_mul(c, od);
    f3 T = mk3(dm_exp(-od.x), dm_exp(-od.y), dm_exp(-od.z));
    if (rsi(pos, dir, PLANET_R).x > 0.0f) T = T * 0.0f;
    return T;
}
// atmos.py:462-473
__global__ void k_trans_lut(uint16_t* lut) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 256 * 128) return;
    int x = i >> 7, y = i & 127;
    const AtmosConst c = atmos_const();
    float cos_theta = ((float)x / 256.0f) * 2.0f - 1.0f;
    float h = ATMOS_H * (float)y / 128.0f;
    float sin_theta = dm_sin(dm_acos(cos_theta));
    const f3 T = ray_transmittance(c, mk3(0.0f, PLANET_R + h, 0.0f), mk3(sin_theta, cos_theta, 0.0f));
    lut[3 * i] = dm_f32_to_f16(T.x);
    lut[3 * i + 1] = dm_f32_to_f16(T.y);
    lut[3 * i + 2] = dm_f32_to_f16(T.z);
}

struct SunArgs { f3 dir, col; float cos_max; f3 bx, by; };

// atmos.py:355-425; DEPTH is ti.template(): depth 2 contributes (0, 1) without marching
template <int DEPTH>
__device__ void atmospheric_scattering(const SkyPrecompute& sp, const AtmosConst& c, const SunArgs& sun, f3 origin, f3 dir, int steps,
                                       dm_rng& rng, f3& in_scatter, f3& trans) {
    float fsteps = 1.0f / (float)steps;
    f2 air = rsi(origin, dir, (float)(6371e3 + 110e3));
    f2 planet = rsi(origin, dir, PLANET_R);
    air.y = (planet.x > 0.0f) ? dm_min(air.y, planet.x) : air.y;
    float step_delta = (air.y - dm_max(air.x, 0.0f)) * fsteps;
    f3 step = dir * step_delta;
    f3 pos = origin + step * 0.5f;
    trans = mk3(1.0f);
    in_scatter = mk3(0.0f);
    if constexpr (DEPTH <= 1) {
        for (int i = 0; i < steps; i++) {
            float h = elevation_of(pos);
            f3 density = density_at(h);
            f3 step_od = extinc_mul(c, density * step_delta);
            f3 step_T = sat3(mk3(dm_exp(-step_od.x), dm_exp(-step_od.y), dm_exp(-step_od.z)));
            f3 visible = trans * sat3((mk3(1.0f) - step_T) / step_od);
            for (int j = 0; j < 8; j++) {
                f3 sd = cone_dir(sun.cos_max, sun.dir, sun.bx, sun.by, rng);
                float ct = dot3(dir, sd);
                float ph_r = rayleigh_phase(ct), ph_m = mie_phase(ct, MIE_G);
                f3 sun_T = read_trans_lut(sp.trans_lut, dot3(norm3(pos), sd), h);
                in_scatter = in_scatter + c.rayleigh * sun.col * sun_T * visible * ph_r * density.x * step_delta / 8.0f;
                in_scatter = in_scatter + c.mie * sun.col * sun_T * visible * ph_m * density.y * step_delta / 8.0f;
            }
            const float ms_energy = 5.3f;
            for (int j = 0; j < 8; j++) {
                f3 sd = unit_vec(mk2(((float)j + 0.5f) / 8.0f, frac1((float)j * 1.618033988749f)));
                float ph_m = mie_phase(dot3(dir, sd), MIE_G);
                f3 amb, amb_T;
                atmospheric_scattering<DEPTH + 1>(sp, c, sun, pos, sd, 5, rng, amb, amb_T);
                in_scatter = in_scatter + ms_energy * c.rayleigh * amb * visible * density.x * step_delta / 8.0f;
                in_scatter = in_scatter + ms_energy * c.mie * amb * visible * ph_m * density.y * step_delta / 8.0f;
            }
            trans = trans * step_T;
            pos = pos + step;
        }
        if (planet.x > 0.0f) trans = trans * 0.0f;
    }
}

__device__ __forceinline__ SunArgs make_sun(f3 dir, f3 col, float cos_max) {
    SunArgs s;
    s.dir = dir; s.col = col; s.cos_max = cos_max;
    ortho_basis(dir, s.bx, s.by);
    return s;
}
__device__ __forceinline__ f3 sky_cam_pos() { return mk3(0.0f, (float)(6371e3 + 0e3 + 1e3), 0.0f); }

// atmos.py:134-138
__global__ void k_cloud_ambient(SkyPrecompute sp, f3 sun_dir, f3 sun_col, float sun_cos) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const AtmosConst c = atmos_const();
    const SunArgs sun = make_sun(sun_dir, sun_col, sun_cos);
    dm_rng rng = dm_rng_init(sp.seed, 0x1000u, 0u, 2u);
    f3 amb, T;
    atmospheric_scattering<0>(sp, c, sun, sky_cam_pos() + mk3(0.0f, CLOUD_H, 0.0f), mk3(0.0f, 1.0f, 0.0f), 64, rng, amb, T);
    sp.cloud_ambient[0] = amb.x; sp.cloud_ambient[1] = amb.y; sp.cloud_ambient[2] = amb.z;
}

// atmos.py:195-224
__device__ float cloud_density_at(const SkyPrecompute& sp, f3 p) {
    const float tile = 29000.0f;
    p.x += (float)(29000.0 * 0.65);
    p.z += (float)(29000.0 * 0.65);
    float ux = (p.x - tile * dm_floor(p.x / tile)) / tile;
    float uz = (p.z - tile * dm_floor(p.z / tile)) / tile;
    int cx = dm_f2i(ux * 256.0f), cy = dm_f2i(uz * 256.0f);
    if (cx < 0) cx += 29000;
    if (cy < 0) cy += 29000;
    if (cx > 255) cx = 255;
    if (cy > 255) cy = 255;
    float rel_h = len3(p) - PLANET_R - 0.0f;
    const uint8_t* t = sp.cloud_tex + (cx * 256 + cy) * 3;
    f3 tex = mk3((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f);
    if (tex.x < 0.7f) tex.x = 0.0f;
    if (tex.y < 0.7f) tex.y = 0.0f;
    if (tex.z < 0.7f) tex.z = 0.0f;
    float cloud = 0.0f;
    if (rel_h < CLOUD_H + CLOUD_T * 0.65f) cloud += tex.x;
    else cloud += tex.y;
    bool in_layer = rel_h > CLOUD_H && rel_h < CLOUD_H + CLOUD_T;
    return in_layer ? CLOUD_DENSITY * tex.z * cloud : 0.0f;
}
// atmos.py:231-260
__device__ float cloud_shadow_od(const SkyPrecompute& sp, f3 origin, f3 dir, float dither) {
    float step_delta = 24.0f / 8.0f;
    float od = 0.0f;
    f3 pos = origin;
    f3 step = dir * step_delta;
    for (int i = 0; i < 8; i++) {
        step = step * 1.6f;
        step_delta *= 1.6f;
        f3 dp = pos + step * dither;
        float rel_h = len3(dp) - PLANET_R - 0.0f;
        if (rel_h < CLOUD_H || rel_h > CLOUD_H + CLOUD_T) continue;
        od += cloud_density_at(sp, dp) * step_delta;
        pos = pos + step;
    }
    return od;
}
__device__ __forceinline__ float cloud_phase(float ct, float an) {  // atmos.py:262-267
    float peak = mie_phase(ct, 0.92f * an), front = mie_phase(ct, 0.4f * an), back = mie_phase(ct, -0.55f * an);
    return lerp1(lerp1(front, back, 0.5f), peak, 0.15f);
}

// atmos.py:269-349
__device__ void clouds_scattering(const SkyPrecompute& sp, const SunArgs& sun, f3 ambient, f3 origin, f3 dir, float dither, dm_rng& rng,
                                  f3& in_scatter, float& transmittance, float& weighted_dist) {
    const float fsteps = 1.0f / 32.0f;
    float bottom = rsi(origin, dir, (float)(6371e3 + 0e3 + 2000.0)).y;
    float top = rsi(origin, dir, (float)(6371e3 + 0e3 + 2000.0 + 340.0)).y;
    transmittance = 1.0f;
    in_scatter = mk3(0.0f);
    float weight_sum = 0.0f;
    weighted_dist = 0.0f;
    f3 start = origin + dir * bottom;
    float step_delta = (top - bottom) * fsteps;
    f3 step = dir * step_delta;
    f3 pos = start + step * dither;
    float traveled = len3(start - origin);
    for (int i = 0; i < 32; i++) {
        float density = cloud_density_at(sp, pos);
        if (density <= 0.0f || transmittance <= 1e-4f) {
            pos = pos + step;
            traveled += step_delta;
            weighted_dist += traveled * transmittance;
            weight_sum += transmittance;
            continue;
        }
        float step_od = CLOUD_EXT * density * step_delta;
        float step_T = dm_saturate(dm_exp(-step_od));
        float step_w = (1.0f - step_T) / CLOUD_EXT;
        float visible = transmittance * step_w;
        for (int j = 0; j < 8; j++) {
            f3 sd = cone_dir(sun.cos_max, sun.dir, sun.bx, sun.by, rng);
            float ct = dot3(dir, sd);
            float sun_od = cloud_shadow_od(sp, pos, sd, dither);
            f3 sun_T = read_trans_lut(sp.trans_lut, dot3(norm3(pos), sd), elevation_of(pos));
            float an = 1.0f;
            for (int k = 0; k < 4; k++) {
                float phase = cloud_phase(ct, an);
                in_scatter = in_scatter + visible * an * CLOUD_EXT * phase * dm_exp(-sun_od * CLOUD_EXT * an) * sun_T * sun.col / 8.0f;
                an *= 0.5f;
            }
        }
        float amb_od = cloud_shadow_od(sp, pos, mk3(0.0f, 1.0f, 0.0f), dither);
        float an = 1.0f;
        for (int k = 0; k < 4; k++) {
            in_scatter = in_scatter + visible * an * CLOUD_EXT / (float)(4.0 * 3.141592653589793) * dm_exp(-amb_od * CLOUD_EXT * an) * ambient;
            an *= 0.5f;
        }
        transmittance *= step_T;
        pos = pos + step;
        traveled += step_delta;
        weighted_dist += traveled * transmittance;
        weight_sum += transmittance;
    }
    weighted_dist /= weight_sum;
}

// atmos.py:140-157
__global__ __launch_bounds__(256) void k_sky_clouds(SkyPrecompute sp, f3 sun_dir, f3 sun_col, float sun_cos, int max_samples, uint32_t pass, int u0, int u1) {
    const int v = blockIdx.x * 64 + (threadIdx.x & 63);
    const int u = u0 + blockIdx.y * 4 + (threadIdx.x >> 6);   // columns [u0, u1): a texel depends on no other (atmos.py:140-157)
    if (u >= u1 || v >= sp.res) return;
    const SunArgs sun = make_sun(sun_dir, sun_col, sun_cos);
    const f3 ambient = mk3(sp.cloud_ambient[0], sp.cloud_ambient[1], sp.cloud_ambient[2]);
    dm_rng rng = dm_rng_init(sp.seed, pass, (uint32_t)(u * sp.res + v), 2u);
    const f3 dir = unproject_sky(sp.fres, mk2(((float)u + 0.5f) * sp.fres, ((float)v + 0.5f) * sp.fres));
    const float dither = dm_rng_f32(&rng);
    f3 in_scatter;
    float transmittance, weighted_dist;
    clouds_scattering(sp, sun, ambient, sky_cam_pos(), dir, dither, rng, in_scatter, transmittance, weighted_dist);

    in_scatter = in_scatter * 1.2f;
    const float fmax = 1.0f / (float)max_samples;
    const int i = u * sp.res + v;
    float* sc = sp.scattering + 3 * i;
    float* tr = sp.transmittance + 3 * i;
    f3 add = in_scatter * fmax;
    sc[0] += add.x; sc[1] += add.y; sc[2] += add.z;
    tr[0] += dm_saturate(transmittance) * fmax;
    tr[1] += weighted_dist * fmax;
}

// atmos.py:159-189
__global__ __launch_bounds__(256) void k_sky_slice(SkyPrecompute sp, f3 sun_dir, f3 sun_col, float sun_cos, int u0, int u1) {
    const int v = blockIdx.x * 64 + (threadIdx.x & 63);
    const int u = u0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= u1 || v >= sp.res) return;
    const AtmosConst c = atmos_const();
    const SunArgs sun = make_sun(sun_dir, sun_col, sun_cos);
    dm_rng rng = dm_rng_init(sp.seed, 0x2000u, (uint32_t)(u * sp.res + v), 2u);
    const f3 dir = unproject_sky(sp.fres, mk2(((float)u + 0.5f) * sp.fres, ((float)v + 0.5f) * sp.fres));
    const int i = u * sp.res + v;
    float* sc = sp.scattering + 3 * i;
    float* tr = sp.transmittance + 3 * i;
    const f3 cloud_in = mk3(sc[0], sc[1], sc[2]);
    const float cloud_T = tr[0], cloud_dist = tr[1];
    const f3 origin = sky_cam_pos();
    f3 total, T_total, from, T_from;
    atmospheric_scattering<0>(sp, c, sun, origin, dir, 64, rng, total, T_total);
    atmospheric_scattering<0>(sp, c, sun, origin + dir * dm_max(cloud_dist, 0.0f), dir, 64, rng, from, T_from);
    f3 T_to = T_total / T_from;
    f3 result = total;
    if (sp.use_clouds == 1) {
        result = result - from * sat3(T_to * dm_max(1.0f - cloud_T, 0.0f));
        result = result + cloud_in * sat3(T_to);
    }
    f3 tout = T_total * cloud_T;
    sc[0] = result.x; sc[1] = result.y; sc[2] = result.z;
    tr[0] = tout.x; tr[1] = tout.y; tr[2] = tout.z;
}

// Test hook (vrt_sky_probe): one function of this file per row of arguments, as oracle/orc_api.cpp's orc_unit_atmos lays them out.
__global__ void k_sky_probe(SkyPrecompute sp, int op, int n, const float* in, int in_stride, float* out, int out_stride, f3 ambient) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* a = in + (size_t)k * in_stride;
    float* o = out + (size_t)k * out_stride;
    const AtmosConst c = atmos_const();
    const f3 p = mk3(a[0], a[1], a[2]), d = mk3(a[3], a[4], a[5]);
    switch (op) {
        case 0: { const f2 r = rsi(p, d, a[6]); o[0] = r.x; o[1] = r.y; break; }
        case 1: o[0] = ozone_density(a[0]); break;
        case 2: { const f3 r = density_at(a[0]); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
        case 3: o[0] = cloud_phase(a[0], a[1]); break;
        case 4: o[0] = cloud_density_at(sp, p); break;
        case 5: o[0] = cloud_shadow_od(sp, p, d, a[6]); break;
        case 6: { const f3 r = ray_transmittance(c, p, d); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
        case 7: {
            const SunArgs sun = make_sun(mk3(a[6], a[7], a[8]), mk3(a[9], a[10], a[11]), a[12]);
            dm_rng rng = dm_rng_init(sp.seed, 0x3000u, (uint32_t)a[14], 2u);
            f3 sc;
            float tr, dist;
            clouds_scattering(sp, sun, ambient, p, d, a[13], rng, sc, tr, dist);
            o[0] = sc.x; o[1] = sc.y; o[2] = sc.z; o[3] = tr; o[4] = dist;
            break;
        }
        case 8: case 9: {
            const SunArgs sun = make_sun(mk3(a[6], a[7], a[8]), mk3(a[9], a[10], a[11]), a[12]);
            dm_rng rng = dm_rng_init(sp.seed, 0x3000u, (uint32_t)a[14], 2u);
            f3 sc, tr;
            if (op == 8) atmospheric_scattering<0>(sp, c, sun, p, d, (int)a[13], rng, sc, tr);
            else atmospheric_scattering<1>(sp, c, sun, p, d, (int)a[13], rng, sc, tr);
            o[0] = sc.x; o[1] = sc.y; o[2] = sc.z; o[3] = tr.x; o[4] = tr.y; o[5] = tr.z;
            break;
        }
    }
}

#define VRT_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

hipError_t launch_sky_probe(hipStream_t st, const SkyPrecompute& sp, int op, int n, const float* in, int in_stride, float* out, int out_stride, f3 ambient) {
    hipLaunchKernelGGL(k_sky_probe, dim3((n + 63) / 64), dim3(64), 0, st, sp, op, n, in, in_stride, out, out_stride, ambient);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_sky_prepare(hipStream_t st, const SkyPrecompute& sp, f3 sun_dir, f3 sun_col, float sun_cos) {
    hipLaunchKernelGGL(k_trans_lut, dim3(256 * 128 / 256), dim3(256), 0, st, sp.trans_lut);
    VRT_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_cloud_ambient, dim3(1), dim3(64), 0, st, sp, sun_dir, sun_col, sun_cos);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_sky_clouds(hipStream_t st, const SkyPrecompute& sp, f3 sun_dir, f3 sun_col, float sun_cos, int max_samples, uint32_t pass, int u0, int u1) {
    if (u1 <= u0) return hipSuccess;
    dim3 g((sp.res + 63) / 64, (u1 - u0 + 3) / 4), b(256);
    hipLaunchKernelGGL(k_sky_clouds, g, b, 0, st, sp, sun_dir, sun_col, sun_cos, max_samples, pass, u0, u1);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}
hipError_t launch_sky_slice(hipStream_t st, const SkyPrecompute& sp, f3 sun_dir, f3 sun_col, float sun_cos, int u0, int u1) {
    if (u1 <= u0) return hipSuccess;
    dim3 g((sp.res + 63) / 64, (u1 - u0 + 3) / 4), b(256);
    hipLaunchKernelGGL(k_sky_slice, g, b, 0, st, sp, sun_dir, sun_col, sun_cos, u0, u1);
    VRT_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace vrt
