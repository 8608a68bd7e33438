/* orc_atmos.h -- ORACLE (test infrastructure, not product code).
 *
 * Literal CPU restatement of /root/reference/renderer/atmos.py: transmittance LUT, cloud and sky
 * precompute, skybox projection and lookup.  The skybox resolution (3840 in the reference,
 * atmos.py:66-67) is a parameter so tests can run small tables.  Python-level constant
 * expressions are folded in double and rounded to f32 once, as Taichi does with captured
 * Python scalars.  pow(x, 1.5) and pow(x, 3.0) are x*sqrt(x) and x*x*x.
 *
 * Deviations on undefined behaviour: cloud-texture coordinates are clamped to 0..255
 * (atmos.py:200 can produce 256 when mod() rounds to the tile size); casts of NaN give 0.
 */
#ifndef ORC_ATMOS_H
#define ORC_ATMOS_H

#include <vector>
#include "orc_math.h"

namespace orc {

/* atmos.py:9-15 */
inline V2 rsi(V3 pos, V3 dir, float r) {
    float b = dot(pos, dir);
    float discr = b * b - dot(pos, pos) + r * r;
    discr = dm_sqrt(discr);
    if (discr < 0.0f) return v2(-1.0f, -1.0f);
    return v2(-b + -discr, -b + discr);
}
/* atmos.py:18-20 */
inline float rayleigh_phase(float cos_theta) {
    return (float)(3.0 / (16.0 * 3.141592653589793)) * (1.0f + cos_theta * cos_theta);
}
/* atmos.py:22-25 */
inline float mie_phase(float cos_theta, float g) {
    float x = 1.0f + g * g - 2.0f * g * cos_theta;
    return (1.0f - g * g) / ((float)(4.0 * 3.141592653589793) * (x * dm_sqrt(x)));
}
/* atmos.py:27-31 */
inline V3 get_unit_vec(V2 rand) {
    rand.x *= DM_TWO_PI;
    rand.y = rand.y * 2.0f - 1.0f;
    float s = dm_sqrt(1.0f - rand.y * rand.y);
    V2 ground = v2(dm_sin(rand.x) * s, dm_cos(rand.x) * s);
    return normalized(v3(ground.x, ground.y, rand.y));
}

struct Atmos {
    /* atmos.py:36-83 */
    V3 rayleigh_coeff, ozone_coeff;
    float mie_coeff;
    float extinc[3][3]; /* extinc[c][k]: channel c, species k (rayleigh, mie*1.11, ozone) */
    float scale_height_rayl = 8500.0f, scale_height_mie = 1200.0f;
    float mie_g = 0.75f;
    float planet_r = 6371e3f, planet_r_offset = 0.0f, atmos_height = 110e3f;
    float cloud_height = 2000.0f, cloud_thickness = 340.0f, cloud_density = 0.27f, cloud_extinc = 0.075f,
          cloud_scatter = 0.075f;
    V3 cloud_ambient = V3{0, 0, 0};
    V3 cam_pos;
    int use_clouds = 0;
    int res = 0;
    float fres = 0.0f;
    std::vector<uint16_t> trans_LUT;        /* [256][128][3] binary16 */
    std::vector<V3> skybox_scattering;      /* [res][res] indexed [u][v] */
    std::vector<V3> skybox_transmittance;
    std::vector<uint8_t> cloud_tex;         /* [256][256][3] indexed [x][y] (atmos.py:80, Appendix A-9) */
    uint32_t seed = 0;
    uint32_t cloud_pass = 0;

    void init(int sky_res, uint32_t seed_) {
        const double air = 2.5035422e25, ozone_peak = 8e-6;
        const double ozone_num = air * 0.012588 * ozone_peak;
        const double ocs[3] = {4.51103766177301e-21 * 0.0001, 3.2854797958699e-21 * 0.0001, 1.96774621921165e-22 * 0.0001};
        rayleigh_coeff = v3(0.00000519673f, 0.0000121427f, 0.0000296453f);
        mie_coeff = 8.6e-6f;
        ozone_coeff = v3((float)(ocs[0] * ozone_num), (float)(ocs[1] * ozone_num), (float)(ocs[2] * ozone_num));
        for (int c = 0; c < 3; c++) {
            extinc[c][0] = rayleigh_coeff[c];
            extinc[c][1] = (float)(8.6e-6 * 1.11);
            extinc[c][2] = ozone_coeff[c];
        }
        cam_pos = v3(0.0f, (float)(6371e3 + 0e3 + 1e3), 0.0f);
        res = sky_res;
        fres = (float)(1.0 / (double)sky_res);
        seed = seed_;
        trans_LUT.assign(256 * 128 * 3, 0);
        skybox_scattering.assign((size_t)res * res, v3(0.0f));
        skybox_transmittance.assign((size_t)res * res, v3(0.0f));
        cloud_tex.assign(256 * 256 * 3, 0);
        cloud_pass = 0;
    }
    V3 extinc_mul(V3 d) const {
        return v3(extinc[0][0] * d.x + extinc[0][1] * d.y + extinc[0][2] * d.z,
                  extinc[1][0] * d.x + extinc[1][1] * d.y + extinc[1][2] * d.z,
                  extinc[2][0] * d.x + extinc[2][1] * d.y + extinc[2][2] * d.z);
    }

    /* atmos.py:525-527 */
    float get_elevation(V3 pos) const { return dm_sqrt(pos.x * pos.x + pos.y * pos.y + pos.z * pos.z) - planet_r; }
    /* atmos.py:500-518 */
    float get_ozone_density(float h) const {
        float h_km = h * 0.001f;
        float rel = h_km - 25.0f;
        rel = rel * rel;
        float d = 0.625f * dm_exp(-rel / 49.0f);
        d += 0.375f * dm_exp(-rel / 256.0f);
        d += dm_max(0.0f, -0.000015f * dm_pow3(h_km - 15.0f));
        return d * 4.0f;
    }
    /* atmos.py:520-523 */
    V3 get_density(float h) const {
        h = dm_max(h, 0.0f);
        return v3(dm_exp(-h / scale_height_rayl), dm_exp(-h / scale_height_mie), get_ozone_density(h));
    }
    /* atmos.py:457-460 */
    V3 read_trans_lut(float cos_theta, float h) const {
        int sx = dm_f2i(dm_clamp((cos_theta * 0.5f + 0.5f) * 256.0f, 0.0f, 255.0f));
        int sy = dm_f2i(dm_clamp((h / atmos_height) * 128.0f, 0.0f, 127.0f));
        const uint16_t* p = &trans_LUT[((size_t)sx * 128 + sy) * 3];
        return v3(dm_f16_to_f32(p[0]), dm_f16_to_f32(p[1]), dm_f16_to_f32(p[2]));
    }
    /* atmos.py:475-498 */
    V3 get_ray_transmittance(V3 ray_pos, V3 ray_dir) const {
        const int steps = 128;
        const float fsteps = 1.0f / 128.0f;
        float step_delta = rsi(ray_pos, ray_dir, (float)(6371e3 + 110e3)).y * fsteps;
        V3 ray_step = ray_dir * step_delta;
        ray_pos = ray_pos + ray_step * (0.5f * (dm_max(ray_dir.y, 0.0f) * 0.5f + 0.5f));
        V3 od = v3(0.0f);
        for (int i = 0; i < steps; i++) {
            float elevation = get_elevation(ray_pos);
            V3 densities = get_density(elevation);
            od += densities * step_delta;
            ray_pos += ray_step;
        }
        od = extinc_mul(od);
        V3 transmittance = v3(dm_exp(-od.x), dm_exp(-od.y), dm_exp(-od.z));
        if (rsi(ray_pos, ray_dir, planet_r).x > 0.0f) transmittance *= 0.0f;
        return transmittance;
    }
    /* atmos.py:462-473 */
    void generate_transmittance_lut() {
        for (int x = 0; x < 256; x++)
            for (int y = 0; y < 128; y++) {
                float cos_theta = ((float)x / 256.0f) * 2.0f - 1.0f;
                float h = atmos_height * (float)y / 128.0f;
                float theta = dm_acos(cos_theta);
                float sin_theta = dm_sin(theta);
                V3 ray_dir = v3(sin_theta, cos_theta, 0.0f);
                V3 ray_pos = v3(0.0f, planet_r + h, 0.0f);
                V3 t = get_ray_transmittance(ray_pos, ray_dir);
                uint16_t* p = &trans_LUT[((size_t)x * 128 + y) * 3];
                p[0] = dm_f32_to_f16(t.x);
                p[1] = dm_f32_to_f16(t.y);
                p[2] = dm_f32_to_f16(t.z);
            }
    }

    /* atmos.py:428-440 */
    V2 project_sky(V3 ray_dir) const {
        V2 projected_dir = normalized(v2(ray_dir.x, ray_dir.z));
        const float horizon_angle = (float)(3.141592653589793 * 0.5);
        float azimuth = PI + dm_atan2(projected_dir.x, -projected_dir.y);
        float elevation = horizon_angle - dm_acos(ray_dir.y);
        V2 coord;
        coord.x = azimuth / DM_TWO_PI;
        coord.y = 0.5f + 0.5f * sign(elevation) * dm_sqrt((float)(2.0 / 3.141592653589793) * dm_abs(elevation));
        return v2(coord.x * (1.0f - fres) + 0.5f * fres, coord.y * (1.0f - fres) + 0.5f * fres);
    }
    /* atmos.py:442-455 */
    V3 unproject_sky(V2 uv) const {
        V2 coord = v2((uv.x - 0.5f * fres) / (1.0f - 1.0f * fres), (uv.y - 0.5f * fres) / (1.0f - 1.0f * fres));
        coord.y = (coord.y < 0.5f) ? -sqr(1.0f - 2.0f * coord.y) : sqr(2.0f * coord.y - 1.0f);
        float azimuth = coord.x * 2.0f * PI - PI;
        float elevation = coord.y * 0.5f * PI;
        float cos_elevation = dm_cos(elevation), sin_elevation = dm_sin(elevation);
        float cos_azimuth = dm_cos(azimuth), sin_azimuth = dm_sin(azimuth);
        return normalized(v3(cos_elevation * sin_azimuth, sin_elevation, -cos_elevation * cos_azimuth));
    }

    V3 bilinear(const std::vector<V3>& tab, int ix, int iy, float fx, float fy) const {
        int ix1 = (ix + 1) % res, iy1 = (iy + 1) % res;
        V3 bl = tab[(size_t)ix * res + iy], br = tab[(size_t)ix1 * res + iy];
        V3 tl = tab[(size_t)ix * res + iy1], tr = tab[(size_t)ix1 * res + iy1];
        return mix(mix(bl, br, fx), mix(tl, tr, fx), fy);
    }
    void lookup_coords(V2 texcoord, int* ix, int* iy, float* fx, float* fy) const {
        float fcx = texcoord.x * (float)res - 0.5f, fcy = texcoord.y * (float)res - 0.5f;
        *ix = dm_f2i(fcx);
        *iy = dm_f2i(fcy);
        /* clamp: the reference indexes the table unchecked; only NaN directions get here */
        if (*ix < 0) *ix = 0; if (*ix > res - 1) *ix = res - 1;
        if (*iy < 0) *iy = 0; if (*iy > res - 1) *iy = res - 1;
        *fx = fract(fcx);
        *fy = fract(fcy);
    }
    /* atmos.py:94-115 */
    void sample_skybox(V3 ray_dir, dm_rng* rng, V3* scatt, V3* trans) const {
        float r0 = dm_rng_f32(rng), r1 = dm_rng_f32(rng), r2 = dm_rng_f32(rng);
        V2 texcoord = project_sky(normalized(ray_dir + v3(r0, r1, r2) * 0.0015f));
        int ix, iy;
        float fx, fy;
        lookup_coords(texcoord, &ix, &iy, &fx, &fy);
        *scatt = bilinear(skybox_scattering, ix, iy, fx, fy);
        *trans = bilinear(skybox_transmittance, ix, iy, fx, fy);
    }
    /* atmos.py:117-131 */
    V3 sample_skybox_transmittance(V3 ray_dir) const {
        V2 texcoord = project_sky(ray_dir);
        int ix, iy;
        float fx, fy;
        lookup_coords(texcoord, &ix, &iy, &fx, &fy);
        return bilinear(skybox_transmittance, ix, iy, fx, fy);
    }

    /* ---- clouds -------------------------------------------------------------------------- */
    /* atmos.py:195-224 */
    float sample_cloud_density(V3 ray_pos) const {
        const float tile_size = 29000.0f;
        ray_pos.x += (float)(29000.0 * 0.65);
        ray_pos.z += (float)(29000.0 * 0.65);
        float ux = (ray_pos.x - tile_size * dm_floor(ray_pos.x / tile_size)) / tile_size;
        float uz = (ray_pos.z - tile_size * dm_floor(ray_pos.z / tile_size)) / tile_size;
        int cx = dm_f2i(ux * 256.0f), cy = dm_f2i(uz * 256.0f);
        if (cx < 0) cx += 29000;
        if (cy < 0) cy += 29000;
        if (cx > 255) cx = 255;
        if (cy > 255) cy = 255;
        float relative_height = length(ray_pos) - planet_r - planet_r_offset;
        const uint8_t* t = &cloud_tex[((size_t)cx * 256 + cy) * 3];
        V3 tex = v3((float)t[0] / 255.0f, (float)t[1] / 255.0f, (float)t[2] / 255.0f);
        if (tex.x < 0.7f) tex.x = 0.0f;
        if (tex.y < 0.7f) tex.y = 0.0f;
        if (tex.z < 0.7f) tex.z = 0.0f;
        float cloud = 0.0f;
        if (relative_height < cloud_height + cloud_thickness * 0.65f) cloud += tex.x;
        else cloud += tex.y;
        float coverage = tex.z;
        bool in_layer = relative_height > cloud_height && relative_height < cloud_height + cloud_thickness;
        return in_layer ? cloud_density * coverage * cloud : 0.0f;
    }
    /* atmos.py:231-260 */
    float clouds_shadow_od(V3 ray_origin, V3 ray_dir, float dither) const {
        const int steps = 8;
        const float exponent = 1.6f;
        float step_delta = 24.0f / (float)steps;
        float od = 0.0f;
        V3 ray_pos = ray_origin;
        V3 ray_step = ray_dir * step_delta;
        for (int i = 0; i < steps; i++) {
            ray_step *= exponent;
            step_delta *= exponent;
            V3 dithered = ray_pos + ray_step * dither;
            float relative_height = length(dithered) - planet_r - planet_r_offset;
            if (relative_height < cloud_height || relative_height > cloud_height + cloud_thickness) continue;
            od += sample_cloud_density(dithered) * step_delta;
            ray_pos += ray_step;
        }
        return od;
    }
    /* atmos.py:262-267 */
    float cloud_phase(float cos_theta, float an) const {
        float peak = mie_phase(cos_theta, 0.92f * an);
        float front = mie_phase(cos_theta, 0.4f * an);
        float back = mie_phase(cos_theta, -0.55f * an);
        return mix(mix(front, back, 0.5f), peak, 0.15f);
    }
    /* atmos.py:269-349 */
    void clouds_scattering(V3 ray_origin, V3 ray_dir, V3 sun_dir, V3 sun_col, float sun_cos, float dither, dm_rng* rng,
                           V3* in_scatter_out, float* transmittance_out, float* dist_out) const {
        const int steps = 32;
        const float fsteps = 1.0f / (float)steps;
        float bottom = rsi(ray_origin, ray_dir, (float)(6371e3 + 0e3 + 2000.0)).y;
        float top = rsi(ray_origin, ray_dir, (float)(6371e3 + 0e3 + 2000.0 + 340.0)).y;
        float transmittance = 1.0f;
        V3 in_scatter = v3(0.0f);
        float distance_traveled = bottom;
        float weight_sum = 0.0f, weighted_dist = 0.0f;
        {
            V3 start = ray_origin + ray_dir * bottom;
            float step_delta = (top - bottom) * fsteps;
            V3 ray_step = ray_dir * step_delta;
            V3 ray_pos = start + ray_step * dither;
            distance_traveled = distance(start, ray_origin);
            for (int i = 0; i < steps; i++) {
                float density = sample_cloud_density(ray_pos);
                if (density <= 0.0f || transmittance <= 1e-4f) {
                    ray_pos += ray_step;
                    distance_traveled += step_delta;
                    weighted_dist += distance_traveled * transmittance;
                    weight_sum += transmittance;
                    continue;
                }
                float step_od = cloud_extinc * density * step_delta;
                float step_transmittance = saturate(dm_exp(-step_od));
                float step_weight = (1.0f - step_transmittance) / cloud_extinc;
                float visible_scattering = transmittance * step_weight;
                const int DIRECT = 8;
                for (int j = 0; j < DIRECT; j++) {
                    V3 sample_dir = sample_cone_oriented(sun_cos, sun_dir, rng);
                    float cos_theta = dot(ray_dir, sample_dir);
                    float sun_ray_od = clouds_shadow_od(ray_pos, sample_dir, dither);
                    V3 sun_atmos_T = read_trans_lut(dot(normalized(ray_pos), sample_dir), get_elevation(ray_pos));
                    float an = 1.0f;
                    for (int k = 0; k < 4; k++) {
                        float phase = cloud_phase(cos_theta, an);
                        in_scatter += visible_scattering * an * cloud_scatter * phase * dm_exp(-sun_ray_od * cloud_extinc * an) *
                                      sun_atmos_T * sun_col / (float)DIRECT;
                        an *= 0.5f;
                    }
                }
                float ambient_od = clouds_shadow_od(ray_pos, v3(0.0f, 1.0f, 0.0f), dither);
                float an = 1.0f;
                for (int k = 0; k < 4; k++) {
                    in_scatter += visible_scattering * an * cloud_scatter / (float)(4.0 * 3.141592653589793) *
                                  dm_exp(-ambient_od * cloud_extinc * an) * cloud_ambient;
                    an *= 0.5f;
                }
                transmittance *= step_transmittance;
                ray_pos += ray_step;
                distance_traveled += step_delta;
                weighted_dist += distance_traveled * transmittance;
                weight_sum += transmittance;
            }
            weighted_dist /= weight_sum;
        }
        *in_scatter_out = in_scatter;
        *transmittance_out = transmittance;
        *dist_out = weighted_dist;
    }

    /* ---- atmosphere ---------------------------------------------------------------------- */
    /* atmos.py:355-425.  depth is ti.template(): depth 2 skips the march and returns (0, 1). */
    void atmospheric_scattering(V3 ray_origin, V3 ray_dir, V3 sun_dir, V3 sun_col, float sun_cos, int depth, int steps,
                                dm_rng* rng, V3* in_scatter_out, V3* transmittance_out) const {
        float fsteps = 1.0f / (float)steps;
        V2 air = rsi(ray_origin, ray_dir, (float)(6371e3 + 110e3));
        V2 planet = rsi(ray_origin, ray_dir, planet_r);
        air.y = (planet.x > 0.0f) ? dm_min(air.y, planet.x) : air.y;
        float step_delta = (air.y - dm_max(air.x, 0.0f)) * fsteps;
        V3 ray_step = ray_dir * step_delta;
        V3 ray_pos = ray_origin + ray_step * 0.5f;
        V3 transmittance = v3(1.0f);
        V3 in_scatter_col = v3(0.0f);
        if (depth <= 1) {
            for (int i = 0; i < steps; i++) {
                float h = get_elevation(ray_pos);
                V3 density = get_density(h);
                V3 step_od = extinc_mul(density * step_delta);
                V3 step_T = saturate(v3(dm_exp(-step_od.x), dm_exp(-step_od.y), dm_exp(-step_od.z)));
                V3 visible = transmittance * saturate((v3(1.0f) - step_T) / step_od);
                const int DIRECT = 8;
                for (int j = 0; j < DIRECT; j++) {
                    V3 sample_dir = sample_cone_oriented(sun_cos, sun_dir, rng);
                    float cos_theta = dot(ray_dir, sample_dir);
                    float ph_r = rayleigh_phase(cos_theta), ph_m = mie_phase(cos_theta, mie_g);
                    V3 sun_T = read_trans_lut(dot(normalized(ray_pos), sample_dir), h);
                    in_scatter_col += rayleigh_coeff * sun_col * sun_T * visible * ph_r * density.x * step_delta / (float)DIRECT;
                    in_scatter_col += mie_coeff * sun_col * sun_T * visible * ph_m * density.y * step_delta / (float)DIRECT;
                }
                const float ms_energy = 5.3f;
                const int MS = 8;
                for (int j = 0; j < MS; j++) {
                    V3 sample_dir = get_unit_vec(v2(((float)j + 0.5f) / (float)MS, fract((float)j * 1.618033988749f)));
                    float cos_theta = dot(ray_dir, sample_dir);
                    float ph_m = mie_phase(cos_theta, mie_g);
                    V3 amb, amb_T;
                    atmospheric_scattering(ray_pos, sample_dir, sun_dir, sun_col, sun_cos, depth + 1, 5, rng, &amb, &amb_T);
                    in_scatter_col += ms_energy * rayleigh_coeff * amb * visible * density.x * step_delta / (float)MS;
                    in_scatter_col += ms_energy * mie_coeff * amb * visible * ph_m * density.y * step_delta / (float)MS;
                }
                transmittance *= step_T;
                ray_pos += ray_step;
            }
            if (planet.x > 0.0f) transmittance *= 0.0f;
        }
        *in_scatter_out = in_scatter_col;
        *transmittance_out = transmittance;
    }

    /* stream 2 = sky precompute; frame tags keep the three kernels' streams apart */
    enum { TAG_AMBIENT = 0x1000, TAG_SKYBOX = 0x2000 };

    /* atmos.py:134-138 */
    void compute_cloud_ambient(V3 sun_dir, V3 sun_col, float sun_cos) {
        dm_rng rng = dm_rng_init(seed, TAG_AMBIENT, 0u, 2u);
        V3 amb, T;
        atmospheric_scattering(cam_pos + v3(0.0f, cloud_height, 0.0f), v3(0.0f, 1.0f, 0.0f), sun_dir, sun_col, sun_cos, 0, 64,
                               &rng, &amb, &T);
        cloud_ambient = amb;
    }
    /* atmos.py:140-157; rows [u0,u1) so the caller can split the work over threads */
    void accumulate_clouds_rows(V3 sun_dir, V3 sun_col, float sun_cos, int max_samples, int u0, int u1) {
        float fmax = 1.0f / (float)max_samples;
        for (int u = u0; u < u1; u++)
            for (int v = 0; v < res; v++) {
                dm_rng rng = dm_rng_init(seed, cloud_pass, (uint32_t)(u * res + v), 2u);
                V2 texcoord = v2(((float)u + 0.5f) * fres, ((float)v + 0.5f) * fres);
                V3 ray_dir = unproject_sky(texcoord);
                float dither = dm_rng_f32(&rng);
                V3 cs;
                float cT, cdist;
                clouds_scattering(cam_pos, ray_dir, sun_dir, sun_col, sun_cos, dither, &rng, &cs, &cT, &cdist);
                cs *= 1.2f;
                size_t i = (size_t)u * res + v;
                skybox_scattering[i] += cs * fmax;
                skybox_transmittance[i].x += saturate(cT) * fmax;
                skybox_transmittance[i].y += cdist * fmax;
            }
    }
    /* atmos.py:159-189 */
    void compute_skybox_rows(V3 sun_dir, V3 sun_col, float sun_cos, int u0, int u1) {
        for (int u = u0; u < u1; u++)
            for (int v = 0; v < res; v++) {
                dm_rng rng = dm_rng_init(seed, TAG_SKYBOX, (uint32_t)(u * res + v), 2u);
                V2 texcoord = v2(((float)u + 0.5f) * fres, ((float)v + 0.5f) * fres);
                V3 ray_dir = unproject_sky(texcoord);
                size_t i = (size_t)u * res + v;
                V3 cloud_in_scatter = skybox_scattering[i];
                float cloud_T = skybox_transmittance[i].x;
                float cloud_dist = skybox_transmittance[i].y;
                V3 sky_total, sky_T_total, sky_from, sky_T_from;
                atmospheric_scattering(cam_pos, ray_dir, sun_dir, sun_col, sun_cos, 0, 64, &rng, &sky_total, &sky_T_total);
                V3 cloud_pos = cam_pos + ray_dir * dm_max(cloud_dist, 0.0f);
                atmospheric_scattering(cloud_pos, ray_dir, sun_dir, sun_col, sun_cos, 0, 64, &rng, &sky_from, &sky_T_from);
                V3 T_to_cloud = sky_T_total / sky_T_from;
                V3 in_scattering = sky_total;
                if (use_clouds == 1) {
                    in_scattering = in_scattering - sky_from * saturate(T_to_cloud * dm_max(1.0f - cloud_T, 0.0f));
                    in_scattering += cloud_in_scatter * saturate(T_to_cloud);
                }
                skybox_scattering[i] = in_scattering;
                skybox_transmittance[i] = sky_T_total * cloud_T;
            }
    }
};

} /* namespace orc */
#endif
