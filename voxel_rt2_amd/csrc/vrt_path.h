// vrt_path.h -- one camera path as a resumable per-lane state machine.
//
// Replaces the body of Renderer.render (reference renderer/pathtracer.py:355-632) together with
// generate_new_sample / get_cast_dir (293-347) and the packing helpers it uses
// (math_utils.py:202-215, 231-247; space_transformations.py).
//
// The reference runs `for depth in range(MAX_RAY_DEPTH)` inside one GPU thread per pixel.  Here a
// path is a small register-resident record (`Path`) that is advanced ONE SEGMENT at a time by
// path_segment(): closest hit -> sun sample + shadow ray -> BSDF sample.  A persistent wave keeps
// 64 such records; lanes whose path ended are refilled with fresh pixels between segments
// (vrt_kernels.hip), so lanes at different depths run side by side instead of idling until the
// longest path of the wave is done.  Per-pixel results do not depend on that schedule: every
// pixel owns its random stream (dm_rng keyed by the global pixel index).
#ifndef VRT_PATH_H
#define VRT_PATH_H

#include <type_traits>
#include "vrt_trace.h"
#include "vrt_bsdf.h"
#include "vrt_sky.h"

namespace vrt {

// per-pixel outputs of the render stage (SoA, index = local_row * W + u)
struct PixelBuffers {
    f3* color_d;           // color_buffer          (pathtracer.py:39)
    f3* color_s;           // color_buffer_specular (:40)
    uint32_t* gb_normal;   // gbuff_normals, 2 x binary16 packed (:113)
    float* gb_depth;       // gbuff_depth (:114)
    float* gb_refl_depth;  // gbuff_depth_reflection (:115)
    f3* gb_position;       // gbuff_position (:116)
    uint32_t* gb_mat;      // gbuff_mat_id (:112)
    struct ReservoirRec* reservoir;  // spatial_reservoirs[..., 0] (:108-109), only when ReSTIR is on
    int sample_stride;     // elements between the colour / reflection-depth planes of consecutive fused samples
};

// reservoir.py:8-19 as a 64-byte record (f16 fields as binary16 codes)
struct ReservoirRec {
    f3 F;
    f3 rc_pos;
    f3 rc_incident_L;
    uint32_t rc_normal_and_nee;   // 4 x 8 bit
    uint32_t rc_incident_dir;     // 2 x binary16
    uint32_t rc_mat_info;
    uint32_t M_W;                 // 2 x binary16
    uint32_t jac_lobes;           // binary16 | lobes << 16
    uint32_t pad[2];
};

// math_utils.py:202-207
VRT_DEV uint32_t oct_encode(f3 v) {
    float s = dm_abs(v.x) + dm_abs(v.y) + dm_abs(v.z);
    v.x /= s;
    v.y /= s;
    float ex, ey;
    if (v.z <= 0.0f) {
        ex = (1.0f - dm_abs(v.y)) * ((v.x >= 0.0f) ? 1.0f : -1.0f);
        ey = (1.0f - dm_abs(v.x)) * ((v.y >= 0.0f) ? 1.0f : -1.0f);
    } else {
        ex = v.x;
        ey = v.y;
    }
    return (uint32_t)dm_f32_to_f16(ex * 0.5f + 0.5f) | ((uint32_t)dm_f32_to_f16(ey * 0.5f + 0.5f) << 16);
}
// math_utils.py:209-215
VRT_DEV f3 oct_decode_f(float ax, float ay) {
    float ex = ax * 2.0f - 1.0f, ey = ay * 2.0f - 1.0f;
    f3 v = mk3(ex, ey, 1.0f - dm_abs(ex) - dm_abs(ey));
    float t = dm_max(-v.z, 0.0f);
    v.x += (v.x >= 0.0f) ? -t : t;
    v.y += (v.y >= 0.0f) ? -t : t;
    return norm3(v);
}
VRT_DEV f3 oct_decode(uint32_t packed) {
    return oct_decode_f(dm_f16_to_f32((uint16_t)(packed & 0xffffu)), dm_f16_to_f32((uint16_t)(packed >> 16)));
}
// math_utils.py:231-236
VRT_DEV uint32_t pack_material(int id, f3 albedo) {
    return dm_f2u32((float)id) | (dm_f2u32(albedo.x * 255.0f) << 8) | (dm_f2u32(albedo.y * 255.0f) << 16) |
           (dm_f2u32(albedo.z * 255.0f) << 24);
}
VRT_DEV f3 unpack_albedo(uint32_t enc) {
    return mk3((float)((enc >> 8) & 255u) / 255.0f, (float)((enc >> 16) & 255u) / 255.0f, (float)((enc >> 24) & 255u) / 255.0f);
}
VRT_DEV Material load_material(const float* mats, int id) {
    const float* p = mats + 14 * (id & 127);
    Material m;
    m.base = mk3(p[0], p[1], p[2]);
    m.subsurface = p[3]; m.metallic = p[4]; m.specular = p[5]; m.specular_tint = p[6]; m.roughness = p[7];
    m.anisotropic = p[8]; m.sheen = p[9]; m.sheen_tint = p[10]; m.clearcoat = p[11]; m.clearcoat_gloss = p[12];
    m.ior_minus_one = p[13];
    return m;
}

// space_transformations.py
VRT_DEV float linearize_depth(float depth, const mat4& ip) { return 1.0f / ((depth * 2.0f - 1.0f) * ip.m[14] + ip.m[15]); }
VRT_DEV float delinearize_depth(float lin, const mat4& p) { return ((-lin * p.m[10] + p.m[11]) / -lin) * -0.5f + 0.5f; }
VRT_DEV f3 screen_to_view(f2 uv, float depth, const mat4& ip) {
    f4 p = mul4(ip, mk4(uv.x * 2.0f - 1.0f, uv.y * 2.0f - 1.0f, depth * 2.0f - 1.0f, 1.0f));
    return mk3(p.x / p.w, p.y / p.w, p.z / p.w);
}
VRT_DEV f3 view_to_screen(f3 v, const mat4& pm) {
    f4 p = mul4(pm, mk4(v.x, v.y, v.z, 1.0f));
    return mk3(p.x / p.w, p.y / p.w, p.z / p.w) * 0.5f + 0.5f;
}
VRT_DEV f3 xform(const mat4& m, f3 p, float w) {
    f4 r = mul4(m, mk4(p.x, p.y, p.z, w));
    return mk3(r.x, r.y, r.z);
}
VRT_DEV f2 pixel_texcoord(const FrameParams& fp, float u, float v) {
    return mk2((u + 0.5f) * fp.inv_res.x / fp.render_scale, (v + 0.5f) * fp.inv_res.y / fp.render_scale);
}
VRT_DEV bool outside_render_area(const FrameParams& fp, float u, float v) {  // pathtracer.py:289-291
    return u > fp.render_scale * (float)fp.W || v > fp.render_scale * (float)fp.H;
}
VRT_DEV float power_heuristic(float a, float b) {  // pathtracer.py:349-353
    float a2 = a * a;
    return a2 / dm_max(a2 + b * b, 1e-4f);
}

// ReSTIR-only part of the path record (pathtracer.py:381-391, reservoir.py:22-38)
struct PathRestir {
    f3 thr_after_rc, first_dir, first_light_dir;
    f3 rc_pos, rc_normal, rc_incident_dir, rc_incident_L, rc_nee_dir;
    uint32_t rc_mat_info;
    float first_light_bsdf_pdf;
    int rc_lobe;
};
struct NoRestir {};

template <bool RESTIR>
struct Path {
    f3 pos, d, thr, contrib, nee_d, nee_s, primary_albedo, primary_pos;
    float first_invpdf, refl_dist;
    uint32_t primary_mat_info;
    int pix_u, pix_v;     // global pixel coordinates
    int sample;           // index of this sample inside a fused accumulate(n) launch (random-stream frame = fp.frame + sample)
    int depth;            // segment about to be traced; < 0 = lane holds no path
    int first_lobe;
    int sky_primary;
    dm_rng rng;
    typename std::conditional<RESTIR, PathRestir, NoRestir>::type rs;
};

// get_cast_dir (pathtracer.py:293-309): the camera ray through pixel (u, v).  The same for every sample of a frame (the
// jitter is drawn once per frame, :264-265).
VRT_DEV f3 camera_ray_dir(const FrameParams& fp, int u, int v) {
    f2 tc = pixel_texcoord(fp, (float)u, (float)v);
    if (fp.camera_is_moving == 0) { tc.x = tc.x + fp.taa_jitter.x * 0.5f; tc.y = tc.y + fp.taa_jitter.y * 0.5f; }
    const f3 dv = norm3(screen_to_view(tc, 1.0f, fp.proj_inv));
    return xform(fp.view_inv, dv, 0.0f);
}
// generate_new_sample + get_cast_dir (pathtracer.py:293-347)
template <bool RESTIR>
VRT_DEV void path_begin_along(const FrameParams& fp, Path<RESTIR>& p, int u, int v, int sample, f3 d) {  // d = camera_ray_dir(fp, u, v)
    p.pix_u = u;
    p.pix_v = v;
    p.sample = sample;
    p.primary_pos = mk3(0.0f);
    p.rng = dm_rng_init(fp.seed, fp.frame + (uint32_t)sample, (uint32_t)(v * fp.W + u), 0u);
    p.d = d;
    p.pos = fp.camera_pos;
    p.thr = mk3(1.0f);
    p.contrib = mk3(0.0f);
    p.nee_d = mk3(0.0f);
    p.nee_s = mk3(0.0f);
    p.primary_albedo = mk3(1.0f);
    p.first_invpdf = 1.0f;
    p.refl_dist = 0.0f;
    p.primary_mat_info = 0u;
    p.depth = 0;
    p.first_lobe = 0;
    p.sky_primary = 0;
    if constexpr (RESTIR) {
        p.rs.thr_after_rc = mk3(1.0f);
        p.rs.first_dir = mk3(0.0f);
        p.rs.first_light_dir = mk3(0.0f);
        p.rs.rc_pos = mk3(0.0f); p.rs.rc_normal = mk3(0.0f); p.rs.rc_incident_dir = mk3(0.0f);
        p.rs.rc_incident_L = mk3(0.0f); p.rs.rc_nee_dir = mk3(0.0f);
        p.rs.rc_mat_info = 0u;
        p.rs.first_light_bsdf_pdf = 1.0f;
        p.rs.rc_lobe = 0;
    }
}
template <bool RESTIR>
VRT_DEV void path_begin(const FrameParams& fp, Path<RESTIR>& p, int u, int v, int sample) { path_begin_along(fp, p, u, v, sample, camera_ray_dir(fp, u, v)); }

// What the caller already knows about the hit handed to path_shade(): lets a stage that only ever sees one kind
// (vrt_pool.h) drop the other kind's code.
enum { HIT_ANY = 0, HIT_SOMETHING = 1, HIT_NOTHING = 2 };

// One iteration of pathtracer.py:396-525 after its closest-hit query `h`.  Returns true when the path is over.
// BLACK_SUN: the caller knows (and the launch parameters say) that the sun does not emit -- scene.py's default, which
// example1.py never changes.  Then nothing can use the light sample except case (b) below, and the compiler drops
// the sun direction (a basis, a sincos, a square root), the light-sample evaluation and what feeds them from the
// common path; the two draws of the cone sampler are still made, so every later draw keeps its place in the stream.
template <bool RESTIR, int KIND, bool BLACK_SUN = false, class PyrT>
VRT_DEV bool path_shade(const FrameParams& fp, const SceneData& sc, const PyrT& P, const PixelBuffers& out, int local_idx,
                        Path<RESTIR>& p, const Hit& h, TraceStats& ts) {
    const int depth = p.depth;
    const f3 hit_pos = p.pos + h.closest * p.d;
    const bool surface = (KIND != HIT_NOTHING) && (!h.hit_light) && (h.closest < DM_INF);

    if (depth == 0) {
        VRT_REGION(8);
        // g-buffer of the primary vertex (pathtracer.py:403-407, 535-541); sky pixels store position 0 (:510)
        const f3 ppos = (h.closest == DM_INF) ? mk3(0.0f) : hit_pos;
        p.primary_mat_info = pack_material(h.mat_id, h.albedo);
        p.primary_albedo = h.albedo;
        p.sky_primary = (h.closest == DM_INF) ? 1 : 0;
        p.primary_pos = ppos;
        if (p.sample == 0) {  // the samples of one accumulate(n) call share camera and jitter: same primary vertex
            stream_store(&out.gb_normal[local_idx], oct_encode(h.normal));
            stream_store3(&out.gb_position[local_idx], ppos);
            stream_store(&out.gb_mat[local_idx], p.primary_mat_info);
            stream_store(&out.gb_depth[local_idx], view_to_screen(xform(fp.view, ppos, 1.0f), fp.proj).z);
        }
    } else if (depth == 1) {
        if (p.first_lobe != LOBE_DIFFUSE) p.refl_dist += h.closest;
        if constexpr (RESTIR) {
            p.rs.rc_pos = hit_pos;
            p.rs.rc_normal = h.normal;
            p.rs.rc_mat_info = pack_material(h.mat_id, h.albedo);
            p.rs.first_dir = p.d;
        }
    } else if (depth == 2) {
        if constexpr (RESTIR) p.rs.rc_incident_dir = p.d;
    }

    if (surface) {
        VRT_REGION(2);
        p.pos = hit_pos + h.normal * VRT_EPS;
        Material m = load_material(sc.mats, h.mat_id);
        m.base = h.albedo;
        Surf s;
        surf_init(s, m, h.normal, -p.d);

        // Sun sample (pathtracer.py:435-476) and next direction (478-497).  The random draws keep the reference's
        // order -- two for the cone, then the BSDF sampler's -- but the shadow ray, which draws nothing, is traced
        // AFTER the BSDF sample and only when its answer can reach the image:
        //   (a) the sun emits (light_color * light_weight != 0): the light sample contributes; or
        //   (b) the sampled bounce direction lies inside the sun cone, where NEE_visible enters its MIS weight (:490-491).
        // With a black sun (scene.py:127's default, e.g. example1.py) the light-sample terms are exact zeros
        // (firefly() maps a NaN product to 0), so the ray and the BSDF evaluation behind it are skipped.
        const float lu0 = dm_rng_f32(&p.rng), lu1 = dm_rng_f32(&p.rng);  // cone_dir's two draws
        f3 ldir = mk3(0.0f);
        float ndl = 0.0f;
        if constexpr (!(BLACK_SUN && !RESTIR)) {
            f3 lx, ly;
            ortho_basis(fp.light_dir, lx, ly);
            ldir = cone_dir_from(fp.light_cos_max, fp.light_dir, lx, ly, lu0, lu1);
            ndl = dot3(ldir, h.normal);
        }
        const bool light_on = !(BLACK_SUN && !RESTIR) &&
                              (fp.light_color.x != 0.0f || fp.light_color.y != 0.0f || fp.light_color.z != 0.0f) && fp.light_weight != 0.0f;
        if constexpr (RESTIR) {
            if (depth == 0) { p.rs.first_light_bsdf_pdf = pdf_all(s, ldir); p.rs.first_light_dir = ldir; }
        }
        const bool final_segment = (depth > 0) && (depth + 1 >= fp.max_depth) && !RESTIR;  // its bounce is never traced
        f3 brdf = mk3(0.0f), next_d = p.d;
        float pdf = 1.0f, bounce_light_pdf = 0.0f;
        int lobe = 0;
        if (!final_segment) {
            VRT_REGION(10);
            next_d = sample_bsdf(s, p.rng, brdf, pdf, lobe);
            bounce_light_pdf = cone_pdf(fp.light_cos_max, dot3(fp.light_dir, next_d));
        }
        if constexpr (BLACK_SUN && !RESTIR) {
            if (depth > 0 && bounce_light_pdf > 0.0f) {  // case (b): the only use of the sun direction under a black sun
                f3 lx, ly;
                ortho_basis(fp.light_dir, lx, ly);
                ldir = cone_dir_from(fp.light_cos_max, fp.light_dir, lx, ly, lu0, lu1);
                ndl = dot3(ldir, h.normal);
            }
        }
        float nee_visible = 0.0f;
        if (ndl > 0.0f && (RESTIR || light_on || (depth > 0 && bounce_light_pdf > 0.0f))) {
            Hit sh;
            next_hit<true>(fp, sc, P, p.pos, ldir, sh, ts);
            if (sh.closest >= DM_INF) {
                nee_visible = 1.0f;
                if constexpr (RESTIR) { if (depth == 1) p.rs.rc_nee_dir = ldir; }
                if (RESTIR || light_on) {
                    VRT_REGION(4);
                    const float light_bsdf_pdf = pdf_all(s, ldir);
                    float w = 1.0f;
                    if (depth > 0) w = power_heuristic(cone_pdf(fp.light_cos_max, 1.0f), light_bsdf_pdf);
                    f3 bd, bs;
                    eval_lobes(s, ldir, LOBE_ALL, bd, bs);
                    f3 sky_t = mk3(1.0f);
                    if (fp.use_sky == 1) { sky_t = sky_transmittance(sc.sky, ldir); ts.sky_lookups += 1u; }
                    const f3 nd = w * bd * sky_t * fp.light_weight * fp.light_color * ndl;
                    const f3 ns = w * bs * sky_t * fp.light_weight * fp.light_color * ndl;
                    if (depth == 0) {
                        p.nee_d = p.nee_d + firefly(p.thr * nd);
                        p.nee_s = p.nee_s + firefly(p.thr * ns);
                        if constexpr (!RESTIR) {
                            // pathtracer.py:567-578: the first-vertex light-sample MIS weight is known here
                            const float lw = power_heuristic(cone_pdf(fp.light_cos_max, 1.0f), light_bsdf_pdf);
                            p.nee_d = p.nee_d * lw;
                            p.nee_s = p.nee_s * lw;
                        }
                    } else {
                        p.contrib = p.contrib + firefly(p.thr * (nd + ns));
                    }
                    if constexpr (RESTIR) { if (depth >= 2) p.rs.rc_incident_L = p.rs.rc_incident_L + p.rs.thr_after_rc * (nd + ns); }
                }
            }
        }
        p.depth = depth + 1;
        if (final_segment) return true;

        p.d = next_d;
        f3 bw = brdf * dm_saturate(dot3(next_d, h.normal));
        if (depth == 0) {
            p.first_invpdf = 1.0f / pdf;
            p.first_lobe = lobe;
        } else {
            bw = bw / pdf;
            bw = bw * power_heuristic(pdf, nee_visible * bounce_light_pdf);
            if constexpr (RESTIR) {
                if (depth == 1) p.rs.rc_lobe = lobe;
                if (depth >= 2) p.rs.thr_after_rc = p.rs.thr_after_rc * bw;
            }
        }
        p.thr = p.thr * bw;
        return p.depth >= fp.max_depth;
    }

    if (KIND != HIT_SOMETHING && h.closest == DM_INF) {
        VRT_REGION(5);
        // escaped: background colour or skybox, plus the sun disc (pathtracer.py:500-517)
        const float hit_sun = (dot3(fp.light_dir, p.d) >= fp.light_cos_max) ? 1.0f : 0.0f;
        f3 scat = fp.background, trans = mk3(1.0f);
        if (fp.use_sky == 1) { sky_lookup(sc.sky, p.d, p.rng, scat, trans); ts.sky_lookups += 2u; }
        const f3 emission = firefly(scat + trans * fp.light_weight * fp.light_color * hit_sun);
        p.contrib = p.contrib + p.thr * emission;
        if constexpr (RESTIR) {
            if (depth == 1) { p.rs.rc_pos = p.d; p.rs.rc_incident_L = emission; }
            if (depth >= 2) p.rs.rc_incident_L = p.rs.rc_incident_L + firefly(p.rs.thr_after_rc * emission);
        }
    } else {
        // emissive voxel / floor terminates the path (pathtracer.py:518-524)
        if (depth > 0) p.contrib = p.contrib + p.thr * h.albedo;
        if constexpr (RESTIR) { if (depth >= 2) p.rs.rc_incident_L = p.rs.rc_incident_L + firefly(p.rs.thr_after_rc * h.albedo); }
    }
    return true;
}

// Advance one segment: closest-hit query, then path_shade().
template <bool RESTIR, class PyrT>
VRT_DEV bool path_segment(const FrameParams& fp, const SceneData& sc, const PyrT& P, const PixelBuffers& out, int local_idx,
                          Path<RESTIR>& p, TraceStats& ts) {
    Hit h;
    next_hit<false>(fp, sc, P, p.pos, p.d, h, ts);
    return path_shade<RESTIR, HIT_ANY>(fp, sc, P, out, local_idx, p, h, ts);
}

// defined in vrt_restir.h: builds the input reservoir (pathtracer.py:549-607, 620-626)
template <class PathT>
VRT_DEV void restir_finish(const FrameParams& fp, const SceneData& sc, const PixelBuffers& out, int local_idx, PathT& p,
                           f3 primary_pos, f3& diffuse, f3& specular, TraceStats& ts);

// Everything after the bounce loop (pathtracer.py:535-632)
template <bool RESTIR>
VRT_DEV void path_finish(const FrameParams& fp, const SceneData& sc, const PixelBuffers& out, int local_idx, Path<RESTIR>& p,
                         TraceStats& ts) {
    VRT_REGION(6);
    const f3 primary_pos = p.primary_pos;
    const int plane = local_idx + p.sample * out.sample_stride;
    // virtual reflection depth (543-547)
    float refl = 0.0f;
    if (p.refl_dist != 0.0f) {
        const f3 pdir = norm3(primary_pos - fp.camera_pos);
        const f3 vp = primary_pos + pdir * p.refl_dist;
        refl = linearize_depth(view_to_screen(xform(fp.view, vp, 1.0f), fp.proj).z, fp.proj_inv);
    }
    stream_store(&out.gb_refl_depth[plane], refl);

    f3 diffuse = mk3(0.0f), specular = mk3(0.0f);
    if constexpr (!RESTIR) {
        // 611-619: BSDF-sampled and light-sampled estimates, split by first-bounce lobe
        const f3 emission = ((p.primary_mat_info & 255u) == 2u) ? unpack_albedo(p.primary_mat_info) : mk3(0.0f);
        diffuse = diffuse + ((p.first_lobe == LOBE_DIFFUSE) ? p.contrib * p.first_invpdf + emission : mk3(0.0f));
        specular = specular + ((p.first_lobe == LOBE_SPEC) ? p.contrib * p.first_invpdf : mk3(0.0f));
        diffuse = diffuse + p.nee_d;
        specular = specular + p.nee_s;
        if (fp.camera_is_moving == 1) diffuse = diffuse / max3s(p.primary_albedo, 1e-2f);
    } else {
        restir_finish(fp, sc, out, local_idx, p, primary_pos, diffuse, specular, ts);
    }
    stream_store3(&out.color_d[plane], diffuse);
    stream_store3(&out.color_s[plane], specular);
}

}  // namespace vrt
#endif
