/* orc_renderer.h -- ORACLE (test infrastructure, not product code).  PARITY: the reference ships
 * no tests / golden vectors and its Taichi runtime is not installable here,
 * so there is no reference build to compare with.  This restatement is pinned (a) against the
 * reference's own source files executed under a host emulation of the Taichi DSL
 * (tests/refexec, tests/golden/make_reference_vectors.py -> tests/golden/reference/*.npz,
 * tests/test_reference_vectors.py: whole frames and single rays, bit for bit) and (b) by
 * hand-derived known-answer and property tests under tests/.  Not pinned: what the Taichi
 * compiler itself does to that source (fast-math, its own elementary functions).
 *
 * Literal CPU restatement of /root/reference/renderer/pathtracer.py class Renderer: ray
 * generation, next_hit, the render kernel, spatial_GRIS / shift, the three temporal kernels,
 * accumulate(), _render_to_image.  One pixel at a time, in the reference's statement order.
 *
 * Semantics fixed where the reference is racy or undefined:
 *  - kernels that read neighbours of a buffer they also write see the values from BEFORE the
 *    kernel (snapshot): temporal_filter_prepass on gbuff_depth_reflection, temporal_filter on
 *    color_buffer;
 *  - out-of-image taps of bilinear_sample clamp to the edge; out-of-image taps of spatial_GRIS
 *    are skipped;
 *  - self.current_frame inside spatial_GRIS is the value baked at first compile (0);
 *  - dead outputs are not produced: specular_mean / specular_stdev (pathtracer.py:1063-1064,
 *    consumer commented out at 1280-1281), history_buffer_specular_depth (only ever feeds
 *    itself), VoxelWorld.bbox, the unused 3x3 variance block of _render_to_image (646-658).
 */
#ifndef ORC_RENDERER_H
#define ORC_RENDERER_H

#include <vector>
#include <thread>
#include <functional>
#include "orc_world.h"
#include "orc_bsdf.h"
#include "orc_reservoir.h"
#include "orc_atmos.h"

namespace orc {

static const float RADIANCE_CLAMP = 300.0f; /* pathtracer.py:20 */
inline V3 firefly_filter(V3 v) { return vclamp(v, 0.0f, RADIANCE_CLAMP); } /* :22-24 */

struct Renderer {
    /* configuration (module constants in the reference: pathtracer.py:15-17, scene.py:11-13) */
    int W = 0, H = 0;
    int max_ray_depth = 4;
    bool use_restir = false;
    float exposure = 3.0f;
    uint32_t seed = 0;
    int n_threads = 1;
    /* rows [row_begin,row_end) are produced by accumulate(); the rest of the image is left
     * untouched.  Used to check that sharded rendering equals the full render. */
    int row_begin = 0, row_end = 0;

    VoxelWorld world;
    VoxelOctreeRaytracer voxel_raytracer;
    DisneyMaterial mat_list[128];
    Atmos atmos;

    /* scalar state (pathtracer.py:50-69, 99-104, 130-136) */
    V3 light_direction = V3{0, 0, 0}, light_color = V3{0, 0, 0};
    float light_cone_cos_theta_max = 1.0f, light_weight = 0.0f;
    V3 camera_pos = V3{0, 0, 0}, prev_camera_pos = V3{0, 0, 0};
    float floor_height = 0.0f;
    V3 floor_color = V3{1, 1, 1};
    int floor_material = 1;
    V3 background_color = V3{0, 0, 0};
    M4 proj_mat{}, proj_mat_inv{}, view_mat{}, view_mat_inv{}, prev_proj_mat{}, prev_view_mat{};
    float max_accum_frames = 0.0f, render_scale = 1.0f;
    int camera_is_moving = 0;
    V2 taa_jitter = V2{0, 0};
    int use_physical_atmosphere = 0;
    int current_spp = 0;
    uint32_t current_frame = 0;

    /* per-pixel buffers, index v*W + u */
    std::vector<V3> color_buffer, color_buffer_specular, gbuff_position;
    std::vector<V4> history_buffer[2], history_buffer_specular[2];
    std::vector<uint32_t> gbuff_mat_id;
    std::vector<uint16_t> gbuff_normals, gbuff_prev_normals; /* 2 per pixel */
    std::vector<float> gbuff_depth, gbuff_prev_depth, gbuff_depth_reflection;
    std::vector<StorageReservoir> spatial_reservoirs[2];
    Stats stats;

    void init(int w, int h, float dx, float voxel_edges, float exposure_, int max_depth, bool restir, uint32_t seed_,
              int sky_res, int threads, int grid_res = 128) {
        W = w; H = h; exposure = exposure_; max_ray_depth = max_depth; use_restir = restir; seed = seed_;
        n_threads = threads < 1 ? 1 : threads;
        row_begin = 0; row_end = h;
        /* pathtracer.py:83-85 fixes voxel_grid_res = 128 at the call site; VoxelWorld and VoxelOctreeRaytracer are
         * parametric in it (n_lods = log2(res), raytracer.py:9; offset -res//2, voxel_world.py:14) */
        world.init(dx, grid_res, voxel_edges);
        voxel_raytracer.init(grid_res);
        for (int i = 0; i < 128; i++) mat_list[i] = default_material();
        atmos.init(sky_res, seed_);
        size_t n = (size_t)w * h;
        color_buffer.assign(n, v3(0.0f)); color_buffer_specular.assign(n, v3(0.0f)); gbuff_position.assign(n, v3(0.0f));
        for (int s = 0; s < 2; s++) {
            history_buffer[s].assign(n, V4{0, 0, 0, 0});
            history_buffer_specular[s].assign(n, V4{0, 0, 0, 0});
            spatial_reservoirs[s].assign(n, StorageReservoir{});
        }
        gbuff_mat_id.assign(n, 0u);
        gbuff_normals.assign(2 * n, 0); gbuff_prev_normals.assign(2 * n, 0);
        gbuff_depth.assign(n, 0.0f); gbuff_prev_depth.assign(n, 0.0f); gbuff_depth_reflection.assign(n, 0.0f);
        render_scale = 1.0f;
    }

    void parallel_rows(int r0, int r1, const std::function<void(int, int, Stats*)>& fn) {
        int rows = r1 - r0;
        int nt = n_threads < rows ? n_threads : (rows > 0 ? rows : 1);
        if (nt <= 1) { fn(r0, r1, &stats); return; }
        std::vector<std::thread> th;
        std::vector<Stats> st(nt);
        /* interleaved strips for load balance: 4 rows, fewer when there are not 4 rows per thread (a few sky-table columns) */
        const int strip = rows >= 4 * nt ? 4 : 1;
        for (int t = 0; t < nt; t++)
            th.emplace_back([&, t]() {
                for (int r = r0 + strip * t; r < r1; r += strip * nt) fn(r, (r + strip < r1) ? r + strip : r1, &st[t]);
            });
        for (auto& x : th) x.join();
        for (auto& s : st) {
            stats.rays += s.rays; stats.iters += s.iters; stats.queries += s.queries;
            stats.closest_hits += s.closest_hits; stats.sky_lookups += s.sky_lookups;
        }
    }

    /* pathtracer.py:139-144 (direction already normalised by the caller) */
    void set_directional_light(V3 direction, float cos_theta_max, V3 color) {
        light_direction = direction;
        light_cone_cos_theta_max = cos_theta_max;
        light_color = color;
        light_weight = 3.0f;
    }
    /* pathtracer.py:262-273: jitter drawn once per call from stream 3 */
    void draw_taa_jitter(uint32_t jitter_index) {
        dm_rng rng = dm_rng_init(seed, jitter_index, 0u, 3u);
        float r0 = dm_rng_f32(&rng), r1 = dm_rng_f32(&rng);
        taa_jitter = v2((r0 * 2.0f - 1.0f) * (float)(1.0 / (double)W), (r1 * 2.0f - 1.0f) * (float)(1.0 / (double)H));
    }
    /* pathtracer.py:283-287 */
    void copy_prev_matrices() {
        prev_proj_mat = proj_mat;
        prev_view_mat = view_mat;
        prev_camera_pos = camera_pos;
    }
    /* pathtracer.py:314-323 */
    void prepare_data() {
        world.make_texture();
        voxel_raytracer.update_lods(world.voxel_material.data());
        if (use_physical_atmosphere == 1) {
            atmos.generate_transmittance_lut();
            atmos.compute_cloud_ambient(light_direction, light_color * light_weight, light_cone_cos_theta_max);
            std::fill(atmos.skybox_scattering.begin(), atmos.skybox_scattering.end(), v3(0.0f));
            std::fill(atmos.skybox_transmittance.begin(), atmos.skybox_transmittance.end(), v3(0.0f));
            atmos.cloud_pass = 0;
        }
    }
    /* pathtracer.py:325-329 */
    void accumulate_clouds(int max_samples) {
        Stats dummy = stats;
        parallel_rows(0, atmos.res, [&](int a, int b, Stats*) {
            atmos.accumulate_clouds_rows(light_direction, light_color * light_weight, light_cone_cos_theta_max, max_samples, a, b);
        });
        stats = dummy;
        atmos.cloud_pass++;
    }
    /* one cloud pass over the columns of one slice (a texel's passes depend on no other texel: atmos.py:140-157) */
    void accumulate_clouds_slice(int max_samples, int slice_idx, int max_slices) {
        int w = atmos.res / max_slices;
        Stats dummy = stats;
        parallel_rows(w * slice_idx, w * (slice_idx + 1), [&](int a, int b, Stats*) {
            atmos.accumulate_clouds_rows(light_direction, light_color * light_weight, light_cone_cos_theta_max, max_samples, a, b);
        });
        stats = dummy;
        atmos.cloud_pass++;
    }
    void compute_atmosphere(int slice_idx, int max_slices) {
        int slice_width = atmos.res / max_slices; /* atmos.py:162 */
        Stats dummy = stats;
        parallel_rows(slice_width * slice_idx, slice_width * (slice_idx + 1), [&](int a, int b, Stats*) {
            atmos.compute_skybox_rows(light_direction, light_color * light_weight, light_cone_cos_theta_max, a, b);
        });
        stats = dummy;
    }

    V3 world_to_voxel(V3 pos) const { /* :165-167 */
        return world.voxel_inv_size * pos - (float)world.voxel_grid_offset;
    }
    /* pathtracer.py:173-190.  `hit_pos - dot(hit_pos, sdf_normal)` broadcasts the scalar. */
    void trace_sdf(V3 pos, V3 dir, float* closest, V3* normal, V3* color, int* is_light, int* mat_id) const {
        float ray_march_dist = (floor_height - pos.y) / dir.y; /* :152-155 */
        if (ray_march_dist > EPS && ray_march_dist < *closest) {
            V3 hit_pos = pos + dir * ray_march_dist;
            V3 sdf_normal = v3(0.0f, 1.0f, 0.0f);
            if (length(hit_pos - dot(hit_pos, sdf_normal)) < 10.0f) {
                *closest = ray_march_dist;
                *normal = sdf_normal;
                if (dot(*normal, dir) > 0.0f) *normal = -*normal;
                *color = floor_color;
                *is_light = (floor_material == 2) ? 1 : 0;
                *mat_id = floor_material;
            }
        }
    }
    /* pathtracer.py:192-216 */
    void trace_voxel(V3 eye_pos, V3 d, float* closest, V3* normal, V3* color, int* is_light, int* mat_id, bool shadow_ray,
                     Stats* st) const {
        V3 eye_pos_scaled = world_to_voxel(eye_pos);
        VoxelOctreeRaytracer::Hit h = voxel_raytracer.raytrace(eye_pos_scaled, d, EPS, INF, st);
        if (h.distance * world.voxel_size < *closest) {
            *closest = h.distance * world.voxel_size;
            if (!shadow_ray) {
                V3 idx_f = v3((float)h.ipos.x, (float)h.ipos.y, (float)h.ipos.z);
                V3 voxel_uv = vclamp(eye_pos_scaled + h.distance * d - idx_f, 0.0f, 1.0f);
                I3 voxel_index = I3{h.ipos.x + world.voxel_grid_offset, h.ipos.y + world.voxel_grid_offset,
                                    h.ipos.z + world.voxel_grid_offset};
                world.voxel_surface_color(voxel_index, voxel_uv, color, is_light, mat_id);
                *normal = h.normal;
                if (st) st->closest_hits++;
            }
        }
    }
    /* pathtracer.py:218-244 (cast_voxel_hit is never set: the highlight branch is dead) */
    void next_hit(V3 pos, V3 d, float max_dist, bool shadow_ray, Stats* st, float* closest, V3* normal, V3* albedo,
                  int* hit_light, int* mat_id) const {
        *closest = max_dist;
        *normal = v3(0.0f);
        *albedo = v3(1.0f);
        *hit_light = 0;
        *mat_id = 0;
        trace_sdf(pos, d, closest, normal, albedo, hit_light, mat_id);
        trace_voxel(pos, d, closest, normal, albedo, hit_light, mat_id, shadow_ray, st);
    }
    bool is_outside_render_area(float u, float v) const { /* :289-291 */
        return u > render_scale * (float)W || v > render_scale * (float)H;
    }
    V2 pixel_texcoord(float u, float v) const {
        return v2((u + 0.5f) * (float)(1.0 / (double)W) / render_scale, (v + 0.5f) * (float)(1.0 / (double)H) / render_scale);
    }
    /* pathtracer.py:293-312 */
    V3 get_cast_dir(float u, float v) const {
        V2 texcoord = pixel_texcoord(u, v);
        if (camera_is_moving == 0) texcoord = texcoord + taa_jitter * 0.5f;
        V3 d = normalized(screen_to_view(texcoord, 1.0f, proj_mat_inv));
        return view_to_world(d, view_mat_inv, 0.0f);
    }
    static float power_heuristic(float a, float b) { /* :349-353 */
        float a_sqr = a * a;
        float p_sum = dm_max(a_sqr + b * b, 1e-4f);
        return a_sqr / p_sum;
    }
    DisneyMaterial decode_material(uint32_t enc, int* id) const { /* math_utils.py:238-247 */
        V3 albedo;
        decode_material_bits(enc, id, &albedo);
        DisneyMaterial m = mat_list[*id & 127];
        m.base_col = albedo;
        return m;
    }

    /* ---- pathtracer.py:355-632 --------------------------------------------------------------- */
    void render_pixel(int ui, int vi, Stats* st) {
        const float u = (float)ui, v = (float)vi;
        const size_t pix = (size_t)vi * W + ui;
        dm_rng rng = dm_rng_init(seed, current_frame, (uint32_t)pix, 0u);
        /* generate_new_sample :331-347 */
        V3 d = get_cast_dir(u, v);
        V3 pos = camera_pos;
        V3 contrib = v3(0.0f), throughput = v3(1.0f);
        int hit_light = 0;
        Reservoir isr;
        isr.init();
        if (is_outside_render_area(u, v)) return;

        uint16_t primary_normal[2] = {0, 0};
        V3 primary_pos = v3(0.0f);
        uint32_t primary_mat_info = 0u;
        V3 primary_albedo = v3(1.0f);
        V3 throughput_after_rc = v3(1.0f);
        int first_bounce_lobe_id = 0;
        float first_bounce_invpdf = 1.0f;
        V3 first_vertex_NEE_diffuse = v3(0.0f), first_vertex_NEE_specular = v3(0.0f);
        V3 first_bounce_dir = v3(0.0f);
        float first_light_sample_bsdf_pdf = 1.0f;
        V3 first_light_sample_dir = v3(0.0f);
        float first_bounce_reflection_dist = 0.0f;
        int rc_bounce_lobe_id = 0;
        bool is_sky_ray = false;

        for (int depth = 0; depth < max_ray_depth; depth++) {
            float closest;
            V3 normal, albedo;
            int mat_id;
            next_hit(pos, d, INF, false, st, &closest, &normal, &albedo, &hit_light, &mat_id);
            DisneyMaterial hit_mat = mat_list[mat_id & 127];
            V3 hit_pos = pos + closest * d;

            if (depth == 0) {
                encode_unit_vector_3x16(normal, primary_normal);
                primary_pos = hit_pos;
                primary_mat_info = encode_material(mat_id, albedo);
                primary_albedo = albedo;
            } else if (depth == 1) {
                isr.z.rc_pos = hit_pos;
                isr.z.rc_normal = normal;
                isr.z.rc_mat_info = encode_material(mat_id, albedo);
                first_bounce_dir = d;
                if (first_bounce_lobe_id != LOBE_DIFFUSE) first_bounce_reflection_dist += closest;
            } else if (depth == 2) {
                isr.z.rc_incident_dir = d;
            }

            if (!hit_light && closest < INF) {
                pos = hit_pos + normal * EPS;
                hit_mat.base_col = albedo;
                V3 view = -d;
                V3 tang, bitang;
                make_orthonormal_basis(normal, &tang, &bitang);
                float NEE_visible = 0.0f;
                { /* use_directional_light, :435-476 */
                    V3 light_dir = sample_cone_oriented(light_cone_cos_theta_max, light_direction, &rng);
                    float dotl = dot(light_dir, normal);
                    float light_sample_bsdf_pdf = pdf_disney(hit_mat, view, normal, light_dir, tang, bitang);
                    if (depth == 0) {
                        first_light_sample_bsdf_pdf = light_sample_bsdf_pdf;
                        first_light_sample_dir = light_dir;
                    }
                    if (dotl > 0.0f) {
                        float dist;
                        V3 n_, a_;
                        int hl_, sm_;
                        next_hit(pos, light_dir, INF, true, st, &dist, &n_, &a_, &hl_, &sm_);
                        if (dist >= INF) {
                            NEE_visible = 1.0f;
                            if (depth == 1) isr.z.rc_NEE_dir = light_dir;
                            float light_sample_mis_weight = 1.0f;
                            if (depth > 0) {
                                float light_sample_light_pdf = cone_sample_pdf(light_cone_cos_theta_max, 1.0f);
                                light_sample_mis_weight = power_heuristic(light_sample_light_pdf, light_sample_bsdf_pdf);
                            }
                            V3 bd, bs;
                            disney_evaluate_split(hit_mat, view, normal, light_dir, tang, bitang, &bd, &bs);
                            V3 sky_T = v3(1.0f);
                            if (use_physical_atmosphere == 1) {
                                sky_T = atmos.sample_skybox_transmittance(light_dir);
                                if (st) st->sky_lookups++;
                            }
                            V3 NEE_d = light_sample_mis_weight * bd * sky_T * light_weight * light_color * dotl;
                            V3 NEE_s = light_sample_mis_weight * bs * sky_T * light_weight * light_color * dotl;
                            if (depth == 0) {
                                first_vertex_NEE_diffuse += firefly_filter(throughput * NEE_d);
                                first_vertex_NEE_specular += firefly_filter(throughput * NEE_s);
                            } else {
                                contrib += firefly_filter(throughput * (NEE_d + NEE_s));
                            }
                            if (depth >= 2) isr.z.rc_incident_L += throughput_after_rc * (NEE_d + NEE_s);
                        }
                    }
                }
                V3 bsdf;
                float pdf;
                int lobe_id;
                d = sample_disney(hit_mat, view, normal, tang, bitang, &bsdf, &pdf, &lobe_id, &rng);
                V3 bounce_weight = bsdf * saturate(dot(d, normal));
                if (depth == 0) {
                    first_bounce_invpdf = 1.0f / pdf;
                    first_bounce_lobe_id = lobe_id;
                } else {
                    bounce_weight /= pdf;
                    float bsdf_sample_light_pdf = cone_sample_pdf(light_cone_cos_theta_max, dot(light_direction, d));
                    bounce_weight *= power_heuristic(pdf, NEE_visible * bsdf_sample_light_pdf);
                    if (depth == 1) rc_bounce_lobe_id = lobe_id;
                    if (depth >= 2) throughput_after_rc *= bounce_weight;
                }
                throughput *= bounce_weight;
            } else {
                if (closest == INF) {
                    float hit_sun = (dot(light_direction, d) >= light_cone_cos_theta_max) ? 1.0f : 0.0f;
                    V3 sky_scattering = background_color;
                    V3 sky_T = v3(1.0f);
                    if (use_physical_atmosphere == 1) {
                        atmos.sample_skybox(d, &rng, &sky_scattering, &sky_T);
                        if (st) st->sky_lookups += 2;
                    }
                    V3 sky_emission = firefly_filter(sky_scattering + sky_T * light_weight * light_color * hit_sun);
                    contrib += throughput * sky_emission;
                    if (depth == 0) {
                        primary_pos = v3(0.0f);
                        is_sky_ray = true;
                    } else if (depth == 1) {
                        isr.z.rc_pos = d;
                        isr.z.rc_incident_L = sky_emission;
                    }
                    if (depth >= 2) isr.z.rc_incident_L += firefly_filter(throughput_after_rc * sky_emission);
                } else {
                    if (depth > 0) contrib += throughput * albedo;
                    if (depth >= 2) isr.z.rc_incident_L += firefly_filter(throughput_after_rc * albedo);
                }
                break;
            }
        }

        V3 primary_pos_view = world_to_view(primary_pos, view_mat);
        gbuff_normals[2 * pix] = primary_normal[0];
        gbuff_normals[2 * pix + 1] = primary_normal[1];
        gbuff_depth[pix] = view_to_screen(primary_pos_view, proj_mat).z;
        gbuff_position[pix] = primary_pos;
        gbuff_mat_id[pix] = primary_mat_info;

        V3 primary_dir = normalized(primary_pos - camera_pos);
        V3 virtual_point = primary_pos + primary_dir * first_bounce_reflection_dist;
        float refl_depth = view_to_screen(world_to_view(virtual_point, view_mat), proj_mat).z;
        gbuff_depth_reflection[pix] = (first_bounce_reflection_dist != 0.0f) ? linearize_depth(refl_depth, proj_mat_inv) : 0.0f;

        isr.z.F = contrib;
        isr.z.lobes = rc_bounce_lobe_id * 10 + first_bounce_lobe_id;
        isr.M = 1.0f;
        isr.update_cached_jacobian_term(primary_pos);

        bool chose_NEE_sample = false;
        if (!is_sky_ray) {
            float bsdf_sample_bsdf_pdf = 1.0f / first_bounce_invpdf;
            float bsdf_sample_light_pdf = cone_sample_pdf(light_cone_cos_theta_max, dot(light_direction, first_bounce_dir));
            if (is_vec_zero(first_vertex_NEE_diffuse + first_vertex_NEE_specular)) bsdf_sample_light_pdf = 0.0f;
            float bsdf_sample_mis_weight = power_heuristic(bsdf_sample_bsdf_pdf, bsdf_sample_light_pdf);
            float light_sample_light_pdf = cone_sample_pdf(light_cone_cos_theta_max, 1.0f);
            float light_sample_mis_weight = power_heuristic(light_sample_light_pdf, first_light_sample_bsdf_pdf);
            if (!use_restir) isr.z.F *= bsdf_sample_mis_weight;
            float p_hat = luminance(isr.z.F);
            isr.weight = bsdf_sample_mis_weight * p_hat * first_bounce_invpdf;
            if (!use_restir) {
                first_vertex_NEE_diffuse *= light_sample_mis_weight;
                first_vertex_NEE_specular *= light_sample_mis_weight;
            }
            float light_sample_weight = light_sample_mis_weight * luminance(first_vertex_NEE_diffuse + first_vertex_NEE_specular);
            V3 sky_T = v3(1.0f);
            if (use_physical_atmosphere == 1) {
                sky_T = atmos.sample_skybox_transmittance(first_light_sample_dir);
                if (st && use_restir) st->sky_lookups++; /* only feeds the reservoir: unobservable with ReSTIR off */
            }
            Sample light_sample;
            light_sample.F = first_vertex_NEE_diffuse + first_vertex_NEE_specular;
            light_sample.rc_pos = first_light_sample_dir;
            light_sample.rc_normal = v3(0.0f);
            light_sample.rc_incident_dir = v3(0.0f);
            light_sample.rc_incident_L = sky_T * light_weight * light_color;
            light_sample.rc_NEE_dir = v3(0.0f);
            light_sample.rc_mat_info = 0u;
            light_sample.cached_jacobian_term = 1.0f;
            light_sample.lobes = LOBE_ALL * 10 + LOBE_ALL;
            chose_NEE_sample = isr.input_sample(light_sample_weight, light_sample, &rng);
            isr.finalize_without_M();
        } else {
            isr.weight = 1.0f;
        }
        spatial_reservoirs[0][pix] = isr.encode();

        V3 diffuse = v3(0.0f), specular = v3(0.0f);
        if (!use_restir) {
            int primary_mat_id;
            DisneyMaterial primary_mat = decode_material(primary_mat_info, &primary_mat_id);
            V3 emission = (primary_mat_id == 2) ? primary_mat.base_col : v3(0.0f);
            diffuse += (first_bounce_lobe_id == LOBE_DIFFUSE) ? contrib * first_bounce_invpdf + emission : v3(0.0f);
            specular += (first_bounce_lobe_id == LOBE_SPEC_REFL) ? contrib * first_bounce_invpdf : v3(0.0f);
            diffuse += first_vertex_NEE_diffuse;
            specular += first_vertex_NEE_specular;
        } else {
            if (!chose_NEE_sample) {
                diffuse += (first_bounce_lobe_id == LOBE_DIFFUSE) ? isr.z.F : v3(0.0f);
                specular += (first_bounce_lobe_id == LOBE_SPEC_REFL) ? isr.z.F : v3(0.0f);
            } else {
                diffuse += first_vertex_NEE_diffuse;
                specular += first_vertex_NEE_specular;
            }
        }
        if (!use_restir)
            if (camera_is_moving == 1) diffuse /= vmax(primary_albedo, 1e-2f);
        color_buffer[pix] = diffuse;
        color_buffer_specular[pix] = specular;
    }

    /* ---- pathtracer.py:672-812 --------------------------------------------------------------- */
    void shift(V3 dst_pos, V3 dst_normal, const DisneyMaterial& dst_material, V3 src_pos, const Reservoir& src, V3* diffuse_out,
               V3* specular_out, float* jacobian_out, Stats* st) const {
        bool rc_is_escape_vertex = is_vec_zero(src.z.rc_normal);
        bool rc_is_last_vertex = is_vec_zero(src.z.rc_incident_dir);
        bool rc_is_NEE_visible = !is_vec_zero(src.z.rc_NEE_dir);
        V3 dir_to_rc_vertex = rc_is_escape_vertex ? src.z.rc_pos : normalized(src.z.rc_pos - dst_pos);
        V3 src_dir_to_rc_vertex = rc_is_escape_vertex ? src.z.rc_pos : normalized(src.z.rc_pos - src_pos);
        float passed_checks = 1.0f;
        if (dot(dst_normal, dir_to_rc_vertex) < 1e-5f || (!rc_is_escape_vertex && dot(src.z.rc_normal, -dir_to_rc_vertex) < 1e-5f))
            passed_checks = 0.0f;
        V3 rc_tang, rc_bitang;
        make_orthonormal_basis(src.z.rc_normal, &rc_tang, &rc_bitang);
        int rc_mat_id;
        DisneyMaterial rc_mat = decode_material(src.z.rc_mat_info, &rc_mat_id);
        V3 rc_brdf = v3(0.0f);
        float dst_rc_pdf = 1.0f, src_rc_pdf = 1.0f;
        if (!rc_is_last_vertex && !rc_is_escape_vertex) {
            rc_brdf = disney_evaluate_lobewise(rc_mat, -dir_to_rc_vertex, src.z.rc_normal, src.z.rc_incident_dir, rc_tang,
                                               rc_bitang, src.z.lobes / 10);
            rc_brdf *= saturate(dot(src.z.rc_normal, src.z.rc_incident_dir));
            dst_rc_pdf = pdf_disney_lobewise(rc_mat, -dir_to_rc_vertex, src.z.rc_normal, src.z.rc_incident_dir, rc_tang,
                                             rc_bitang, src.z.lobes / 10);
            src_rc_pdf = pdf_disney_lobewise(rc_mat, -src_dir_to_rc_vertex, src.z.rc_normal, src.z.rc_incident_dir, rc_tang,
                                             rc_bitang, src.z.lobes / 10);
        }
        (void)src_rc_pdf;
        V3 rc_nee_brdf = v3(0.0f);
        if (rc_is_NEE_visible) {
            rc_nee_brdf = disney_evaluate(rc_mat, -dir_to_rc_vertex, src.z.rc_normal, src.z.rc_NEE_dir, rc_tang, rc_bitang);
            rc_nee_brdf *= saturate(dot(src.z.rc_normal, src.z.rc_NEE_dir));
        }
        V3 dst_tang, dst_bitang;
        make_orthonormal_basis(dst_normal, &dst_tang, &dst_bitang);
        V3 view = normalized(camera_pos - dst_pos);
        V3 primary_brdf_d, primary_brdf_s;
        disney_evaluate_lobewise_split(dst_material, view, dst_normal, dir_to_rc_vertex, dst_tang, dst_bitang, src.z.lobes % 10,
                                       &primary_brdf_d, &primary_brdf_s);
        primary_brdf_d *= saturate(dot(dst_normal, dir_to_rc_vertex));
        primary_brdf_s *= saturate(dot(dst_normal, dir_to_rc_vertex));

        V3 contrib = v3(0.0f);
        if (!rc_is_escape_vertex && !rc_is_last_vertex) {
            float rc_bsdf_sample_light_pdf = cone_sample_pdf(light_cone_cos_theta_max, dot(light_direction, src.z.rc_incident_dir));
            float rc_bsdf_mis_weight = power_heuristic(dst_rc_pdf, rc_bsdf_sample_light_pdf * (rc_is_NEE_visible ? 1.0f : 0.0f));
            contrib += firefly_filter(rc_bsdf_mis_weight * rc_brdf / dst_rc_pdf * src.z.rc_incident_L);
        }
        if (rc_is_escape_vertex) contrib += firefly_filter(src.z.rc_incident_L);
        if (rc_is_NEE_visible && !rc_is_escape_vertex) {
            float rc_light_sample_bsdf_pdf = pdf_disney(rc_mat, -dir_to_rc_vertex, src.z.rc_normal, src.z.rc_NEE_dir, rc_tang, rc_bitang);
            float rc_light_sample_light_pdf = cone_sample_pdf(light_cone_cos_theta_max, 1.0f);
            float rc_light_sample_mis_weight = power_heuristic(rc_light_sample_light_pdf, rc_light_sample_bsdf_pdf);
            V3 sky_T = v3(1.0f);
            if (use_physical_atmosphere == 1) {
                sky_T = atmos.sample_skybox_transmittance(src.z.rc_NEE_dir);
                if (st) st->sky_lookups++;
            }
            contrib += firefly_filter(rc_light_sample_mis_weight * rc_nee_brdf * sky_T * light_weight * light_color);
        }
        contrib += (rc_mat_id != 2) ? v3(0.0f) : rc_mat.base_col;

        V3 diffuse = primary_brdf_d * contrib;
        V3 specular = primary_brdf_s * contrib;
        float jacobian = 1.0f;
        if (!rc_is_escape_vertex) {
            jacobian = src.z.cached_jacobian_term;
            V3 dir_y1_to_x2 = src.z.rc_pos - dst_pos;
            jacobian *= dm_abs(dot(normalized(dir_y1_to_x2), src.z.rc_normal)) / dot(dir_y1_to_x2, dir_y1_to_x2);
        }
        if (jacobian < 0.0f || dm_isnan(jacobian) || dm_isinf(jacobian)) {
            jacobian = 0.0f;
            if (dm_max(jacobian, 1.0f / jacobian) > 11.0f) { /* :801: inside the jacobian = 0 branch, as shipped */
                diffuse = v3(0.0f);
                specular = v3(0.0f);
            }
        }
        *diffuse_out = diffuse;
        *specular_out = specular;
        *jacobian_out = jacobian * passed_checks;
    }

    /* ---- pathtracer.py:815-989 --------------------------------------------------------------- */
    void spatial_gris_pixel(int ui, int vi, int pass_id, float max_radius, int max_taps, int pass_total, Stats* st,
                            std::vector<V3>& out_d, std::vector<V3>& out_s) {
        const float u = (float)ui, v = (float)vi;
        const size_t pix = (size_t)vi * W + ui;
        if (is_outside_render_area(u, v)) return;
        dm_rng rng = dm_rng_init(seed, current_frame, (uint32_t)pix, 1u);
        V2 texcoord = pixel_texcoord(u, v);
        uint32_t start_index = dm_f2u32(dm_rng_f32(&rng) * (float)max_taps);
        (void)start_index;
        uint32_t seed_x = (pass_id == 0) ? ((uint32_t)ui >> 3) : 2u;
        uint32_t seed_y = (pass_id == 0) ? ((uint32_t)vi >> 3) : 2u;
        uint32_t hseed = hash3(seed_x, seed_y, 0u * 2u + (uint32_t)pass_id); /* current_frame baked at 0 */
        float angle_shift = (float)((hseed & 0x007FFFFFu) | 0x3F800000u) / 4294967295.0f * PI;
        float radius_shift = dm_rng_f32(&rng);

        const std::vector<StorageReservoir>& src_slot = spatial_reservoirs[pass_id % 2];
        Reservoir center_reservoir;
        center_reservoir.init();
        center_reservoir.decode(src_slot[pix]);
        Reservoir output_reservoir;
        output_reservoir.init();

        float center_depth = gbuff_depth[pix];
        V3 center_x1 = screen_to_view(texcoord, center_depth, proj_mat_inv);
        center_x1 = view_to_world(center_x1, view_mat_inv);
        float center_dist = distance(center_x1, camera_pos);
        V3 center_n1 = decode_unit_vector_3x16(&gbuff_normals[2 * pix]);

        if (is_vec_zero(center_x1)) {
            out_d[pix] = center_reservoir.z.F;
            return;
        }
        int center_mat_id;
        DisneyMaterial center_mat = decode_material(gbuff_mat_id[pix], &center_mat_id);
        int valid_samples = 0;
        float canonical_mis_weight = 1.0f;
        V3 chosen_F_d = v3(0.0f), chosen_F_s = v3(0.0f);

        for (int i = 0; i < max_taps; i++) {
            const float golden_angle = 2.399963229728f;
            float angle = ((float)i + angle_shift) * golden_angle;
            float offset_radius = dm_sqrt(((float)i + radius_shift) / (float)max_taps) * max_radius;
            int ox = dm_f2i(dm_cos(angle) * offset_radius), oy = dm_f2i(dm_sin(angle) * offset_radius);
            if (ox == 0 && oy == 0) continue;
            int tx = ui + ox, ty = vi + oy;
            if (tx < 0 || ty < 0 || tx >= W || ty >= H) continue; /* reference reads out of bounds here */
            size_t tpix = (size_t)ty * W + tx;
            V2 tap_texcoord = pixel_texcoord((float)tx, (float)ty);
            V3 neighbour_n1 = decode_unit_vector_3x16(&gbuff_normals[2 * tpix]);
            float neighbour_depth = gbuff_depth[tpix];
            V3 neighbour_x1 = screen_to_view(tap_texcoord, neighbour_depth, proj_mat_inv);
            neighbour_x1 = view_to_world(neighbour_x1, view_mat_inv);
            float neighbour_dist = distance(neighbour_x1, camera_pos);
            Reservoir neighbour_reservoir;
            neighbour_reservoir.init();
            neighbour_reservoir.decode(src_slot[tpix]);
            if (dm_abs(neighbour_dist - center_dist) > 0.1f * center_dist || dot(center_n1, neighbour_n1) < 0.5f) continue;
            int neighbour_mat_id;
            DisneyMaterial neighbour_mat = decode_material(gbuff_mat_id[tpix], &neighbour_mat_id);

            V3 center_integrand_d, center_integrand_s, shifted_d, shifted_s;
            float c_jacobian, jacobian;
            shift(neighbour_x1, neighbour_n1, neighbour_mat, center_x1, center_reservoir, &center_integrand_d, &center_integrand_s,
                  &c_jacobian, st);
            shift(center_x1, center_n1, center_mat, neighbour_x1, neighbour_reservoir, &shifted_d, &shifted_s, &jacobian, st);

            float center_p_hat = luminance(center_integrand_d + center_integrand_s) * c_jacobian;
            float canonical_weight = center_p_hat * neighbour_reservoir.M;
            canonical_weight /= center_p_hat * neighbour_reservoir.M +
                                luminance(center_reservoir.z.F) * center_reservoir.M / (float)max_taps;
            canonical_mis_weight += 1.0f - canonical_weight;

            float p_hat = luminance(shifted_d + shifted_s);
            float p_hat_from_neighbour = p_hat / jacobian;
            float neighbour_mis_weight = p_hat_from_neighbour * neighbour_reservoir.M;
            neighbour_mis_weight /= p_hat_from_neighbour * neighbour_reservoir.M + p_hat * center_reservoir.M / (float)max_taps;
            if (dm_isinf(neighbour_mis_weight) || dm_isnan(neighbour_mis_weight)) neighbour_mis_weight = 0.0f;

            neighbour_reservoir.z.F = shifted_d + shifted_s;
            bool selected = output_reservoir.merge(neighbour_reservoir,
                                                   neighbour_reservoir.weight * p_hat * jacobian * neighbour_mis_weight, &rng);
            if (selected) {
                chosen_F_d = shifted_d;
                chosen_F_s = shifted_s;
            }
            valid_samples += 1;
        }

        bool force_add_canonical = false;
        bool out_escape = is_vec_zero(output_reservoir.z.rc_normal);
        V3 dir_to_rc_vertex = out_escape ? output_reservoir.z.rc_pos : normalized(output_reservoir.z.rc_pos - center_x1);
        float dist;
        V3 n_, a_;
        int hl_, sm_;
        next_hit(center_x1 + center_n1 * 0.003f * center_dist, dir_to_rc_vertex, INF, true, st, &dist, &n_, &a_, &hl_, &sm_);
        float actual_dist = out_escape ? INF : distance(center_x1, output_reservoir.z.rc_pos);
        if (dist < INF && dm_abs(dist - actual_dist) > 0.1f * actual_dist) {
            output_reservoir.weight = 0.0f;
            force_add_canonical = true;
        }
        float center_p_hat = luminance(center_reservoir.z.F);
        bool selected = output_reservoir.merge(center_reservoir, center_reservoir.weight * center_p_hat * canonical_mis_weight, &rng,
                                               force_add_canonical);
        if (selected) {
            chosen_F_d = color_buffer[pix];
            chosen_F_s = color_buffer_specular[pix];
        }
        output_reservoir.finalize_without_M();
        output_reservoir.weight /= (float)(valid_samples + 1);

        if (pass_id == pass_total - 1) {
            V3 emission = (center_mat_id == 2) ? center_mat.base_col : v3(0.0f);
            if (camera_is_moving == 1) chosen_F_d /= vmax(center_mat.base_col, 1e-2f);
            float wc = dm_clamp(output_reservoir.weight, 0.0f, 50.0f);
            out_d[pix] = chosen_F_d * wc + emission;
            out_s[pix] = chosen_F_s * wc;
        }
        output_reservoir.update_cached_jacobian_term(center_x1);
        spatial_reservoirs[(pass_id + 1) % 2][pix] = output_reservoir.encode();
    }

    /* ---- temporal kernels -------------------------------------------------------------------- */
    void ires(int* ix, int* iy) const { /* cast(vec2(W, H) * render_scale, i32) */
        *ix = dm_f2i((float)W * render_scale);
        *iy = dm_f2i((float)H * render_scale);
    }
    /* pathtracer.py:1077-1090; out-of-image taps clamp to the edge */
    V3 bilinear_sample(const std::vector<V3>& buffer, V2 uv) const {
        int rx, ry;
        ires(&rx, &ry);
        float fcx = uv.x * (float)rx - 0.5f, fcy = uv.y * (float)ry - 0.5f;
        int ix = dm_f2i(fcx), iy = dm_f2i(fcy);
        float fx = fract(fcx), fy = fract(fcy);
        auto at = [&](int x, int y) {
            x = x < 0 ? 0 : (x > W - 1 ? W - 1 : x);
            y = y < 0 ? 0 : (y > H - 1 ? H - 1 : y);
            return buffer[(size_t)y * W + x];
        };
        V3 bl = at(ix, iy), br = at(ix + 1, iy), tl = at(ix, iy + 1), tr = at(ix + 1, iy + 1);
        return mix(mix(bl, br, fx), mix(tl, tr, fx), fy);
    }
    static bool scrub_needed(V3 c) { /* :1072 */
        for (int i = 0; i < 3; i++)
            if (dm_isnan(c[i]) || dm_isinf(c[i]) || c[i] < 0.0f) return true;
        return false;
    }
    /* pathtracer.py:1020-1075 (statistics outputs are dead and not produced) */
    void temporal_filter_prepass_pixel(int ui, int vi, const std::vector<float>& gdr_old) {
        if (is_outside_render_area((float)ui, (float)vi)) return;
        int rx, ry;
        ires(&rx, &ry);
        size_t pix = (size_t)vi * W + ui;
        float refl_depth_sum = 0.0f, valid = 0.0f;
        for (int x = -1; x < 3; x++)
            for (int y = -1; y < 3; y++) {
                int tx = ui + x, ty = vi + y;
                if (tx < 0 || ty < 0 || tx > rx - 1 || ty > ry - 1) continue;
                float rd = gdr_old[(size_t)ty * W + tx];
                if (rd != 0.0f) {
                    valid += 1.0f;
                    refl_depth_sum += rd;
                }
            }
        gbuff_depth_reflection[pix] = (valid > 0.01f) ? refl_depth_sum / valid : 0.0f;
        if (scrub_needed(color_buffer[pix])) color_buffer[pix] = v3(0.0f);
        if (scrub_needed(color_buffer_specular[pix])) color_buffer_specular[pix] = v3(0.0f);
    }
    V3 reproject(V3 world_pos) const { /* :993-1000 */
        V4 pos = v4(world_pos.x, world_pos.y, world_pos.z, 1.0f);
        pos = mul(prev_view_mat, pos);
        pos = mul(prev_proj_mat, pos);
        V3 p = v3(pos.x / pos.w, pos.y / pos.w, pos.z / pos.w);
        return p * 0.5f + 0.5f;
    }
    static float catmullrom(float x) { /* :1002-1014 */
        float x2 = x * x, x3 = x * x * x, fx = 0.0f;
        if (x < 1.0f) fx = 1.5f * x3 - 2.5f * x2 + 1.0f;
        else if (x < 2.0f) fx = -0.5f * x3 + 2.5f * x2 - 4.0f * x + 2.0f;
        return fx;
    }
    /* pathtracer.py:1092-1130 and 1132-1183 share this body; specular skips the depth test */
    float history_filter(const std::vector<V4>& hist, V2 uv, float center_depth, V3 center_normal, bool depth_test, V4* out) const {
        int rx, ry;
        ires(&rx, &ry);
        float fcx = uv.x * (float)rx - 0.5f, fcy = uv.y * (float)ry - 0.5f;
        int icx = dm_f2i(fcx), icy = dm_f2i(fcy);
        float fx = fract(fcx), fy = fract(fcy);
        V4 col_sum = V4{0, 0, 0, 0}, col_max = V4{0, 0, 0, 0}, col_min = V4{999999.0f, 999999.0f, 999999.0f, 999999.0f};
        float weight_sum = 0.0f;
        for (int x = -1; x < 3; x++)
            for (int y = -1; y < 3; y++) {
                int tx = icx + x, ty = icy + y;
                if (tx < 0 || ty < 0 || tx > rx - 1 || ty > ry - 1) continue;
                size_t t = (size_t)ty * W + tx;
                float w = catmullrom(dm_abs((float)x - fx)) * catmullrom(dm_abs((float)y - fy));
                V3 tap_normal = decode_unit_vector_3x16(&gbuff_prev_normals[2 * t]);
                if (camera_is_moving == 1) {
                    if (depth_test) {
                        float tap_depth = linearize_depth(gbuff_prev_depth[t], proj_mat_inv);
                        w *= (dm_abs(tap_depth - center_depth) / center_depth < 0.05f) ? 1.0f : 0.0f;
                    }
                    w *= (dot(center_normal, tap_normal) > 0.642f) ? 1.0f : 0.0f;
                }
                V4 col = hist[t];
                for (int k = 0; k < 4; k++) {
                    col_max[k] = dm_max(col_max[k], col[k]);
                    col_min[k] = dm_min(col_min[k], col[k]);
                    col_sum[k] += col[k] * w;
                }
                weight_sum += w;
            }
        const float lo[4] = {0.0f, 0.0f, 0.0f, 1.0f};
        for (int k = 0; k < 4; k++) {
            float c = col_sum[k] / weight_sum;
            (*out)[k] = dm_max(dm_clamp(c, col_min[k], col_max[k]), lo[k]);
        }
        return weight_sum;
    }
    /* pathtracer.py:1185-1230; `cb_old` is the color_buffer snapshot taken after the prepass */
    void temporal_filter_pixel(int ui, int vi, const std::vector<V3>& cb_old) {
        if (is_outside_render_area((float)ui, (float)vi)) return;
        size_t pix = (size_t)vi * W + ui;
        V2 texcoord = pixel_texcoord((float)ui, (float)vi);
        float center_nonlinear_depth = gbuff_depth[pix];
        float center_depth = linearize_depth(center_nonlinear_depth, proj_mat_inv);
        V3 center_n1 = decode_unit_vector_3x16(&gbuff_normals[2 * pix]);
        V3 center_x1 = view_to_world(screen_to_view(texcoord, center_nonlinear_depth, proj_mat_inv), view_mat_inv);
        if (is_vec_zero(center_x1)) return;
        V3 current = bilinear_sample(cb_old, texcoord);
        float w_sum = 1.0f;
        V4 history;
        if (camera_is_moving == 0) {
            history = history_buffer[0][pix];
        } else {
            V3 rp = reproject(center_x1);
            w_sum = history_filter(history_buffer[0], v2(rp.x, rp.y), linearize_depth(rp.z, proj_mat_inv), center_n1, true, &history);
        }
        (void)center_depth;
        if (w_sum > 1e-3f) {
            history.w = dm_min(history.w + 1.0f, max_accum_frames);
            V3 m = mix(v3(history.x, history.y, history.z), current, 1.0f / history.w);
            history.x = m.x; history.y = m.y; history.z = m.z;
        } else {
            history = v4(current.x, current.y, current.z, 1.0f);
        }
        int center_mat_id;
        DisneyMaterial center_mat = decode_material(gbuff_mat_id[pix], &center_mat_id);
        history_buffer[1][pix] = history;
        V3 hx = v3(history.x, history.y, history.z);
        if (camera_is_moving == 1) hx *= center_mat.base_col;
        color_buffer[pix] = hx;
    }
    /* pathtracer.py:1242-1295 */
    void temporal_filter_specular_pixel(int ui, int vi) {
        if (is_outside_render_area((float)ui, (float)vi)) return;
        size_t pix = (size_t)vi * W + ui;
        V2 texcoord = pixel_texcoord((float)ui, (float)vi);
        float center_nonlinear_depth = gbuff_depth[pix];
        V3 center_n1 = decode_unit_vector_3x16(&gbuff_normals[2 * pix]);
        V3 center_x1 = view_to_world(screen_to_view(texcoord, center_nonlinear_depth, proj_mat_inv), view_mat_inv);
        float center_refl_depth = gbuff_depth_reflection[pix];
        if (is_vec_zero(center_x1)) return;
        V3 current = bilinear_sample(color_buffer_specular, texcoord);
        float w_sum = 1.0f;
        V4 history;
        if (camera_is_moving == 0) {
            history = history_buffer_specular[0][pix];
        } else {
            float nl = delinearize_depth(center_refl_depth, proj_mat);
            V3 center_refl_pos = view_to_world(screen_to_view(texcoord, nl, proj_mat_inv), view_mat_inv);
            V3 rp = reproject((center_refl_depth != 0.0f) ? center_refl_pos : center_x1);
            w_sum = history_filter(history_buffer_specular[0], v2(rp.x, rp.y), linearize_depth(rp.z, proj_mat_inv), center_n1,
                                   false, &history);
        }
        if (w_sum > 1e-3f) {
            history.w = dm_min(history.w + 1.0f, max_accum_frames);
            V3 m = mix(v3(history.x, history.y, history.z), current, 1.0f / history.w);
            history.x = m.x; history.y = m.y; history.z = m.z;
        } else {
            history = v4(current.x, current.y, current.z, 1.0f);
        }
        history_buffer_specular[1][pix] = history;
        color_buffer[pix] += v3(history.x, history.y, history.z);
    }

    /* ---- pathtracer.py:1310-1319 ------------------------------------------------------------- */
    void accumulate() {
        /* rows needed by this shard's filters: one halo row for bilinear_sample, 24 for the
         * spatial reuse radius (pathtracer.py:1313) plus the prepass taps */
        int halo = use_restir ? 26 : 2;
        int r0 = row_begin - halo < 0 ? 0 : row_begin - halo;
        int r1 = row_end + halo > H ? H : row_end + halo;
        parallel_rows(r0, r1, [&](int a, int b, Stats* st) {
            for (int v = a; v < b; v++)
                for (int u = 0; u < W; u++) render_pixel(u, v, st);
        });
        int f0 = row_begin - 2 < 0 ? 0 : row_begin - 2, f1 = row_end + 2 > H ? H : row_end + 2;
        if (use_restir) {
            std::vector<V3> out_d = color_buffer, out_s = color_buffer_specular;
            parallel_rows(f0, f1, [&](int a, int b, Stats* st) {
                for (int v = a; v < b; v++)
                    for (int u = 0; u < W; u++) spatial_gris_pixel(u, v, 0, 24.0f, 32, 1, st, out_d, out_s);
            });
            color_buffer.swap(out_d);
            color_buffer_specular.swap(out_s);
        }
        {
            std::vector<float> gdr_old = gbuff_depth_reflection;
            parallel_rows(f0, f1, [&](int a, int b, Stats*) {
                for (int v = a; v < b; v++)
                    for (int u = 0; u < W; u++) temporal_filter_prepass_pixel(u, v, gdr_old);
            });
        }
        {
            std::vector<V3> cb_old = color_buffer;
            parallel_rows(row_begin, row_end, [&](int a, int b, Stats*) {
                for (int v = a; v < b; v++)
                    for (int u = 0; u < W; u++) temporal_filter_pixel(u, v, cb_old);
            });
        }
        parallel_rows(row_begin, row_end, [&](int a, int b, Stats*) {
            for (int v = a; v < b; v++)
                for (int u = 0; u < W; u++) temporal_filter_specular_pixel(u, v);
        });
        /* :1298-1303 */
        for (int v = row_begin; v < row_end; v++)
            for (int u = 0; u < W; u++) {
                size_t p = (size_t)v * W + u;
                history_buffer[0][p] = history_buffer[1][p];
                gbuff_prev_depth[p] = gbuff_depth[p];
                gbuff_prev_normals[2 * p] = gbuff_normals[2 * p];
                gbuff_prev_normals[2 * p + 1] = gbuff_normals[2 * p + 1];
                history_buffer_specular[0][p] = history_buffer_specular[1][p];
            }
        current_spp += 1;
        current_frame += 1;
    }
    /* pathtracer.py:664-668 */
    void reset_framebuffer() {
        current_spp = 0;
        for (int s = 0; s < 2; s++) {
            std::fill(history_buffer[s].begin(), history_buffer[s].end(), V4{0, 0, 0, 0});
            std::fill(history_buffer_specular[s].begin(), history_buffer_specular[s].end(), V4{0, 0, 0, 0});
        }
    }
    /* pathtracer.py:634-662: out[v][u][4] */
    void render_to_image(float* out) const {
        for (int j = 0; j < H; j++)
            for (int i = 0; i < W; i++) {
                V2 uv = v2((float)i / (float)W, (float)j / (float)H);
                float dx = uv.x - 0.5f, dy = uv.y - 0.5f;
                float dist = dm_sqrt(dx * dx + dy * dy);
                float darken = 1.0f - 0.9f * dm_max(dist - 0.0f, 0.0f);
                int sx = dm_f2i((float)i * render_scale), sy = dm_f2i((float)j * render_scale);
                V3 hdr = color_buffer[(size_t)sy * W + sx];
                V3 t = uchimura(hdr * darken * exposure);
                const float g = (float)(1.0 / 2.2);
                V3 ldr = saturate(v3(dm_pow(t.x, g), dm_pow(t.y, g), dm_pow(t.z, g)));
                float* o = out + ((size_t)j * W + i) * 4;
                o[0] = ldr.x; o[1] = ldr.y; o[2] = ldr.z; o[3] = 1.0f;
            }
    }
};

} /* namespace orc */
#endif
