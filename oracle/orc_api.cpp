/* orc_api.cpp -- ORACLE (test infrastructure, not product code).  Parity status: see
 * orc_renderer.h.  C entry points over the CPU restatement, loaded with ctypes by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY.  The product library
 * (voxel_rt2_amd/csrc) never includes, links or calls anything in this directory.
 *
 * The entry points mirror include/vrt_api.h one for one (orc_ instead of vrt_) so that a parity
 * test drives both sides with the same calls; the extra orc_unit_* functions expose single
 * reference functions for the known-answer and property tests.
 */
#include <cstring>
#include <cstdlib>
#include "orc_renderer.h"
#include "../include/vrt_api.h"

using namespace orc;

struct orc_ctx {
    Renderer r;
};

static M4 load_m4(const float* m) {
    M4 out;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) out.m[i][j] = m[i * 4 + j];
    return out;
}

extern "C" {

orc_ctx* orc_create(const vrt_config* cfg, int n_threads) {
    const int g = cfg ? cfg->grid_res : 0;
    if (!cfg || g < 8 || g > 512 || (g & (g - 1)) != 0 || cfg->width <= 0 || cfg->height <= 0) return nullptr;
    orc_ctx* c = new orc_ctx();
    c->r.init(cfg->width, cfg->height, cfg->dx, cfg->voxel_edges, cfg->exposure, cfg->max_depth, cfg->use_restir != 0, cfg->seed,
              cfg->sky_res > 0 ? cfg->sky_res : 64, n_threads, g);
    if (cfg->row_end > cfg->row_begin) {
        c->r.row_begin = cfg->row_begin;
        c->r.row_end = cfg->row_end;
    }
    return c;
}
void orc_destroy(orc_ctx* c) { delete c; }
/* vrt_set_reference_indexing: occupancy queries outside the grid follow the reference's own index arithmetic (orc_world.h) */
int orc_set_reference_indexing(orc_ctx* c, int on) { c->r.voxel_raytracer.reference_indexing = on != 0; return 0; }

int orc_upload_voxels(orc_ctx* c, const int8_t* mat, const uint8_t* rgb) {
    size_t n = c->r.world.voxel_material.size();
    memcpy(c->r.world.voxel_material.data(), mat, n);
    memcpy(c->r.world.voxel_color.data(), rgb, n * 3);
    return 0;
}
int orc_upload_materials(orc_ctx* c, const float* table) {
    for (int i = 0; i < 128; i++) memcpy(&c->r.mat_list[i], table + 14 * i, 14 * sizeof(float));
    return 0;
}
int orc_upload_cloud_texture(orc_ctx* c, const uint8_t* rgb) {
    memcpy(c->r.atmos.cloud_tex.data(), rgb, 256 * 256 * 3);
    return 0;
}
/* test plumbing: install sky tables computed elsewhere (the GPU's, whose equality with this oracle's own tables is tested at
 * small sizes) so that a full-size frame can be checked without the hours the 3840^2 precompute takes on a CPU */
int orc_upload_sky(orc_ctx* c, const float* scat, const float* trans) {
    memcpy(c->r.atmos.skybox_scattering.data(), scat, c->r.atmos.skybox_scattering.size() * 12);
    memcpy(c->r.atmos.skybox_transmittance.data(), trans, c->r.atmos.skybox_transmittance.size() * 12);
    return 0;
}
int orc_set_scene(orc_ctx* c, const vrt_scene_params* s) {
    Renderer& r = c->r;
    r.floor_height = s->floor_height;
    r.floor_color = v3(s->floor_color[0], s->floor_color[1], s->floor_color[2]);
    r.floor_material = s->floor_material;
    r.background_color = v3(s->background_color[0], s->background_color[1], s->background_color[2]);
    r.light_direction = v3(s->light_direction[0], s->light_direction[1], s->light_direction[2]);
    r.light_cone_cos_theta_max = s->light_cos_theta_max;
    r.light_color = v3(s->light_color[0], s->light_color[1], s->light_color[2]);
    r.light_weight = s->light_weight;
    r.use_physical_atmosphere = s->use_physical_sky;
    r.atmos.use_clouds = s->use_clouds;
    return 0;
}
int orc_set_camera(orc_ctx* c, const vrt_camera* cam) {
    Renderer& r = c->r;
    r.view_mat = load_m4(cam->view);
    r.proj_mat = load_m4(cam->proj);
    r.view_mat_inv = load_m4(cam->view_inv);
    r.proj_mat_inv = load_m4(cam->proj_inv);
    r.camera_pos = v3(cam->pos[0], cam->pos[1], cam->pos[2]);
    r.camera_is_moving = cam->camera_is_moving;
    r.render_scale = cam->render_scale;
    r.max_accum_frames = cam->max_accum_frames;
    r.draw_taa_jitter(cam->jitter_index);
    return 0;
}
int orc_prepare(orc_ctx* c) { c->r.prepare_data(); return 0; }
int orc_sky_accumulate_clouds(orc_ctx* c, int max_samples) { c->r.accumulate_clouds(max_samples); return 0; }
int orc_sky_compute_slice(orc_ctx* c, int slice, int max_slices) { c->r.compute_atmosphere(slice, max_slices); return 0; }
/* the two entry points of the sharded precompute (include/vrt_api.h), on host memory */
int orc_sky_accumulate_clouds_slice(orc_ctx* c, int max_samples, int slice, int max_slices) {
    c->r.accumulate_clouds_slice(max_samples, slice, max_slices);
    return 0;
}
int orc_sky_table_io(orc_ctx* c, int which, int u0, int u1, void* ptr, int to_library) {
    Atmos& a = c->r.atmos;
    if (which != VRT_BUF_SKY_SCATTERING && which != VRT_BUF_SKY_TRANSMITTANCE) return -1;
    if (u0 < 0 || u1 > a.res || u1 <= u0) return -1;
    V3* table = (which == VRT_BUF_SKY_SCATTERING ? a.skybox_scattering.data() : a.skybox_transmittance.data()) + (size_t)u0 * a.res;
    size_t bytes = (size_t)(u1 - u0) * a.res * sizeof(V3);
    if (to_library) memcpy(table, ptr, bytes); else memcpy(ptr, table, bytes);
    return 0;
}
int orc_accumulate(orc_ctx* c, int n) {
    for (int i = 0; i < n; i++) c->r.accumulate();
    return 0;
}
int orc_reset(orc_ctx* c) { c->r.reset_framebuffer(); return 0; }
int orc_end_frame(orc_ctx* c) { c->r.copy_prev_matrices(); return 0; }
int orc_fetch_hdr(orc_ctx* c, float* out) { /* rows outside [row_begin,row_end) are zero, like vrt_fetch_hdr */
    const Renderer& r = c->r;
    memset(out, 0, (size_t)r.W * r.H * sizeof(V3));
    memcpy(out + (size_t)r.row_begin * r.W * 3, r.color_buffer.data() + (size_t)r.row_begin * r.W,
           (size_t)(r.row_end - r.row_begin) * r.W * sizeof(V3));
    return 0;
}
int orc_fetch_ldr(orc_ctx* c, float* out) { c->r.render_to_image(out); return 0; }
/* which: see VRT_BUF_* in vrt_api.h */
int orc_fetch_buffer(orc_ctx* c, int which, void* out) {
    Renderer& r = c->r;
    size_t n = (size_t)r.W * r.H;
    switch (which) {
        case VRT_BUF_GBUF_DEPTH: memcpy(out, r.gbuff_depth.data(), n * 4); return 0;
        case VRT_BUF_GBUF_NORMAL: memcpy(out, r.gbuff_normals.data(), n * 4); return 0;
        case VRT_BUF_GBUF_POSITION: memcpy(out, r.gbuff_position.data(), n * 12); return 0;
        case VRT_BUF_GBUF_MAT: memcpy(out, r.gbuff_mat_id.data(), n * 4); return 0;
        case VRT_BUF_GBUF_REFL_DEPTH: memcpy(out, r.gbuff_depth_reflection.data(), n * 4); return 0;
        case VRT_BUF_HISTORY_DIFFUSE: memcpy(out, r.history_buffer[0].data(), n * 16); return 0;
        case VRT_BUF_HISTORY_SPECULAR: memcpy(out, r.history_buffer_specular[0].data(), n * 16); return 0;
        case VRT_BUF_SKY_SCATTERING:
            memcpy(out, r.atmos.skybox_scattering.data(), r.atmos.skybox_scattering.size() * 12); return 0;
        case VRT_BUF_SKY_TRANSMITTANCE:
            memcpy(out, r.atmos.skybox_transmittance.data(), r.atmos.skybox_transmittance.size() * 12); return 0;
        case VRT_BUF_TRANS_LUT: memcpy(out, r.atmos.trans_LUT.data(), r.atmos.trans_LUT.size() * 2); return 0;
        default: return -1;
    }
}
int orc_get_stats(orc_ctx* c, vrt_stats* s) {
    memset(s, 0, sizeof(*s));
    s->path_samples = (uint64_t)c->r.current_frame * (uint64_t)c->r.W * (uint64_t)(c->r.row_end - c->r.row_begin);
    s->rays = c->r.stats.rays;
    s->dda_iters = c->r.stats.iters;
    s->occupancy_queries = c->r.stats.queries;
    s->closest_hits = c->r.stats.closest_hits;
    s->sky_lookups = c->r.stats.sky_lookups;
    return 0;
}

/* ---- single-function entry points for known-answer / property tests ------------------------ */
int orc_unit_query_occupancy(orc_ctx* c, int x, int y, int z, int lod) {
    return c->r.voxel_raytracer.query_occupancy(I3{x, y, z}, lod) ? 1 : 0;
}
/* out: [0] distance, [1..3] cell, [4..6] normal, [7] iters (all as float) */
void orc_unit_raytrace(orc_ctx* c, const float* o, const float* d, float tmin, float tmax, float* out) {
    auto h = c->r.voxel_raytracer.raytrace(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), tmin, tmax);
    out[0] = h.distance;
    out[1] = (float)h.ipos.x; out[2] = (float)h.ipos.y; out[3] = (float)h.ipos.z;
    out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z;
    out[7] = (float)h.iters;
}
/* out: [0] closest, [1..3] normal, [4..6] albedo, [7] hit_light, [8] mat_id */
void orc_unit_next_hit(orc_ctx* c, const float* o, const float* d, int shadow, float* out) {
    float closest;
    V3 n, a;
    int hl, m;
    c->r.next_hit(v3(o[0], o[1], o[2]), v3(d[0], d[1], d[2]), INF, shadow != 0, nullptr, &closest, &n, &a, &hl, &m);
    out[0] = closest; out[1] = n.x; out[2] = n.y; out[3] = n.z; out[4] = a.x; out[5] = a.y; out[6] = a.z;
    out[7] = (float)hl; out[8] = (float)m;
}
void orc_unit_cast_dir(orc_ctx* c, int u, int v, float* out) {
    V3 d = c->r.get_cast_dir((float)u, (float)v);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
static DisneyMaterial mat_from(const float* m);
/* pathtracer.py:672-812 on an UNSTORED sample: shift(dst_pos, dst_normal, dst_material, src_pos, sample).  `sample` = 23 floats:
 * F, rc_pos, rc_normal, rc_incident_dir, rc_incident_L, rc_NEE_dir (3 each), rc_mat_info (bit pattern), cached_jacobian_term,
 * lobes.  out: diffuse rgb, specular rgb, jacobian. */
void orc_unit_shift(orc_ctx* c, const float* dst_pos, const float* dst_n, const float* dst_mat, const float* src_pos, const float* sample,
                    float* out) {
    Reservoir r;
    r.init();
    const float* p = sample;
    auto rd = [&]() { V3 v = v3(p[0], p[1], p[2]); p += 3; return v; };
    r.z.F = rd(); r.z.rc_pos = rd(); r.z.rc_normal = rd(); r.z.rc_incident_dir = rd(); r.z.rc_incident_L = rd(); r.z.rc_NEE_dir = rd();
    r.z.rc_mat_info = dm_f2u(p[0]); r.z.cached_jacobian_term = p[1]; r.z.lobes = (int)p[2];
    r.M = 1.0f; r.weight = 1.0f;
    V3 d = v3(0.0f), sp = v3(0.0f);
    float jac = 0.0f;
    c->r.shift(v3(dst_pos[0], dst_pos[1], dst_pos[2]), v3(dst_n[0], dst_n[1], dst_n[2]), mat_from(dst_mat), v3(src_pos[0], src_pos[1], src_pos[2]), r,
               &d, &sp, &jac, nullptr);
    out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = sp.x; out[4] = sp.y; out[5] = sp.z; out[6] = jac;
}
static DisneyMaterial mat_from(const float* m) {
    DisneyMaterial dm;
    memcpy(&dm, m, 14 * sizeof(float));
    return dm;
}
/* BSDF: v, n, l are unit vectors; out = diffuse rgb, specular rgb, pdf */
void orc_unit_bsdf_eval(const float* mat, const float* v, const float* n, const float* l, float* out) {
    DisneyMaterial m = mat_from(mat);
    V3 N = v3(n[0], n[1], n[2]), t, b;
    make_orthonormal_basis(N, &t, &b);
    V3 d, s;
    disney_evaluate_split(m, v3(v[0], v[1], v[2]), N, v3(l[0], l[1], l[2]), t, b, &d, &s);
    out[0] = d.x; out[1] = d.y; out[2] = d.z; out[3] = s.x; out[4] = s.y; out[5] = s.z;
    out[6] = pdf_disney(m, v3(v[0], v[1], v[2]), N, v3(l[0], l[1], l[2]), t, b);
}
/* n samples from sample_disney with stream (seed, 0, i, 0); out[i] = dir xyz, brdf rgb, pdf, lobe */
void orc_unit_bsdf_sample(const float* mat, const float* v, const float* n, uint32_t seed, int count, float* out) {
    DisneyMaterial m = mat_from(mat);
    V3 N = v3(n[0], n[1], n[2]), t, b;
    make_orthonormal_basis(N, &t, &b);
    for (int i = 0; i < count; i++) {
        dm_rng rng = dm_rng_init(seed, 0u, (uint32_t)i, 0u);
        V3 brdf;
        float pdf;
        int lobe;
        V3 d = sample_disney(m, v3(v[0], v[1], v[2]), N, t, b, &brdf, &pdf, &lobe, &rng);
        float* o = out + 8 * i;
        o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = brdf.x; o[4] = brdf.y; o[5] = brdf.z; o[6] = pdf; o[7] = (float)lobe;
    }
}
void orc_unit_lobe_pdf(const float* mat, const float* v, const float* n, const float* l, int lobe, float* out) {
    DisneyMaterial m = mat_from(mat);
    V3 N = v3(n[0], n[1], n[2]), t, b;
    make_orthonormal_basis(N, &t, &b);
    out[0] = pdf_disney_lobewise(m, v3(v[0], v[1], v[2]), N, v3(l[0], l[1], l[2]), t, b, lobe);
}
void orc_unit_sample_cone(float cos_max, const float* n, uint32_t seed, int count, float* out) {
    for (int i = 0; i < count; i++) {
        dm_rng rng = dm_rng_init(seed, 0u, (uint32_t)i, 0u);
        V3 d = sample_cone_oriented(cos_max, v3(n[0], n[1], n[2]), &rng);
        out[3 * i] = d.x; out[3 * i + 1] = d.y; out[3 * i + 2] = d.z;
    }
}
void orc_unit_oct_encode(const float* v, uint16_t* out) { encode_unit_vector_3x16(v3(v[0], v[1], v[2]), out); }
void orc_unit_oct_decode(const uint16_t* h, float* out) {
    V3 d = decode_unit_vector_3x16(h);
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
uint32_t orc_unit_encode_material(int id, const float* a) { return encode_material(id, v3(a[0], a[1], a[2])); }
uint32_t orc_unit_hash3(uint32_t x, uint32_t y, uint32_t z) { return hash3(x, y, z); }
void orc_unit_uchimura(const float* x, int n, float* out) {
    for (int i = 0; i < n; i++) out[i] = uchimura1(x[i]);
}
void orc_unit_project_sky(orc_ctx* c, const float* d, float* out) {
    V2 t = c->r.atmos.project_sky(v3(d[0], d[1], d[2]));
    out[0] = t.x; out[1] = t.y;
}
void orc_unit_unproject_sky(orc_ctx* c, const float* uv, float* out) {
    V3 d = c->r.atmos.unproject_sky(v2(uv[0], uv[1]));
    out[0] = d.x; out[1] = d.y; out[2] = d.z;
}
/* reservoir encode/decode round trip of one Sample given as 23 floats */
/* ---- atmos.py, one function at a time (tests/golden/reference/functions.npz: sky_*) ----------------------------------------
 * op: 0 rsi(pos, dir, r) -> 2 | 1 get_ozone_density(h) -> 1 | 2 get_density(h) -> 3 | 3 cloud_phase(cos, an) -> 1 |
 *     4 sample_cloud_density(pos) -> 1 | 5 clouds_shadow_od(origin, dir, dither) -> 1 | 6 get_ray_transmittance(pos, dir) -> 3 |
 *     7 clouds_scattering(origin, dir, sun_dir, sun_col, sun_cos, dither; stream index) -> scatter 3, transmittance, distance |
 *     8 / 9 atmospheric_scattering at template depth 0 / 1 (origin, dir, sun_dir, sun_col, sun_cos, steps; stream index) -> scatter 3, transmittance 3
 * `in` holds n rows of in_stride floats, `out` n rows of out_stride.  The functions that draw random numbers (7, 8, 9: the cone
 * sampler) take them from stream 2 of (seed, frame 0x3000, the row's stream index).  orc_set_trans_lut / orc_set_cloud_ambient
 * install what they read besides the cloud tile (orc_upload_cloud_texture). */
int orc_set_trans_lut(orc_ctx* c, const uint16_t* lut) { memcpy(c->r.atmos.trans_LUT.data(), lut, 256 * 128 * 3 * sizeof(uint16_t)); return 0; }
int orc_set_cloud_ambient(orc_ctx* c, const float* a) { c->r.atmos.cloud_ambient = v3(a[0], a[1], a[2]); return 0; }
int orc_unit_atmos(orc_ctx* c, int op, int n, const float* in, int in_stride, float* out, int out_stride) {
    const Atmos& A = c->r.atmos;
    for (int k = 0; k < n; k++) {
        const float* a = in + (size_t)k * in_stride;
        float* o = out + (size_t)k * out_stride;
        const V3 p = v3(a[0], a[1], a[2]), d = v3(a[3], a[4], a[5]);
        switch (op) {
            case 0: { V2 r = rsi(p, d, a[6]); o[0] = r.x; o[1] = r.y; break; }
            case 1: o[0] = A.get_ozone_density(a[0]); break;
            case 2: { V3 r = A.get_density(a[0]); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
            case 3: o[0] = A.cloud_phase(a[0], a[1]); break;
            case 4: o[0] = A.sample_cloud_density(p); break;
            case 5: o[0] = A.clouds_shadow_od(p, d, a[6]); break;
            case 6: { V3 r = A.get_ray_transmittance(p, d); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
            case 7: {
                dm_rng rng = dm_rng_init(A.seed, 0x3000u, (uint32_t)a[14], 2u);
                V3 sc; float tr, dist;
                A.clouds_scattering(p, d, v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), a[12], a[13], &rng, &sc, &tr, &dist);
                o[0] = sc.x; o[1] = sc.y; o[2] = sc.z; o[3] = tr; o[4] = dist;
                break;
            }
            case 8: case 9: {
                dm_rng rng = dm_rng_init(A.seed, 0x3000u, (uint32_t)a[14], 2u);
                V3 sc, tr;
                A.atmospheric_scattering(p, d, v3(a[6], a[7], a[8]), v3(a[9], a[10], a[11]), a[12], op - 8, (int)a[13], &rng, &sc, &tr);
                o[0] = sc.x; o[1] = sc.y; o[2] = sc.z; o[3] = tr.x; o[4] = tr.y; o[5] = tr.z;
                break;
            }
            default: return -1;
        }
    }
    return 0;
}
void orc_unit_reservoir_roundtrip(const float* in, float* out) {
    Reservoir r;
    r.init();
    const float* p = in;
    auto rd = [&]() { V3 v = v3(p[0], p[1], p[2]); p += 3; return v; };
    r.z.F = rd(); r.z.rc_pos = rd(); r.z.rc_normal = rd(); r.z.rc_incident_dir = rd(); r.z.rc_incident_L = rd(); r.z.rc_NEE_dir = rd();
    r.z.rc_mat_info = dm_f2u(p[0]); r.z.cached_jacobian_term = p[1]; r.z.lobes = (int)p[2]; r.M = p[3]; r.weight = p[4];
    StorageReservoir e = r.encode();
    Reservoir q;
    q.init();
    q.decode(e);
    float* o = out;
    auto wr = [&](V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o += 3; };
    wr(q.z.F); wr(q.z.rc_pos); wr(q.z.rc_normal); wr(q.z.rc_incident_dir); wr(q.z.rc_incident_L); wr(q.z.rc_NEE_dir);
    o[0] = dm_u2f(q.z.rc_mat_info); o[1] = q.z.cached_jacobian_term; o[2] = (float)q.z.lobes; o[3] = q.M; o[4] = q.weight;
}

/* detmath evaluators: op 0 sin 1 cos 2 exp 3 log 4 pow(a,b) 5 acos 6 atan2(a,b) 7 min 8 max 9 f16 round trip */
void orc_unit_detmath(int op, int n, const float* a, const float* b, float* out) {
    for (int i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = dm_sin(a[i]); break;
            case 1: out[i] = dm_cos(a[i]); break;
            case 2: out[i] = dm_exp(a[i]); break;
            case 3: out[i] = dm_log(a[i]); break;
            case 4: out[i] = dm_pow(a[i], b[i]); break;
            case 5: out[i] = dm_acos(a[i]); break;
            case 6: out[i] = dm_atan2(a[i], b[i]); break;
            case 7: out[i] = dm_min(a[i], b[i]); break;
            case 8: out[i] = dm_max(a[i], b[i]); break;
            case 9: out[i] = dm_round_f16(a[i]); break;
            default: out[i] = 0.0f;
        }
    }
}
void orc_unit_rng(uint32_t seed, uint32_t frame, uint32_t index, uint32_t stream, int n, float* out) {
    dm_rng r = dm_rng_init(seed, frame, index, stream);
    for (int i = 0; i < n; i++) out[i] = dm_rng_f32(&r);
}

} /* extern "C" */
