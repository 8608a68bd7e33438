// emul.cpp -- TEST TOOLING: a g++ build of the product's per-path device functions
// (voxel_rt2_amd/csrc/*.h with VRT_DEV = inline), stepped one pixel at a time on the host.
// It exists so that the CPU test-suite (no GPU in CI) can check the device code's arithmetic
// against the oracle bit for bit, and so that sanitizers can run over it.  It is NOT a backend:
// nothing in voxel_rt2_amd/ links or loads it.
#include <cstring>
#include <vector>
#include "../../include/vrt_api.h"
#include "../../voxel_rt2_amd/csrc/vrt_types.h"
#include "../../voxel_rt2_amd/csrc/vrt_trace.h"
#include "../../voxel_rt2_amd/csrc/vrt_bsdf.h"
#include "../../voxel_rt2_amd/csrc/vrt_sky.h"
#include "../../voxel_rt2_amd/csrc/vrt_path.h"
#include "../../voxel_rt2_amd/csrc/vrt_pool.h"
#include <cstdlib>
#include "../../voxel_rt2_amd/csrc/vrt_restir.h"
#include "../../voxel_rt2_amd/csrc/vrt_temporal.h"

using namespace vrt;

struct Emu {
    vrt_config cfg;
    vrt_scene_params scene;
    vrt_camera cam;
    std::vector<uint32_t> grid;
    std::vector<unsigned long long> l0, l1, l2, l3, l0c;
    std::vector<uint32_t> l0c_base;  // [512] + count
    float cull[16];                  // k_cull_box's grown box + flag, then the same with the flag off
    std::vector<float> mats, mats_x, sky_scat, sky_trans;
    std::vector<GrisGeo> gris_geo;
    std::vector<GrisSrc> gris_src;
    std::vector<GrisTest> gris_tst;
    int buf0, buf1, own0, own1;
    size_t n;
    std::vector<f3> cbuf[2], color_s, color_d2, color_s2, gb_pos;
    int cidx = 0;
    std::vector<uint32_t> gb_normal[2], gb_mat;
    std::vector<float> gb_depth[2], gb_refl, gb_refl_f;
    std::vector<f4> hist_d[2], hist_s[2];
    std::vector<ReservoirRec> res[2];
    int cur = 0, hist_in = 0;
    uint32_t frame = 0;
    mat4 prev_view{}, prev_proj{};
    TraceStats ts;
    bool ref_oob = false;            // vrt_set_reference_indexing
};

static FrameParams frame_params(const Emu* c) {
    FrameParams fp;
    memset(&fp, 0, sizeof(fp));
    memcpy(fp.view.m, c->cam.view, 64); memcpy(fp.proj.m, c->cam.proj, 64);
    memcpy(fp.view_inv.m, c->cam.view_inv, 64); memcpy(fp.proj_inv.m, c->cam.proj_inv, 64);
    fp.camera_pos = mk3(c->cam.pos[0], c->cam.pos[1], c->cam.pos[2]);
    const int W = c->cfg.width, H = c->cfg.height;
    fp.inv_res = mk2((float)(1.0 / (double)W), (float)(1.0 / (double)H));
    dm_rng rng = dm_rng_init(c->cfg.seed, c->cam.jitter_index, 0u, 3u);
    float r0 = dm_rng_f32(&rng), r1 = dm_rng_f32(&rng);
    fp.taa_jitter = mk2((r0 * 2.0f - 1.0f) * fp.inv_res.x, (r1 * 2.0f - 1.0f) * fp.inv_res.y);
    fp.W = W; fp.H = H; fp.row0 = c->buf0; fp.row1 = c->buf1;
    fp.camera_is_moving = c->cam.camera_is_moving;
    fp.render_scale = c->cam.render_scale; fp.max_accum_frames = c->cam.max_accum_frames;
    fp.light_dir = mk3(c->scene.light_direction[0], c->scene.light_direction[1], c->scene.light_direction[2]);
    fp.light_color = mk3(c->scene.light_color[0], c->scene.light_color[1], c->scene.light_color[2]);
    fp.light_cos_max = c->scene.light_cos_theta_max; fp.light_weight = c->scene.light_weight;
    fp.floor_height = c->scene.floor_height;
    fp.floor_color = mk3(c->scene.floor_color[0], c->scene.floor_color[1], c->scene.floor_color[2]);
    fp.floor_material = c->scene.floor_material;
    fp.background = mk3(c->scene.background_color[0], c->scene.background_color[1], c->scene.background_color[2]);
    fp.use_sky = c->scene.use_physical_sky;
    fp.voxel_edges = c->cfg.voxel_edges; fp.exposure = c->cfg.exposure; fp.max_depth = c->cfg.max_depth;
    fp.seed = c->cfg.seed; fp.frame = c->frame;
    return fp;
}

extern "C" {

Emu* emu_create(const vrt_config* cfg) {
    Emu* c = new Emu();
    c->cfg = *cfg;
    c->own0 = 0; c->own1 = cfg->height;
    if (cfg->row_end > cfg->row_begin) { c->own0 = cfg->row_begin; c->own1 = cfg->row_end; }
    int halo = cfg->use_restir ? 26 : 2;
    c->buf0 = c->own0 - halo < 0 ? 0 : c->own0 - halo;
    c->buf1 = c->own1 + halo > cfg->height ? cfg->height : c->own1 + halo;
    size_t n = c->n = (size_t)(c->buf1 - c->buf0) * cfg->width;
    if (cfg->grid_res != 128 && cfg->grid_res != 256) { delete c; return nullptr; }
    const size_t nv = (size_t)cfg->grid_res * cfg->grid_res * cfg->grid_res;
    c->grid.assign(nv, 0); c->l0.assign(nv / 64, 0); c->l1.assign(nv / 4096, 0); c->l2.assign(nv / 262144, 0); c->l3.assign(1, 0);
    c->mats.assign(128 * 14, 0.0f);
    f3 z = mk3(0.0f);
    c->cbuf[0].assign(n, z); c->cbuf[1].assign(n, z); c->color_s.assign(n, z); c->color_d2.assign(n, z); c->color_s2.assign(n, z);
    c->gb_pos.assign(n, z); c->gb_mat.assign(n, 0); c->gb_refl.assign(n, 0); c->gb_refl_f.assign(n, 0);
    for (int s = 0; s < 2; s++) {
        c->gb_normal[s].assign(n, 0); c->gb_depth[s].assign(n, 0);
        c->hist_d[s].assign(n, mk4(0, 0, 0, 0)); c->hist_s[s].assign(n, mk4(0, 0, 0, 0));
        c->res[s].assign(n, ReservoirRec{});
    }
    if (cfg->sky_res > 0) {
        c->sky_scat.assign((size_t)cfg->sky_res * cfg->sky_res * 3, 0.0f);
        c->sky_trans.assign((size_t)cfg->sky_res * cfg->sky_res * 3, 0.0f);
    }
    stats_zero(c->ts);
    return c;
}
void emu_destroy(Emu* c) { delete c; }
int emu_upload_voxels(Emu* c, const int8_t* mat, const uint8_t* rgb) {
    const int G = c->cfg.grid_res, n = G * G * G, n0 = G / 4;
    for (int i = 0; i < n; i++) {  // k_pack_grid
        int m = mat[i];
        uint32_t a = (m < 0) ? 0u : (uint32_t)m;
        const int z = i % G, y = (i / G) % G, x = i / (G * G);
        c->grid[G == 256 ? texel_index<256>(x, y, z) : texel_index<128>(x, y, z)] =
            (uint32_t)rgb[3 * i] | ((uint32_t)rgb[3 * i + 1] << 8) | ((uint32_t)rgb[3 * i + 2] << 16) | (a << 24);
    }
    for (int b = 0; b < n0 * n0 * n0; b++) {  // k_build_l0
        int bx = b % n0, by = (b / n0) % n0, bz = b / (n0 * n0);
        unsigned long long w = 0;
        for (int z = 0; z < 4; z++) for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++)
            if (mat[((size_t)(bx * 4 + x) * G + (by * 4 + y)) * G + (bz * 4 + z)] > 0) w |= 1ULL << (z * 16 + y * 4 + x);
        c->l0[b] = w;
    }
    auto coarse = [](const std::vector<unsigned long long>& fine, std::vector<unsigned long long>& out, int nc) {
        int nf = nc * 4;
        for (int b = 0; b < nc * nc * nc; b++) {
            int bx = b % nc, by = (b / nc) % nc, bz = b / (nc * nc);
            unsigned long long w = 0;
            for (int z = 0; z < 4; z++) for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++)
                if (fine[((bz * 4 + z) * nf + (by * 4 + y)) * nf + (bx * 4 + x)] != 0) w |= 1ULL << (z * 16 + y * 4 + x);
            out[b] = w;
        }
    };
    {   // k_cull_box
        int lo[3] = {1 << 20, 1 << 20, 1 << 20}, hi[3] = {-(1 << 20), -(1 << 20), -(1 << 20)};
        for (int b = 0; b < n0 * n0 * n0; b++) {
            if (c->l0[b] == 0ULL) continue;
            const int cc[3] = {b % n0, (b / n0) % n0, b / (n0 * n0)};
            for (int a = 0; a < 3; a++) { lo[a] = lo[a] < cc[a] * 4 ? lo[a] : cc[a] * 4; hi[a] = hi[a] > cc[a] * 4 + 4 ? hi[a] : cc[a] * 4 + 4; }
        }
        bool some = false;
        for (int a = 0; a < 3; a++) {
            c->cull[a] = (float)lo[a] - VRT_CULL_MARGIN; c->cull[3 + a] = (float)hi[a] + VRT_CULL_MARGIN;
            c->cull[8 + a] = -1e30f; c->cull[11 + a] = 1e30f;
            some = some || c->cull[a] > 0.0f || c->cull[3 + a] < (float)G;
        }
        c->cull[6] = some ? 1.0f : 0.0f; c->cull[7] = 0.0f; c->cull[14] = 0.0f; c->cull[15] = 0.0f;
    }
    coarse(c->l0, c->l1, G / 16);
    coarse(c->l1, c->l2, G / 64);
    c->l0c.assign(32768, 0ULL);
    c->l0c_base.assign(513, 0u);
    if (G == 256) { coarse(c->l2, c->l3, 1); return 0; }
    // k_build_l0c: the fine level without its empty words (128^3 only)
    uint32_t k = 0;
    for (int i = 0; i < 512; i++) {
        c->l0c_base[i] = k;
        int bx1 = i & 7, by1 = (i >> 3) & 7, bz1 = i >> 6;
        for (int b = 0; b < 64; b++)
            if ((c->l1[i] >> b) & 1ULL)
                c->l0c[k++] = c->l0[((bz1 * 4 + (b >> 4)) * 32 + (by1 * 4 + ((b >> 2) & 3))) * 32 + (bx1 * 4 + (b & 3))];
    }
    c->l0c_base[512] = k;
    return 0;
}
static void derive_materials(Emu* c) {
    c->mats_x.assign(128 * 8, 0.0f);
    for (int id = 0; id < 128; id++) { const Material m = load_material(c->mats.data(), id); store_mat_derived(c->mats_x.data(), id, mat_derive(m), material_unit_range(m)); }
}
int emu_upload_materials(Emu* c, const float* t) { memcpy(c->mats.data(), t, 128 * 14 * 4); derive_materials(c); return 0; }
int emu_upload_cloud_texture(Emu*, const uint8_t*) { return 0; }
int emu_set_scene(Emu* c, const vrt_scene_params* s) { c->scene = *s; return 0; }
int emu_set_camera(Emu* c, const vrt_camera* cam) { c->cam = *cam; return 0; }
int emu_prepare(Emu*) { return 0; }
int emu_set_reference_indexing(Emu* c, int on) { c->ref_oob = on != 0; return 0; }
// The rows a striped launch renders (vrt_set_row_stripes): the render kernels' enumeration -- tile rows through launch_tile_row(), pixels
// through launch_renders_row() (vrt_types.h) -- marked in out[H].  Returns the number of tile rows enumerated.
int emu_unit_stripe_rows(int H, int stripe_rows, int n_parts, int part, uint8_t* out) {
    FrameParams fp;
    memset(&fp, 0, sizeof(fp));
    fp.H = H; fp.row0 = 0; fp.row1 = H;
    fp.stripe_rows = stripe_rows; fp.stripe_period = stripe_rows * n_parts; fp.stripe_first = part * stripe_rows;
    int n = 0;
    for (int s0 = fp.stripe_first; s0 < H; s0 += fp.stripe_period) n++;
    fp.stripe_tile_rows = n * (stripe_rows / 8 + 2);
    for (int t = 0; t < launch_tile_rows(fp); t++)
        for (int r = 0; r < 8; r++) {
            const int v = launch_tile_row(fp, t) + r;
            if (v < fp.row1 && launch_renders_row(fp, v)) out[v] += 1;
        }
    return launch_tile_rows(fp);
}
int emu_upload_sky(Emu* c, const float* scat, const float* trans) {
    memcpy(c->sky_scat.data(), scat, c->sky_scat.size() * 4);
    memcpy(c->sky_trans.data(), trans, c->sky_trans.size() * 4);
    return 0;
}

}  // extern "C"

template <int G, bool RESTIR>
static void render_all(Emu* c, const FrameParams& fp, const SceneData& sc, const PixelBuffers& out) {
    GlobalPyramid<G> P;
    P.p = sc.pyr;
    for (int v = fp.row0; v < fp.row1; v++)
        for (int u = 0; u < fp.W; u++) {
            if (outside_render_area(fp, (float)u, (float)v)) continue;
            Path<RESTIR> p;
            path_begin(fp, p, u, v, 0);
            int idx = (v - fp.row0) * fp.W + u;
            while (!path_segment<RESTIR>(fp, sc, P, out, idx, p, c->ts)) {}
            path_finish<RESTIR>(fp, sc, out, idx, p, c->ts);
        }
}

// The pool kernel's stage functions (vrt_pool.h) stepped one path at a time: the record goes through the same
// packed slot (here a 25-dword array) and scratch line as on the device, and every walk is suspended and
// resumed every third step so that the packing of a half-done walk is exercised too.
template <int G, bool RESTIR>
static void render_all_pool(Emu* c, const FrameParams& fp, const SceneData& sc, const PixelBuffers& out) {
    GlobalPyramid<G> P;
    P.p = sc.pyr;
    // the launcher's choice of kernel variant (vrt_kernels.hip, launch_render_pool)
    const bool black_sun = !RESTIR && !((fp.light_color.x != 0.0f || fp.light_color.y != 0.0f || fp.light_color.z != 0.0f) && fp.light_weight != 0.0f);
    for (int v = fp.row0; v < fp.row1; v++)
        for (int u = 0; u < fp.W; u++) {
            if (outside_render_area(fp, (float)u, (float)v)) continue;
            uint32_t slot[PF_COUNT] = {0}, cold[ColdLine<RESTIR>::count] = {0};
            SlotRef s{slot, 1};
            int st = pool_begin<G>(fp, sc.cull, s, u, v, 0, c->ts);
            while (st != SLOT_EMPTY) {
                if (st == SLOT_RAY) {
                    RayWalk w;
                    walk_load<G>(s, w);
                    BrickCache bc{-1, 0ULL};
                    CoarseWords cw;
                    coarse_fetch(P, w.ix, w.iy, w.iz, cw);
                    const int iters0 = w.iters;
                    for (int k = 1;; k++) {
                        int nq;
                        const bool fin = walk_trip(P, w, bc, cw, nq);
                        c->ts.queries += (unsigned)nq;
                        if (fin) break;
                        if (k % 3 == 0) { walk_store(s, w); walk_load<G>(s, w); bc.key = -1; coarse_fetch(P, w.ix, w.iy, w.iz, cw); }
                    }
                    c->ts.iters += (unsigned)(w.iters - iters0);
                    walk_store(s, w);
                    st = slot_state_after_walk<G>(w.t, s.f(PF_FLOOR_T));
                } else if (st == SLOT_SHADE) {
                    if constexpr (RESTIR) st = pool_shade<HIT_SOMETHING, false, true>(fp, sc, P, out, s, cold, c->ts);
                    else st = black_sun ? pool_shade<HIT_SOMETHING, true>(fp, sc, P, out, s, cold, c->ts)
                                        : pool_shade<HIT_SOMETHING, false>(fp, sc, P, out, s, cold, c->ts);
                } else {
                    st = pool_shade<HIT_NOTHING, false, RESTIR>(fp, sc, P, out, s, cold, c->ts);
                }
            }
        }
}

template <int G>
static int accumulate_g(Emu* c, int n_samples) {
    for (int s = 0; s < n_samples; s++) {
        FrameParams fp = frame_params(c);
        SceneData sc;
        sc.pyr.l0 = c->l0.data(); sc.pyr.l1 = c->l1.data(); sc.pyr.l2 = c->l2.data(); sc.pyr.l3 = c->l3.data();
        sc.pyr.l0c = c->l0c.data(); sc.pyr.l0c_base = c->l0c_base.data(); sc.pyr.l0c_count = c->l0c_base.data() + 512;
        sc.pyr.ref_oob = c->ref_oob ? 1 : 0;
        sc.grid = c->grid.data(); sc.mats = c->mats.data();
        sc.sky.scattering = c->sky_scat.data(); sc.sky.transmittance = c->sky_trans.data();
        sc.sky.res = c->cfg.sky_res; sc.sky.fres = c->cfg.sky_res > 0 ? (float)(1.0 / (double)c->cfg.sky_res) : 0.0f;
        sc.counters = nullptr;
        // off by default: the CPU tests compare traversal counters with the oracle's; and off with the reference's indexing, where
        // a ray clear of every solid voxel can still "hit" outside the grid (vrt_api.hip, culling())
        sc.cull = c->cull + (getenv("VRT_EMU_CULL") && !c->ref_oob ? 0 : 8);
        PixelBuffers out;
        f3* rt = c->cbuf[c->cidx].data();
        f3* hdr = c->cbuf[c->cidx ^ 1].data();
        out.color_d = rt; out.color_s = c->color_s.data();
        if (c->cam.render_scale != 1.0f) {  // as vrt_api.hip: a partial pass starts from the last pass's g-buffer
            c->gb_normal[c->cur] = c->gb_normal[c->cur ^ 1];
            c->gb_depth[c->cur] = c->gb_depth[c->cur ^ 1];
        }
        out.gb_normal = c->gb_normal[c->cur].data(); out.gb_depth = c->gb_depth[c->cur].data();
        out.gb_refl_depth = c->gb_refl.data(); out.gb_position = c->gb_pos.data(); out.gb_mat = c->gb_mat.data();
        out.reservoir = c->res[0].data();
        out.sample_stride = 0;
        const f3* cd = rt;
        const f3* cs = c->color_s.data();
        if (c->cfg.use_restir) {
            if (getenv("VRT_EMU_POOL")) render_all_pool<G, true>(c, fp, sc, out);
            else render_all<G, true>(c, fp, sc, out);
            GrisBuffers gb;
            gb.color_d_in = rt; gb.color_s_in = c->color_s.data();
            gb.color_d_out = c->color_d2.data(); gb.color_s_out = c->color_s2.data();
            gb.gb_normal = out.gb_normal; gb.gb_depth = out.gb_depth; gb.gb_mat = out.gb_mat;
            gb.res_in = c->res[0].data(); gb.res_out = c->res[1].data();
            c->gris_geo.resize((size_t)(fp.row1 - fp.row0) * fp.W); c->gris_src.resize(c->gris_geo.size()); c->gris_tst.resize(c->gris_geo.size());
            if (c->mats_x.empty()) derive_materials(c);
            gb.geo = c->gris_geo.data(); gb.src = c->gris_src.data(); gb.tst = c->gris_tst.data(); gb.mats_x = c->mats_x.data();
            for (int v = fp.row0; v < fp.row1; v++)
                for (int u = 0; u < fp.W; u++) gris_prepare_pixel(fp, sc, gb, u, v);
            GlobalPyramid<G> P;
            P.p = sc.pyr;
            int g0 = c->own0 - 2 < c->buf0 ? c->buf0 : c->own0 - 2, g1 = c->own1 + 2 > c->buf1 ? c->buf1 : c->own1 + 2;
            // VRT_EMU_GRIS_SPLIT: the pass as the GPU runs it, two kernels (every pixel's first half, then every pixel's second half)
            const bool split = getenv("VRT_EMU_GRIS_SPLIT") != nullptr;
            for (int phase = split ? 1 : 0; phase <= (split ? 3 : 0); phase++)      // split: classify (here 1), first half (2), second half (3)
                for (int v = g0; v < g1; v++)
                    for (int u = 0; u < fp.W; u++) {
                        float cs[64];
                        uint16_t off[32];
                        for (int i = 0; i < 32; i++) gris_tap_cs(u, v, 0, i, cs);
                        GrisTaps taps;
                        taps.cs = cs; taps.off = off; taps.off_stride = 1;
                        if (split && phase == 1) { gris_classify_pixel(fp, gb, taps, u, v, 24.0f, 32, c->ts); continue; }
                        if (split) { if (phase == 2) gris_pixel<1>(fp, sc, P, gb, taps, u, v, 0, 24.0f, 32, 1, c->ts); else gris_pixel<2>(fp, sc, P, gb, taps, u, v, 0, 24.0f, 32, 1, c->ts); continue; }
                        if (phase == 0) gris_pixel<0>(fp, sc, P, gb, taps, u, v, 0, 24.0f, 32, 1, c->ts);
                        else if (phase == 1) gris_pixel<1>(fp, sc, P, gb, taps, u, v, 0, 24.0f, 32, 1, c->ts);
                        else gris_pixel<2>(fp, sc, P, gb, taps, u, v, 0, 24.0f, 32, 1, c->ts);
                    }
            cd = c->color_d2.data();
            cs = c->color_s2.data();
        } else if (getenv("VRT_EMU_POOL")) {
            render_all_pool<G, false>(c, fp, sc, out);
        } else {
            render_all<G, false>(c, fp, sc, out);
        }
        TemporalBuffers tb;
        tb.color_d = cd; tb.color_s = cs;
        tb.gb_normal = out.gb_normal; tb.gb_depth = out.gb_depth; tb.gb_mat = out.gb_mat;
        tb.gb_refl_raw = c->gb_refl.data(); tb.gb_refl_filtered = c->gb_refl_f.data();
        tb.hist_d_in = c->hist_d[c->hist_in].data(); tb.hist_d_out = c->hist_d[c->hist_in ^ 1].data();
        tb.hist_s_in = c->hist_s[c->hist_in].data(); tb.hist_s_out = c->hist_s[c->hist_in ^ 1].data();
        tb.prev_normal = c->gb_normal[c->cur ^ 1].data(); tb.prev_depth = c->gb_depth[c->cur ^ 1].data();
        tb.hdr = hdr;
        tb.sample_stride = 0;
        tb.prev_view = c->prev_view; tb.prev_proj = c->prev_proj;
        tb.tile = nullptr; tb.tile_row0 = 0;
        for (int v = c->own0; v < c->own1; v++)
            for (int u = 0; u < fp.W; u++) temporal_pixel(fp, tb, u, v, 1);
        c->hist_in ^= 1;
        c->cur ^= 1;
        c->cidx ^= 1;
        c->frame += 1;
    }
    return 0;
}

extern "C" {

int emu_accumulate(Emu* c, int n_samples) { return c->cfg.grid_res == 256 ? accumulate_g<256>(c, n_samples) : accumulate_g<128>(c, n_samples); }
int emu_reset(Emu* c) {
    for (int s = 0; s < 2; s++) {
        std::fill(c->hist_d[s].begin(), c->hist_d[s].end(), mk4(0, 0, 0, 0));
        std::fill(c->hist_s[s].begin(), c->hist_s[s].end(), mk4(0, 0, 0, 0));
    }
    return 0;
}
int emu_end_frame(Emu* c) { memcpy(c->prev_view.m, c->cam.view, 64); memcpy(c->prev_proj.m, c->cam.proj, 64); return 0; }
static void fetch_rows(Emu* c, const void* buf, size_t elem, void* out) {
    size_t W = c->cfg.width;
    memset(out, 0, (size_t)c->cfg.height * W * elem);
    memcpy((char*)out + (size_t)c->own0 * W * elem, (const char*)buf + (size_t)(c->own0 - c->buf0) * W * elem,
           (size_t)(c->own1 - c->own0) * W * elem);
}
int emu_fetch_hdr(Emu* c, float* out) { fetch_rows(c, c->cbuf[c->cidx].data(), 12, out); return 0; }
int emu_fetch_ldr(Emu* c, float* out) {
    FrameParams fp = frame_params(c);
    std::vector<f4> ldr(c->n);
    for (int v = c->own0; v < c->own1; v++)
        for (int u = 0; u < fp.W; u++) ldr[(v - fp.row0) * fp.W + u] = tonemap_pixel(fp, c->cbuf[c->cidx].data(), u, v);
    fetch_rows(c, ldr.data(), 16, out);
    return 0;
}
int emu_fetch_buffer(Emu* c, int which, void* out) {
    int last = c->cur ^ 1;
    switch (which) {
        case VRT_BUF_GBUF_DEPTH: fetch_rows(c, c->gb_depth[last].data(), 4, out); return 0;
        case VRT_BUF_GBUF_NORMAL: fetch_rows(c, c->gb_normal[last].data(), 4, out); return 0;
        case VRT_BUF_GBUF_POSITION: fetch_rows(c, c->gb_pos.data(), 12, out); return 0;
        case VRT_BUF_GBUF_MAT: fetch_rows(c, c->gb_mat.data(), 4, out); return 0;
        case VRT_BUF_GBUF_REFL_DEPTH: fetch_rows(c, c->gb_refl_f.data(), 4, out); return 0;
        case VRT_BUF_HISTORY_DIFFUSE: fetch_rows(c, c->hist_d[c->hist_in].data(), 16, out); return 0;
        case VRT_BUF_HISTORY_SPECULAR: fetch_rows(c, c->hist_s[c->hist_in].data(), 16, out); return 0;
    }
    return -1;
}
int emu_get_stats(Emu* c, vrt_stats* s) {
    memset(s, 0, sizeof(*s));
    s->path_samples = (uint64_t)c->frame * c->cfg.width * (c->own1 - c->own0);
    s->rays = c->ts.rays; s->dda_iters = c->ts.iters; s->occupancy_queries = c->ts.queries;
    s->closest_hits = c->ts.closest_hits; s->sky_lookups = c->ts.sky_lookups;
    return 0;
}

// The spatial-reuse kernel evaluates the BSDF through bsdf_eval_pdf() with per-vertex shared terms and a per-material
// table (vrt_bsdf.h); the render kernels through eval_lobes() / pdf_lobe() / pdf_all() / surf_init().  Both must give
// the same bits: random materials (all parameters live), normals, view and light directions on either side of the
// surface, every lobe code.  Returns the number of mismatching cases.
int emu_bsdf_selftest(int n, uint32_t seed) {
    dm_rng rng = dm_rng_init(seed, 0u, 0u, 7u);
    auto rnd = [&]() { return dm_rng_f32(&rng); };
    auto unit = [&]() { f3 v = mk3(rnd() * 2.0f - 1.0f, rnd() * 2.0f - 1.0f, rnd() * 2.0f - 1.0f); return norm3(v + mk3(1e-3f, 0.0f, 0.0f)); };
    auto same = [](float a, float b) { return dm_f2u(a) == dm_f2u(b) || (a != a && b != b); };
    int bad = 0;
    for (int k = 0; k < n; k++) {
        Material m;
        const bool plain = (k & 3) == 0;  // a quarter of the cases with the defaults' zeros (sheen, clearcoat, metallic ...)
        m.base = mk3(rnd(), rnd(), rnd());
        if ((k & 15) == 5) m.base = mk3(0.0f);
        m.subsurface = plain ? 0.0f : rnd(); m.metallic = plain ? 0.0f : rnd(); m.specular = rnd(); m.specular_tint = plain ? 0.0f : rnd();
        m.roughness = (k & 7) == 3 ? 0.0f : rnd(); m.anisotropic = plain ? 0.0f : rnd(); m.sheen = plain ? 0.0f : rnd(); m.sheen_tint = rnd();
        m.clearcoat = plain ? 0.0f : rnd(); m.clearcoat_gloss = rnd(); m.ior_minus_one = 0.0f;
        const f3 n1 = unit(), v = unit(), l = unit();
        float table[8];
        store_mat_derived(table, 0, mat_derive(m));
        Surf a, b;
        surf_init(a, m, n1, v);
        f3 tx, ty;
        ortho_basis(n1, tx, ty);
        surf_set(b, m, load_mat_derived(table, 0), n1, v, cross3(n1, ty), ty);
        const int lobes[6] = {LOBE_DIFFUSE, LOBE_SPEC, LOBE_CLEARCOAT, LOBE_ALL, 5, -3};
        for (int q = 0; q < 6; q++) {
            const int lobe = lobes[q];
            f3 rd, rs, xd, xs;
            eval_lobes(a, l, lobe, rd, rs);
            const float rp_lobe = pdf_lobe(a, l, lobe), rp_all = pdf_all(a, l);
            for (int mode = 0; mode < 3; mode++) {
                const bool all = mode == PDF_ALL;
                const SurfShared c = surf_shared(b, all || lobe_has(lobe, LOBE_DIFFUSE), all || lobe_has(lobe, LOBE_SPEC), all || lobe_has(lobe, LOBE_CLEARCOAT));
                float xp;
                bsdf_eval_pdf(b, c, l, lobe, mode, xd, xs, xp);
                bool ok = same(rd.x, xd.x) && same(rd.y, xd.y) && same(rd.z, xd.z) && same(rs.x, xs.x) && same(rs.y, xs.y) && same(rs.z, xs.z);
                if (mode == PDF_LOBE) ok = ok && same(rp_lobe, xp);
                if (mode == PDF_ALL) ok = ok && same(rp_all, xp);
                if (!ok) bad++;
                // the same with the terms that do not depend on the view vector taken from the prepare pass's helpers
                const SurfShared cv = surf_shared_view(b, mat_colours(m), all || lobe_has(lobe, LOBE_DIFFUSE), all || lobe_has(lobe, LOBE_SPEC),
                                                       all || lobe_has(lobe, LOBE_CLEARCOAT));
                f3 yd, ys;
                float yp;
                bsdf_eval_pdf_pre(b, cv, l, dir_terms(b.n, b.tx, b.ty, b.ax, b.ay, l), lobe, mode, yd, ys, yp);
                bool ok2 = same(yd.x, xd.x) && same(yd.y, xd.y) && same(yd.z, xd.z) && same(ys.x, xs.x) && same(ys.y, xs.y) && same(ys.z, xs.z) && same(yp, xp);
                if (!ok2) bad++;
            }
        }
    }
    return bad;
}
// host definition of the two half conversions (include/vrt_detmath.h), for tests that pin numpy's against it
void emu_half_probe(int op, int n, const uint32_t* in, uint32_t* out) {
    for (int i = 0; i < n; i++) out[i] = op == 14 ? (uint32_t)dm_f32_to_f16(dm_u2f(in[i])) : dm_f2u(dm_f16_to_f32((uint16_t)(in[i] & 0xffffu)));
}
}
