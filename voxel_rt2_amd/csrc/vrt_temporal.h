// vrt_temporal.h -- temporal accumulation, one fused pass per pixel.
//
// Replaces Renderer.temporal_filter_prepass, temporal_filter, temporal_filter_specular and the
// trailing copy loop (reference renderer/pathtracer.py:1020-1075, 1077-1130, 1132-1183,
// 1185-1230, 1242-1303).  The reference launches three kernels plus a copy loop and moves about
// 440 B per pixel per sample through HBM; fused, a static-camera sample reads the two colour
// values (+ the bilinear neighbours, which hit L2), two 16-byte histories and writes the two
// histories and the 12-byte HDR pixel.  The copy loop is replaced by swapping the in/out history
// and current/previous g-buffer pointers on the host.
//
// Ordering semantics kept from the reference's kernel sequence: the NaN/Inf/negative scrub of the
// prepass is applied to every colour tap as it is loaded (the prepass finishes before the filters
// start), neighbour reads see the values the render stage wrote (snapshot), a pixel whose
// reconstructed world position is ~0 is skipped by both filters and keeps its history.
// Dead outputs of the reference are not produced (specular_mean/stdev, specular depth history).
#ifndef VRT_TEMPORAL_H
#define VRT_TEMPORAL_H

#include "vrt_path.h"

namespace vrt {

struct TemporalBuffers {
    const f3* color_d;
    const f3* color_s;
    const uint32_t* gb_normal;
    const float* gb_depth;
    const uint32_t* gb_mat;
    const float* gb_refl_raw;
    float* gb_refl_filtered;      // gbuff_depth_reflection after the prepass
    const f4* hist_d_in;
    f4* hist_d_out;
    const f4* hist_s_in;
    f4* hist_s_out;
    const uint32_t* prev_normal;  // gbuff_prev_normals / gbuff_prev_depth
    const float* prev_depth;
    f3* hdr;                      // color_buffer after accumulate(); becomes the next pass's render target
    int sample_stride;            // elements between the planes of consecutive fused samples (colours, raw reflection depth)
    mat4 prev_view, prev_proj;    // prev_view_mat / prev_proj_mat (pathtracer.py:103-104, 283-287)
    f3* tile;                     // or null: the caller's copy of the HDR rows [tile_row0, ...) (vrt_set_hdr_targets: the tile a
    int tile_row0;                // multi-GPU rank hands to the gather), written with the frame instead of copied afterwards
};
// STRIPED: a context with row stripes (vrt_set_row_stripes; a kernel of its own, so that the usual one carries none of this)
template <bool STRIPED>
VRT_DEV void store_hdr(const FrameParams& fp, const TemporalBuffers& tb, int idx, int u, int v, f3 c) {
    tb.hdr[idx] = c;
    if (tb.tile) {   // the context's rows, one after the other (stripe after stripe on a striped context)
        int row = v - tb.tile_row0;
        if (STRIPED) { const int d = v - fp.stripe_first; row = (d / fp.stripe_period) * fp.stripe_rows + d % fp.stripe_period; }
        tb.tile[row * fp.W + u] = c;
    }
}

VRT_DEV f3 scrub(f3 c) {  // pathtracer.py:1069-1075
#if defined(__HIP_DEVICE_COMPILE__)
    // "NaN, infinite or negative" is one class test per component (v_cmp_class_f32: either NaN, either infinity, negative
    // normal or subnormal; -0 is not < 0), without the branches the short-circuit form compiles to -- this runs on 32
    // taps per pixel of the temporal pass
    const int bad_classes = 0x001 | 0x002 | 0x004 | 0x008 | 0x010 | 0x200;  // sNaN qNaN -inf -normal -subnormal +inf
    const bool bad = (bool)((int)__builtin_isfpclass(c.x, bad_classes) | (int)__builtin_isfpclass(c.y, bad_classes) |
                            (int)__builtin_isfpclass(c.z, bad_classes));
#else
    bool bad = dm_isnan(c.x) || dm_isinf(c.x) || c.x < 0.0f || dm_isnan(c.y) || dm_isinf(c.y) || c.y < 0.0f ||
               dm_isnan(c.z) || dm_isinf(c.z) || c.z < 0.0f;
#endif
    return bad ? mk3(0.0f) : c;
}
VRT_DEV void render_res(const FrameParams& fp, int& rx, int& ry) {
    rx = dm_f2i((float)fp.W * fp.render_scale);
    ry = dm_f2i((float)fp.H * fp.render_scale);
}
// local index of global pixel (x, y), clamped to the image (bilinear taps past the edge)
VRT_DEV int clamped_index(const FrameParams& fp, int x, int y) {
    x = x < 0 ? 0 : (x > fp.W - 1 ? fp.W - 1 : x);
    y = y < 0 ? 0 : (y > fp.H - 1 ? fp.H - 1 : y);
    return (y - fp.row0) * fp.W + x;
}
// pathtracer.py:1077-1090 on a scrubbed colour buffer
VRT_DEV f3 bilinear_color(const FrameParams& fp, const f3* buf, f2 uv) {
    int rx, ry;
    render_res(fp, rx, ry);
    float fcx = uv.x * (float)rx - 0.5f, fcy = uv.y * (float)ry - 0.5f;
    int ix = dm_f2i(fcx), iy = dm_f2i(fcy);
    float fx = frac1(fcx), fy = frac1(fcy);
    f3 bl = scrub(buf[clamped_index(fp, ix, iy)]), br = scrub(buf[clamped_index(fp, ix + 1, iy)]);
    f3 tl = scrub(buf[clamped_index(fp, ix, iy + 1)]), tr = scrub(buf[clamped_index(fp, ix + 1, iy + 1)]);
    return lerp3(lerp3(bl, br, fx), lerp3(tl, tr, fx), fy);
}
// the same fetch from two buffers at one texcoord (a sample's diffuse and specular planes): the eight taps fetched together
VRT_DEV void bilinear_color2(const FrameParams& fp, const f3* buf_a, const f3* buf_b, f2 uv, f3& out_a, f3& out_b) {
    int rx, ry;
    render_res(fp, rx, ry);
    float fcx = uv.x * (float)rx - 0.5f, fcy = uv.y * (float)ry - 0.5f;
    int ix = dm_f2i(fcx), iy = dm_f2i(fcy);
    float fx = frac1(fcx), fy = frac1(fcy);
    const int i0 = clamped_index(fp, ix, iy), i1 = clamped_index(fp, ix + 1, iy), i2 = clamped_index(fp, ix, iy + 1), i3 = clamped_index(fp, ix + 1, iy + 1);
    const f3 a0 = buf_a[i0], a1 = buf_a[i1], a2 = buf_a[i2], a3 = buf_a[i3];
    const f3 b0 = buf_b[i0], b1 = buf_b[i1], b2 = buf_b[i2], b3 = buf_b[i3];
    out_a = lerp3(lerp3(scrub(a0), scrub(a1), fx), lerp3(scrub(a2), scrub(a3), fx), fy);
    out_b = lerp3(lerp3(scrub(b0), scrub(b1), fx), lerp3(scrub(b2), scrub(b3), fx), fy);
}
VRT_DEV f3 reproject(const FrameParams& fp, const TemporalBuffers& tb, f3 wp) {  // pathtracer.py:993-1000
    f4 p = mul4(tb.prev_proj, mul4(tb.prev_view, mk4(wp.x, wp.y, wp.z, 1.0f)));
    return mk3(p.x / p.w, p.y / p.w, p.z / p.w) * 0.5f + 0.5f;
}
VRT_DEV float catmullrom(float x) {  // pathtracer.py:1002-1014
    float x2 = x * x, x3 = x * x * x, fx = 0.0f;
    if (x < 1.0f) fx = 1.5f * x3 - 2.5f * x2 + 1.0f;
    else if (x < 2.0f) fx = -0.5f * x3 + 2.5f * x2 - 4.0f * x + 2.0f;
    return fx;
}
// pathtracer.py:1092-1130 (DEPTH_TEST) and 1132-1183 (without): 4x4 Catmull-Rom history resample
template <bool DEPTH_TEST>
VRT_DEV float history_resample(const FrameParams& fp, const TemporalBuffers& tb, const f4* hist, f2 uv, float center_depth,
                               f3 center_normal, f4& out) {
    int rx, ry;
    render_res(fp, rx, ry);
    float fcx = uv.x * (float)rx - 0.5f, fcy = uv.y * (float)ry - 0.5f;
    int icx = dm_f2i(fcx), icy = dm_f2i(fcy);
    float fx = frac1(fcx), fy = frac1(fcy);
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mx[4] = {0.0f, 0.0f, 0.0f, 0.0f}, mn[4] = {999999.0f, 999999.0f, 999999.0f, 999999.0f};
    float wsum = 0.0f;
    for (int x = -1; x < 3; x++)
        for (int y = -1; y < 3; y++) {
            int tx = icx + x, ty = icy + y;
            if (tx < 0 || ty < 0 || tx > rx - 1 || ty > ry - 1) continue;
            int t = (ty - fp.row0) * fp.W + tx;
            float w = catmullrom(dm_abs((float)x - fx)) * catmullrom(dm_abs((float)y - fy));
            f3 tn = oct_decode(tb.prev_normal[t]);
            if (fp.camera_is_moving == 1) {
                if (DEPTH_TEST) {
                    float td = linearize_depth(tb.prev_depth[t], fp.proj_inv);
                    w *= (dm_abs(td - center_depth) / center_depth < 0.05f) ? 1.0f : 0.0f;
                }
                w *= (dot3(center_normal, tn) > 0.642f) ? 1.0f : 0.0f;
            }
            f4 c = hist[t];
            float cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                mx[k] = dm_max(mx[k], cv[k]);
                mn[k] = dm_min(mn[k], cv[k]);
                sum[k] += cv[k] * w;
            }
            wsum += w;
        }
    out.x = dm_max(dm_clamp(sum[0] / wsum, mn[0], mx[0]), 0.0f);
    out.y = dm_max(dm_clamp(sum[1] / wsum, mn[1], mx[1]), 0.0f);
    out.z = dm_max(dm_clamp(sum[2] / wsum, mn[2], mx[2]), 0.0f);
    out.w = dm_max(dm_clamp(sum[3] / wsum, mn[3], mx[3]), 1.0f);
    return wsum;
}

VRT_DEV void blend_history(const FrameParams& fp, float wsum, f4& h, f3 cur) {  // :1216-1220, 1283-1289
    if (wsum > 1e-3f) {
        h.w = dm_min(h.w + 1.0f, fp.max_accum_frames);
        f3 m = lerp3(mk3(h.x, h.y, h.z), cur, 1.0f / h.w);
        h.x = m.x; h.y = m.y; h.z = m.z;
    } else {
        h = mk4(cur.x, cur.y, cur.z, 1.0f);
    }
}

// One pixel (u, v) of this shard's rows.  n_samples > 1: the samples of one accumulate(n) call, rendered by one
// fused k_render launch into consecutive planes (static camera only): the running means are advanced n times in
// registers, in sample order, exactly as n separate passes would, and the histories / HDR are written once.
template <bool STRIPED = false>
VRT_DEV void temporal_pixel(const FrameParams& fp, const TemporalBuffers& tb, int u, int v, int n_samples) {
    const int idx = (v - fp.row0) * fp.W + u;
    if (outside_render_area(fp, (float)u, (float)v)) {
        // not rendered at this render_scale: the reference leaves color_buffer (= the last HDR value) untouched.
        // The render target and the HDR target swap roles every pass, so carry the value across.
        store_hdr<STRIPED>(fp, tb, idx, u, v, tb.color_d[idx]);
        tb.hist_d_out[idx] = tb.hist_d_in[idx];
        tb.hist_s_out[idx] = tb.hist_s_in[idx];
        return;
    }
    int rx, ry;
    render_res(fp, rx, ry);
    const int last = (n_samples - 1) * tb.sample_stride;  // plane of the last sample

    // prepass: reflection-depth average over the valid taps of a 4x4 window (:1040-1066)
    // (the 16 taps are fetched before any is looked at -- a tap outside the window fetches the pixel's own word, which is not
    // used: fetched one by one, each behind the test of the one before, the loop was 16 memory round trips in a row, and beside a
    // render launch this pass runs at one wave per SIMD with nothing to hide them behind; the sums run in the reference's order)
    float rsum = 0.0f, rcount = 0.0f;
    {
        float rd[16];
        bool in[16];
#pragma unroll
        for (int t = 0; t < 16; t++) {
            const int tx = u + (t >> 2) - 1, ty = v + (t & 3) - 1;
            in[t] = !(tx < 0 || ty < 0 || tx > rx - 1 || ty > ry - 1) && !(ty < fp.row0 || ty >= fp.row1);  // (beyond the shard's halo: reflection depth is unused there)
            rd[t] = tb.gb_refl_raw[last + (in[t] ? (ty - fp.row0) * fp.W + tx : idx)];
        }
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (in[t] && rd[t] != 0.0f) { rcount += 1.0f; rsum += rd[t]; }
    }
    const float refl_depth = (rcount > 0.01f) ? rsum / rcount : 0.0f;
    tb.gb_refl_filtered[idx] = refl_depth;

    const f2 tc = pixel_texcoord(fp, (float)u, (float)v);
    const float nl_depth = tb.gb_depth[idx];
    const f3 x1 = xform(fp.view_inv, screen_to_view(tc, nl_depth, fp.proj_inv), 1.0f);
    if (near_zero3(x1)) {  // both filters `continue`: colour stays the scrubbed diffuse sample, histories persist
        store_hdr<STRIPED>(fp, tb, idx, u, v, scrub(tb.color_d[last + idx]));
        tb.hist_d_out[idx] = tb.hist_d_in[idx];
        tb.hist_s_out[idx] = tb.hist_s_in[idx];
        return;
    }
    f4 hd, hs;
    if (fp.camera_is_moving == 0) {
        hd = tb.hist_d_in[idx];
        hs = tb.hist_s_in[idx];
        for (int k = 0; k < n_samples; k++) {
            f3 cur_d, cur_s;
            bilinear_color2(fp, tb.color_d + k * tb.sample_stride, tb.color_s + k * tb.sample_stride, tc, cur_d, cur_s);
            blend_history(fp, 1.0f, hd, cur_d);
            blend_history(fp, 1.0f, hs, cur_s);
        }
    } else {
        const f3 cur_d = bilinear_color(fp, tb.color_d, tc);
        const f3 cur_s = bilinear_color(fp, tb.color_s, tc);
        const f3 n1 = oct_decode(tb.gb_normal[idx]);
        f3 rp = reproject(fp, tb, x1);
        float wd = history_resample<true>(fp, tb, tb.hist_d_in, mk2(rp.x, rp.y), linearize_depth(rp.z, fp.proj_inv), n1, hd);
        float nl = delinearize_depth(refl_depth, fp.proj);
        f3 refl_pos = xform(fp.view_inv, screen_to_view(tc, nl, fp.proj_inv), 1.0f);
        f3 rps = reproject(fp, tb, (refl_depth != 0.0f) ? refl_pos : x1);
        float ws = history_resample<false>(fp, tb, tb.hist_s_in, mk2(rps.x, rps.y), linearize_depth(rps.z, fp.proj_inv), n1, hs);
        blend_history(fp, wd, hd, cur_d);
        blend_history(fp, ws, hs, cur_s);
    }
    tb.hist_d_out[idx] = hd;
    tb.hist_s_out[idx] = hs;
    f3 col = mk3(hd.x, hd.y, hd.z);
    if (fp.camera_is_moving == 1) col = col * unpack_albedo(tb.gb_mat[idx]);  // re-modulate albedo (:1227-1228)
    store_hdr<STRIPED>(fp, tb, idx, u, v, col + mk3(hs.x, hs.y, hs.z));
}

// Renderer._render_to_image (pathtracer.py:634-662) with uchimura (math_utils.py:163-186)
VRT_DEV float uchimura1(float x) {
    const double P = 1.0, a = 1.0, m = 0.22, l = 0.4, c = 1.33, b = 0.0;
    const double l0 = ((P - m) * l) / a, S0 = m + l0, S1 = m + a * l0, C2 = (a * P) / (P - S1), CP = -C2 / P;
    float t0 = dm_clamp((x - 0.0f) / ((float)m - 0.0f), 0.0f, 1.0f);
    float w0 = 1.0f - t0 * t0 * (3.0f - 2.0f * t0);
    float w2 = (x < (float)(m + l0)) ? 0.0f : 1.0f;
    float w1 = 1.0f - w0 - w2;
    float T = (float)m * dm_pow(x / (float)m, (float)c) + (float)b;
    float S = (float)P - (float)(P - S1) * dm_exp((float)CP * (x - (float)S0));
    float L = (float)m + (float)a * (x - (float)m);
    return T * w0 + L * w1 + S * w2;
}
VRT_DEV f4 tonemap_pixel(const FrameParams& fp, const f3* hdr, int i, int j) {
    float ux = (float)i / (float)fp.W, uy = (float)j / (float)fp.H;
    float dx = ux - 0.5f, dy = uy - 0.5f;
    float darken = 1.0f - 0.9f * dm_max(dm_sqrt(dx * dx + dy * dy) - 0.0f, 0.0f);
    int sx = dm_f2i((float)i * fp.render_scale), sy = dm_f2i((float)j * fp.render_scale);
    f3 c = hdr[(sy - fp.row0) * fp.W + sx] * darken * fp.exposure;
    const float g = (float)(1.0 / 2.2);
    return mk4(dm_saturate(dm_pow(uchimura1(c.x), g)), dm_saturate(dm_pow(uchimura1(c.y), g)),
               dm_saturate(dm_pow(uchimura1(c.z), g)), 1.0f);
}

}  // namespace vrt
#endif
