"""CPU-side parity: the product's device headers (voxel_rt2_amd/csrc/*.h) compiled for the host and
stepped pixel by pixel (tests/emul/) against the oracle.  No GPU needed; this is what catches an
arithmetic divergence between the two implementations before a GPU run.  Bit-exact everywhere,
including the traversal counters (same sequence of DDA states through the bit-brick pyramid as
through the reference's flat per-LOD bitmap)."""
import numpy as np
import pytest

import emu
import orc
from voxel_rt2_amd import _abi, host, scenes, camera

BUFS = (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_POSITION, _abi.BUF_GBUF_MAT, _abi.BUF_GBUF_REFL_DEPTH,
        _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR)


@pytest.fixture(autouse=True, params=["fused", "pool", "fused+cull", "pool+cull"])
def render_schedule(request, monkeypatch):
    """Both schedules of the render stage: path_segment per pixel (vrt_path.h) and the pooled kernel's stage
    functions stepped through their packed LDS slot, walks suspended and resumed every third step (vrt_pool.h) -- each
    with and without the culling of rays that cannot hit a voxel (cull_ray, vrt_trace.h: the shipped kernels cull; with
    it on, rays / steps / queries are no longer the reference's, so only the images are compared)."""
    if request.param.startswith("pool"):
        monkeypatch.setenv("VRT_EMU_POOL", "1")
    else:
        monkeypatch.delenv("VRT_EMU_POOL", raising=False)
    if request.param.endswith("+cull"):
        monkeypatch.setenv("VRT_EMU_CULL", "1")
    else:
        monkeypatch.delenv("VRT_EMU_CULL", raising=False)
    return request.param


def culling():
    import os
    return bool(os.environ.get("VRT_EMU_CULL"))


def pair(scene, W, H, depth, seed, restir=False, rows=None):
    mat, rgb, params = scenes.SCENES[scene](0)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=seed,
                           use_restir=restir, rows=rows)
    o, e = orc.Oracle(cfg, threads=4), emu.Emulated(cfg)
    for s in (o, e):
        orc.setup(s, mat, rgb, params)
    return o, e


def assert_same(o, e, stats=True):
    a, b = o.fetch_hdr(), e.fetch_hdr()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} HDR values differ"
    for which in BUFS:
        x, y = o.fetch_buffer(which), e.fetch_buffer(which)
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"buffer {which}"
    if stats and not culling():
        so, se = o.stats(), e.stats()
        for k in ("rays", "dda_iters", "occupancy_queries", "closest_hits", "sky_lookups"):
            assert so[k] == se[k], k
    assert np.array_equal(o.fetch_ldr().view(np.uint32), e.fetch_ldr().view(np.uint32))


@pytest.mark.parametrize("scene,W,H,depth,spp", [("s1", 128, 96, 4, 2), ("sunlit", 144, 88, 8, 3), ("dense", 96, 64, 8, 2),
                                                 ("s6", 96, 64, 6, 1), ("sunlit", 50, 30, 2, 1)])
def test_render_and_temporal(scene, W, H, depth, spp):
    mat, rgb, params = scenes.SCENES[scene](0)
    params = dict(params, use_physical_sky=0, use_clouds=0)  # sky tables are covered in test_sky_lookup
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=3)
    o, e = orc.Oracle(cfg, threads=4), emu.Emulated(cfg)
    for s in (o, e):
        orc.setup(s, mat, rgb, params)
        s.accumulate(spp)
    # with a black sun the product skips shadow rays whose visibility cannot reach the image (vrt_path.h);
    # the oracle traces them all, so only the images -- not the ray counts -- are equal there
    sun_on = any(c != 0 for c in params["light_color"])
    assert_same(o, e, stats=sun_on)
    if not sun_on:
        assert e.stats()["rays"] < o.stats()["rays"]
    if culling():
        assert e.stats()["dda_iters"] <= o.stats()["dda_iters"] and (scene == "dense" or e.stats()["dda_iters"] < o.stats()["dda_iters"])


def test_restir():
    o, e = pair("sunlit", 112, 72, 5, seed=7, restir=True)
    for s in (o, e):
        s.accumulate(2)
    # the product does not trace the unobservable visibility ray of escape-vertex samples (vrt_restir.h)
    assert_same(o, e, stats=False)
    assert e.stats()["rays"] <= o.stats()["rays"]


def test_moving_camera_sequence():
    W, H = 112, 72
    o, e = pair("sunlit", W, H, 4, seed=9)
    for s in (o, e):
        s.accumulate(2)
        s.end_frame()
        for k in range(3):
            pos = (0.4 + 0.05 * (k + 1), 0.5, 2.0)
            view, proj = camera.default_matrices(W, H, pos=pos)
            s.set_camera(host.make_camera(view, proj, pos, jitter_index=k + 1, moving=True, render_scale=0.5, max_accum_frames=50.0))
            if k == 0:
                s.reset()
            s.accumulate(1)
            s.end_frame()
    assert_same(o, e)


def test_row_shard():
    o, e = pair("sunlit", 96, 60, 5, seed=2, rows=(20, 41))
    for s in (o, e):
        s.accumulate(2)
    a, b = o.fetch_hdr(), e.fetch_hdr()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not a[:20].any() and not a[41:].any() and a[20:41].any()


def test_sky_lookup():
    """Render with physical sky against synthetic sky tables (the lookup path of atmos.py:94-131)."""
    mat, rgb, params = scenes.scene_s6(0)
    W, H, R = 96, 64, 48
    cfg = host.make_config(W, H, voxel_edges=0.0, exposure=2.0, max_depth=5, seed=5, sky_res=R)
    o, e = orc.Oracle(cfg, threads=4), emu.Emulated(cfg)
    for s in (o, e):
        orc.setup(s, mat, rgb, params, cloud=np.zeros((256, 256, 3), dtype=np.uint8))
    # run the oracle's precompute at tiny size, hand the same tables to the emulated device code
    o.sky_accumulate_clouds(1)
    for sl in range(4):
        o.sky_compute_slice(sl, 4)
    scat, trans = o.fetch_buffer(_abi.BUF_SKY_SCATTERING), o.fetch_buffer(_abi.BUF_SKY_TRANSMITTANCE)
    assert np.isfinite(scat).all() and scat.max() > 0
    e.upload_sky(scat, trans)
    for s in (o, e):
        s.accumulate(2)
    a, b = o.fetch_hdr(), e.fetch_hdr()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert o.stats()["sky_lookups"] == e.stats()["sky_lookups"] > 0


def test_spatial_reuse_bsdf_path_equals_render_bsdf_path():
    """bsdf_eval_pdf + surf_shared + the material-derived table (k_gris) against eval_lobes / pdf_lobe / pdf_all /
    surf_init (render kernels): same bits for random materials, directions and lobe codes."""
    import ctypes as C
    lib = emu.lib()
    lib.emu_bsdf_selftest.argtypes = [C.c_int, C.c_uint32]
    lib.emu_bsdf_selftest.restype = C.c_int
    assert lib.emu_bsdf_selftest(20000, 11) == 0


@pytest.mark.parametrize("fill", ["empty", "full", "one_voxel", "far_corner_blocks"])
def test_degenerate_grids(fill):
    """The culling box at its extremes: no solid voxel (an empty box: every ray a miss), all solid (the box is the grid),
    a single voxel / two small blocks in opposite corners (a box that is mostly air) -- over a lit floor."""
    mat, rgb = scenes.empty()
    if fill == "full":
        mat[...] = 1
        rgb[...] = (180, 140, 90)
    elif fill == "one_voxel":
        mat[127, 64, 0] = 11
        rgb[127, 64, 0] = (255, 64, 32)
    elif fill == "far_corner_blocks":
        mat[2:9, 60:70, 3:8] = 21
        rgb[2:9, 60:70, 3:8] = (40, 200, 90)
        mat[118:126, 66:72, 119:127] = 1
        rgb[118:126, 66:72, 119:127] = (220, 210, 60)
    params = dict(exposure=1.0, voxel_edges=0.06, floor_height=-0.3, floor_color=(0.7, 0.6, 0.5), floor_material=1,
                  background_color=(0.2, 0.3, 0.5), light_direction=(0.3, 1.0, 0.2), light_cone=0.1, light_color=(1.0, 0.9, 0.8),
                  use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(88, 56, voxel_edges=0.06, exposure=1.0, max_depth=5, seed=17)
    o, e = orc.Oracle(cfg, threads=4), emu.Emulated(cfg)
    for s in (o, e):
        orc.setup(s, mat, rgb, params)
        s.accumulate(2)
    assert_same(o, e)


def test_row_stripe_enumeration_covers_its_rows_once():
    """vrt_set_row_stripes: the render kernels' tile-row enumeration (launch_tile_row / launch_renders_row, vrt_types.h) visits exactly
    the part's stripes plus two rows either side, each row once; the parts' own stripes tile the frame."""
    import ctypes as C
    lib = emu.lib()
    for H, S, N in ((1080, 8, 8), (1080, 32, 8), (2160, 64, 8), (100, 8, 3), (136, 32, 2), (72, 16, 5), (8, 8, 1)):
        owned_all = np.zeros(H, int)
        for part in range(N):
            out = np.zeros(H, np.uint8)
            lib.emu_unit_stripe_rows(H, S, N, part, out.ctypes.data_as(C.c_void_p))
            own = np.zeros(H, bool)
            for a in range(part * S, H, S * N):
                own[a:a + S] = True
            need = own.copy()
            for d in (1, 2):
                need[:-d] |= own[d:]
                need[d:] |= own[:-d]
            assert np.array_equal(out > 0, need) and out.max() == 1, (H, S, N, part)
            owned_all += own
        assert (owned_all == 1).all()
