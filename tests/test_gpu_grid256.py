"""GPU parity at 256^3 (BASELINE config 5): libvrt_hip.so through the C ABI against the CPU oracle, bit for bit.
The oracle's semantics at this size are the reference's own parametric code (tests/test_grid256.py pins them)."""
import functools

import numpy as np
import pytest

import orc
from voxel_rt2_amd import _abi, _lib, host, scenes
from voxel_rt2_amd._session import NativeSession

pytestmark = pytest.mark.gpu
BUFS = (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_POSITION, _abi.BUF_GBUF_MAT, _abi.BUF_GBUF_REFL_DEPTH,
        _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR)


@functools.lru_cache(maxsize=None)
def scene(name):
    return scenes.SCENES[name](0)


@pytest.fixture(params=["pool", "fused"])
def render_schedule(request, monkeypatch):
    monkeypatch.setenv("VRT_RENDER", request.param)
    return request.param


def config(name, W, H, depth, seed, **kw):
    _, _, params = scene(name)
    return host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=seed,
                            grid_res=256, **kw)


def pair(name, cfg, threads=None):
    mat, rgb, params = scene(name)
    g, o = NativeSession(_lib.load(), "vrt_", cfg), orc.Oracle(cfg, threads=threads)
    for s in (g, o):
        orc.setup(s, mat, rgb, params)
    return g, o


def assert_same(g, o, rows=None, bufs=BUFS):
    sl = slice(*rows) if rows else slice(None)
    a, b = g.fetch_hdr()[sl], o.fetch_hdr()[sl]
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} HDR values differ"
    for which in bufs:
        assert np.array_equal(g.fetch_buffer(which)[sl].view(np.uint8), o.fetch_buffer(which)[sl].view(np.uint8)), f"buffer {which}"


@pytest.mark.parametrize("name,W,H,depth,spp", [("dense256", 160, 96, 8, 2), ("sponge256", 192, 112, 6, 3), ("s1_256", 256, 144, 5, 4),
                                                 ("sponge256", 100, 60, 3, 1)])
def test_hdr_matches_oracle_256(render_schedule, name, W, H, depth, spp):
    g, o = pair(name, config(name, W, H, depth, seed=11))
    for s in (g, o):
        s.accumulate(spp)
    assert_same(g, o)
    assert np.array_equal(g.fetch_ldr().view(np.uint32), o.fetch_ldr().view(np.uint32))


def test_traversal_counters_match_oracle_256(render_schedule):
    """Every ray, DDA step and occupancy query of the reference's eight-LOD walk, counted on both sides."""
    g, o = pair("sponge256", config("sponge256", 192, 128, 8, seed=5))
    assert _lib.load().vrt_set_instrumented(g._ctx, 1) == 0
    for s in (g, o):
        s.accumulate(2)
    sg, so = g.stats(), o.stats()
    for k in ("rays", "dda_iters", "occupancy_queries", "closest_hits"):
        assert sg[k] == so[k], (k, sg[k], so[k])
    assert_same(g, o, bufs=())


def test_restir_256():
    g, o = pair("sponge256", config("sponge256", 128, 80, 5, seed=7, use_restir=True))
    for s in (g, o):
        s.accumulate(2)
    assert_same(g, o, bufs=())


def test_config5_full_size_shard_matches_oracle():
    """Config 5 at its real frame -- 3840x2160, dense 256^3 fill, 8 bounces, one fused call of 4 samples -- on the 16-row
    shard that holds the last row and column (u = 3839, v = 2159: the pooled kernel packs u, v in 12 bits), against the
    oracle; then the whole frame, whose rows must equal the shard's, twice (determinism across schedules of 2048 waves)."""
    W, H, rows = 3840, 2160, (2144, 2160)
    g, o = pair("dense256", config("dense256", W, H, 8, seed=0, rows=rows), threads=16)
    for s in (g, o):
        s.accumulate(4)
    assert_same(g, o, rows=rows, bufs=(_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_MAT, _abi.BUF_HISTORY_DIFFUSE))
    shard = g.fetch_hdr()[rows[0]:rows[1]].copy()
    g.close(); o.close()
    mat, rgb, params = scene("dense256")
    frames = []
    for _ in range(2):
        f = NativeSession(_lib.load(), "vrt_", config("dense256", W, H, 8, seed=0))
        orc.setup(f, mat, rgb, params)
        f.accumulate(4)
        frames.append(f.fetch_hdr())
        f.close()
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    assert np.array_equal(frames[0][rows[0]:rows[1]].view(np.uint32), shard.view(np.uint32))
    assert np.isfinite(frames[0]).all() and frames[0].mean() > 0.01


@pytest.mark.parametrize("grid", [128, 256])
@pytest.mark.parametrize("fill", ["empty", "full", "one_voxel"])
def test_degenerate_grids(render_schedule, grid, fill):
    """No solid voxel at all (the culling box is empty: every ray is a miss without a walk), every voxel solid (nothing to cull,
    every ray stops in its first cell), a single voxel in a corner (a tiny box far from the camera axis) -- each over a lit floor,
    at both grid sizes, against the oracle."""
    mat, rgb = scenes.empty(grid)
    if fill == "full":
        mat[...] = 1
        rgb[...] = (180, 140, 90)
    elif fill == "one_voxel":
        mat[grid - 1, grid // 2, 0] = 11
        rgb[grid - 1, grid // 2, 0] = (255, 64, 32)
    params = dict(exposure=1.0, voxel_edges=0.06, floor_height=-0.3, floor_color=(0.7, 0.6, 0.5), floor_material=1,
                  background_color=(0.2, 0.3, 0.5), light_direction=(0.3, 1.0, 0.2), light_cone=0.1, light_color=(1.0, 0.9, 0.8),
                  use_physical_sky=0, use_clouds=0)
    cfg = host.make_config(136, 84, voxel_edges=0.06, exposure=1.0, max_depth=5, seed=17, grid_res=grid)
    g, o = NativeSession(_lib.load(), "vrt_", cfg), orc.Oracle(cfg)
    for s in (g, o):
        orc.setup(s, mat, rgb, params)
        s.accumulate(3)
    assert_same(g, o)
    assert np.isfinite(g.fetch_hdr()).all() and g.fetch_hdr().mean() > 0.01
