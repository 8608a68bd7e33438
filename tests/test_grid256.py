"""BASELINE config 5: the 256^3 grid.  The reference fixes voxel_grid_res = 128 at one call site (pathtracer.py:83) but
VoxelWorld / VoxelOctreeRaytracer take it as a parameter (n_lods = log2(res) = 8, raytracer.py:9; lod cap n_lods - 1, :147;
offset -res//2, voxel_world.py:14), so the oracle defines the semantics and the product (one more brick level, brick-tiled
texels, eight-wave workgroups: vrt_types.h, vrt_kernels.hip) has to reproduce it bit for bit.  CPU only: oracle KATs and
properties, then the device headers compiled for the host against the oracle."""
import functools

import numpy as np
import pytest

import emu
import orc
from voxel_rt2_amd import _abi, host, scenes

INF = np.float32(np.inf)
BUFS = (_abi.BUF_GBUF_DEPTH, _abi.BUF_GBUF_NORMAL, _abi.BUF_GBUF_POSITION, _abi.BUF_GBUF_MAT, _abi.BUF_GBUF_REFL_DEPTH,
        _abi.BUF_HISTORY_DIFFUSE, _abi.BUF_HISTORY_SPECULAR)


@functools.lru_cache(maxsize=None)
def scene(name):
    return scenes.SCENES[name](0)


def oracle256(mat, rgb, W=64, H=64, **params):
    p = dict(exposure=1.0, voxel_edges=0.06, floor_height=-100.0, floor_color=(1, 1, 1), floor_material=1, background_color=(0, 0, 0),
             light_direction=(1, 1, 1), light_cone=0.1, light_color=(0, 0, 0))
    p.update(params)
    cfg = host.make_config(W, H, voxel_edges=p["voxel_edges"], exposure=p["exposure"], max_depth=4, seed=0, grid_res=256)
    assert cfg.dx == np.float32(1.0 / 128.0)
    o = orc.Oracle(cfg, threads=2)
    orc.setup(o, mat, rgb, p)
    return o


# ---- oracle: raytracer.py at res = 256 ----------------------------------------------------------------
def test_pyramid_has_eight_lods_and_matches_bruteforce():
    rng = np.random.default_rng(5)
    mat, rgb = scenes.empty(256)
    pts = rng.integers(0, 256, size=(400, 3))
    mat[pts[:, 0], pts[:, 1], pts[:, 2]] = 1
    mat[7, 7, 7] = -5  # signed byte: not solid (raytracer.py:50)
    o = oracle256(mat, rgb)
    solid = mat > 0
    for lod in range(8):  # n_lods = int(log2(256)) = 8 (raytracer.py:9)
        s, r = 1 << lod, 256 >> lod
        blocks = solid.reshape(r, s, r, s, r, s).any(axis=(1, 3, 5))
        for x, y, z in rng.integers(0, r, size=(200, 3)):
            assert o.query_occupancy(x, y, z, lod) == bool(blocks[x, y, z])
        for x, y, z in (pts[:40] >> lod):
            assert o.query_occupancy(x, y, z, lod)
    assert not o.query_occupancy(7, 7, 7, 0)


def test_raytrace_axis_ray_known_answer_256():
    """Ray along +x from (-10, 10.5, 10.5) at the single solid voxel (200,10,10), raytracer.py:72-155 with res = 256.
    By hand: the box is entered at t = 10; empty-space steps double the cell (lod = min(7, lod + 1), :147):
    t = 11, 12, 14, 18, 26, 42, 74, 138 (the last one a whole LOD-7 cell, 128 wide); the LOD-7 cell [128,256) is occupied,
    its LOD-6 child [128,192) is not: t = 202; then [192,256) descends 7 -> 6 -> 5 -> 4 to the empty LOD-3 cell [192,200):
    t = 210, and from x = 200 the descent reaches the solid voxel: 10 steps, distance 210, normal -x."""
    mat, rgb = scenes.empty(256)
    mat[200, 10, 10] = 1
    o = oracle256(mat, rgb)
    h = o.raytrace((-10.0, 10.5, 10.5), (1.0, 0.0, 0.0))
    assert h["distance"] == np.float32(210.0)
    assert tuple(h["cell"]) == (200, 10, 10)
    assert tuple(h["normal"]) == (-1.0, 0.0, 0.0)
    assert h["iters"] == 10
    m = o.raytrace((-10.0, 12.5, 10.5), (1.0, 0.0, 0.0))
    assert m["distance"] == INF
    # boundary-start normal uses res / 2 = 128 (raytracer.py:99)
    mat[250, 128, 128] = 1
    o2 = oracle256(mat, rgb)
    s = o2.raytrace((250.5, 128.2, 128.3), (-1.0, 0.0, 0.0))
    assert s["iters"] == 0 and tuple(s["normal"]) == (1.0, 0.0, 0.0) and tuple(s["cell"]) == (250, 128, 128)


def test_world_scale_of_the_256_grid():
    """world_to_voxel = pos / dx + res / 2 (pathtracer.py:165-167) with dx = 1/128: the grid spans [-1,1]^3, and a hit
    distance comes back in world units (:205-207)."""
    mat, rgb = scenes.empty(256)
    mat[128 + 20, 128 + 5, 128 - 7] = 11
    rgb[128 + 20, 128 + 5, 128 - 7] = (255, 128, 0)
    o = oracle256(mat, rgb, voxel_edges=0.0)
    centre = (np.array([20.5, 5.5, -6.5]) / 128.0).astype(np.float32)
    h = o.next_hit(centre + np.float32([0.0, 0.0, 1.0]), (0.0, 0.0, -1.0))
    assert abs(h["closest"] - (1.0 - 0.5 / 128.0)) < 1e-6
    assert h["mat_id"] == 11 and tuple(h["normal"]) == (0.0, 0.0, 1.0)
    assert np.allclose(h["albedo"], (1.0, 128 / 255.0, 0.0))


def test_same_scene_on_both_grids_has_the_same_primary_surface():
    """Scene S1 and its 2x refinement describe the same solid in world space: the primary g-buffer (depth, material) of the
    256 render equals that of the 128 render except where float rounding moves a ray across a voxel boundary."""
    W, H = 160, 96
    m1, c1, p = scenes.scene_s1(0)
    m2, c2 = scenes.upsample2(m1, c1)
    p = dict(p, voxel_edges=0.0)
    out = []
    for mat, rgb, g in ((m1, c1, 128), (m2, c2, 256)):
        cfg = host.make_config(W, H, voxel_edges=0.0, exposure=p["exposure"], max_depth=2, seed=1, grid_res=g)
        o = orc.Oracle(cfg, threads=4)
        orc.setup(o, mat, rgb, p)
        o.accumulate(1)
        out.append((o.fetch_buffer(_abi.BUF_GBUF_DEPTH)[..., 0], o.fetch_buffer(_abi.BUF_GBUF_MAT)[..., 0]))
    (d1, id1), (d2, id2) = out
    same = id1 == id2
    assert same.mean() > 0.995
    assert np.abs(d1[same] - d2[same]).max() < 2e-4


# ---- product device code (host build) == oracle at 256^3 ----------------------------------------------
@pytest.fixture(params=["fused", "pool", "pool+cull"])
def render_schedule(request, monkeypatch):
    if request.param.startswith("pool"):
        monkeypatch.setenv("VRT_EMU_POOL", "1")
    else:
        monkeypatch.delenv("VRT_EMU_POOL", raising=False)
    if request.param.endswith("+cull"):   # rays that cannot hit a voxel are not walked (cull_ray, vrt_trace.h): images only
        monkeypatch.setenv("VRT_EMU_CULL", "1")
    else:
        monkeypatch.delenv("VRT_EMU_CULL", raising=False)
    return request.param


def assert_same(o, e, stats):
    a, b = o.fetch_hdr(), e.fetch_hdr()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{(a != b).sum()} HDR values differ"
    for which in BUFS:
        assert np.array_equal(o.fetch_buffer(which).view(np.uint8), e.fetch_buffer(which).view(np.uint8)), f"buffer {which}"
    if stats:
        so, se = o.stats(), e.stats()
        for k in ("rays", "dda_iters", "occupancy_queries", "closest_hits"):
            assert so[k] == se[k], k
    assert np.array_equal(o.fetch_ldr().view(np.uint32), e.fetch_ldr().view(np.uint32))


@pytest.mark.parametrize("name,W,H,depth,spp", [("s1_256", 112, 72, 5, 2), ("sponge256", 96, 64, 6, 2), ("dense256", 72, 48, 8, 2)])
def test_emulated_device_code_matches_oracle_256(render_schedule, name, W, H, depth, spp):
    mat, rgb, params = scene(name)
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=depth, seed=3, grid_res=256)
    o, e = orc.Oracle(cfg, threads=4), emu.Emulated(cfg)
    for s in (o, e):
        orc.setup(s, mat, rgb, params)
        s.accumulate(spp)
    sun_on = any(c != 0 for c in params["light_color"])  # black sun: the product skips unobservable shadow rays
    assert_same(o, e, stats=sun_on and not render_schedule.endswith("+cull"))
    assert e.stats()["occupancy_queries"] > 0


def test_emulated_restir_256():
    mat, rgb, params = scene("sponge256")
    cfg = host.make_config(80, 56, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=4, seed=7, use_restir=True,
                           grid_res=256)
    o, e = orc.Oracle(cfg, threads=4), emu.Emulated(cfg)
    for s in (o, e):
        orc.setup(s, mat, rgb, params)
        s.accumulate(2)
    assert_same(o, e, stats=False)


def test_emulated_row_shard_256():
    mat, rgb, params = scene("dense256")
    full = host.make_config(64, 48, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=4, seed=2, grid_res=256)
    part = host.make_config(64, 48, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=4, seed=2, grid_res=256,
                            rows=(16, 32))
    a, b = emu.Emulated(full), emu.Emulated(part)
    for s in (a, b):
        orc.setup(s, mat, rgb, params)
        s.accumulate(2)
    assert np.array_equal(a.fetch_hdr()[16:32].view(np.uint32), b.fetch_hdr()[16:32].view(np.uint32))
