"""CPU-side checks of the boundary: libvrt_hip.so loads and exports every symbol include/vrt_api.h
declares, fails loudly without a GPU, and the host logic (camera, materials, scenes) is right."""
import ctypes as C
import numpy as np
import pytest

from voxel_rt2_amd import _abi, _lib, camera, host, materials, scenes


def test_library_exports_declared_symbols():
    lib = _lib.load()
    names = _lib.exported_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_struct_layouts_match_header():
    # sizes the C side compiles to (include/vrt_api.h); a drift here corrupts every call
    assert C.sizeof(_abi.VrtConfig) == 13 * 4
    assert C.sizeof(_abi.VrtSceneParams) == 4 * (1 + 3 + 1 + 3 + 3 + 1 + 3 + 1 + 1 + 1)
    assert C.sizeof(_abi.VrtCamera) == 4 * (64 + 3 + 1 + 1 + 1 + 1)
    assert C.sizeof(_abi.VrtStats) == 8 * 6 + 8 * 3 + 4 * 4


def test_no_cpu_fallback():
    """Without a usable MI355X the product refuses to create a context (and says why)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load()
    cfg = host.make_config(64, 64)
    assert lib.vrt_create(C.byref(cfg)) is None
    assert b"no HIP device" in lib.vrt_last_error() or b"gfx950" in lib.vrt_last_error()
    from voxel_rt2_amd._session import NativeError
    from voxel_rt2_amd.renderer import Renderer
    with pytest.raises(NativeError):
        Renderer(dx=1 / 64, image_res=(64, 64), up=(0, 1, 0), voxel_edges=0.06, exposure=3, sky_res=0)


def test_create_rejects_bad_config():
    lib = _lib.load()
    # grid_res is 128 or 256 and the voxel size follows from it (dx = 2 / grid_res: include/vrt_api.h)
    for field, value in (("grid_res", 64), ("grid_res", 256), ("dx", 1.0 / 128.0), ("width", 0), ("max_depth", 0), ("sky_res", 7)):
        cfg = host.make_config(64, 64)
        setattr(cfg, field, value)
        assert lib.vrt_create(C.byref(cfg)) is None
        assert lib.vrt_last_error()


def test_camera_matrices():
    proj = camera.perspective(np.deg2rad(50.0), 16 / 9, 0.01, 10.0)
    f = 1 / np.tan(np.deg2rad(25.0))
    assert np.isclose(proj[1, 1], f) and np.isclose(proj[0, 0], f / (16 / 9)) and proj[3, 2] == -1
    # a point on the near / far plane maps to NDC z = -1 / +1 (GL convention, space_transformations.py:17)
    for z, ndc in ((-0.01, -1.0), (-10.0, 1.0)):
        clip = proj @ np.array([0, 0, z, 1.0])
        assert np.isclose(clip[2] / clip[3], ndc)
    view = camera.look_at((0.4, 0.5, 2.0), (0, 0, 0), (0, 1, 0))
    eye = view @ np.array([0.4, 0.5, 2.0, 1.0])
    assert np.allclose(eye[:3], 0, atol=1e-12)
    origin = view @ np.array([0, 0, 0, 1.0])
    assert np.allclose(origin[:2], 0, atol=1e-12) and origin[2] < 0  # camera looks down -z
    assert np.allclose(view[:3, :3] @ view[:3, :3].T, np.eye(3), atol=1e-12)
    g = camera.to_glm_memory(view)
    assert np.array_equal(camera.from_glm_memory(g), view.astype(np.float32))
    inv = camera.inverse_f32(view.astype(np.float32))
    assert np.allclose(inv.astype(np.float64) @ view, np.eye(4), atol=1e-6)


def test_material_table():
    t = materials.load_table()
    assert t.shape == (128, 14) and t.dtype == np.float32
    np.testing.assert_array_equal(t[0], materials.default_row())
    np.testing.assert_array_equal(t[2], materials.default_row())          # "emissive" is a convention, not a row
    np.testing.assert_allclose(t[52, [4, 5, 7]], [1.0, 0.8, 1.0])          # mirror: metallic 1, specular .8, roughness 1.0
    np.testing.assert_allclose(t[54, [4, 11, 12]], [0.7, 0.7, 0.9])        # car paint
    np.testing.assert_allclose(t[82, [3, 9, 10]], [0.95, 0.9, 0.4])        # cloth
    assert (t[:, 0:3] == 1).all()


def test_scenes_are_deterministic_and_shaped():
    for name, fn in scenes.SCENES.items():
        if name in ("sponge256", "dense256"):
            continue  # 256^3 builders take seconds each: tests/test_grid256.py renders them
        m1, c1, p1 = fn(0) if name != "dense" else fn(12345)
        m2, c2, _ = fn(0) if name != "dense" else fn(12345)
        g = 256 if name.endswith("256") else 128
        assert m1.shape == (g, g, g) and m1.dtype == np.int8 and c1.shape == (g, g, g, 3) and c1.dtype == np.uint8
        assert np.array_equal(m1, m2) and np.array_equal(c1, c2)
        assert (m1 > 0).any() and not c1[m1 == 0].any()
    m, _, _ = scenes.scene_dense(12345)
    assert abs((m > 0).mean() - 0.5) < 0.005
    m, c, _ = scenes.scene_s1(0)
    assert 3000 < (m > 0).sum() < 4200 and (m[64:114, 64, 64:114] > 0).all()  # the 50x50 slab
    assert (m[64, 64, 64:114] == 2).all() and tuple(c[64 + 5, 64, 64 + 5]) == (229, 25, 25)


def test_scene_param_struct():
    s = host.make_scene_params(light_direction=(1, 1, -1), light_cone=0.025, light_color=(1.3, 1.2, 1.1))
    d = np.array(s.light_direction[:])
    assert abs(np.linalg.norm(d) - 1) < 1e-6 and d[2] < 0
    assert np.isclose(s.light_cos_theta_max, np.cos(0.0125))
    assert s.light_weight == 3.0
