"""Host-side construction of the vrt_api.h argument structs from plain Python values."""
import math
import numpy as np

from . import _abi, camera as cam_mod


def make_config(width, height, *, voxel_edges=0.06, exposure=3.0, max_depth=4, use_restir=False, seed=0, sky_res=0,
                device=0, rows=None, dx=None, grid_res=128):
    cfg = _abi.VrtConfig()
    cfg.width, cfg.height, cfg.grid_res = int(width), int(height), int(grid_res)
    if dx is None:
        dx = 2.0 / int(grid_res)   # the grid spans [-1,1]^3: scene.py:11's 1/64 at 128
    cfg.dx, cfg.voxel_edges, cfg.exposure = float(dx), float(voxel_edges), float(exposure)
    cfg.max_depth, cfg.use_restir, cfg.seed = int(max_depth), int(bool(use_restir)), int(seed) & 0xFFFFFFFF
    cfg.sky_res, cfg.device = int(sky_res), int(device)
    cfg.row_begin, cfg.row_end = (int(rows[0]), int(rows[1])) if rows else (0, 0)
    return cfg


def normalize3(v):
    """ti.Vector(direction).normalized() as set_directional_light evaluates it (pathtracer.py:140): in PYTHON scope, where Taichi's
    vectors hold Python numbers and compute in double; the result is rounded to float32 once, by the store into the f32 field."""
    v = [float(x) for x in v]
    n = (v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) ** 0.5
    inv = 1.0 / n
    return np.array([inv * x for x in v], dtype=np.float32)


def make_scene_params(*, floor_height=0.0, floor_color=(1.0, 1.0, 1.0), floor_material=1, background_color=(0.0, 0.0, 0.0),
                      light_direction=(1.0, 1.0, 1.0), light_cone=0.1, light_color=(0.0, 0.0, 0.0), light_weight=3.0,
                      use_physical_sky=0, use_clouds=0, **_ignored):
    s = _abi.VrtSceneParams()
    s.floor_height = float(floor_height)
    s.floor_color[:] = [float(x) for x in floor_color]
    s.floor_material = int(floor_material)
    s.background_color[:] = [float(x) for x in background_color]
    s.light_direction[:] = [float(x) for x in normalize3(light_direction)]
    s.light_cos_theta_max = math.cos(float(light_cone) * 0.5)  # pathtracer.py:142
    s.light_color[:] = [float(x) for x in light_color]
    s.light_weight = float(light_weight)
    s.use_physical_sky, s.use_clouds = int(bool(use_physical_sky)), int(bool(use_clouds))
    return s


def make_camera(view, proj, pos, *, jitter_index=0, moving=False, render_scale=1.0, max_accum_frames=999999999.0):
    """view / proj: float32 mathematical (row, col) matrices."""
    view = np.ascontiguousarray(view, dtype=np.float32)
    proj = np.ascontiguousarray(proj, dtype=np.float32)
    c = _abi.VrtCamera()
    c.view[:] = view.reshape(-1).tolist()
    c.proj[:] = proj.reshape(-1).tolist()
    c.view_inv[:] = cam_mod.inverse_f32(view).reshape(-1).tolist()
    c.proj_inv[:] = cam_mod.inverse_f32(proj).reshape(-1).tolist()
    c.pos[:] = [float(x) for x in pos]
    c.jitter_index = int(jitter_index) & 0xFFFFFFFF
    c.camera_is_moving = int(bool(moving))
    c.render_scale, c.max_accum_frames = float(render_scale), float(max_accum_frames)
    return c


def default_camera(width, height, **kw):
    view, proj = cam_mod.default_matrices(width, height)
    return make_camera(view, proj, cam_mod.DEFAULT_POS, **kw)
