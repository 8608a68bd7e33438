"""`Renderer`: the drop-in for the reference's renderer.pathtracer.Renderer, on libvrt_hip.so.

Same constructor and method names as /root/reference/renderer/pathtracer.py (the surface that
scene.py drives); every method forwards to the C ABI of include/vrt_api.h.
Scalar fields the reference pokes with ``field[None] = x`` (floor_height, floor_color,
floor_material, background_color, use_physical_atmosphere, atmos.use_clouds, fov) are `_Field`
objects supporting the same syntax.  The module constants of the reference are keyword arguments
/ environment variables here: VRT_MAX_DEPTH (MAX_RAY_DEPTH, default 4), VRT_RESTIR (USE_RESTIR_PT,
default 0), VRT_SEED, VRT_SKY_RES (default 3840).
"""
import ctypes as C
import math
import os
import struct
import numpy as np

from . import _abi, _lib, camera as cam_mod, host, materials
from ._session import NativeSession, NativeError  # noqa: F401


class _Field:
    """Stand-in for a 0-d Taichi field: ``f[None] = v`` / ``f[None]``."""

    def __init__(self, value, on_change=None):
        self._v, self._cb = value, on_change

    def __getitem__(self, _):
        return self._v

    def __setitem__(self, _, value):
        self._v = value
        if self._cb:
            self._cb()


class _AtmosProxy:
    def __init__(self, on_change):
        self.use_clouds = _Field(0, on_change)


_F32 = struct.Struct("f")


def _f32(x):
    """x rounded to binary32 (the reference's voxel colours are f32 vectors)."""
    return _F32.unpack(_F32.pack(x))[0]


class VoxelStore:
    """Renderer.set_voxel / get_voxel and their storage (pathtracer.py:1325-1334, voxel_world.py:7-18):
    int8 material + uint8 rgb per voxel, index (x+G/2, y+G/2, z+G/2), G = voxel_grid_res (128 in the reference,
    pathtracer.py:83).  Kept on the host because the example kernels author the scene on the host; bytearray-backed
    so a per-voxel call costs ~1 us."""

    def _init_voxels(self, grid_res=128):
        g = self.voxel_grid_res = int(grid_res)
        self._mat = bytearray(g * g * g)
        self._rgb = bytearray(g * g * g * 3)
        self.voxel_material = np.frombuffer(self._mat, dtype=np.int8).reshape(g, g, g)
        self.voxel_color = np.frombuffer(self._rgb, dtype=np.uint8).reshape(g, g, g, 3)
        self._voxels_dirty = True

    def set_voxel(self, idx, mat, color):
        g = self.voxel_grid_res
        h = g >> 1
        x, y, z = int(idx[0]) + h, int(idx[1]) + h, int(idx[2]) + h
        if not (0 <= x < g and 0 <= y < g and 0 <= z < g):
            return  # the reference writes out of bounds here (undefined behaviour)
        i = (x * g + y) * g + z
        self._mat[i] = int(mat) & 0xFF  # ti.cast(mat, ti.i8)
        rgb = self._rgb
        j = 3 * i
        for k in (0, 1, 2):  # math_utils.py:86-92: u8(clamp(c, 0, 1) * 255), evaluated in f32
            c = _f32(color[k])
            c = 0.0 if c < 0.0 else (1.0 if c > 1.0 else c)
            rgb[j + k] = int(_f32(c * 255.0))
        self._voxels_dirty = True

    def get_voxel(self, ijk):
        g = self.voxel_grid_res
        h = g >> 1
        x, y, z = int(ijk[0]) + h, int(ijk[1]) + h, int(ijk[2]) + h
        if not (0 <= x < g and 0 <= y < g and 0 <= z < g):
            return 0, (0.0, 0.0, 0.0)
        i = (x * g + y) * g + z
        m = self._mat[i]
        j = 3 * i
        return (m - 256 if m > 127 else m), (_f32(self._rgb[j] / 255.0), _f32(self._rgb[j + 1] / 255.0), _f32(self._rgb[j + 2] / 255.0))

    def set_voxel_arrays(self, mat, rgb):
        self.voxel_material[...] = mat
        self.voxel_color[...] = rgb
        self._voxels_dirty = True


class Renderer(VoxelStore):
    def __init__(self, dx, image_res, up, voxel_edges, exposure=3, *, max_depth=None, use_restir=None, seed=None, sky_res=None,
                 device=0, rows=None, grid_res=None):
        self.image_res = tuple(int(x) for x in image_res)
        self.aspect_ratio = self.image_res[0] / self.image_res[1]
        self.exposure = exposure
        self.up = tuple(up)
        self.max_depth = int(os.environ.get("VRT_MAX_DEPTH", 4)) if max_depth is None else int(max_depth)
        self.use_restir = bool(int(os.environ.get("VRT_RESTIR", 0))) if use_restir is None else bool(use_restir)
        self.seed = int(os.environ.get("VRT_SEED", 0)) if seed is None else int(seed)
        self.sky_res = int(os.environ.get("VRT_SKY_RES", 3840)) if sky_res is None else int(sky_res)
        cfg = host.make_config(self.image_res[0], self.image_res[1], voxel_edges=voxel_edges, exposure=exposure,
                               max_depth=self.max_depth, use_restir=self.use_restir, seed=self.seed, sky_res=self.sky_res,
                               device=device, rows=rows, dx=dx,
                               grid_res=int(os.environ.get("VRT_GRID_RES", 128)) if grid_res is None else int(grid_res))
        self._s = NativeSession(_lib.load(), "vrt_", cfg)   # the library insists on dx == 2 / grid_res
        if int(os.environ.get("VRT_REFERENCE_INDEXING", 0)):
            self._s.set_reference_indexing(True)

        # voxel storage the user kernels write through set_voxel (voxel_world.py:7-18)
        self._init_voxels(cfg.grid_res)

        dirty = self._mark_scene_dirty
        self.floor_height = _Field(0.0, dirty)       # pathtracer.py:91-93
        self.floor_color = _Field((1.0, 1.0, 1.0), dirty)
        self.floor_material = _Field(1, dirty)
        self.background_color = _Field((0.0, 0.0, 0.0), dirty)
        self.use_physical_atmosphere = _Field(0, dirty)
        self.atmos = _AtmosProxy(dirty)
        self.fov = _Field(float(np.deg2rad(50.0)))    # pathtracer.py:89
        self._light = dict(direction=(1.0, 1.0, 1.0), cone=0.1, color=(0.0, 0.0, 0.0), weight=0.0)
        self._scene_dirty = True

        self._pos = cam_mod.DEFAULT_POS
        self._look_at = cam_mod.DEFAULT_LOOK_AT
        self._view = self._proj = None
        self._jitter_index = 0
        self._moving = False
        self._render_scale = 1.0
        self._max_samples = 999999999.0
        self._cam_dirty = True
        self.current_spp = 0
        self.current_frame = 0
        self._s.upload_materials(materials.load_table())
        if self.sky_res > 0:
            self._s.upload_cloud_texture(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cloud_texture.npy")))

    # -- scalar state -----------------------------------------------------------------------
    def _mark_scene_dirty(self):
        self._scene_dirty = True

    def set_directional_light(self, direction, light_cone_angle, light_color):  # pathtracer.py:139-144
        self._light = dict(direction=tuple(direction), cone=float(light_cone_angle), color=tuple(light_color), weight=3.0)
        self._scene_dirty = True

    def set_camera_is_moving(self, val):
        self._moving, self._cam_dirty = bool(val), True

    def set_render_scale(self, val):
        self._render_scale, self._cam_dirty = float(val), True

    def set_max_samples(self, max_samples):
        self._max_samples, self._cam_dirty = float(max_samples), True

    def set_camera_pos(self, x, y, z):
        self._pos, self._cam_dirty = (float(x), float(y), float(z)), True

    def set_look_at(self, x, y, z):
        self._look_at = (float(x), float(y), float(z))

    def set_up(self, x, y, z):
        self.up = (float(x), float(y), float(z))

    def set_fov(self, fov):
        self.fov[None] = float(fov)

    def set_proj_mat(self, M):  # M in glm memory order, like ti.ui.Camera.get_projection_matrix
        self._proj = cam_mod.from_glm_memory(M)
        self._jitter_index += 1  # one TAA jitter draw per call (pathtracer.py:264-265)
        self._cam_dirty = True

    def set_view_mat(self, M):
        self._view = cam_mod.from_glm_memory(M)
        self._cam_dirty = True

    def copy_prev_matrices(self):
        self._push()
        self._s.end_frame()

    # -- state push -------------------------------------------------------------------------
    def _push(self):
        if self._scene_dirty:
            self._s.set_scene(host.make_scene_params(
                floor_height=self.floor_height[None], floor_color=self.floor_color[None], floor_material=self.floor_material[None],
                background_color=self.background_color[None], light_direction=self._light["direction"],
                light_cone=self._light["cone"], light_color=self._light["color"], light_weight=self._light["weight"],
                use_physical_sky=self.use_physical_atmosphere[None], use_clouds=self.atmos.use_clouds[None]))
            self._scene_dirty = False
        if self._cam_dirty:
            if self._view is None or self._proj is None:
                self._view, self._proj = cam_mod.default_matrices(*self.image_res, pos=self._pos, look=self._look_at, fov=self.fov[None])
            self._s.set_camera(host.make_camera(self._view, self._proj, self._pos, jitter_index=self._jitter_index,
                                                moving=self._moving, render_scale=self._render_scale,
                                                max_accum_frames=self._max_samples))
            self._cam_dirty = False

    # -- work (pathtracer.py:314-329, 664-668, 1310-1323) -------------------------------------
    def prepare_data(self):
        self._push()
        if self._voxels_dirty:
            self._s.upload_voxels(self.voxel_material, self.voxel_color)
            self._voxels_dirty = False
        self._s.prepare()

    def accumulate_clouds(self, max_samples):
        self._push()
        self._s.sky_accumulate_clouds(max_samples)

    def compute_atmosphere(self, slice_idx, max_slices):
        self._push()
        self._s.sky_compute_slice(slice_idx, max_slices)

    def reset_framebuffer(self):
        self.current_spp = 0
        self._s.reset()

    def accumulate(self, n=1):
        self._push()
        self._s.accumulate(n)
        self.current_spp += n
        self.current_frame += n

    def fetch_image(self):
        """LDR rgba float32 [H, W, 4] (row 0 = bottom, like the reference's (u, v) image)."""
        self._push()
        return self._s.fetch_ldr()

    def fetch_hdr(self):
        return self._s.fetch_hdr()

    def set_reference_indexing(self, on=True):
        """Occupancy queries outside the grid read the bit the reference's index arithmetic addresses (raytracer.py:17-38)
        instead of "empty" (the default).  Also `VRT_REFERENCE_INDEXING=1` in the environment."""
        self._s.set_reference_indexing(on)

    # The reference shows every frame (scene.py:255-262: accumulate, fetch_image, canvas.set_image).  A blocking fetch_image
    # waits for every launch queued so far; these two queue the tonemap and the copy of an 8-bit image behind the frame and
    # return, so the caller can queue the next frame and pick the image up a frame later (two slots alternate).
    def present_async(self, slot=0):
        self._push()
        if not hasattr(self, "_present"):
            self._present = [self._s.host_alloc((self.image_res[1], self.image_res[0], 4), np.uint8) for _ in range(2)]
        self._s.fetch_ldr8_async(self._present[slot & 1], slot & 1)

    def present_wait(self, slot=0):
        """rgba8 [H, W, 4] (row 0 = bottom) of the frame present_async(slot) was called on.  A copy: the page-locked buffer
        behind it is reused two frames later and is freed when the session closes."""
        self._s.fetch_wait(slot & 1)
        return self._present[slot & 1].copy()

    def stats(self):
        return self._s.stats()

    @property
    def session(self):
        return self._s
