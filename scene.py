"""`Scene`: the public API of voxel-rt2's scene.py (reference scene.py:112-169) on the MI355X renderer.

Same names and argument meaning -- Scene(voxel_edges, exposure), set_voxel / get_voxel, set_floor,
set_directional_light, set_background_color, set_use_physical_sky, set_use_clouds, finish() -- so the
reference's example1..10.py and main.py run unmodified (`from scene import Scene`).  Differences, all in
finish() (reference scene.py:171-297, an endless GGUI window loop):

  * headless: renders VRT_FRAMES frames (default 64) of VRT_SPP samples each (default 1, the
    reference's samples_per_frame), then writes the tonemapped image as PNG into ./screenshot/
    (or $VRT_OUT) and returns;
  * the sky phase machine (32 cloud passes, 32 skybox slices, scene.py:243-253) runs up front;
  * the camera stays at the reference's initial pose (scene.py:28-29); no WASD / mouse.

Environment: VRT_RES=WxH (default 1920x1080 = SCREEN_RES), VRT_FRAMES, VRT_SPP, VRT_OUT, VRT_SEED,
VRT_MAX_DEPTH, VRT_RESTIR, VRT_SKY_RES, VRT_DEVICE; VRT_PRESENT=1 copies every frame's 8-bit image to the host
like the reference's window loop shows it (asynchronously, a frame behind: Renderer.present_async).
"""
import os
import sys
import time
from datetime import datetime

import numpy as np

import taichi as ti
from taichi.math import Vector
from voxel_rt2_amd import camera as cam_mod
from voxel_rt2_amd.renderer import Renderer

VOXEL_DX = 1 / 64
SCREEN_RES = tuple(int(v) for v in os.environ.get("VRT_RES", "1920x1080").lower().split("x"))
UP_DIR = (0, 1, 0)


class Camera:
    """The reference's initial pose (scene.py:25-30); the interactive part is out of scope."""

    def __init__(self, up=UP_DIR):
        self._camera_pos = np.array((0.4, 0.5, 2.0))
        self._lookat_pos = np.array((0.0, 0.0, 0.0))
        self._up = np.array(up, dtype=np.float64)

    @property
    def position(self):
        return self._camera_pos

    @property
    def look_at(self):
        return self._lookat_pos

    def update_camera(self, delta_time):
        return False


class Scene:
    def __init__(self, voxel_edges=0.06, exposure=3):
        ti.init(arch=ti.vulkan)
        self.camera = Camera(up=UP_DIR)
        self.renderer = Renderer(dx=VOXEL_DX, image_res=SCREEN_RES, up=UP_DIR, voxel_edges=voxel_edges, exposure=exposure,
                                 device=int(os.environ.get("VRT_DEVICE", 0)))
        self.renderer.set_camera_pos(*self.camera.position)
        self.renderer.set_directional_light((1, 1, 1), 0.1, (0.0, 0.0, 0.0))  # default values (scene.py:127)
        self.hdr = None
        self.image = None

    @staticmethod
    def round_idx(idx_):  # scene.py:131-137: ti.cast(idx_, ti.f32), ti.round, cast to i32
        return [x if x.__class__ is int else int(ti.round(float(x))) for x in (idx_[0], idx_[1], idx_[2])]

    def set_voxel(self, idx, mat, color):
        self.renderer.set_voxel(self.round_idx(idx), mat, color)

    def get_voxel(self, idx):
        mat, color = self.renderer.get_voxel(self.round_idx(idx))
        return mat, Vector(list(color))

    def set_floor(self, height, color, material=1):
        self.renderer.floor_height[None] = height
        self.renderer.floor_color[None] = tuple(float(c) for c in color)
        self.renderer.floor_material[None] = int(material)

    def set_directional_light(self, direction, direction_noise, color):
        self.renderer.set_directional_light(tuple(float(c) for c in direction), direction_noise, tuple(float(c) for c in color))

    def set_background_color(self, color):
        self.renderer.background_color[None] = tuple(float(c) for c in color)

    def set_use_physical_sky(self, use):
        self.renderer.use_physical_atmosphere[None] = 1 if use else 0

    def set_use_clouds(self, use):
        self.renderer.atmos.use_clouds[None] = 1 if use else 0

    def finish(self):
        r = self.renderer
        frames = int(os.environ.get("VRT_FRAMES", 64))
        samples_per_frame = int(os.environ.get("VRT_SPP", 1))
        t_start = time.time()
        r.prepare_data()
        if r.use_physical_atmosphere[None] == 1:
            print("Computing clouds")
            max_samples = max_slices = 32
            for _ in range(max_samples):
                r.accumulate_clouds(max_samples)
            for s in range(max_slices):
                r.compute_atmosphere(s, max_slices)
            r.session.sync()
            print(f"Done atmosphere & clouds ({time.time() - t_start:.1f} s)")

        aspect = SCREEN_RES[0] / SCREEN_RES[1]
        proj = cam_mod.perspective(r.fov[None], aspect, cam_mod.Z_NEAR, cam_mod.Z_FAR)
        view = cam_mod.look_at(self.camera.position, self.camera.look_at, UP_DIR)
        present = bool(int(os.environ.get("VRT_PRESENT", 0)))
        t0 = time.time()
        for k in range(frames):
            r.set_max_samples(999999999.0)
            r.set_render_scale(1.0)
            r.set_camera_is_moving(False)
            r.set_proj_mat(cam_mod.to_glm_memory(proj))
            r.set_view_mat(cam_mod.to_glm_memory(view))
            r.accumulate(samples_per_frame)
            if present:   # scene.py:260 fetch_image() / canvas.set_image(): frame k - 1 is on the host while frame k renders
                r.present_async(k)
                if k:
                    self.presented = r.present_wait(k - 1)
            r.copy_prev_matrices()
        if present and frames:
            self.presented = r.present_wait(frames - 1)
        r.session.sync()
        dt = time.time() - t0
        n = frames * samples_per_frame
        print(f"{n} samples took {dt:.3f} s ({SCREEN_RES[0] * SCREEN_RES[1] * n / max(dt, 1e-9) / 1e6:.1f} Mpath-samples/s)")

        self.image = r.fetch_image()
        self.hdr = r.fetch_hdr()
        out = os.environ.get("VRT_OUT")
        if out is None:
            os.makedirs("screenshot", exist_ok=True)
            main = os.path.split(getattr(sys.modules.get("__main__"), "__file__", "scene"))[1]
            out = os.path.join("screenshot", f"{main}-{datetime.today().strftime('%Y-%m-%d-%H%M%S')}.png")
        if out:
            save_image(self.image, out)
            print(f"Image has been saved to {out}")
        return self.image


def save_image(ldr_rgba, path):
    """LDR float rgba [H, W, 4] with row 0 at the bottom -> 8-bit PNG."""
    from PIL import Image
    img = (np.clip(ldr_rgba[::-1, :, :3], 0.0, 1.0) * 255.0 + 0.5).astype(np.uint8)
    Image.fromarray(img).save(path)
