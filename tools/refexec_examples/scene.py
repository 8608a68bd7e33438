"""Harness stand-in for the reference's scene.py (which opens a window): the voxel-authoring surface the example scripts use, bound to the
reference's own Renderer executed under tests/refexec."""
import taichi as ti
import numpy as np
import renderer.pathtracer as pt

INSTANCES = []


@ti.data_oriented
class Scene:
    def __init__(self, voxel_edges=0.06, exposure=3):
        self.renderer = pt.Renderer(dx=1 / 64, image_res=(16, 8), up=(0, 1, 0), voxel_edges=voxel_edges, exposure=exposure)
        self.args = dict(voxel_edges=voxel_edges, exposure=exposure)
        INSTANCES.append(self)

    @staticmethod
    @ti.func
    def round_idx(idx_):          # scene.py:131-137
        idx = ti.cast(idx_, ti.f32)
        return ti.Vector([ti.round(idx[0]), ti.round(idx[1]), ti.round(idx[2])]).cast(ti.i32)

    @ti.func
    def set_voxel(self, idx, mat, color):      # scene.py:139-141
        self.renderer.set_voxel(self.round_idx(idx), mat, color)

    @ti.func
    def get_voxel(self, idx):                  # scene.py:143-146
        mat, color = self.renderer.get_voxel(self.round_idx(idx))
        return mat, color

    def set_floor(self, height, color, material=1): pass
    def set_directional_light(self, direction, direction_noise, color): pass
    def set_background_color(self, color): pass
    def set_use_physical_sky(self, use): pass
    def set_use_clouds(self, use): pass
    def finish(self): pass
