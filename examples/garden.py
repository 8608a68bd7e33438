"""A small scene written against the same Scene API + kernel DSL as voxel-rt2's examples (this file is new
code, not one of the reference's scripts).  Run from the repo root:

    VRT_RES=1280x720 VRT_FRAMES=128 python examples/garden.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scene import Scene  # noqa: E402
import taichi as ti  # noqa: E402
from taichi.math import *  # noqa: E402,F401,F403

scene = Scene(voxel_edges=0.04, exposure=1.5)
scene.set_floor(-0.3, (0.9, 0.9, 0.85), 10)
scene.set_directional_light((0.6, 1.0, 0.4), 0.1, (1.0, 0.95, 0.85))
scene.set_background_color((0.35, 0.5, 0.75))


@ti.func
def pillar(p, h, mat, color):
    for I in ti.grouped(ti.ndrange((-2, 3), (0, h), (-2, 3))):
        if abs(I.x) + abs(I.z) < 4:
            scene.set_voxel(p + ivec3(I.x, I.y, I.z), mat, color * (0.9 + 0.1 * ti.random()))


@ti.kernel
def build():
    for i, j in ti.ndrange((-40, 40), (-40, 40)):
        h = int(2 + 2 * ti.sin(i * 0.2) * ti.cos(j * 0.17))
        for k in range(h):
            scene.set_voxel(vec3(i, -19 + k, j), 80, vec3(0.15, 0.45 + 0.1 * ti.random(), 0.1))
    mats = [50, 51, 53, 54, 21, 32, 11, 20]
    for n in range(8):
        a = n / 8 * 2 * pi
        p = ivec3(int(28 * ti.cos(a)), -17, int(28 * ti.sin(a)))
        pillar(p, 12 + 3 * (n % 3), mats[n], vec3(0.9 - 0.08 * n, 0.4 + 0.05 * n, 0.3 + 0.08 * n))
    for I in ti.grouped(ti.ndrange((-6, 7), (-6, 7), (-6, 7))):
        if I.norm() < 6.5:
            scene.set_voxel(ivec3(0, -4, 0) + I, 2 if I.norm() > 5.5 and I.y > 3 else 52, vec3(1.0, 0.9, 0.7))


build()
scene.finish()
