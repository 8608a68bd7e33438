"""Harness module the example scripts import as `scene` under tools/refexec_examples/run.py: the REFERENCE'S OWN `Scene` class
(/root/reference/scene.py, imported from where it lies: round_idx, set_voxel, get_voxel and the setters are its code, executed under
tests/refexec) with the two methods that need a window replaced -- the constructor (reference scene.py:113-129 without the GGUI window and
camera) and finish() (the render loop)."""
import importlib.util
import sys
import types

import taichi as ti

# scene.py:23 reaches into Taichi for a window-system handle it only uses in the interactive loop
for name in ("taichi.lang", "taichi.lang.impl"):
    sys.modules.setdefault(name, types.ModuleType(name))
sys.modules["taichi.lang.impl"]._ti_core = None
_spec = importlib.util.spec_from_file_location("reference_scene", "/root/reference/scene.py")
reference_scene = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(reference_scene)

INSTANCES = []


class Scene(reference_scene.Scene):
    def __init__(self, voxel_edges=0.06, exposure=3):
        self.renderer = reference_scene.Renderer(dx=reference_scene.VOXEL_DX, image_res=(16, 8), up=reference_scene.UP_DIR,
                                                 voxel_edges=voxel_edges, exposure=exposure)
        self.renderer.set_directional_light((1, 1, 1), 0.1, (0.0, 0.0, 0.0))   # scene.py:127
        self.args = dict(voxel_edges=voxel_edges, exposure=exposure)
        INSTANCES.append(self)

    def finish(self):
        pass
