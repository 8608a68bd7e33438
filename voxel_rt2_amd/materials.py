"""The 128-entry Disney material table (reference: renderer/materials.py:48-112).

Row layout = bsdf.py:26-37: base_col rgb, subsurface, metallic, specular, specular_tint,
roughness, anisotropic, sheen, sheen_tint, clearcoat, clearcoat_gloss, ior_minus_one (14 f32).
Ids 0, 1, 2 keep the defaults; id 2 is "emissive" by convention only (voxel_world.py:53).
"""
import csv
import os
import numpy as np

N_MATERIALS = 128
N_FIELDS = 14
_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "default_material_set.csv")


def default_row():
    # materials.py:50-63
    return np.array([1.0, 1.0, 1.0, 0.0, 0.0, 0.04, 0.0, 0.9, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0], dtype=np.float32)


def load_table(path=None):
    table = np.tile(default_row(), (N_MATERIALS, 1))
    with open(path or _DATA, newline="") as f:
        rows = list(csv.reader(f))[1:]
    for row in rows:
        vals = np.array([float(x) for x in row], dtype=np.float32)
        table[int(vals[0])] = vals[1:15]
    return np.ascontiguousarray(table, dtype=np.float32)
