#!/usr/bin/env python3
"""Writes the sparse (S1) and dense scenes' material arrays and runs tools/probes/walk_alone on each (GPU box)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from voxel_rt2_amd import scenes
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
for name, seed in (("s1", 0), ("dense", 12345)):
    mat, _, _ = scenes.SCENES[name](seed)
    path = os.path.join(out, f"walk_{name}.bin")
    np.ascontiguousarray(mat, dtype=np.int8).tofile(path)
    print(f"== scene {name}", flush=True)
    subprocess.run([os.path.join(ROOT, "tools", "probes", "walk_alone"), path], check=False, timeout=300)
