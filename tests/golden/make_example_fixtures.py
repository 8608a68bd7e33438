#!/usr/bin/env python3
"""Fixtures that stand in for the reference's example scripts where the reference tree is absent (the GPU box).

Runs main.py and example1..10.py FROM /root/reference (read at run time, never copied) through this repo's `scene.py` +
host-side `taichi` shim under ti.seed(0), with a recorder in place of the GPU renderer, and writes

    tests/golden/examples.json        per script: sha256 of the authored voxel_material / voxel_color arrays, solid-voxel count,
                                      and every scene parameter the script set (floor, light, background, sky flags, edges, exposure)
    tests/golden/example6_grid.npz    the two arrays example6.py authors (data, not script text): the scene of BASELINE config 3

    python tests/golden/make_example_fixtures.py      # build container only (needs /root/reference)

The hashes pin this repo's DSL shim and Scene API (its random stream, rounding, swizzles, kernel semantics): a change that moves
one voxel of any example shows up in tests/test_examples_shim.py as a hash mismatch instead of slipping through a range check.
"""
import hashlib
import json
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
for p in (ROOT, os.path.dirname(HERE)):
    sys.path.insert(0, p)
import numpy as np

SCRIPTS = ["main.py"] + [f"example{i}.py" for i in range(1, 11)]


def record(name, tmp):
    """Run one reference script; return the recorder that stood in for the renderer (tests/test_examples_shim.py)."""
    import scene
    import taichi
    from test_examples_shim import RecorderRenderer
    RecorderRenderer.instances.clear()
    old_renderer, old_finish, cwd = scene.Renderer, scene.Scene.finish, os.getcwd()
    scene.Renderer = RecorderRenderer
    scene.Scene.finish = lambda self: self.renderer.prepare_data()
    os.chdir(tmp)
    try:
        taichi.seed(0)
        runpy.run_path(os.path.join(REF, name), run_name="__main__")
    finally:
        scene.Renderer, scene.Scene.finish = old_renderer, old_finish
        os.chdir(cwd)
    assert len(RecorderRenderer.instances) == 1
    return RecorderRenderer.instances[0]


def describe(r):
    m, c = np.ascontiguousarray(r.voxel_material), np.ascontiguousarray(r.voxel_color)
    return dict(
        material_sha256=hashlib.sha256(m.tobytes()).hexdigest(), color_sha256=hashlib.sha256(c.tobytes()).hexdigest(),
        solid=int((m != 0).sum()), materials=sorted(int(x) for x in np.unique(m)),
        voxel_edges=float(r.args["voxel_edges"]), exposure=float(r.args["exposure"]),
        floor_height=float(r.floor_height[None]), floor_color=[float(x) for x in r.floor_color[None]], floor_material=int(r.floor_material[None]),
        background_color=[float(x) for x in r.background_color[None]],
        light=None if r.light is None else dict(direction=[float(x) for x in r.light[0]], cone=float(r.light[1]), color=[float(x) for x in r.light[2]]),
        use_physical_sky=int(r.use_physical_atmosphere[None]), use_clouds=int(r.atmos.use_clouds[None]))


def record_to(name, outdir):
    """One script, for tests/test_examples_shim.py (which runs the eleven scripts as parallel processes): the authored arrays and
    what describe() says about them, into outdir/<script>.npz / .json."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        r = record(name, tmp)
        d = describe(r)
        d["calls"] = list(r.calls)
        np.savez(os.path.join(outdir, name + ".npz"), voxel_material=r.voxel_material, voxel_color=r.voxel_color)
        json.dump(d, open(os.path.join(outdir, name + ".json"), "w"))


if __name__ == "__main__" and len(sys.argv) == 4 and sys.argv[1] == "--record":
    record_to(sys.argv[2], sys.argv[3])
elif __name__ == "__main__":
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in SCRIPTS:
            r = record(name, tmp)
            out[name] = describe(r)
            print(name, out[name]["solid"], out[name]["material_sha256"][:12], flush=True)
            if name == "example6.py":
                np.savez_compressed(os.path.join(HERE, "example6_grid.npz"), voxel_material=r.voxel_material.copy(), voxel_color=r.voxel_color.copy())
    json.dump(out, open(os.path.join(HERE, "examples.json"), "w"), indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "examples.json"))
