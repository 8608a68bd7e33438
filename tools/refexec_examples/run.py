"""Cross-check of this repo's DSL shim (taichi/, scene.py: what runs the reference's example scripts in the product) against an
independent reading of Taichi: the same script, read from /root/reference, executed under tests/refexec with the reference's own Scene and
Renderer.set_voxel (scene.py here subclasses the reference's class), fed what Taichi leaves undefined exactly as the product shim defines
it -- ti.random(): the shim's stream under ti.seed(0); sin / cos / ...: libm's double routines rounded once to binary32 (taichi/math.py) --
and the authored voxel arrays are compared with the hashes of tests/golden/examples.json.

    python tools/refexec_examples/run.py example4.py      (build container only; minutes per script)"""
import sys, os, hashlib, json, math, runpy, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, os.path.join(ROOT, "tests", "refexec"), "/root/reference"]
os.chdir("/root/reference")
import numpy as np
import taichi as ti
state = [(0 * 747796405 + 2891336453) & 0xFFFFFFFF]     # the product shim's stream under ti.seed(0) (taichi/__init__.py)
def rnd(index):
    s = (state[0] * 747796405 + 2891336453) & 0xFFFFFFFF
    state[0] = s
    w = (((s >> ((s >> 28) + 4)) ^ s) * 277803737) & 0xFFFFFFFF
    w = (w >> 22) ^ w
    return (w >> 8) * (1.0 / 16777216.0)
ti.set_random_source(rnd)
def once(f):     # a double routine of libm on f32 arguments, its result rounded once
    def g(*a):
        try:
            return np.float32(f(*[float(x) for x in a]))
        except (ValueError, OverflowError, ZeroDivisionError):
            return np.float32(np.nan)
    return g
ti.set_elementary(sin=once(math.sin), cos=once(math.cos), tan=once(math.tan), asin=once(math.asin), acos=once(math.acos),
                  atan2=once(math.atan2), exp=once(math.exp), log=once(math.log), pow=once(math.pow))
ti.set_out_of_bounds_reads("zero")
import scene
name = sys.argv[1]
t = time.time()
runpy.run_path(os.path.join("/root/reference", name), run_name="__main__")
s = scene.INSTANCES[0]
m, c = np.ascontiguousarray(s.renderer.world.voxel_material.a), np.ascontiguousarray(s.renderer.world.voxel_color.a)
want = json.load(open(os.path.join(ROOT, "tests", "golden", "examples.json")))[name]
got = dict(material_sha256=hashlib.sha256(m.tobytes()).hexdigest(), color_sha256=hashlib.sha256(c.tobytes()).hexdigest(), solid=int((m != 0).sum()))
print(name, "solid", got["solid"], want["solid"], "material", got["material_sha256"] == want["material_sha256"], "colour", got["color_sha256"] == want["color_sha256"], f"{time.time() - t:.0f} s")
if os.environ.get("REFEXEC_EXAMPLES_OUT"): np.savez_compressed(os.path.join(os.environ["REFEXEC_EXAMPLES_OUT"], name + ".npz"), m=m, c=c)
