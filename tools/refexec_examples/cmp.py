"""The voxel arrays a reference example script authors through the PRODUCT's DSL shim (taichi/, scene.py) against the arrays
tools/refexec_examples/run.py saved for the same script ($REFEXEC_EXAMPLES_OUT/<script>.npz: the script under tests/refexec with the
reference's own Scene).  Exit code 1 if a voxel differs.      python tools/refexec_examples/cmp.py example4.py"""
import sys, os, tempfile
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/tests/golden"]
import numpy as np
import make_example_fixtures as mef
name = sys.argv[1]
with tempfile.TemporaryDirectory() as tmp:
    r = mef.record(name, tmp)
pm, pc = np.array(r.voxel_material), np.array(r.voxel_color)
e = np.load(os.path.join(os.environ.get("REFEXEC_EXAMPLES_OUT", "."), name + ".npz"))
dm = np.argwhere(pm != e["m"]); dc = np.argwhere((pc != e["c"]).any(-1))
print(name, "solid voxels:", int((pm != 0).sum()), " material cells differing:", len(dm), " colour cells differing:", len(dc),
      " max colour byte difference:", int(np.abs(pc.astype(int) - e["c"].astype(int)).max()), flush=True)
for x, y, z in dm[:5]: print("  mat", (x - 64, y - 64, z - 64), "product", pm[x, y, z], "refexec", e["m"][x, y, z])
for x, y, z in dc[:5]: print("  col", (x - 64, y - 64, z - 64), "product", pc[x, y, z], "refexec", e["c"][x, y, z], "mat", pm[x, y, z], e["m"][x, y, z])
sys.exit(1 if len(dm) or len(dc) else 0)
