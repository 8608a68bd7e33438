#!/usr/bin/env python3
"""Where do k_render's issue slots go?  Needs a library built with -DVRT_DIAG_REGIONS (VRT_LIB_PATH).
Prints, per instrumented region, wave-level entries per path-sample and the mean active lanes per entry."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")): sys.path.insert(0, p)
import numpy as np
from voxel_rt2_amd import host, scenes, materials, _lib
from voxel_rt2_amd._session import NativeSession
NAMES = {0: "path_begin", 1: "DDA loop trip (WALK stage)", 7: "DDA loop trip (raytrace(): fused kernel, inline shadow rays)", 2: "surface shading (frame, sun sample)", 10: "BSDF sample", 3: "shadow ray set-up/result",
         9: "closest ray set-up/result", 11: "pool SHADE stage", 12: "pool ESCAPE stage", 13: "pool BEGIN stage",
         14: "pool WALK stage", 15: "pool WALK refill", 16: "pool WALK hand-over", 17: "pool WALK suspend", 19: "flat descent: fine brick word needed", 18: "flat descent: fine brick word loaded (L2)", 4: "light-sample evaluation", 5: "escape / sky", 6: "path_finish", 8: "g-buffer (depth 0)"}
lib = _lib.load()
lib.vrt_diag_regions.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for scene in sys.argv[1:] or ["s1", "sunlit", "dense"]:
    mat, rgb, params = scenes.SCENES[scene](12345 if scene == "dense" else 0)
    params = dict(params, use_physical_sky=0, use_clouds=0)
    W, H = 1920, 1080
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0)
    s = NativeSession(lib, "vrt_", cfg)
    s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
    s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(W, H, jitter_index=1)); s.prepare()
    out = np.zeros(64, dtype=np.uint64)
    s.accumulate(4); s.sync()
    lib.vrt_diag_regions(C.c_void_p(s._ctx), out.ctypes.data_as(C.c_void_p), 1)
    s.accumulate(4); s.sync()
    assert lib.vrt_diag_regions(C.c_void_p(s._ctx), out.ctypes.data_as(C.c_void_p), 1) == 0
    n = W * H * 4
    print(f"== {scene}: per path-sample")
    for rid, name in sorted(NAMES.items()):
        ent, lanes = int(out[2 * rid]), int(out[2 * rid + 1])
        if ent:
            print(f"  {name:38s} wave entries {ent / n:8.4f}   lanes/entry {lanes / ent:5.1f}   lane-work {lanes / n:7.3f}")
    cyc = {k: int(out[2 * (20 + k)]) for k in range(6)}
    if sum(cyc.values()):
        tot = sum(cyc.values())
        print("  pool wave-cycles by stage: " + "  ".join(f"{nm} {100 * cyc[k] / tot:.1f}%" for k, nm in
              ((0, "BEGIN"), (1, "WALK"), (2, "SHADE"), (3, "ESCAPE"), (5, "census"), (4, "start/exit"))))
    sub = [int(out[52 + k]) for k in range(4)]
    if sum(sub) and sum(cyc.values()):
        print("  BEGIN, of all pool wave-cycles: " + "  ".join(f"{nm} {100 * sub[k] / tot:.1f}%" for k, nm in
              enumerate(("work reservation (atomic round trip)", "record read + check", "set-up from a record", "set-up with a ray"))))
    s.close()
