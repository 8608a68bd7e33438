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


def gris(scene):
    """A library built with -DVRT_DIAG_REGIONS -DVRT_DIAG_GRIS: the regions of the two spatial-reuse kernels (vrt_restir.h, VRT_GREGION)."""
    mat, rgb, params = scenes.SCENES[scene](0)
    sky = int(bool(params.get("use_physical_sky")))
    W, H = 1920, 1080
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0, use_restir=True, sky_res=1024 if sky else 0)
    s = NativeSession(lib, "vrt_", cfg)
    s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
    if sky:
        s.upload_cloud_texture(np.load(os.path.join(ROOT, "voxel_rt2_amd", "data", "cloud_texture.npy")))
    s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(W, H, jitter_index=1)); s.prepare()
    if sky:
        for _ in range(4):
            s.sky_accumulate_clouds(4)
        for sl in range(32):
            s.sky_compute_slice(sl, 32)
    out = np.zeros(64, dtype=np.uint64)
    s.accumulate(1); s.sync()
    lib.vrt_diag_regions(C.c_void_p(s._ctx), out.ctypes.data_as(C.c_void_p), 1)
    s.accumulate(1); s.sync()
    assert lib.vrt_diag_regions(C.c_void_p(s._ctx), out.ctypes.data_as(C.c_void_p), 1) == 0
    waves = W * H / 64
    names = {0: "shift (wave trips)", 1: "reconnection vertex evaluated (not an escape sample)", 2: "... its continuation direction", 3: "... its sun sample",
             4: "destination BSDF not an exact zero", 5: "... diffuse lobe", 6: "... specular lobe", 7: "visibility ray of the chosen sample"}
    print(f"== {scene}: spatial-reuse kernels, per wave (= 64 pixels) of one pass")
    for base, title in ((0, "first kernel (centre's sample into the neighbours' domains)"), (8, "second kernel (neighbours' samples into the centre's domain)")):
        print("  " + title)
        for k in range(8):
            ent, lanes = int(out[2 * (base + k)]), int(out[2 * (base + k) + 1])
            if ent:
                print(f"    {names[k]:56s} wave entries {ent / waves:7.2f}   lanes/entry {lanes / ent:5.1f}   lane-work per pixel {lanes / (W * H):6.2f}")
    s.close()


if len(sys.argv) > 1 and sys.argv[1] == "gris":
    for scene in sys.argv[2:] or ["s6", "sunlit"]:
        gris(scene)
    sys.exit(0)
for scene in sys.argv[1:] or ["s1", "sunlit", "dense"]:
    # NAME or NAME@row0:row1 (a rank's rows of a multi-GPU split) or NAME@row0:row1xK (K calls queued together, the pipeline in flight)
    rows, calls, stripes = None, 1, None     # NAME@S32/8/3[xK]: rank 3's share of 8 in stripes of 32 rows
    if "@" in scene:
        scene, r = scene.split("@")
        if "x" in r:
            r, k = r.split("x"); calls = int(k)
        if r.startswith("S"):
            stripes = tuple(int(x) for x in r[1:].split("/"))
        else:
            rows = tuple(int(x) for x in r.split(":"))
    mat, rgb, params = scenes.SCENES[scene](12345 if scene == "dense" else 0)
    params = dict(params, use_physical_sky=0, use_clouds=0)
    W, H = 1920, 1080
    cfg = host.make_config(W, H, voxel_edges=params["voxel_edges"], exposure=params["exposure"], max_depth=8, seed=0, rows=rows)
    s = NativeSession(lib, "vrt_", cfg)
    if stripes:
        s.set_row_stripes(*stripes)
    s.upload_voxels(mat, rgb); s.upload_materials(materials.load_table())
    s.set_scene(host.make_scene_params(**params)); s.set_camera(host.default_camera(W, H, jitter_index=1)); s.prepare()
    out = np.zeros(64, dtype=np.uint64)
    for _ in range(max(calls, 2)):
        s.accumulate(4)
    s.sync()
    lib.vrt_diag_regions(C.c_void_p(s._ctx), out.ctypes.data_as(C.c_void_p), 1)
    for _ in range(calls):
        s.accumulate(4)
    s.sync()
    assert lib.vrt_diag_regions(C.c_void_p(s._ctx), out.ctypes.data_as(C.c_void_p), 1) == 0
    n = W * (len(s.owned_rows()) if stripes else (rows[1] - rows[0]) if rows else H) * 4 * calls
    print(f"== {scene}{'' if not stripes else f' stripes {stripes}'}{'' if not rows else f' rows {rows[0]}-{rows[1]}'}{'' if calls == 1 else f', {calls} calls queued together'}: per path-sample")
    for rid, name in sorted(NAMES.items()):
        ent, lanes = int(out[2 * rid]), int(out[2 * rid + 1])
        if ent:
            print(f"  {name:38s} wave entries {ent / n:8.4f}   lanes/entry {lanes / ent:5.1f}   lane-work {lanes / n:7.3f}")
    cyc = {k: int(out[2 * (20 + k)]) for k in range(6)}
    if sum(cyc.values()):
        tot = sum(cyc.values())
        print(f"  pool wave-cycles per path-sample: {tot / n:.1f}")
        print("  pool wave-cycles by stage: " + "  ".join(f"{nm} {100 * cyc[k] / tot:.1f}%" for k, nm in
              ((0, "BEGIN"), (1, "WALK"), (2, "SHADE"), (3, "ESCAPE"), (5, "census"), (4, "start/exit"))))
    sub = [int(out[52 + k]) for k in range(4)]
    if sum(sub) and sum(cyc.values()):
        print("  BEGIN, of all pool wave-cycles: " + "  ".join(f"{nm} {100 * sub[k] / tot:.1f}%" for k, nm in
              enumerate(("work reservation (atomic round trip)", "record read + check", "set-up from a record", "set-up with a ray"))))
    s.close()
