"""tests/golden/reference/*.npz hold what THE REFERENCE'S OWN SOURCE FILES compute for a few small cases: they were written by
tests/golden/make_reference_vectors.py, which imports /root/reference/renderer/*.py from where they lie and executes them under
the Taichi-DSL emulation of tests/refexec (the Taichi JIT itself is not installed; that script's header lists what the emulation
stands for and what the reference leaves undefined).  Here the oracle, the product's device code compiled for the host and -- on
a GPU box -- libvrt_hip.so through the C ABI are driven through the same cases and must reproduce every buffer bit for bit:
HDR, LDR, g-buffer (depth, packed normal, position, packed material, reflection depth) and both temporal histories."""
import importlib.util
import os

import numpy as np
import pytest

import emu
import orc

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, "golden", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


mg = _load("make_golden")
CASES = _load("reference_cases").CASES


def check(session, name, reference_indexing=False):
    """Every buffer, every pixel, bit for bit.  DEFAULT mode leaves out the pixels a fixture lists in `undefined_px`: there one of
    the path's rays left the grid a rounding error before the reference's own `far` test caught it, and the reference's bit index
    for the cell outside the grid addressed another cell's SET bit (make_reference_vectors.py; only the dense grid has such
    pixels) -- the build reads "empty" outside the grid.  With `reference_indexing` the session reads such a
    cell the reference's way (vrt_set_reference_indexing, include/vrt_api.h) and NOTHING is left out."""
    if reference_indexing:
        session.set_reference_indexing(True)
    got = mg.run_case(session, CASES[name])
    want = np.load(os.path.join(HERE, "golden", "reference", name + ".npz"))
    skip = np.zeros((got["hdr"].shape[0], got["hdr"].shape[1]), bool)
    if "undefined_px" in want.files and not reference_indexing:
        skip = want["undefined_px"]
    assert skip.sum() <= 0.01 * skip.size
    spread = skip.copy()    # the reflection-depth prepass averages a 4x4 box (offsets -1..2, pathtracer.py:1030-1045)
    for dy in range(-2, 2):
        for dx in range(-2, 2):
            spread |= np.roll(np.roll(skip, dy, 0), dx, 1)
    assert sorted(f for f in want.files if f != "undefined_px") == sorted(got.keys())
    for key in got:
        a, b = np.ascontiguousarray(got[key]), want[key]
        assert a.shape == b.shape and a.dtype == b.dtype, key
        same = (a.view(np.uint8) == b.view(np.uint8)).reshape(a.shape[0], a.shape[1], -1).all(-1)
        if same.shape == skip.shape:
            same |= spread if key == "gbuf_refl_depth" else skip
        assert same.all(), f"{name}: {key} differs from the reference's output at {int((~same).sum())} of {same.size} pixels, first {np.argwhere(~same)[:4].tolist()}"


DENSE = sorted(n for n in CASES if "undefined_px" in np.load(os.path.join(HERE, "golden", "reference", n + ".npz")).files)


def test_every_case_has_a_fixture():
    have = {f[:-4] for f in os.listdir(os.path.join(HERE, "golden", "reference")) if f.endswith(".npz")}
    have -= {"rays_sunlit", "functions", "functions_sky"}    # single rays and single functions: the tests at the end
    assert have == set(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_equals_reference_source(name):
    o = orc.Oracle(mg.config_of(CASES[name]), threads=4)
    check(o, name)
    o.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_emulated_device_code_equals_reference_source(name):
    if len(CASES[name]) > 8 and CASES[name][8][0] != "given":
        pytest.skip("the sky precompute kernels (vrt_sky_kernels.hip) have no host build: the GPU test covers them")
    e = emu.Emulated(mg.config_of(CASES[name]))
    check(e, name)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["pool", "fused"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_equals_reference_source(name, schedule, monkeypatch):
    from voxel_rt2_amd import _lib
    from voxel_rt2_amd._session import NativeSession
    monkeypatch.setenv("VRT_RENDER", schedule)
    g = NativeSession(_lib.load(), "vrt_", mg.config_of(CASES[name]))
    check(g, name)
    g.close()


# ---- the reference's own reading of cells outside the grid (vrt_set_reference_indexing): no pixel is left out -------------------
def test_dense_cases_have_pixels_the_default_mode_leaves_out():
    assert len(DENSE) >= 2 and all(np.load(os.path.join(HERE, "golden", "reference", n + ".npz"))["undefined_px"].any() for n in DENSE)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_with_reference_indexing_equals_reference_source_everywhere(name):
    """EVERY case, not only the dense ones: the reference's source made those reads in all of them (at coarse levels they only
    cost it a descent), so the mode must leave the other frames as they are."""
    o = orc.Oracle(mg.config_of(CASES[name]), threads=4)
    check(o, name, reference_indexing=True)
    o.close()


@pytest.mark.parametrize("pooled", [False, True])
@pytest.mark.parametrize("name", DENSE + ["ref_sunlit_48x24_d5"])
def test_emulated_device_code_with_reference_indexing(name, pooled, monkeypatch):
    if pooled:
        monkeypatch.setenv("VRT_EMU_POOL", "1")      # the pooled schedule's stage functions (flat descent)
    e = emu.Emulated(mg.config_of(CASES[name]))
    check(e, name, reference_indexing=True)
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("schedule", ["pool", "fused"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_with_reference_indexing_equals_reference_source_everywhere(name, schedule, monkeypatch):
    from voxel_rt2_amd import _lib
    from voxel_rt2_amd._session import NativeSession
    monkeypatch.setenv("VRT_RENDER", schedule)
    g = NativeSession(_lib.load(), "vrt_", mg.config_of(CASES[name]))
    check(g, name, reference_indexing=True)
    g.close()


def test_oracle_rays_equal_reference_source():
    """600 rays through VoxelOctreeRaytracer.raytrace as the reference's raytracer.py computes them (rays_sunlit.npz): distance and
    iteration count for every ray -- NaN distances included (axis-parallel rays that start outside their slab: 0 * inf, raytracer.py:94,
    133) -- and, for hits, the voxel and the face normal.  On a miss the reference's cell / normal are whatever its last step left
    OUTSIDE the grid, where its own occupancy reads are out of bounds (undefined; the build reads "empty" there):
    not compared in the default mode, nothing downstream reads them (pathtracer.py:205).  With the reference's indexing
    (orc_set_reference_indexing) they ARE compared, on every ray: the walk then ends in the reference's own last state."""
    from voxel_rt2_amd import host, scenes
    v = np.load(os.path.join(HERE, "golden", "reference", "rays_sunlit.npz"))
    mat, rgb, params = scenes.scene_sunlit(0)
    n = len(v["distance"])
    for ref_idx in (False, True):
        o = orc.Oracle(host.make_config(16, 8, max_depth=2), threads=1)
        o.set_reference_indexing(ref_idx)
        orc.setup(o, mat, rgb, params)
        hits = 0
        for k in range(n):
            got = o.raytrace(v["origin"][k], v["direction"][k], 1e-6, np.inf)
            want = v["distance"][k]
            assert np.float32(got["distance"]).view(np.uint32) == want.view(np.uint32) or (np.isnan(got["distance"]) and np.isnan(want)), k
            assert got["iters"] == v["iters"][k], k
            if np.isfinite(want) or ref_idx:
                hits += bool(np.isfinite(want))
                assert list(got["cell"]) == list(v["cell"][k]), k
                assert np.array_equal(got["normal"], v["normal"][k]), k
        assert hits > 300 and np.isnan(v["distance"]).sum() > 10 and np.isinf(v["distance"]).sum() > 100
        o.close()


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _same(got, want):
    """Bit equality, any NaN equal to any NaN (payloads are not part of the contract)."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    return bool(np.all((_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))))


def test_oracle_functions_equal_reference_source():
    """functions.npz: single functions of the reference on random arguments (make_reference_vectors.function_vectors) -- the Disney
    BSDF's evaluation, pdfs and sampler on 300 random materials, cone sampling, the packing helpers, hash3, the tone curve, a
    reservoir's storage round trip and 150 reconnection shifts -- against the oracle's probes, bit for bit."""
    import ctypes as C
    from voxel_rt2_amd import host, scenes
    v = np.load(os.path.join(HERE, "golden", "reference", "functions.npz"))
    L, f = orc.lib(), orc.fptr
    n = len(v["mat"])
    for k in range(n):
        a = [np.ascontiguousarray(v[key][k]) for key in ("mat", "v", "n", "l")]
        out = np.zeros(7, np.float32)
        L.orc_unit_bsdf_eval(*[f(x) for x in a], f(out))
        assert _same(out, v["eval"][k]), ("eval", k, out, v["eval"][k])
        o1 = np.zeros(1, np.float32)
        L.orc_unit_lobe_pdf(*[f(x) for x in a], int(v["lobe"][k]), f(o1))
        assert _same(o1[0], v["lobe_pdf"][k]), ("lobe pdf", k)
        smp = np.zeros((4, 8), np.float32)
        L.orc_unit_bsdf_sample(f(a[0]), f(a[1]), f(a[2]), C.c_uint32(int(v["sample_seed"]) + k), 4, f(smp))
        assert _same(smp, v["sample"][k]), ("sample", k, smp, v["sample"][k])
    assert len({int(x) for x in v["sample"][:, :, 7].ravel()}) == 3     # every lobe was sampled
    for k in range(len(v["cone_cos"])):
        out = np.zeros((3, 3), np.float32)
        L.orc_unit_sample_cone(C.c_float(float(v["cone_cos"][k])), f(np.ascontiguousarray(v["cone_n"][k])), C.c_uint32(4), 3, f(out))
        assert _same(out, v["cone"][k]), ("cone", k)
    for k in range(len(v["oct_in"])):
        code, dec = np.zeros(2, np.uint16), np.zeros(3, np.float32)
        L.orc_unit_oct_encode(f(np.ascontiguousarray(v["oct_in"][k])), f(code))
        want = v["oct_code"][k]
        nan16 = lambda h: (h & 0x7C00) == 0x7C00 and (h & 0x3FF) != 0  # noqa: E731
        assert all(int(c) == int(w) or (nan16(int(c)) and nan16(int(w))) for c, w in zip(code, want)), ("oct encode", k, code, want)
        L.orc_unit_oct_decode(f(np.ascontiguousarray(want)), f(dec))
        assert _same(dec, v["oct_out"][k]), ("oct decode", k)
    for k in range(len(v["matenc"])):
        assert L.orc_unit_encode_material(int(v["matenc_id"][k]), f(np.ascontiguousarray(v["matenc_albedo"][k]))) == int(v["matenc"][k])
    for k in range(len(v["hash_out"])):
        assert L.orc_unit_hash3(*[int(x) for x in v["hash_in"][k]]) == int(v["hash_out"][k])
    out = np.zeros(len(v["uchimura_in"]), np.float32)
    L.orc_unit_uchimura(f(np.ascontiguousarray(v["uchimura_in"])), len(out), f(out))
    assert _same(out, v["uchimura_out"])
    for k in range(len(v["res_in"])):
        out = np.zeros(23, np.float32)
        L.orc_unit_reservoir_roundtrip(f(np.ascontiguousarray(v["res_in"][k])), f(out))
        assert _same(out, v["res_out"][k]), ("reservoir", k, out, v["res_out"][k])
    mat, rgb, params = scenes.scene_sunlit(0)
    o = orc.Oracle(host.make_config(16, 8, max_depth=2), threads=1)
    orc.setup(o, mat, rgb, params)
    zero_jac = 0
    for k in range(len(v["shift_out"])):
        a = [np.ascontiguousarray(v[key][k]) for key in ("shift_dst_pos", "shift_dst_n", "shift_dst_mat", "shift_src_pos", "shift_sample")]
        out = np.zeros(7, np.float32)
        L.orc_unit_shift(C.c_void_p(o._ctx), *[f(x) for x in a], f(out))
        assert _same(out, v["shift_out"][k]), ("shift", k, out, v["shift_out"][k])
        zero_jac += out[6] == 0.0
    assert 10 < zero_jac < len(v["shift_out"]) - 10      # both rejected and accepted shifts
    o.close()
    for k in range(len(v["light_in"])):      # the boundary's own arithmetic: set_directional_light's Python-scope normalisation and cosine
        sp = host.make_scene_params(light_direction=tuple(v["light_in"][k]), light_cone=float(v["light_cone"][k]))
        assert _same(np.array(list(sp.light_direction), np.float32), v["light_dir"][k]), ("light direction", k)
        assert _same(np.float32(sp.light_cos_theta_max), v["light_cos"][k]), ("cone cosine", k)
    o = orc.Oracle(host.make_config(16, 8, max_depth=2, sky_res=64), threads=1)     # project_sky / unproject_sky (atmos.py:428-455)
    for k in range(len(v["sky_dir"])):
        out = np.zeros(2, np.float32)
        L.orc_unit_project_sky(C.c_void_p(o._ctx), f(np.ascontiguousarray(v["sky_dir"][k])), f(out))
        assert _same(out, v["sky_uv"][k]), ("project_sky", k, out, v["sky_uv"][k])
    for k in range(len(v["sky_uv_in"])):
        out = np.zeros(3, np.float32)
        L.orc_unit_unproject_sky(C.c_void_p(o._ctx), f(np.ascontiguousarray(v["sky_uv_in"][k])), f(out))
        assert _same(out, v["sky_dir_out"][k]), ("unproject_sky", k)
    o.close()


SKY_OPS = {"rsi": (0, 2), "ozone": (1, 1), "density": (2, 3), "cloud_phase": (3, 1), "cloud_density": (4, 1), "cloud_shadow_od": (5, 1),
           "ray_transmittance": (6, 3), "clouds_scattering": (7, 5), "atmos_scattering_d0": (8, 6), "atmos_scattering_d1": (9, 6)}   # name -> (probe op, floats out)


def _sky_inputs():
    from voxel_rt2_amd import host
    v = np.load(os.path.join(HERE, "golden", "reference", "functions_sky.npz"))
    cloud = np.load(os.path.join(os.path.dirname(HERE), "voxel_rt2_amd", "data", "cloud_texture.npy"))
    cfg = host.make_config(16, 8, max_depth=2, sky_res=64, seed=int(v["seed"]))
    return v, cloud, cfg


def _check_sky(v, run):
    """run(op, rows of arguments, floats per result) -> results; every row of every function, bit for bit (any NaN equals any NaN)."""
    interesting = 0
    for name, (op, n_out) in SKY_OPS.items():
        got = run(op, np.ascontiguousarray(v[name + "_in"], np.float32), n_out)
        want = v[name + "_out"]
        bad = [k for k in range(len(want)) if not _same(got[k], want[k])]
        assert not bad, f"{name}: {len(bad)} of {len(want)} rows differ, first {bad[:3]}: got {got[bad[0]]}, reference {want[bad[0]]}"
        interesting += int((np.abs(want) > 0).any(axis=1).sum() > len(want) // 4)
    assert interesting == len(SKY_OPS)      # no function was only ever asked for zeros


def test_oracle_sky_functions_equal_reference_source():
    """functions_sky.npz: atmos.py one function at a time (make_reference_vectors.sky_function_vectors: rsi, the density profiles, the
    cloud tile lookup, the cloud shadow march, cloud_phase, get_ray_transmittance, clouds_scattering and atmospheric_scattering at both
    template depths -- atmos.py:9-15, 195-349, 355-425, 475-523) against the oracle's orc_unit_atmos probe, bit for bit."""
    import ctypes as C
    v, cloud, cfg = _sky_inputs()
    o = orc.Oracle(cfg, threads=1)
    o.upload_cloud_texture(cloud)
    L, f = orc.lib(), orc.fptr
    L.orc_set_trans_lut(C.c_void_p(o._ctx), f(np.ascontiguousarray(v["trans_lut"])))
    L.orc_set_cloud_ambient(C.c_void_p(o._ctx), f(np.ascontiguousarray(v["cloud_ambient"])))

    def run(op, args, n_out):
        out = np.zeros((len(args), n_out), np.float32)
        assert L.orc_unit_atmos(C.c_void_p(o._ctx), op, len(args), f(args), args.shape[1], f(out), n_out) == 0
        return out
    _check_sky(v, run)
    o.close()


@pytest.mark.gpu
def test_gpu_sky_functions_equal_reference_source():
    """The same rows through the device functions of vrt_sky_kernels.hip (vrt_sky_probe, include/vrt_api.h)."""
    import ctypes as C
    from voxel_rt2_amd import _lib
    from voxel_rt2_amd._session import NativeSession
    v, cloud, cfg = _sky_inputs()
    lib = _lib.load()
    g = NativeSession(lib, "vrt_", cfg)
    g.upload_cloud_texture(cloud)
    lut, amb = np.ascontiguousarray(v["trans_lut"]), np.ascontiguousarray(v["cloud_ambient"])

    def run(op, args, n_out):
        out = np.zeros((len(args), n_out), np.float32)
        rc = lib.vrt_sky_probe(C.c_void_p(g._ctx), op, len(args), orc.fptr(args), args.shape[1], orc.fptr(out), n_out, orc.fptr(lut), orc.fptr(amb))
        assert rc == 0, lib.vrt_last_error()
        return out
    _check_sky(v, run)
    g.close()


def test_voxel_authoring_equals_reference_source():
    """What an example script's kernel does to the grid -- Scene.set_voxel -> round_idx -> Renderer.set_voxel (scene.py:131-141,
    pathtracer.py:1325-1328): rounding half away from zero, the i8 cast of (possibly fractional) material ids, u8(clamp(c) * 255) in
    f32 -- through this repo's scene.py and voxel store against the reference's own set_voxel (functions.npz: 1 500 calls)."""
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import scene as product_scene
    from voxel_rt2_amd.renderer import VoxelStore
    v = np.load(os.path.join(HERE, "golden", "reference", "functions.npz"))
    vs = VoxelStore()
    vs._init_voxels(128)
    for idx, mat, col in zip(v["author_idx"], v["author_mat"], v["author_color"]):
        vs.set_voxel(product_scene.Scene.round_idx(idx), float(mat), [float(c) for c in col])
    cells = np.argwhere(vs.voxel_material != 0)
    assert np.array_equal(cells, v["author_cells"].astype(np.int64))
    assert np.array_equal(vs.voxel_material[tuple(cells.T)], v["author_cell_mat"])
    assert np.array_equal(vs.voxel_color[tuple(cells.T)], v["author_cell_rgb"])
