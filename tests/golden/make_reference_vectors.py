#!/usr/bin/env python3
"""Writes tests/golden/reference/*.npz: outputs of THE REFERENCE'S OWN SOURCE FILES -- test infrastructure, build container only.

The reference (/root/reference/renderer/*.py) is Python that only runs through the Taichi JIT, which is not installed and
cannot be fetched.  What is possible is to execute the reference's source text, imported from where it
lies and never copied, under a host-side emulation of the Taichi DSL subset it uses (tests/refexec/taichi: one reading of
Taichi's semantics -- f32 / i32 defaults and Taichi's type promotion, value semantics, typed locals, ti.template() parameters by
reference, constant whole-number powers by repeated multiplication, maxnum / minnum, unorm8 textures).  A `ReferenceSession`
below wraps the reference's `Renderer` in the session interface the oracle, the emulated device code and libvrt_hip.so already
share (voxel_rt2_amd/_session.py), so the cases of this file are driven through `make_golden.run_case` exactly like theirs.

What the reference leaves undefined is supplied here the way DESIGN.md section 5 records it (and nowhere else):
  * ti.random(): the build's counter-based streams (include/vrt_detmath.h dm_rng: stream 0 = render, 1 = spatial_GRIS, per pixel
    and frame; stream 3 = the TAA jitter draw of set_proj_mat), read through the oracle's probe `orc_unit_rng`;
  * sin / cos / exp / log / pow / acos / atan2: the numeric contract's functions (include/vrt_detmath.h, probe `orc_unit_detmath`;
    their accuracy is bounded separately by tests/test_detmath.py).  `--libm` uses numpy's instead and reports how far the
    images move (nothing is written);
  * reads outside the image: the nearest pixel for bilinear / Catmull-Rom taps, unwritten memory (zero) for spatial_GRIS' g-buffer taps,
    which its own distance test then rejects (= the build's "taps outside the image are skipped");
  * `self.current_frame` inside spatial_GRIS: the value at first compile, 0;
  * the camera matrices' inverses: inverted in float64, rounded once (voxel_rt2_amd/camera.py), as the boundary hands them over.
What these vectors pin: the oracle (and through it the HIP library) against the reference's source text under those semantics.
What they cannot pin: Taichi's code generation itself (fast-math reassociation, its own elementary functions).

    python tests/golden/make_reference_vectors.py [case ...]      # from the repo root; ~4 min per case (2 M-voxel Python loops)
    ... --verify [case ...]     runs the cases again and compares with the committed files; --check-golden runs make_golden.py's cases
    ... --libm [case ...]       numpy's elementary functions instead of the contract's: reports the drift
"""
import ctypes as C
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
OUT = os.path.join(HERE, "reference")
sys.path[:0] = [os.path.join(ROOT, "tests", "refexec"), REFERENCE, ROOT, os.path.join(ROOT, "tests"), HERE]

import numpy as np  # noqa: E402
import taichi  # noqa: E402,F401   (tests/refexec/taichi, before anything puts the repo root -- and the product's DSL shim -- first)

from reference_cases import ALL_CASES as CASES  # noqa: E402


_PREPARED = {}


class ReferenceSession:
    """The reference's Renderer (pathtracer.py:27) behind the NativeSession interface."""

    def __init__(self, cfg, libm=False):
        import taichi as ti
        import orc
        assert "refexec" in ti.__file__, "tests/refexec must shadow the product's DSL shim"
        os.chdir(REFERENCE)      # materials.py:97 and atmos.py:86 open files relative to the working directory
        import renderer.pathtracer as pt
        self.ti, self.pt, self.cfg = ti, pt, cfg
        self.W, self.H = cfg.width, cfg.height
        assert cfg.grid_res == 128
        pt.MAX_RAY_DEPTH = cfg.max_depth            # module constants of the reference (pathtracer.py:15-17), set from outside
        pt.USE_RESTIR_PT = bool(cfg.use_restir)
        self.L = orc.lib()
        self._fptr = orc.fptr
        if not libm:
            ti.set_elementary(**{n: self._dm(op) for op, n in enumerate(["sin", "cos", "exp", "log", "pow", "acos", "atan2"])})
        ti.set_random_source(self._random)
        self.stream, self.jitter_index, self._cache = 3, 0, {}
        r = self.r = pt.Renderer(dx=cfg.dx, image_res=(self.W, self.H), up=(0, 1, 0), voxel_edges=cfg.voxel_edges, exposure=cfg.exposure)
        # reads outside the image are undefined in the reference; DESIGN.md section 5: bilinear / Catmull-Rom taps see the nearest
        # pixel, spatial_GRIS taps outside the image are skipped -- which is what the reference's own distance test (:912) does to a
        # tap whose g-buffer reads as memory nobody wrote (depth 0 = the near plane, 0.01 from the camera)
        ti.set_out_of_bounds_reads("clamp", r.color_buffer, r.color_buffer_specular, r.history_buffer, r.history_buffer_specular,
                                   r.history_buffer_specular_depth, r.gbuff_prev_depth, r.gbuff_prev_normals, r.gbuff_depth_reflection)
        ti.set_out_of_bounds_reads("zero", r.gbuff_normals, r.gbuff_depth, r.gbuff_mat_id, r.spatial_reservoirs)
        if cfg.sky_res:
            # the skybox tables' size is a constant of Atmos.__init__ (atmos.py:66-69: 3840, far beyond a Python loop); the kernels
            # read it from these instance attributes, which are replaced here by the same expressions at the case's size
            R, a = int(cfg.sky_res), r.atmos
            a.skybox_fres = ti.Vector([1.0 / R, 1.0 / R])
            a.skybox_res = ti.Vector([R, R])
            a.skybox_scattering = ti.Vector.field(3, dtype=ti.f32, shape=(R, R))
            a.skybox_transmittance = ti.Vector.field(3, dtype=ti.f32, shape=(R, R))
        self._cloud_pass = 0
        # Where the reference's own indexing leaves the grid.  A ray that reaches the grid's far face with hit_distance a rounding
        # error short of `far` (raytracer.py:104) takes one more step with a cell coordinate of -1 or 128; linearize_index (:34-37)
        # then addresses ANOTHER cell's bit (x = 128 is x = 0 of the next row) or memory outside the level.  Undefined in the
        # reference (its levels >= 2 already live outside the allocation); the build reads "empty" there (DESIGN.md
        # section 5).  A set bit read that way at a COARSE level only makes the walk descend where it could have stepped (frequent, and
        # without effect on what is hit); at level 0 it is reported as a hit on a voxel outside the grid.  Pixels whose paths got
        # such a hit are recorded: the reference's value there is an artefact of that indexing (a black speck on the grid's
        # face), and the comparison leaves them out.
        self.undefined_px = np.zeros((self.H, self.W), bool)
        rt, inner = r.voxel_raytracer, type(r.voxel_raytracer).query_occupancy

        def query_occupancy(ipos, lod):
            hit = inner(rt, ipos, lod)
            if hit and int(lod) == 0:
                res = rt.voxel_grid_res >> int(lod)
                if not all(0 <= int(c) < res for c in ipos):
                    idx = ti._loop_index[0]
                    if idx is not None and self.stream in (0, 1):
                        self.undefined_px[int(idx[1]), int(idx[0])] = True
            return hit
        rt.query_occupancy = query_occupancy

    def _dm(self, op):
        def f(a, b=0.0):
            if op == 4 and b == 1.5:       # the contract's pow(x, 1.5) is x * sqrt(x) (atmos.py:25)
                return np.float32(a) * np.sqrt(np.float32(a))
            x, y, out = np.array([a], np.float32), np.array([b], np.float32), np.zeros(1, np.float32)
            self.L.orc_unit_detmath(op, 1, self._fptr(x), self._fptr(y), self._fptr(out))
            return out[0]
        return f

    def _random(self, index):
        if self.stream == 3:
            key, frame, pix = ("jitter", self.jitter_index), self.jitter_index, 0
        elif self.stream == 2:    # sky precompute (oracle/orc_atmos.h: frame = the kernel's tag or the cloud pass, texel u * R + v)
            pix = int(index[0]) * int(self.cfg.sky_res) + int(index[1]) if index is not None else 0
            frame = self._frame
            key = (2, frame, pix)
        else:
            pix = int(index[1]) * self.W + int(index[0])
            frame = self._frame
            key = (self.stream, frame, pix)
        st = self._cache.get(key)
        if st is None or st[1] >= len(st[0]):
            n = 1024 if st is None else 8 * len(st[0])      # (the probe restarts the stream: ask for a longer prefix)
            out = np.zeros(n, np.float32)
            self.L.orc_unit_rng(C.c_uint32(self.cfg.seed), C.c_uint32(frame), C.c_uint32(pix), C.c_uint32(self.stream), n, self._fptr(out))
            st = self._cache[key] = [out, 0 if st is None else st[1]]
        st[1] += 1
        return st[0][st[1] - 1]

    # -- the session interface ------------------------------------------------------------------------------------------
    def upload_voxels(self, mat, rgb):
        self.r.world.voxel_material.a[...] = mat       # what the example scripts' kernels write through set_voxel (pathtracer.py:1325-1328)
        self.r.world.voxel_color.a[...] = rgb

    def upload_materials(self, table):
        ml = self.r.mats.mat_list
        names = ["subsurface", "metallic", "specular", "specular_tint", "roughness", "anisotropic", "sheen", "sheen_tint", "clearcoat",
                 "clearcoat_gloss", "ior_minus_one"]
        own = np.array([[*ml[i].base_col.to_list(), *[float(getattr(ml[i], n)) for n in names]] for i in range(128)], np.float32)
        assert np.array_equal(own, np.asarray(table, np.float32)), "the product's material table differs from the reference's MaterialList"

    def upload_cloud_texture(self, tex):
        """atmos.py:85-90 reads textures/cloud_texture.jpg through ti.tools.imread; the product ships the decoded array
        (voxel_rt2_amd/data/cloud_texture.npy).  They must be the same bytes."""
        a = self.r.atmos
        a.load_textures()
        assert np.array_equal(a.cloud_tex.to_numpy(), np.asarray(tex)), "data/cloud_texture.npy differs from the reference's texture as imread decodes it"

    def set_scene(self, s):
        r = self.r
        assert bool(s.use_physical_sky) == bool(self.cfg.sky_res)
        r.light_direction[None] = list(s.light_direction)       # set_directional_light (pathtracer.py:139-144) with the values the
        r.light_cone_cos_theta_max[None] = s.light_cos_theta_max  # boundary hands over (host.make_scene_params)
        r.light_color[None] = list(s.light_color)
        r.light_weight = float(s.light_weight)
        r.floor_height[None] = s.floor_height
        r.floor_color[None] = list(s.floor_color)
        r.floor_material[None] = s.floor_material
        r.background_color[None] = list(s.background_color)
        r.use_physical_atmosphere[None] = int(s.use_physical_sky)
        r.atmos.use_clouds[None] = int(s.use_clouds)

    def set_camera(self, cam):
        from voxel_rt2_amd import camera
        r = self.r
        view = np.array(cam.view, np.float32).reshape(4, 4)
        proj = np.array(cam.proj, np.float32).reshape(4, 4)
        r.set_camera_pos(*[float(x) for x in cam.pos])
        r.set_max_samples(float(cam.max_accum_frames))
        r.set_render_scale(float(cam.render_scale))
        r.set_camera_is_moving(int(cam.camera_is_moving))
        self.stream, self.jitter_index = 3, int(cam.jitter_index)
        self._cache.pop(("jitter", self.jitter_index), None)
        r.set_proj_mat(camera.to_glm_memory(proj))
        r.set_view_mat(camera.to_glm_memory(view))
        assert np.array_equal(self._m4(r.proj_mat_inv[None]), np.array(cam.proj_inv, np.float32).reshape(4, 4))
        assert np.array_equal(self._m4(r.view_mat_inv[None]), np.array(cam.view_inv, np.float32).reshape(4, 4))

    @staticmethod
    def _m4(m):
        return np.array([[float(m[i, j]) for j in range(4)] for i in range(4)], np.float32)

    def prepare(self):
        """prepare_data (pathtracer.py:314-323).  Its three 128^3 loops take minutes here: what they produce for one voxel grid is
        kept for the later cases of this run (and, for development, across runs in $REFEXEC_PREP_CACHE)."""
        import hashlib
        import pickle
        w = self.r.world
        key = hashlib.sha256(w.voxel_material.a.tobytes() + w.voxel_color.a.tobytes()).hexdigest()[:16]
        disk = os.environ.get("REFEXEC_PREP_CACHE")
        path = os.path.join(disk, key + ".pkl") if disk else None
        if key not in _PREPARED and path and os.path.exists(path):
            _PREPARED[key] = pickle.load(open(path, "rb"))
        if key not in _PREPARED:
            sky, self.r.use_physical_atmosphere[None] = self.r.use_physical_atmosphere[None], 0   # (the sky part runs below, with its stream)
            self.r.prepare_data()
            self.r.use_physical_atmosphere[None] = sky
            _PREPARED[key] = (w.voxel_color_texture.a.copy(), self.r.voxel_raytracer.occupancy.a.copy())
            if path:
                pickle.dump(_PREPARED[key], open(path, "wb"))
        w.voxel_color_texture.a, self.r.voxel_raytracer.occupancy.a = (x.copy() for x in _PREPARED[key])
        r = self.r
        if r.use_physical_atmosphere[None] == 1 and not getattr(self, "tables_given", False):     # the rest of prepare_data (pathtracer.py:317-323; the textures are loaded above)
            self.stream, self._frame = 2, 0x1000     # TAG_AMBIENT
            self._cache.clear()
            t = time.time()
            lut = os.path.join(disk, "trans_lut.npy") if disk else None
            if "lut" not in _PREPARED and lut and os.path.exists(lut):
                _PREPARED["lut"] = np.load(lut)
            if "lut" not in _PREPARED:
                r.atmos.generate_transmittance_lut()
                _PREPARED["lut"] = r.atmos.trans_LUT.a.copy()
                if lut:
                    np.save(lut, _PREPARED["lut"])
            r.atmos.trans_LUT.a = _PREPARED["lut"].copy()
            print(f"  transmittance LUT {time.time() - t:.0f} s", flush=True)
            r.atmos.compute_cloud_ambient(r.light_direction[None], r.light_color[None] * r.light_weight, r.light_cone_cos_theta_max[None])
            r.atmos.skybox_scattering.fill(self.ti.Vector([0., 0., 0.]))
            r.atmos.skybox_transmittance.fill(self.ti.Vector([0., 0., 0.]))

    def upload_sky(self, scat, trans):
        self.r.atmos.skybox_scattering.from_numpy(scat)
        self.r.atmos.skybox_transmittance.from_numpy(trans)

    def sky_accumulate_clouds(self, max_samples):
        self.stream, self._frame = 2, self._cloud_pass
        self._cache.clear()
        self.r.accumulate_clouds(max_samples)
        self._cloud_pass += 1

    def sky_compute_slice(self, slice_idx, max_slices):
        self.stream, self._frame = 2, 0x2000         # TAG_SKYBOX
        self._cache.clear()
        t = time.time()
        self.r.compute_atmosphere(slice_idx, max_slices)
        print(f"  sky slice {slice_idx + 1} / {max_slices} {time.time() - t:.0f} s", flush=True)

    def accumulate(self, n=1):
        """Renderer.accumulate (pathtracer.py:1310-1319), spelt out so that ti.random()'s stream follows the kernel."""
        r, pt = self.r, self.pt
        for _ in range(n):
            self._cache.clear()
            self.stream, self._frame = 0, r.current_frame
            r.render(r.world.voxel_color_texture)
            if pt.USE_RESTIR_PT:
                self.stream = 1
                r.current_frame = 0    # `self.current_frame` is a Python int: baked into the kernel at first compile (pathtracer.py:834)
                try:
                    r.spatial_GRIS(0, 24.0, 32, 1, r.world.voxel_color_texture)
                finally:
                    r.current_frame = self._frame
            r.temporal_filter_prepass()
            r.temporal_filter()
            r.temporal_filter_specular()
            r.current_spp += 1
            r.current_frame += 1

    def reset(self):
        self.r.reset_framebuffer()

    def end_frame(self):
        self.r.copy_prev_matrices()

    def fetch_hdr(self):
        return np.ascontiguousarray(self.r.color_buffer.to_numpy().transpose(1, 0, 2))

    def fetch_ldr(self):
        img = self.r.fetch_image()
        return np.ascontiguousarray(img.to_numpy().transpose(1, 0, 2))

    def fetch_buffer(self, which):
        from voxel_rt2_amd import _abi
        r = self.r
        t2 = lambda a: np.ascontiguousarray(a.transpose(1, 0) if a.ndim == 2 else a.transpose(1, 0, 2))  # noqa: E731
        if which == _abi.BUF_GBUF_DEPTH:
            return t2(r.gbuff_depth.to_numpy())[..., None]
        if which == _abi.BUF_GBUF_NORMAL:
            return t2(r.gbuff_normals.to_numpy()).view(np.uint16)
        if which == _abi.BUF_GBUF_POSITION:
            return t2(r.gbuff_position.to_numpy())
        if which == _abi.BUF_GBUF_MAT:
            return t2(r.gbuff_mat_id.to_numpy())[..., None]
        if which == _abi.BUF_GBUF_REFL_DEPTH:
            return t2(r.gbuff_depth_reflection.to_numpy())[..., None]
        if which == _abi.BUF_HISTORY_DIFFUSE:
            return np.ascontiguousarray(r.history_buffer.to_numpy()[:, :, 0, :].transpose(1, 0, 2))
        if which == _abi.BUF_HISTORY_SPECULAR:
            return np.ascontiguousarray(r.history_buffer_specular.to_numpy()[:, :, 0, :].transpose(1, 0, 2))
        if which == _abi.BUF_SKY_SCATTERING:
            return r.atmos.skybox_scattering.to_numpy()
        if which == _abi.BUF_SKY_TRANSMITTANCE:
            return r.atmos.skybox_transmittance.to_numpy()
        if which == _abi.BUF_TRANS_LUT:
            return r.atmos.trans_LUT.to_numpy().view(np.uint16)
        raise ValueError(which)

    def close(self):
        pass


def ray_vectors(path, n=600, seed=4):
    """VoxelOctreeRaytracer.raytrace (raytracer.py:72-155, the hot loop) on its own: rays from inside and outside the grid of the
    sun-lit scene, toward it and past it, axis-parallel ones (a zero direction component: the 0 * inf of :94, 133), grazing ones
    along cell faces, and origins exactly on cell boundaries."""
    import taichi as ti
    import make_golden
    from voxel_rt2_amd import scenes
    case = ("sunlit", 0, 16, 8, 2, 0, False, [])
    sess = ReferenceSession(make_golden.config_of(case))
    mat, rgb, _ = scenes.scene_sunlit(0)
    sess.upload_voxels(mat, rgb)
    sess.prepare()
    rt = sess.r.voxel_raytracer
    rng = np.random.default_rng(seed)
    org = rng.uniform(-24.0, 152.0, (n, 3)).astype(np.float32)
    org[: n // 3] = rng.uniform(0.0, 128.0, (n // 3, 3)).astype(np.float32)           # inside the grid
    solid = np.argwhere(mat > 0)
    tgt = (solid[rng.integers(0, len(solid), n)] + rng.uniform(-3.0, 4.0, (n, 3))).astype(np.float32)   # toward (or just past) solid voxels
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    for k in range(0, n, 7):
        d[k, rng.integers(0, 3)] = 0.0                                              # axis-parallel component
    for k in range(3, n, 11):
        org[k] = np.floor(org[k])                                                    # exactly on cell boundaries
    for k in range(5, n, 13):
        a = rng.integers(0, 3)
        d[k] = 0.0
        d[k, a] = rng.choice([-1.0, 1.0])                                            # along an axis
    out = dict(origin=org, direction=d, distance=np.zeros(n, np.float32), cell=np.zeros((n, 3), np.int32),
               normal=np.zeros((n, 3), np.float32), iters=np.zeros(n, np.int32))
    eps = 1e-6   # math_utils.py:5, as _trace_voxel passes it (pathtracer.py:201-202)
    for k in range(n):
        dist, cell, nor, iters = rt.raytrace(ti.Vector([x for x in org[k]]), ti.Vector([x for x in d[k]]), eps, np.inf)
        out["distance"][k], out["iters"][k] = dist, iters
        out["cell"][k], out["normal"][k] = cell.to_list(), nor.to_list()
    np.savez_compressed(path, **out)
    hit = np.isfinite(out["distance"])
    print(os.path.basename(path), f"{n} rays: {int(hit.sum())} hits, {int(np.isinf(out['distance']).sum())} misses, {int(np.isnan(out['distance']).sum())} NaN, "
          f"iterations {out['iters'].min()}..{out['iters'].max()}", flush=True)


MAT_FIELDS = ["subsurface", "metallic", "specular", "specular_tint", "roughness", "anisotropic", "sheen", "sheen_tint", "clearcoat",
              "clearcoat_gloss", "ior_minus_one"]


def function_vectors(path, n=300, seed=9):
    """Single functions of the reference on random arguments -- the frames above only meet the few materials their scenes use:
    DisneyBSDF.disney_evaluate_split / pdf_disney / pdf_disney_lobewise / sample_disney (bsdf.py:139-458) on random materials (all
    twelve parameters, with 0 / 1 extremes) and directions on both sides of the surface; sample_cone_oriented, the octahedral
    and material packing helpers, hash3, uchimura (math_utils.py), a Reservoir's encode -> decode (reservoir.py:96-141) and
    Renderer.shift (pathtracer.py:672-812) on random samples.  tests/test_reference_vectors.py feeds the same arguments to the
    oracle's orc_unit_* probes."""
    import taichi as ti
    import make_golden
    from taichi.math import vec3
    from voxel_rt2_amd import scenes
    import renderer.math_utils as mu
    import renderer.reservoir as rs
    case = ("sunlit", 0, 16, 8, 2, 0, False, [])
    sess = ReferenceSession(make_golden.config_of(case))
    mat_s, rgb_s, params = scenes.scene_sunlit(0)
    import orc
    orc.setup(sess, mat_s, rgb_s, params)    # scene parameters, camera, (cached) grid: what shift() reads
    r, bsdf = sess.r, sess.r.bsdf
    rng = np.random.default_rng(seed)
    L, fptr = sess.L, sess._fptr

    def unit(k):
        v = rng.normal(size=(k, 3))
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    def stream(i, count, sd):
        out = np.zeros(count, np.float32)
        L.orc_unit_rng(C.c_uint32(sd), C.c_uint32(0), C.c_uint32(i), C.c_uint32(0), count, fptr(out))
        return out

    V = lambda a: ti.Vector([x for x in a])  # noqa: E731
    mats = rng.uniform(0.0, 1.0, (n, 14)).astype(np.float32)
    ext = rng.random((n, 14)) < 0.15
    mats[ext] = rng.integers(0, 2, int(ext.sum())).astype(np.float32)
    nrm = unit(n)
    view = unit(n)
    flip = (view * nrm).sum(1) < 0
    view[flip & (rng.random(n) < 0.9)] *= -1.0          # mostly the surface's own side
    lgt = unit(n)
    lobes = rng.integers(0, 3, n).astype(np.int32)

    def material(row):
        return bsdf.disney_material(base_col=vec3(row[0], row[1], row[2]), **{k: row[3 + j] for j, k in enumerate(MAT_FIELDS)})

    out = dict(mat=mats, n=nrm, v=view, l=lgt, lobe=lobes, eval=np.zeros((n, 7), np.float32), lobe_pdf=np.zeros(n, np.float32),
               sample=np.zeros((n, 4, 8), np.float32), sample_seed=np.uint32(77))   # (material k draws from seed 77 + k)
    for k in range(n):
        m = material(mats[k])
        t, b = mu.make_orthonormal_basis(V(nrm[k]))
        d, sp = bsdf.disney_evaluate_split(m, V(view[k]), V(nrm[k]), V(lgt[k]), t, b)
        out["eval"][k] = d.to_list() + sp.to_list() + [bsdf.pdf_disney(m, V(view[k]), V(nrm[k]), V(lgt[k]), t, b)]
        out["lobe_pdf"][k] = bsdf.pdf_disney_lobewise(m, V(view[k]), V(nrm[k]), V(lgt[k]), t, b, int(lobes[k]))
        for i in range(4):
            ti.set_random_source(stream(i, 16, 77 + k))
            sd, brdf, pdf, lobe = bsdf.sample_disney(m, V(view[k]), V(nrm[k]), t, b)
            out["sample"][k, i] = sd.to_list() + brdf.to_list() + [pdf, lobe]
    # cone sampling
    out["cone_cos"] = rng.uniform(0.5, 0.99999, 40).astype(np.float32)
    out["cone_n"] = unit(40)
    out["cone"] = np.zeros((40, 3, 3), np.float32)
    for k in range(40):
        for i in range(3):
            ti.set_random_source(stream(i, 4, 4))
            out["cone"][k, i] = mu.sample_cone_oriented(out["cone_cos"][k], V(out["cone_n"][k])).to_list()
    # packing helpers
    vecs = np.concatenate([unit(200), np.eye(3, dtype=np.float32), -np.eye(3, dtype=np.float32), np.zeros((1, 3), np.float32)])
    out["oct_in"] = vecs
    out["oct_code"] = np.zeros((len(vecs), 2), np.uint16)
    out["oct_out"] = np.zeros((len(vecs), 3), np.float32)
    for k, v in enumerate(vecs):
        code = mu.encode_unit_vector_3x16(V(v))
        out["oct_code"][k] = np.array(code.to_list(), np.float16).view(np.uint16)
        out["oct_out"][k] = mu.decode_unit_vector_3x16(code).to_list()
    out["matenc_id"] = rng.integers(0, 128, 100).astype(np.int32)
    out["matenc_albedo"] = rng.uniform(0, 1, (100, 3)).astype(np.float32)
    out["matenc"] = np.array([int(mu.encode_material(int(i), V(a))) for i, a in zip(out["matenc_id"], out["matenc_albedo"])], np.uint32)
    out["hash_in"] = rng.integers(0, 2**32, (200, 3), dtype=np.uint64).astype(np.uint32)
    out["hash_out"] = np.array([int(mu.hash3(*[np.uint32(x) for x in row])) for row in out["hash_in"]], np.uint32)
    out["uchimura_in"] = np.concatenate([rng.uniform(0, 4, 300), [0.0, 1e-6, 0.22, 0.532, 1.0, 100.0]]).astype(np.float32)
    out["uchimura_out"] = np.array([float(mu.uchimura(V([x, x, x])).x) for x in out["uchimura_in"]], np.float32)
    # reservoir encode -> decode: 23 floats in (Sample + M, weight), 23 out (orc_unit_reservoir_roundtrip's layout)
    K = 120
    smp = np.zeros((K, 23), np.float32)
    smp[:, 0:3] = rng.uniform(0, 5, (K, 3)); smp[:, 3:6] = rng.uniform(-1, 1, (K, 3)); smp[:, 6:9] = unit(K); smp[:, 9:12] = unit(K)
    smp[:, 12:15] = rng.uniform(0, 9, (K, 3)); smp[:, 15:18] = unit(K)
    smp[::7, 6:9] = 0.0; smp[::5, 9:12] = 0.0; smp[::3, 15:18] = 0.0       # escape vertex / last vertex / sun not visible
    info = rng.integers(0, 2**32, K, dtype=np.uint64).astype(np.uint32)
    smp[:, 18] = info.view(np.float32); smp[:, 19] = rng.uniform(0, 30, K); smp[:, 20] = rng.integers(0, 3, K) * 10 + rng.integers(0, 3, K)
    smp[:, 21] = rng.integers(1, 40, K); smp[:, 22] = rng.uniform(0, 60, K)
    out["res_in"], out["res_out"] = smp, np.zeros((K, 23), np.float32)

    def reservoir(row):
        q = rs.Reservoir()
        q.init()
        q.z.F, q.z.rc_pos, q.z.rc_normal, q.z.rc_incident_dir = V(row[0:3]), V(row[3:6]), V(row[6:9]), V(row[9:12])
        q.z.rc_incident_L, q.z.rc_NEE_dir = V(row[12:15]), V(row[15:18])
        q.z.rc_mat_info, q.z.cached_jacobian_term, q.z.lobes = row[18:19].view(np.uint32)[0], row[19], int(row[20])
        q.M, q.weight = row[21], row[22]
        return q
    for k in range(K):
        q = rs.Reservoir()
        q.init()
        q.decode(reservoir(smp[k]).encode())
        o = q.z.F.to_list() + q.z.rc_pos.to_list() + q.z.rc_normal.to_list() + q.z.rc_incident_dir.to_list() + q.z.rc_incident_L.to_list() + q.z.rc_NEE_dir.to_list()
        out["res_out"][k, :18] = o
        out["res_out"][k, 18] = np.array([q.z.rc_mat_info], np.uint32).view(np.float32)[0]
        out["res_out"][k, 19:] = [q.z.cached_jacobian_term, q.z.lobes, q.M, q.weight]
    # shift(): destination / source primary vertices near each other, samples of every kind
    S = 150
    dst_pos = rng.uniform(-0.6, 0.6, (S, 3)).astype(np.float32)
    src_pos = (dst_pos + rng.uniform(-0.05, 0.05, (S, 3))).astype(np.float32)
    dst_n = unit(S)
    dst_mat = rng.uniform(0, 1, (S, 14)).astype(np.float32)
    sm = np.zeros((S, 21), np.float32)
    sm[:, 0:3] = rng.uniform(0, 3, (S, 3)); sm[:, 3:6] = dst_pos + unit(S) * rng.uniform(0.1, 1.5, (S, 1)).astype(np.float32)
    sm[:, 6:9] = unit(S); sm[:, 9:12] = unit(S); sm[:, 12:15] = rng.uniform(0, 4, (S, 3)); sm[:, 15:18] = unit(S)
    esc = np.arange(S) % 4 == 0
    sm[esc, 6:9] = 0.0; sm[esc, 3:6] = unit(int(esc.sum()))               # escape vertex: rc_pos is a direction
    sm[np.arange(S) % 5 == 1, 9:12] = 0.0; sm[np.arange(S) % 3 == 2, 15:18] = 0.0
    ids = rng.choice([1, 2, 10, 11, 20, 21, 30, 40, 50, 80], S).astype(np.uint32)
    alb = rng.integers(0, 256, (S, 3)).astype(np.uint32)
    sm[:, 18] = (ids | (alb[:, 0] << 8) | (alb[:, 1] << 16) | (alb[:, 2] << 24)).astype(np.uint32).view(np.float32)
    sm[:, 19] = rng.uniform(0.01, 20, S); sm[:, 20] = rng.integers(0, 3, S) * 10 + rng.integers(0, 3, S)
    out.update(shift_dst_pos=dst_pos, shift_dst_n=dst_n, shift_dst_mat=dst_mat, shift_src_pos=src_pos, shift_sample=sm, shift_out=np.zeros((S, 7), np.float32))
    for k in range(S):
        row = np.concatenate([sm[k], [1.0, 1.0]]).astype(np.float32)
        q = reservoir(np.concatenate([row[:21], [1.0, 1.0]]).astype(np.float32))
        d, sp, jac = r.shift(V(dst_pos[k]), V(dst_n[k]), material(dst_mat[k]), V(src_pos[k]), V(dst_n[k]), material(dst_mat[k]), q)
        out["shift_out"][k] = d.to_list() + sp.to_list() + [jac]
    # set_directional_light (pathtracer.py:139-144) runs in PYTHON scope: the direction is normalised in double and rounded by the store
    # into the f32 field, the cone's cosine likewise -- what host.make_scene_params hands the library
    ldirs = np.concatenate([rng.uniform(-3.0, 3.0, (200, 3)), [[1, 1, 1], [1, 1, -1], [0, 1, 0], [1e-3, 2, -5e2]]]).astype(np.float64)
    cones = np.concatenate([rng.uniform(0.0, 0.5, 200), [0.1, 0.025, 0.0, 3.0]]).astype(np.float64)
    out["light_in"], out["light_cone"] = ldirs, cones
    out["light_dir"], out["light_cos"] = np.zeros((len(ldirs), 3), np.float32), np.zeros(len(ldirs), np.float32)
    for k in range(len(ldirs)):
        r.set_directional_light(tuple(float(x) for x in ldirs[k]), float(cones[k]), (1.0, 1.0, 1.0))
        out["light_dir"][k] = r.light_direction[None].to_list()
        out["light_cos"][k] = r.light_cone_cos_theta_max[None]
    # the skybox parameterisation (atmos.py:428-455) at the table size of the sky-lookup case
    import make_golden as mg2
    sky_case = ("s6", 0, 16, 8, 2, 0, False, [], ("given", 64, 5))
    sky = ReferenceSession(mg2.config_of(sky_case))
    dirs = np.concatenate([unit(300), np.eye(3, dtype=np.float32), -np.eye(3, dtype=np.float32)])
    out["sky_dir"] = dirs
    out["sky_uv"] = np.array([sky.r.atmos.project_sky(V(d)).to_list() for d in dirs], np.float32)
    uv = rng.uniform(0.0, 1.0, (300, 2)).astype(np.float32)
    out["sky_uv_in"] = uv
    out["sky_dir_out"] = np.array([sky.r.atmos.unproject_sky(V(t)).to_list() for t in uv], np.float32)
    # voxel authoring as the example scripts do it: the reference's OWN Scene.set_voxel -> Scene.round_idx (scene.py:131-141; the class is
    # imported from the reference's scene.py, an instance made without its window-opening constructor) -> Renderer.set_voxel
    # (pathtracer.py:1325-1328, math_utils.py:86-92) -- ties and near-ties of the rounding, float material ids (example8.py:7), colours
    # outside [0, 1] and on exact byte fractions
    global _author_scene
    _author_scene = object.__new__(_reference_scene_class())
    _author_scene.renderer = sky.r
    A = 1500
    aidx = rng.uniform(-63.4, 62.4, (A, 3))
    aidx[::5] = np.round(aidx[::5]) + rng.choice([0.5, -0.5, 0.49999997, 0.0], (len(aidx[::5]), 3))
    aidx = np.clip(aidx, -63.4, 62.4).astype(np.float32)
    amat = np.where(np.arange(A) % 3 != 0, rng.integers(0, 100, A).astype(np.float64), rng.uniform(0, 99, A)).astype(np.float32)
    acol = rng.uniform(-0.1, 1.1, (A, 3))
    acol[::7] = np.round(acol[::7] * 255) / 255
    acol = acol.astype(np.float32)
    for k in range(A):
        _author(*[float(x) for x in aidx[k]], float(amat[k]), *[float(c) for c in acol[k]])
    w = sky.r.world
    cells = np.argwhere(w.voxel_material.a != 0)
    out.update(author_idx=aidx, author_mat=amat, author_color=acol, author_cells=cells.astype(np.int16),
               author_cell_mat=w.voxel_material.a[tuple(cells.T)], author_cell_rgb=w.voxel_color.a[tuple(cells.T)])
    np.savez_compressed(path, **out)
    print(os.path.basename(path), {k: v.shape for k, v in out.items() if hasattr(v, "shape") and v.shape}, flush=True)


SKY_OPS = {"rsi": (0, 7, 2), "ozone": (1, 1, 1), "density": (2, 1, 3), "cloud_phase": (3, 2, 1), "cloud_density": (4, 3, 1),
           "cloud_shadow_od": (5, 7, 1), "ray_transmittance": (6, 6, 3), "clouds_scattering": (7, 15, 5),
           "atmos_scattering_d0": (8, 15, 6), "atmos_scattering_d1": (9, 15, 6)}    # name -> (probe op, floats in, floats out)
SKY_SEED = 31


def sky_function_vectors(path, seed=12):
    """atmos.py one function at a time on random arguments (functions_sky.npz): rsi, get_ozone_density, get_density, cloud_phase,
    sample_cloud_density, clouds_shadow_od, get_ray_transmittance, clouds_scattering and atmospheric_scattering at both template
    depths (atmos.py:9-15, 195-349, 355-425, 475-523) -- the whole-table case ref_s6_sky_precompute only meets 64 texels' worth of
    arguments.  What the functions read besides their arguments is GIVEN: a synthetic transmittance LUT (so that the hour the
    real one takes is not part of this) and a cloud ambient colour, stored in the file; the cloud tile is the reference's own
    texture.  Random numbers (the cone sampler of the two scattering functions): stream 2 of (SKY_SEED, frame 0x3000, row)."""
    import taichi as ti
    import orc
    os.chdir(REFERENCE)
    import renderer.atmos as at
    from taichi.math import vec3
    L = orc.lib()

    def dm(op):
        def f(a, b=0.0):
            if op == 4 and b == 1.5:
                return np.float32(a) * np.sqrt(np.float32(a))
            x, y, out = np.array([a], np.float32), np.array([b], np.float32), np.zeros(1, np.float32)
            L.orc_unit_detmath(op, 1, orc.fptr(x), orc.fptr(y), orc.fptr(out))
            return out[0]
        return f
    ti.set_elementary(**{n: dm(op) for op, n in enumerate(["sin", "cos", "exp", "log", "pow", "acos", "atan2"])})
    a = at.Atmos()
    a.load_textures()
    rng = np.random.default_rng(seed)
    R = 6371e3
    # a smooth positive table in [0, 1] with texel-to-texel variation: every read_trans_lut index matters
    lut = (0.5 + 0.5 * np.sin(np.arange(256)[:, None, None] * 0.11 + np.arange(128)[None, :, None] * 0.23 + np.arange(3)[None, None, :] * 1.7)) \
        * rng.uniform(0.6, 1.0, (256, 128, 3))
    lut = lut.astype(np.float16)
    a.trans_LUT.a[...] = lut
    ambient = np.array([0.21, 0.34, 0.55], np.float32)
    a.cloud_ambient[None] = vec3(*[float(x) for x in ambient])
    V = lambda x: ti.Vector([np.float32(v) for v in x])  # noqa: E731

    def unit(k, up=None):
        v = rng.normal(size=(k, 3))
        if up is not None:
            v[:, 1] = np.abs(v[:, 1]) * up + (1 - up) * v[:, 1]
        return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)

    def stream(row):
        out = np.zeros(131072, np.float32)     # (a 64-step depth-0 march draws 42 000 numbers)
        L.orc_unit_rng(C.c_uint32(SKY_SEED), C.c_uint32(0x3000), C.c_uint32(row), C.c_uint32(2), len(out), orc.fptr(out))
        return out

    out = dict(trans_lut=lut.view(np.uint16), cloud_ambient=ambient, seed=np.uint32(SKY_SEED))
    t0 = time.time()
    # rsi
    n = 160
    pos = np.stack([rng.uniform(-5e4, 5e4, n), R + rng.uniform(0, 1.2e5, n), rng.uniform(-5e4, 5e4, n)], 1).astype(np.float32)
    rad = rng.choice([R, R + 2000.0, R + 2340.0, R + 110e3], n).astype(np.float32)
    arg = np.concatenate([pos, unit(n), rad[:, None]], 1).astype(np.float32)
    res = np.zeros((n, 2), np.float32)
    for k in range(n):
        res[k] = at.rsi(V(arg[k, 0:3]), V(arg[k, 3:6]), np.float32(arg[k, 6])).to_list()
    out["rsi_in"], out["rsi_out"] = arg, res
    # densities
    h = np.concatenate([rng.uniform(-2e3, 1.3e5, 150), [0.0, 15e3, 25e3, -1.0]]).astype(np.float32)
    out["ozone_in"], out["density_in"] = h[:, None].copy(), h[:, None].copy()
    out["ozone_out"] = np.array([[a.get_ozone_density(np.float32(x))] for x in h], np.float32)
    out["density_out"] = np.array([a.get_density(np.float32(x)).to_list() for x in h], np.float32)
    # cloud phase
    arg = np.stack([rng.uniform(-1, 1, 160), rng.choice([1.0, 0.5, 0.25, 0.125], 160)], 1).astype(np.float32)
    out["cloud_phase_in"] = arg
    out["cloud_phase_out"] = np.array([[a.cloud_phase(np.float32(x), np.float32(y))] for x, y in arg], np.float32)
    # cloud density: points in and around the layer (relative height 1 900 .. 2 450 m), over several tiles -- half of them above
    # texels that hold a cloud (a tenth of the tile does), at a height inside the layer
    tex = a.cloud_tex.to_numpy().astype(np.float32) / 255.0
    cloudy = np.argwhere((tex[..., 2] >= 0.7) & ((tex[..., 0] >= 0.7) | (tex[..., 1] >= 0.7)))

    def over_clouds(k, lo, hi):
        c = cloudy[rng.integers(0, len(cloudy), k)]
        tile = 29000.0
        x = (c[:, 0] + rng.uniform(0.1, 0.9, k)) / 256.0 * tile - 0.65 * tile + rng.integers(-1, 2, k) * tile
        z = (c[:, 1] + rng.uniform(0.1, 0.9, k)) / 256.0 * tile - 0.65 * tile + rng.integers(-1, 2, k) * tile
        hgt = rng.uniform(lo, hi, k)
        return np.stack([x, np.sqrt((R + hgt) ** 2 - x * x - z * z), z], 1)
    n = 200
    pos = np.stack([rng.uniform(-6e4, 6e4, n), R + rng.uniform(1900.0, 2450.0, n), rng.uniform(-6e4, 6e4, n)], 1)
    pos[: n // 2] = over_clouds(n // 2, 1950.0, 2400.0)
    pos = pos.astype(np.float32)
    out["cloud_density_in"] = pos
    out["cloud_density_out"] = np.array([[a.sample_cloud_density(V(p))] for p in pos], np.float32)
    # shadow optical depth: from inside the layer
    n = 150
    pos = np.stack([rng.uniform(-4e4, 4e4, n), R + rng.uniform(2000.0, 2340.0, n), rng.uniform(-4e4, 4e4, n)], 1)
    pos[: 2 * n // 3] = over_clouds(2 * n // 3, 2005.0, 2330.0)
    pos = pos.astype(np.float32)
    arg = np.concatenate([pos, unit(n, up=0.8), rng.random((n, 1))], 1).astype(np.float32)
    out["cloud_shadow_od_in"] = arg
    out["cloud_shadow_od_out"] = np.array([[a.clouds_shadow_od(V(r[0:3]), V(r[3:6]), np.float32(r[6]))] for r in arg], np.float32)
    print(f"  small functions {time.time() - t0:.0f} s", flush=True)
    # ray transmittance: the LUT's own rays and others
    n = 100
    th = np.arccos(rng.uniform(-1, 1, n))
    pos = np.stack([np.zeros(n), R + rng.uniform(0, 1.05e5, n), np.zeros(n)], 1).astype(np.float32)
    dirs = np.stack([np.sin(th), np.cos(th), np.zeros(n)], 1).astype(np.float32)
    dirs[n // 2:] = unit(n - n // 2)
    arg = np.concatenate([pos, dirs], 1).astype(np.float32)
    out["ray_transmittance_in"] = arg
    out["ray_transmittance_out"] = np.array([a.get_ray_transmittance(V(r[0:3]), V(r[3:6])).to_list() for r in arg], np.float32)
    print(f"  ray transmittance {time.time() - t0:.0f} s", flush=True)

    def sun_rows(n):
        d = unit(n, up=0.9)
        sun = unit(n, up=0.7)
        col = rng.uniform(0.5, 4.0, (n, 3))
        cosm = np.cos(rng.choice([0.0125, 0.05, 0.2], n))
        return d, np.concatenate([sun, col, cosm[:, None], np.zeros((n, 1)), np.arange(n)[:, None]], 1)
    # clouds_scattering from the camera position (atmos.py:149)
    n = 48
    d, rest = sun_rows(n)
    rest[:, 7] = rng.random(n)      # (this function's 14th argument is the dither, not a step count)
    arg = np.concatenate([np.tile(np.array([[0.0, R + 1e3, 0.0]]), (n, 1)), d, rest], 1).astype(np.float32)
    res = np.zeros((n, 5), np.float32)
    for k in range(n):
        ti.set_random_source(stream(k))
        r = arg[k]
        sc, tr, dist = a.clouds_scattering(V(r[0:3]), V(r[3:6]), V(r[6:9]), V(r[9:12]), np.float32(r[12]), np.float32(r[13]))
        res[k] = sc.to_list() + [tr, dist]
    out["clouds_scattering_in"], out["clouds_scattering_out"] = arg, res
    print(f"  clouds_scattering {time.time() - t0:.0f} s", flush=True)
    # atmospheric_scattering: template depth 1 with its 5 steps (atmos.py:409) and other counts, depth 0 with few steps and with the real 64
    for name, depth, counts in (("atmos_scattering_d1", 1, [5] * 60 + [3] * 10 + [9] * 10), ("atmos_scattering_d0", 0, [2] * 8 + [5] * 4 + [64] * 2)):
        n = len(counts)
        d, rest = sun_rows(n)
        rest[:, 7] = counts
        pos = np.stack([rng.uniform(-3e4, 3e4, n), R + rng.uniform(500.0, 6e4, n), rng.uniform(-3e4, 3e4, n)], 1)
        pos[: n // 3] = [0.0, R + 1e3, 0.0]
        arg = np.concatenate([pos, d, rest], 1).astype(np.float32)
        res = np.zeros((n, 6), np.float32)
        for k in range(n):
            ti.set_random_source(stream(k))
            r = arg[k]
            sc, tr = a.atmospheric_scattering(V(r[0:3]), V(r[3:6]), V(r[6:9]), V(r[9:12]), np.float32(r[12]), depth, int(r[13]))
            res[k] = sc.to_list() + tr.to_list()
            if depth == 0:
                print(f"    depth 0, {int(r[13])} steps: {time.time() - t0:.0f} s", flush=True)
        out[name + "_in"], out[name + "_out"] = arg, res
    np.savez_compressed(path, **out)
    print("functions_sky", {k: v.shape for k, v in out.items() if hasattr(v, "shape")}, os.path.getsize(path), "bytes", f"{time.time() - t0:.0f} s", flush=True)


_author_scene = None


def _reference_scene_class():
    """`Scene` of /root/reference/scene.py, imported from where it lies.  (scene.py:23 reaches into Taichi for a window-system handle
    that only its interactive loop uses: an empty module stands in for that import.)"""
    import importlib.util
    import types
    for name in ("taichi.lang", "taichi.lang.impl"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["taichi.lang.impl"]._ti_core = None
    spec = importlib.util.spec_from_file_location("reference_scene", os.path.join(REFERENCE, "scene.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Scene


@taichi.kernel
def _author(i: taichi.f32, j: taichi.f32, k: taichi.f32, mat: taichi.f32, cr: taichi.f32, cg: taichi.f32, cb: taichi.f32):
    _author_scene.set_voxel(taichi.Vector([i, j, k]), mat, taichi.Vector([cr, cg, cb]))


def config_of(case):
    import make_golden
    return make_golden.config_of(case)


def check_golden(names):
    """tests/golden/*.npz are written by the ORACLE (make_golden.py).  This runs the reference's own source on those same cases,
    where their frame fits its blocked layout, and reports whether it reproduces them -- the larger frames, fused launches and
    longer scripts of that set, checked once in the build container (log: tests/golden/reference/CHECKED.txt)."""
    import make_golden
    for name in names or list(make_golden.CASES):
        case = make_golden.CASES[name]
        W, H = case[2], case[3]
        if W % 16 or H % 8:
            print(f"{name}: {W}x{H} does not fit the reference's 16x8 blocks (pathtracer.py:74) -- not runnable", flush=True)
            continue
        t = time.time()
        sess = ReferenceSession(make_golden.config_of(case))
        got = make_golden.run_case(sess, case)
        want = np.load(os.path.join(HERE, name + ".npz"))
        bad = {}
        for key in want.files:
            a, b = np.ascontiguousarray(got[key]), want[key]
            same = (a.view(np.uint8) == b.view(np.uint8)).reshape(H, W, -1).all(-1)
            if not same.all():
                bad[key] = np.argwhere(~same).tolist()
        undefined = np.argwhere(sess.undefined_px).tolist()
        verdict = "every buffer equal, bit for bit" if not bad else f"DIFFERS: { {k: v[:6] for k, v in bad.items()} }"
        print(f"{name}: {W}x{H}, {sum(s[1] for s in case[7] if s[0] == 'accumulate')} passes: {verdict}; pixels with a level-0 hit outside the grid: {undefined} "
              f"({time.time() - t:.0f} s)", flush=True)


def spot_check(rows=25):
    """The first rows of functions.npz worked out again from the reference's source (seconds: no voxel grid involved)."""
    import taichi as ti
    import orc
    os.chdir(REFERENCE)
    from renderer.bsdf import DisneyBSDF
    import renderer.math_utils as mu
    from taichi.math import vec3
    L = orc.lib()

    def dm(op):
        def f(a, b=0.0):
            if op == 4 and b == 1.5:
                return np.float32(a) * np.sqrt(np.float32(a))
            x, y, out = np.array([a], np.float32), np.array([b], np.float32), np.zeros(1, np.float32)
            L.orc_unit_detmath(op, 1, orc.fptr(x), orc.fptr(y), orc.fptr(out))
            return out[0]
        return f
    ti.set_elementary(**{n: dm(op) for op, n in enumerate(["sin", "cos", "exp", "log", "pow", "acos", "atan2"])})
    v = np.load(os.path.join(OUT, "functions.npz"))
    bsdf = DisneyBSDF()
    V = lambda a: ti.Vector([x for x in a])  # noqa: E731
    bad = 0
    for k in range(rows):
        row = v["mat"][k]
        m = bsdf.disney_material(base_col=vec3(row[0], row[1], row[2]), **{n: row[3 + j] for j, n in enumerate(MAT_FIELDS)})
        t, b = mu.make_orthonormal_basis(V(v["n"][k]))
        d, sp = bsdf.disney_evaluate_split(m, V(v["v"][k]), V(v["n"][k]), V(v["l"][k]), t, b)
        got = np.array(d.to_list() + sp.to_list() + [bsdf.pdf_disney(m, V(v["v"][k]), V(v["n"][k]), V(v["l"][k]), t, b)], np.float32)
        lp = np.float32(bsdf.pdf_disney_lobewise(m, V(v["v"][k]), V(v["n"][k]), V(v["l"][k]), t, b, int(v["lobe"][k])))
        same = lambda a, c: bool(np.all((np.asarray(a, np.float32).view(np.uint32) == np.asarray(c, np.float32).view(np.uint32)) | (np.isnan(a) & np.isnan(c))))  # noqa: E731
        bad += not same(got, v["eval"][k]) or not same(np.array([lp]), np.array([v["lobe_pdf"][k]]))
        bad += int(mu.hash3(*[np.uint32(x) for x in v["hash_in"][k]])) != int(v["hash_out"][k])
    print("spot check:", "ok" if bad == 0 else f"{bad} rows differ", f"({rows} rows of functions.npz)")
    return bad


def main(argv):
    import make_golden
    if "--spot-check" in argv:
        sys.exit(1 if spot_check() else 0)
    if "--check-golden" in argv:
        return check_golden([a for a in argv if not a.startswith("--")])
    libm = "--libm" in argv
    names = [a for a in argv if not a.startswith("--")] or list(CASES) + ["rays", "functions", "functions_sky"]
    os.makedirs(OUT, exist_ok=True)
    if "rays" in names:
        names.remove("rays")
        if not os.path.exists(os.path.join(OUT, "rays_sunlit.npz")) or "--force" in argv:
            ray_vectors(os.path.join(OUT, "rays_sunlit.npz"))
    if "functions_sky" in names:
        names.remove("functions_sky")
        if not os.path.exists(os.path.join(OUT, "functions_sky.npz")) or "--force" in argv:
            sky_function_vectors(os.path.join(OUT, "functions_sky.npz"))
    if "functions" in names:
        names.remove("functions")
        if "--verify" in argv:      # into a scratch file, compared array by array with the committed one
            import tempfile
            tmp = os.path.join(tempfile.mkdtemp(), "functions.npz")
            function_vectors(tmp)
            a, b = np.load(tmp), np.load(os.path.join(OUT, "functions.npz"))
            same = sorted(a.files) == sorted(b.files) and all(np.array_equal(a[k].view(np.uint8) if a[k].dtype.kind == "f" else a[k], b[k].view(np.uint8) if b[k].dtype.kind == "f" else b[k]) for k in b.files)
            print(f"functions: the committed fixture is {'reproduced' if same else 'NOT reproduced'}", flush=True)
        elif not os.path.exists(os.path.join(OUT, "functions.npz")) or "--force" in argv:
            function_vectors(os.path.join(OUT, "functions.npz"))
    for name in names:
        case = CASES[name]
        if os.path.exists(os.path.join(OUT, name + ".npz")) and not libm and "--force" not in argv and "--verify" not in argv:
            print(name, "exists (--force rewrites it)")
            continue
        t = time.time()
        sess = ReferenceSession(make_golden.config_of(case), libm=libm)
        sess.tables_given = len(case) > 8 and case[8][0] == "given"    # (then the LUT and the cloud ambient feed nothing)
        out = make_golden.run_case(sess, case)
        if sess.undefined_px.any():
            out["undefined_px"] = sess.undefined_px
            print(f"  {int(sess.undefined_px.sum())} pixels hit a 'voxel' outside the grid (a set level-0 bit read through an index that left the grid): {np.argwhere(sess.undefined_px).tolist()}")
        path = os.path.join(OUT, name + ".npz")
        if "--verify" in argv:     # run again, compare with the committed file, write nothing
            want = np.load(path)
            same = sorted(want.files) == sorted(out) and all(np.array_equal(np.ascontiguousarray(out[k]).view(np.uint8), want[k].view(np.uint8)) for k in want.files)
            print(f"{name}: the committed fixture is {'reproduced' if same else 'NOT reproduced'} ({time.time() - t:.0f} s)", flush=True)
            continue
        if libm:
            want = np.load(path)
            for key in ("hdr", "ldr"):
                a, b = out[key].astype(np.float64), want[key].astype(np.float64)
                d = np.abs(a - b)
                print(f"{name} [{key}] numpy's elementary functions instead of the contract's: {int((d > 0).sum())} of {d.size} values differ, "
                      f"max |diff| {d.max():.3g}, max relative {np.nanmax(d / np.maximum(np.abs(b), 1e-6)):.3g}")
            continue
        np.savez_compressed(path, **out)
        print(name, {k: (v.shape, str(v.dtype)) for k, v in out.items()}, os.path.getsize(path), "bytes", f"{time.time() - t:.0f} s", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
