"""Thin ctypes session over a library exporting the include/vrt_api.h entry points.

In the product the only caller is renderer.Renderer, which passes libvrt_hip.so (loaded by
_lib.py, which raises if the HIP library is missing -- there is no CPU fallback).  The parity
tests reuse this class for the test oracle, whose C entry points have the same shapes.
"""
import ctypes as C
import numpy as np

from . import _abi


class NativeError(RuntimeError):
    pass


class NativeSession:
    def __init__(self, lib, prefix, cfg, create_extra=()):
        self._lib, self._p = lib, prefix
        _abi.declare(lib, prefix)
        create = getattr(lib, prefix + "create")
        create.restype = C.c_void_p
        self.cfg = cfg
        self.W, self.H = cfg.width, cfg.height
        self.rows = (cfg.row_begin, cfg.row_end) if cfg.row_end > cfg.row_begin else (0, cfg.height)
        self._ctx = create(C.byref(cfg), *create_extra)
        if not self._ctx:
            raise NativeError(f"{prefix}create failed: {self._err()}")

    def _err(self):
        fn = getattr(self._lib, self._p + "last_error", None)
        if fn is None:
            return "(no message)"
        msg = fn()
        return msg.decode() if msg else "(no message)"

    def _call(self, name, *args):
        if not self._ctx:
            raise NativeError("session is closed")
        rc = getattr(self._lib, self._p + name)(C.c_void_p(self._ctx), *args)
        if rc != 0:
            raise NativeError(f"{self._p}{name} failed ({rc}): {self._err()}")

    def close(self):
        """Ends the session.  Page-locked arrays from host_alloc() die with it (see there): outstanding asynchronous fetches are
        waited for first, so that no copy is still writing into memory that is being freed."""
        if self._ctx:
            pinned = getattr(self, "_pinned", [])
            if pinned:
                wait = getattr(self._lib, self._p + "fetch_wait", None)
                if wait is not None:
                    for slot in range(4):       # VRT_FETCH_SLOTS; a slot without a fetch returns at once
                        wait(C.c_void_p(self._ctx), slot)
                for ptr, _ in pinned:
                    getattr(self._lib, self._p + "host_free")(C.c_void_p(self._ctx), C.c_void_p(ptr))
            self._pinned = []
            getattr(self._lib, self._p + "destroy")(C.c_void_p(self._ctx))
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- uploads ---------------------------------------------------------------------------
    def upload_voxels(self, mat, rgb):
        mat = np.ascontiguousarray(mat, dtype=np.int8)
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        g = int(self.cfg.grid_res)
        if mat.shape != (g, g, g) or rgb.shape != (g, g, g, 3):
            raise ValueError(f"voxel arrays must be int8[{g},{g},{g}] and uint8[{g},{g},{g},3] (grid_res = {g})")
        self._call("upload_voxels", mat.ctypes.data_as(C.c_void_p), rgb.ctypes.data_as(C.c_void_p))

    def upload_materials(self, table):
        table = np.ascontiguousarray(table, dtype=np.float32)
        if table.shape != (128, 14):
            raise ValueError("material table must be float32[128,14]")
        self._call("upload_materials", table.ctypes.data_as(C.c_void_p))

    def upload_cloud_texture(self, tex):
        tex = np.ascontiguousarray(tex, dtype=np.uint8)
        if tex.shape != (256, 256, 3):
            raise ValueError("cloud texture must be uint8[256,256,3]")
        self._call("upload_cloud_texture", tex.ctypes.data_as(C.c_void_p))

    def set_scene(self, scene):
        self._call("set_scene", C.byref(scene))

    def set_camera(self, cam):
        self._call("set_camera", C.byref(cam))

    def set_reference_indexing(self, on=True):
        """Occupancy queries outside the grid read the bit the reference's own index arithmetic addresses (raytracer.py:17-44)
        instead of "empty": include/vrt_api.h, vrt_set_reference_indexing."""
        self._call("set_reference_indexing", int(bool(on)))

    def set_row_stripes(self, stripe_rows, n_parts, part):
        """This (whole-frame) context produces every n_parts-th stripe of stripe_rows rows: include/vrt_api.h, vrt_set_row_stripes."""
        self._call("set_row_stripes", int(stripe_rows), int(n_parts), int(part))
        self.stripes = (int(stripe_rows), int(n_parts), int(part)) if stripe_rows else None

    def owned_rows(self):
        """Row indices this session produces, in the order its device tiles hold them."""
        st = getattr(self, "stripes", None)
        if not st:
            return np.arange(self.rows[0], self.rows[1])
        s_, n_, p_ = st
        return np.concatenate([np.arange(a, min(a + s_, self.H)) for a in range(p_ * s_, self.H, s_ * n_)])

    # -- work ------------------------------------------------------------------------------
    def prepare(self):
        self._call("prepare")

    def sky_accumulate_clouds(self, max_samples):
        self._call("sky_accumulate_clouds", int(max_samples))

    def sky_compute_slice(self, slice_idx, max_slices):
        self._call("sky_compute_slice", int(slice_idx), int(max_slices))

    def sky_accumulate_clouds_slice(self, max_samples, slice_idx, max_slices):
        self._call("sky_accumulate_clouds_slice", int(max_samples), int(slice_idx), int(max_slices))

    def sky_table_io(self, which, u0, u1, ptr, to_library):
        """Columns [u0, u1) of a sky table <-> caller memory at `ptr` (device memory for the HIP library, host for the oracle)."""
        self._call("sky_table_io", int(which), int(u0), int(u1), C.c_void_p(int(ptr)), int(bool(to_library)))

    def accumulate(self, n=1):
        self._call("accumulate", int(n))

    def reset(self):
        self._call("reset")

    def end_frame(self):
        self._call("end_frame")

    def sync(self):
        if getattr(self._lib, self._p + "sync", None) is not None:
            self._call("sync")

    # -- results ---------------------------------------------------------------------------
    def fetch_hdr(self):
        out = np.empty((self.H, self.W, 3), dtype=np.float32)
        self._call("fetch_hdr", out.ctypes.data_as(C.c_void_p))
        return out

    def fetch_hdr_device(self, device_ptr):
        self._call("fetch_hdr_device", C.c_void_p(int(device_ptr)))

    def fetch_hdr_device_async(self, device_ptr):
        self._call("fetch_hdr_device_async", C.c_void_p(int(device_ptr)))

    def reserve_cus(self, n_cus):
        """Leave n_cus CUs' worth of workgroup slots out of the persistent render grid (multi-GPU: room for RCCL's kernels)."""
        if getattr(self._lib, self._p + "reserve_cus", None) is not None:
            self._call("reserve_cus", int(n_cus))

    def set_stream(self, hip_stream):
        self._call("set_stream", C.c_void_p(int(hip_stream) if hip_stream else None))

    def fetch_ldr(self):
        out = np.empty((self.H, self.W, 4), dtype=np.float32)
        self._call("fetch_ldr", out.ctypes.data_as(C.c_void_p))
        return out

    # -- presenting every frame (the reference's accumulate / fetch_image / copy_prev_matrices loop, scene.py:255-262) ---
    def host_alloc(self, shape, dtype=np.float32):
        """Page-locked host array (vrt_host_alloc): the target of the asynchronous fetches.  The memory belongs to the SESSION:
        close() frees it, after which the array must not be touched -- copy what has to outlive the session
        (Renderer.present_wait returns copies)."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = C.c_void_p()
        self._call("host_alloc", C.c_uint64(n), C.byref(ptr))
        buf = (C.c_char * n).from_address(ptr.value)
        arr = np.frombuffer(buf, dtype=dtype).reshape(shape)
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append((ptr.value, arr))
        return arr

    def fetch_hdr_async(self, out, slot=0):
        assert out.dtype == np.float32 and out.shape == (self.H, self.W, 3) and out.flags.c_contiguous
        self._call("fetch_hdr_async", out.ctypes.data_as(C.c_void_p), int(slot))

    def fetch_ldr_async(self, out, slot=0):
        assert out.dtype == np.float32 and out.shape == (self.H, self.W, 4) and out.flags.c_contiguous
        self._call("fetch_ldr_async", out.ctypes.data_as(C.c_void_p), int(slot))

    def fetch_ldr8_async(self, out, slot=0):
        assert out.dtype == np.uint8 and out.shape == (self.H, self.W, 4) and out.flags.c_contiguous
        self._call("fetch_ldr8_async", out.ctypes.data_as(C.c_void_p), int(slot))

    def fetch_wait(self, slot=0):
        self._call("fetch_wait", int(slot))

    # -- multi-GPU hand-over: the temporal pass writes the rank's HDR tile itself (vrt_set_hdr_targets) ---------------
    def set_hdr_targets(self, device_ptrs):
        arr = (C.c_void_p * len(device_ptrs))(*[int(p) for p in device_ptrs])
        self._call("set_hdr_targets", arr, len(device_ptrs))

    def hdr_targets_written(self):
        n = C.c_uint64(0)
        self._call("hdr_targets_written", C.byref(n))
        return int(n.value)

    _BUF = {
        _abi.BUF_GBUF_DEPTH: (np.float32, 1), _abi.BUF_GBUF_NORMAL: (np.uint16, 2), _abi.BUF_GBUF_POSITION: (np.float32, 3),
        _abi.BUF_GBUF_MAT: (np.uint32, 1), _abi.BUF_GBUF_REFL_DEPTH: (np.float32, 1),
        _abi.BUF_HISTORY_DIFFUSE: (np.float32, 4), _abi.BUF_HISTORY_SPECULAR: (np.float32, 4),
    }

    def fetch_buffer(self, which):
        if which in self._BUF:
            dt, k = self._BUF[which]
            out = np.empty((self.H, self.W, k), dtype=dt)
        elif which in (_abi.BUF_SKY_SCATTERING, _abi.BUF_SKY_TRANSMITTANCE):
            r = self.cfg.sky_res
            out = np.empty((r, r, 3), dtype=np.float32)
        elif which == _abi.BUF_TRANS_LUT:
            out = np.empty((256, 128, 3), dtype=np.uint16)
        else:
            raise ValueError(f"unknown buffer id {which}")
        self._call("fetch_buffer", int(which), out.ctypes.data_as(C.c_void_p))
        return out

    def stats(self):
        s = _abi.VrtStats()
        self._call("get_stats", C.byref(s))
        return s.as_dict()
