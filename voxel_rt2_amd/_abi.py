"""ctypes mirror of include/vrt_api.h (structs and constants only; no library is loaded here)."""
import ctypes as C

VRT_OK = 0
VRT_E_INVALID, VRT_E_DEVICE, VRT_E_STATE = -1, -2, -3

BUF_GBUF_DEPTH, BUF_GBUF_NORMAL, BUF_GBUF_POSITION, BUF_GBUF_MAT, BUF_GBUF_REFL_DEPTH = 1, 2, 3, 4, 5
BUF_HISTORY_DIFFUSE, BUF_HISTORY_SPECULAR, BUF_SKY_SCATTERING, BUF_SKY_TRANSMITTANCE, BUF_TRANS_LUT = 6, 7, 8, 9, 10


class VrtConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("grid_res", C.c_int32),
        ("dx", C.c_float), ("voxel_edges", C.c_float), ("exposure", C.c_float),
        ("max_depth", C.c_int32), ("use_restir", C.c_int32), ("seed", C.c_uint32),
        ("sky_res", C.c_int32), ("device", C.c_int32),
        ("row_begin", C.c_int32), ("row_end", C.c_int32),
    ]


class VrtSceneParams(C.Structure):
    _fields_ = [
        ("floor_height", C.c_float), ("floor_color", C.c_float * 3), ("floor_material", C.c_int32),
        ("background_color", C.c_float * 3), ("light_direction", C.c_float * 3),
        ("light_cos_theta_max", C.c_float), ("light_color", C.c_float * 3), ("light_weight", C.c_float),
        ("use_physical_sky", C.c_int32), ("use_clouds", C.c_int32),
    ]


class VrtCamera(C.Structure):
    _fields_ = [
        ("view", C.c_float * 16), ("proj", C.c_float * 16), ("view_inv", C.c_float * 16), ("proj_inv", C.c_float * 16),
        ("pos", C.c_float * 3), ("jitter_index", C.c_uint32), ("camera_is_moving", C.c_int32),
        ("render_scale", C.c_float), ("max_accum_frames", C.c_float),
    ]


class VrtStats(C.Structure):
    _fields_ = [
        ("path_samples", C.c_uint64), ("rays", C.c_uint64), ("dda_iters", C.c_uint64),
        ("occupancy_queries", C.c_uint64), ("closest_hits", C.c_uint64), ("sky_lookups", C.c_uint64),
        ("render_ms", C.c_double), ("temporal_ms", C.c_double), ("gris_ms", C.c_double),
        ("render_launches", C.c_uint32), ("temporal_launches", C.c_uint32), ("gris_launches", C.c_uint32),
        ("pipeline_flags", C.c_uint32),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def declare(lib, prefix):
    """Attach argtypes/restype for the API shared by libvrt_hip.so (prefix 'vrt_') and the test
    oracle (prefix 'orc_').  Entry points one library lacks are skipped."""
    P = C.c_void_p

    def sig(name, res, *args):
        fn = getattr(lib, prefix + name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = list(args)

    sig("destroy", None, P)
    sig("upload_voxels", C.c_int, P, P, P)
    sig("upload_materials", C.c_int, P, P)
    sig("upload_cloud_texture", C.c_int, P, P)
    sig("set_scene", C.c_int, P, C.POINTER(VrtSceneParams))
    sig("set_camera", C.c_int, P, C.POINTER(VrtCamera))
    sig("prepare", C.c_int, P)
    sig("sky_accumulate_clouds", C.c_int, P, C.c_int)
    sig("sky_compute_slice", C.c_int, P, C.c_int, C.c_int)
    sig("sky_accumulate_clouds_slice", C.c_int, P, C.c_int, C.c_int, C.c_int)
    sig("sky_table_io", C.c_int, P, C.c_int, C.c_int, C.c_int, P, C.c_int)
    sig("accumulate", C.c_int, P, C.c_int)
    sig("reset", C.c_int, P)
    sig("end_frame", C.c_int, P)
    sig("fetch_hdr", C.c_int, P, P)
    sig("fetch_hdr_device", C.c_int, P, P)
    sig("fetch_hdr_device_async", C.c_int, P, P)
    sig("set_stream", C.c_int, P, P)
    sig("reserve_cus", C.c_int, P, C.c_int)
    sig("fetch_ldr", C.c_int, P, P)
    sig("fetch_hdr_async", C.c_int, P, P, C.c_int)
    sig("fetch_ldr_async", C.c_int, P, P, C.c_int)
    sig("fetch_ldr8_async", C.c_int, P, P, C.c_int)
    sig("fetch_wait", C.c_int, P, C.c_int)
    sig("host_alloc", C.c_int, P, C.c_uint64, C.POINTER(C.c_void_p))
    sig("host_free", C.c_int, P, P)
    sig("set_hdr_targets", C.c_int, P, C.POINTER(C.c_void_p), C.c_int)
    sig("hdr_targets_written", C.c_int, P, C.POINTER(C.c_uint64))
    sig("fetch_buffer", C.c_int, P, C.c_int, P)
    sig("sync", C.c_int, P)
    sig("get_stats", C.c_int, P, C.POINTER(VrtStats))
    sig("last_error", C.c_char_p)
    sig("is_instrumented", C.c_int)
    sig("set_reference_indexing", C.c_int, P, C.c_int)
    sig("set_row_stripes", C.c_int, P, C.c_int, C.c_int, C.c_int)
