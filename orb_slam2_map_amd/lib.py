"""ctypes binding of liborbgpu.so (include/orbgpu.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is missing,
or no HIP device is visible, calls fail loudly (OrbGpuError / OSError).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liborbgpu.so")

OK, EINVAL, ENOMEM, EHIP, ECAPACITY, ELEVEL = 0, -1, -2, -3, -4, -5
MAX_LEVELS = 16
GRID_COLS, GRID_ROWS = 64, 48
TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")])
POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])

DBG_PYRAMID_PADDED, DBG_BLURRED_PADDED, DBG_CANDIDATES, DBG_SELECTED = 0, 1, 2, 3


class OrbGpuError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("orbgpu status %d: %s" % (status, msg))
        self.status = status


class ExtractorParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("device_id", C.c_int32),
                ("max_batch", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p), ("kp_octave", C.c_void_p),
                ("kp_angle", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("scale_factors", C.c_void_p),
                ("nlevels", C.c_int32), ("cell_start", C.c_void_p), ("cell_items", C.c_void_p)]


class MapPointView(C.Structure):
    _fields_ = [("m", C.c_int32), ("in_view", C.c_void_p), ("bad", C.c_void_p), ("obs_pos", C.c_void_p),
                ("level", C.c_void_p), ("view_cos", C.c_void_p), ("proj_x", C.c_void_p),
                ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p), ("desc", C.c_void_p)]


class KeyFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("has_mp", C.c_void_p), ("bad", C.c_void_p), ("already_found", C.c_void_p),
                ("world_pos", C.c_void_p), ("min_dist_inv", C.c_void_p), ("max_dist_inv", C.c_void_p),
                ("max_dist", C.c_void_p), ("desc", C.c_void_p), ("kp_angle", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("dist", C.c_float * 5),
                ("mbf", C.c_float), ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float),
                ("max_y", C.c_float)]


class DeviceFrameView(C.Structure):
    _fields_ = [("cap", C.c_int32), ("n", C.c_void_p), ("kps", C.c_void_p), ("desc", C.c_void_p),
                ("u_right", C.c_void_p), ("cell_start", C.c_void_p), ("cell_items", C.c_void_p),
                ("nlevels", C.c_int32), ("scale_factors", C.c_void_p), ("min_x", C.c_float), ("max_x", C.c_float),
                ("min_y", C.c_float), ("max_y", C.c_float)]


class DeviceMapPointTable(C.Structure):
    _fields_ = [("m", C.c_int32), ("world_pos", C.c_void_p), ("normal", C.c_void_p), ("min_dist", C.c_void_p),
                ("max_dist", C.c_void_p), ("desc", C.c_void_p), ("skip", C.c_void_p), ("obs_pos", C.c_void_p)]


class LocalPointsProblem(C.Structure):
    _fields_ = [("frame", C.c_void_p), ("table", C.c_void_p), ("Tcw", C.c_void_p), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("mbf", C.c_float), ("log_scale_factor", C.c_float),
                ("d_kp_to_mp", C.c_void_p), ("d_counts", C.c_void_p), ("d_track", C.c_void_p)]


class DeviceLastFrameView(C.Structure):
    _fields_ = [("cap", C.c_int32), ("n", C.c_void_p), ("kps", C.c_void_p), ("has_mp", C.c_void_p),
                ("outlier", C.c_void_p), ("obs_pos", C.c_void_p), ("world_pos", C.c_void_p), ("desc", C.c_void_p)]


class TrackScratch(C.Structure):
    _fields_ = [("in_view", C.c_void_p), ("proj_x", C.c_void_p), ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p),
                ("view_cos", C.c_void_p), ("level", C.c_void_p)]


class PointsView(C.Structure):
    _fields_ = [("m", C.c_int32), ("bad", C.c_void_p), ("world_pos", C.c_void_p), ("normal", C.c_void_p),
                ("min_dist", C.c_void_p), ("max_dist", C.c_void_p), ("desc", C.c_void_p)]


class LastFrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("has_mp", C.c_void_p), ("outlier", C.c_void_p), ("obs_pos", C.c_void_p),
                ("world_pos", C.c_void_p), ("desc", C.c_void_p), ("kp_octave", C.c_void_p),
                ("kp_angle", C.c_void_p), ("Tcw", C.c_void_p)]


_LIB = None

# every symbol include/orbgpu.h declares (checked by tests/test_abi.py against the header text)
ABI_SYMBOLS = [
    "orbgpu_last_error_string", "orbgpu_abi_version", "orbgpu_device_count", "orbgpu_measure_copy_bandwidth",
    "orbgpu_set_trig_mode", "orbgpu_get_trig_mode", "orbgpu_trig_table_info", "orbgpu_trig_host_eval", "orbgpu_trig_eval",
    "orbgpu_extractor_create", "orbgpu_extractor_destroy", "orbgpu_extractor_get_levels",
    "orbgpu_extractor_get_scale_factor", "orbgpu_extractor_get_scale_factors",
    "orbgpu_extractor_get_inv_scale_factors", "orbgpu_extractor_get_sigma2", "orbgpu_extractor_get_inv_sigma2",
    "orbgpu_extractor_get_quotas", "orbgpu_extractor_max_keypoints", "orbgpu_extract", "orbgpu_extract_batch",
    "orbgpu_extract_batch_device", "orbgpu_extractor_get_pyramid_level", "orbgpu_extractor_debug_read",
    "orbgpu_extractor_graph_state", "orbgpu_extractor_debug_quadtree_config", "orbgpu_extractor_set_profiling", "orbgpu_extractor_set_concurrent_blur", "orbgpu_extractor_set_fast_early_out", "orbgpu_extractor_set_stage_signal", "orbgpu_extractor_stage_count", "orbgpu_extractor_stage_name",
    "orbgpu_extractor_stage_times",
    "orbgpu_pipeline_create", "orbgpu_pipeline_destroy", "orbgpu_pipeline_parts", "orbgpu_pipeline_part",
    "orbgpu_pipeline_extract_device", "orbgpu_pipeline_wait",
    "orbgpu_hamming256", "orbgpu_match_bf", "orbgpu_matcher_create", "orbgpu_matcher_destroy",
    "orbgpu_match_bf_batch_device", "orbgpu_matcher_last_sweeps", "orbgpu_assign_features_to_grid",
    "orbgpu_frame_glue_batch_device", "orbgpu_undistort_points", "orbgpu_search_local_points_device", "orbgpu_search_local_points_batch_device", "orbgpu_search_by_projection_last_device", "orbgpu_projection_last_sweeps", "orbgpu_distinctive_descriptors", "orbgpu_search_by_projection_sim3",
    "orbgpu_search_by_projection", "orbgpu_search_by_projection_last", "orbgpu_search_by_projection_keyframe",
    "orbgpu_mappoint_table_create", "orbgpu_mappoint_table_destroy", "orbgpu_mappoint_table_rows",
    "orbgpu_mappoint_table_upsert", "orbgpu_mappoint_table_set_bad", "orbgpu_mappoint_table_set_observations",
    "orbgpu_mappoint_table_read", "orbgpu_mappoint_table_last_unknown", "orbgpu_mappoint_table_retain", "orbgpu_frame_create", "orbgpu_frame_destroy", "orbgpu_frame_upload",
    "orbgpu_frame_device_view", "orbgpu_search_local_points_table", "orbgpu_search_by_projection_last_table",
    "orbgpu_vocabulary_create", "orbgpu_vocabulary_destroy", "orbgpu_vocabulary_size", "orbgpu_bow_transform",
    "orbgpu_bow_transform_batch_device", "orbgpu_search_by_bow", "orbgpu_search_by_bow_batch_device",
    "orbgpu_search_by_bow_keyframes", "orbgpu_search_for_triangulation", "orbgpu_search_for_initialization", "orbgpu_fuse", "orbgpu_fuse_sim3",
    "orbgpu_search_by_sim3",
    "orbgpu_mappoint_record_bytes", "orbgpu_write_mappoint_record", "orbgpu_keyframe_record_bytes",
    "orbgpu_write_keyframe_record", "orbgpu_pcd_binary_header", "orbgpu_write_pcd_binary", "orbgpu_cloud_save_pcd",
    "orbgpu_cloud_create", "orbgpu_cloud_destroy", "orbgpu_cloud_insert", "orbgpu_cloud_insert_device",
    "orbgpu_cloud_clear", "orbgpu_cloud_append_filtered", "orbgpu_cloud_last_path", "orbgpu_cloud_set_profiling", "orbgpu_cloud_last_insert_ms", "orbgpu_cloud_rebuild",
    "orbgpu_cloud_size", "orbgpu_cloud_download", "orbgpu_cloud_last_overflow", "orbgpu_backproject",
    "orbgpu_voxel_filter", "orbgpu_cloud_remove_outliers", "orbgpu_statistical_outlier_removal",
]


def lib():
    """Loads liborbgpu.so. Raises OSError if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if "torch" not in sys.modules:
        # torch wheels bundle their own HIP runtime; if liborbgpu.so pulls in /opt/rocm's copy first,
        # a later `import torch` in the same process cannot see the GPU ("No HIP GPUs are available").
        # Loading torch first makes both share one runtime.  Pure plumbing: nothing here uses torch.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    if not os.path.exists(LIB_PATH):
        raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(the HIP extension is the product; there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int32, C.c_float, C.c_size_t
    L.orbgpu_last_error_string.restype = C.c_char_p
    L.orbgpu_extractor_stage_name.restype = C.c_char_p
    L.orbgpu_extractor_stage_name.argtypes = [i32]
    sigs = {
        "orbgpu_extractor_create": [vp, vp],
        "orbgpu_set_trig_mode": [i32],
        "orbgpu_trig_table_info": [vp, vp],
        "orbgpu_extractor_destroy": [vp],
        "orbgpu_extractor_get_levels": [vp, vp],
        "orbgpu_extractor_get_scale_factor": [vp, vp],
        "orbgpu_extractor_get_scale_factors": [vp, vp],
        "orbgpu_extractor_get_inv_scale_factors": [vp, vp],
        "orbgpu_extractor_get_sigma2": [vp, vp],
        "orbgpu_extractor_get_inv_sigma2": [vp, vp],
        "orbgpu_extractor_get_quotas": [vp, vp],
        "orbgpu_extractor_max_keypoints": [vp, i32, i32, vp],
        "orbgpu_extract": [vp, vp, i32, i32, sz, vp, vp, i32, vp],
        "orbgpu_extract_batch": [vp, vp, i32, i32, i32, sz, sz, vp, vp, i32, vp],
        "orbgpu_extract_batch_device": [vp, vp, i32, i32, i32, sz, sz, vp, vp, i32, vp, vp],
        "orbgpu_extractor_get_pyramid_level": [vp, i32, i32, vp, sz, vp, vp],
        "orbgpu_extractor_debug_read": [vp, i32, i32, i32, vp, sz, vp, vp],
        "orbgpu_extractor_set_profiling": [vp, i32],
        "orbgpu_extractor_set_stage_signal": [vp, i32, vp],
        "orbgpu_extractor_stage_times": [vp, vp],
        "orbgpu_pipeline_create": [vp, i32, vp],
        "orbgpu_pipeline_destroy": [vp],
        "orbgpu_pipeline_parts": [vp, vp],
        "orbgpu_pipeline_part": [vp, i32, vp],
        "orbgpu_pipeline_extract_device": [vp, vp, i32, i32, i32, sz, sz, vp, vp, i32, vp, vp, vp],
        "orbgpu_pipeline_wait": [vp, vp],
        "orbgpu_hamming256": [vp, vp, i32, vp, i32],
        "orbgpu_match_bf": [vp, vp, vp, i32, vp, vp, i32, i32, f32, i32, vp, vp, i32],
        "orbgpu_matcher_create": [i32, i32, i32, vp],
        "orbgpu_matcher_destroy": [vp],
        "orbgpu_match_bf_batch_device": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, sz, i32, f32, i32, vp, vp, vp],
        "orbgpu_matcher_last_sweeps": [vp, vp],
        "orbgpu_assign_features_to_grid": [i32, vp, vp, f32, f32, f32, f32, vp, vp],
        "orbgpu_frame_glue_batch_device": [i32, i32, i32, vp, vp, vp, sz, sz, vp, vp, vp, vp, vp, vp, vp],
        "orbgpu_undistort_points": [i32, vp, vp, vp, i32],
        "orbgpu_search_local_points_device": [vp, vp, vp, f32, f32, f32, f32, f32, f32, f32, f32, f32, vp, vp, vp, i32, vp],
        "orbgpu_projection_last_sweeps": [vp, vp],
        "orbgpu_distinctive_descriptors": [i32, vp, vp, vp, i32],
        "orbgpu_search_by_projection_sim3": [vp, vp, f32, f32, f32, f32, f32, vp, i32, vp, vp, i32],
        "orbgpu_search_by_projection": [vp, vp, f32, f32, vp, vp, i32],
        "orbgpu_search_by_projection_last": [vp, vp, f32, f32, f32, f32, f32, f32, vp, f32, i32, i32, vp, vp, i32],
        "orbgpu_search_by_projection_keyframe": [vp, vp, f32, f32, f32, f32, f32, vp, f32, i32, i32, vp, vp, i32],
        "orbgpu_cloud_create": [C.c_double, i32, vp],
        "orbgpu_cloud_destroy": [vp],
        "orbgpu_cloud_insert": [vp, vp, sz, vp, sz, i32, i32, f32, f32, f32, f32, vp],
        "orbgpu_cloud_insert_device": [vp, vp, sz, vp, sz, i32, i32, f32, f32, f32, f32, vp],
        "orbgpu_cloud_last_path": [vp, vp],
        "orbgpu_cloud_set_profiling": [vp, i32],
        "orbgpu_cloud_last_insert_ms": [vp, vp],
        "orbgpu_cloud_rebuild": [vp, i32, vp, sz, vp, sz, i32, i32, f32, f32, f32, f32, vp],
        "orbgpu_cloud_size": [vp, vp],
        "orbgpu_cloud_download": [vp, vp, C.c_int64, vp],
        "orbgpu_cloud_last_overflow": [vp, vp],
        "orbgpu_backproject": [vp, sz, vp, sz, i32, i32, f32, f32, f32, f32, vp, vp, C.c_int64, vp, i32],
        "orbgpu_voxel_filter": [vp, C.c_int64, C.c_double, vp, C.c_int64, vp, vp, i32],
        "orbgpu_cloud_remove_outliers": [vp, i32, C.c_double, vp],
        "orbgpu_statistical_outlier_removal": [vp, C.c_int64, i32, C.c_double, vp, C.c_int64, vp, vp, i32],
    }
    for name, args in sigs.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes = args
            fn.restype = C.c_int
    _LIB = L
    return L


def check(status):
    if status != OK:
        raise OrbGpuError(status, lib().orbgpu_last_error_string().decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count():
    return lib().orbgpu_device_count()


def measure_copy_bandwidth(nbytes=1 << 30, reps=10, device_id=0):
    """Achieved GB/s (read + write) of a plain device-to-device copy kernel: the practical HBM roofline."""
    L = lib()
    L.orbgpu_measure_copy_bandwidth.argtypes = [C.c_size_t, C.c_int32, C.c_int32, C.c_void_p]
    v = C.c_float()
    check(L.orbgpu_measure_copy_bandwidth(nbytes, reps, device_id, C.byref(v)))
    return v.value


TRIG_HOST_LIBM, TRIG_ROUNDED_DOUBLE = 0, 1


def set_trig_mode(mode):
    """cos / sin of computeOrbDescriptor (ORBextractor.cc:112-113) for extractors created from now on: TRIG_HOST_LIBM
    (default: this host's cosf / sinf, bit for bit) or TRIG_ROUNDED_DOUBLE ((float)cos((double)angle))."""
    check(lib().orbgpu_set_trig_mode(int(mode)))


def get_trig_mode():
    return lib().orbgpu_get_trig_mode()


def trig_table_info():
    """(entries of the host-libm exception table or -1 if it has not been built, milliseconds its scan took)."""
    n, ms = C.c_int64(), C.c_double()
    check(lib().orbgpu_trig_table_info(C.byref(n), C.byref(ms)))
    return n.value, ms.value


def trig_host_eval(x):
    """(cos, sin, from_table) of the float x as TRIG_HOST_LIBM delivers them; needs no device."""
    L = lib()
    L.orbgpu_trig_host_eval.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    c, s, t = C.c_float(), C.c_float(), C.c_int32()
    check(L.orbgpu_trig_host_eval(float(x), C.byref(c), C.byref(s), C.byref(t)))
    return np.float32(c.value), np.float32(s.value), t.value


def trig_eval(x, device_id=0):
    """cos / sin of the float32 array x as the DEVICE computes them for the descriptor stage (current trig mode)."""
    L = lib()
    L.orbgpu_trig_eval.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    x = np.ascontiguousarray(x, np.float32)
    c, s = np.zeros(len(x), np.float32), np.zeros(len(x), np.float32)
    check(L.orbgpu_trig_eval(_p(x), len(x), _p(c), _p(s), device_id))
    return c, s


# --------------------------------------------------------------------------------------------
# ORBextractor (reference include/ORBextractor.h:46-110)
# --------------------------------------------------------------------------------------------
class ORBextractor:
    """Mirror of ORB_SLAM2::ORBextractor; __call__ mirrors operator() (ORBextractor.h:59-61)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device_id=0,
                 max_batch=1):
        self.L = lib()
        self.params = ExtractorParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id, max_batch)
        h = C.c_void_p()
        check(self.L.orbgpu_extractor_create(C.byref(self.params), C.byref(h)))
        self.h = h
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def close(self):
        if getattr(self, "h", None):
            self.L.orbgpu_extractor_destroy(self.h)
            self.h = None

    __del__ = close

    def GetLevels(self):
        n = C.c_int32()
        check(self.L.orbgpu_extractor_get_levels(self.h, C.byref(n)))
        return n.value

    def GetScaleFactor(self):
        s = C.c_float()
        check(self.L.orbgpu_extractor_get_scale_factor(self.h, C.byref(s)))
        return s.value

    def _vec(self, fn, dtype):
        out = np.zeros(self.nlevels, dtype)
        check(fn(self.h, _p(out)))
        return out

    def GetScaleFactors(self):
        return self._vec(self.L.orbgpu_extractor_get_scale_factors, np.float32)

    def GetInverseScaleFactors(self):
        return self._vec(self.L.orbgpu_extractor_get_inv_scale_factors, np.float32)

    def GetScaleSigmaSquares(self):
        return self._vec(self.L.orbgpu_extractor_get_sigma2, np.float32)

    def GetInverseScaleSigmaSquares(self):
        return self._vec(self.L.orbgpu_extractor_get_inv_sigma2, np.float32)

    def quotas(self):
        return self._vec(self.L.orbgpu_extractor_get_quotas, np.int32)

    def max_keypoints(self, width, height):
        cap = C.c_int32()
        check(self.L.orbgpu_extractor_max_keypoints(self.h, width, height, C.byref(cap)))
        return cap.value

    def __call__(self, image, mask=None):
        """(keypoints, descriptors) of one 8-bit gray image; mask is ignored as in the reference."""
        k, d = self.extract_batch(np.asarray(image)[None])
        return k[0], d[0]

    def extract_batch(self, images):
        images = np.ascontiguousarray(images, np.uint8)
        if images.ndim != 3:
            raise ValueError("images must be [batch, h, w] uint8")
        b, h, w = images.shape
        if h == 0 or w == 0:
            return [np.zeros(0, KEYPOINT_DTYPE)] * b, [np.zeros((0, 32), np.uint8)] * b
        cap = self.max_keypoints(w, h)
        kps = np.zeros((b, cap), KEYPOINT_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        check(self.L.orbgpu_extract_batch(self.h, _p(images), b, w, h, images.strides[1], images.strides[0], _p(kps),
                                          _p(desc), cap, _p(n)))
        return [kps[i, :n[i]].copy() for i in range(b)], [desc[i, :n[i]].copy() for i in range(b)]

    def extract_batch_device(self, d_gray_ptr, batch, width, height, stride, frame_stride, d_kps_ptr, d_desc_ptr, cap,
                             d_nout_ptr, stream=0):
        check(self.L.orbgpu_extract_batch_device(self.h, d_gray_ptr, batch, width, height, stride, frame_stride,
                                                 d_kps_ptr, d_desc_ptr, cap, d_nout_ptr, stream))

    def pyramid_level(self, frame, level):
        """mvImagePyramid[level] of `frame` of the last call (ORBextractor.h:85)."""
        raw, pitch = self.debug_read(DBG_PYRAMID_PADDED, frame, level)
        img = raw.reshape(-1, pitch)
        return img

    def debug_read(self, what, frame, level):
        cap = 64 << 20
        buf = np.zeros(cap, np.uint8)
        n = C.c_size_t()
        aux = C.c_int32()
        check(self.L.orbgpu_extractor_debug_read(self.h, what, frame, level, _p(buf), cap, C.byref(n), C.byref(aux)))
        if what in (DBG_PYRAMID_PADDED, DBG_BLURRED_PADDED):
            return buf[:n.value].copy(), aux.value
        return buf[:n.value * 12].view(np.int32).reshape(-1, 3).copy(), 0

    def graph_state(self):
        v = C.c_int32()
        self.L.orbgpu_extractor_graph_state.argtypes = [C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_extractor_graph_state(self.h, C.byref(v)))
        return v.value

    def quadtree_config(self, batch=1):
        """(keys of a level kept in LDS, threads per workgroup, dynamic LDS bytes) of k_quadtree for `batch` frames."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self.L.orbgpu_extractor_debug_quadtree_config.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_extractor_debug_quadtree_config(self.h, batch, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_concurrent_blur(self, on):
        check(self.L.orbgpu_extractor_set_concurrent_blur(self.h, int(on)))

    def get_pyramid_level(self, frame, level):
        """mvImagePyramid[level] of frame `frame` of the last call (ORBextractor.h:85) through the drop-in getter
        orbgpu_extractor_get_pyramid_level: (pixels HxW, width, height)."""
        buf = np.zeros((4112, 4112), np.uint8)  # the largest level the library accepts
        w, h = C.c_int32(), C.c_int32()
        check(self.L.orbgpu_extractor_get_pyramid_level(self.h, int(frame), int(level), _p(buf), buf.strides[0], C.byref(w),
                                                        C.byref(h)))
        return buf[:h.value, :w.value].copy(), w.value, h.value

    def set_fast_early_out(self, on):
        """Exact wave-level early-out of the FAST score network (orbgpu_extractor_set_fast_early_out)."""
        self.L.orbgpu_extractor_set_fast_early_out.argtypes = [C.c_void_p, C.c_int32]
        check(self.L.orbgpu_extractor_set_fast_early_out(self.h, int(on)))

    def set_profiling(self, on):
        check(self.L.orbgpu_extractor_set_profiling(self.h, int(on)))

    def set_stage_signal(self, stage_name, hip_event):
        """hip_event (raw hipEvent_t handle or None) is recorded after the named stage of every later device-batch call."""
        names = [self.L.orbgpu_extractor_stage_name(i).decode() for i in range(self.L.orbgpu_extractor_stage_count())]
        check(self.L.orbgpu_extractor_set_stage_signal(self.h, names.index(stage_name), C.c_void_p(hip_event or 0)))

    def stage_times(self):
        n = self.L.orbgpu_extractor_stage_count()
        ms = np.zeros(n, np.float32)
        check(self.L.orbgpu_extractor_stage_times(self.h, _p(ms)))
        return {self.L.orbgpu_extractor_stage_name(i).decode(): float(ms[i]) for i in range(n)}


class _PipelinePart(ORBextractor):
    """View of one part's handle (set_profiling / stage_times / set_stage_signal / debug reads); the pipeline owns it."""

    def __init__(self, L, h, nfeatures, nlevels):
        self.L, self.h, self.nfeatures, self.nlevels = L, h, nfeatures, nlevels

    def close(self):
        self.h = None

    __del__ = close


class ExtractorPipeline:
    """orbgpu_pipeline: a resident batch extracted as `parts` staggered sub-batches on streams of their own (include/orbgpu.h).
    Calls only enqueue; order consumers with done_event / wait(stream)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device_id=0, max_batch=1,
                 parts=2):
        self.L = lib()
        self.params = ExtractorParams(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device_id, max_batch)
        h = C.c_void_p()
        check(self.L.orbgpu_pipeline_create(C.byref(self.params), parts, C.byref(h)))
        self.h = h
        self.parts = []
        for k in range(parts):
            ph = C.c_void_p()
            check(self.L.orbgpu_pipeline_part(self.h, k, C.byref(ph)))
            self.parts.append(_PipelinePart(self.L, ph, nfeatures, nlevels))

    def close(self):
        if getattr(self, "h", None):
            for e in self.parts:
                e.h = None
            self.L.orbgpu_pipeline_destroy(self.h)
            self.h = None

    __del__ = close

    def max_keypoints(self, width, height):
        return self.parts[0].max_keypoints(width, height)

    def extract_batch_device(self, d_gray_ptr, batch, width, height, stride, frame_stride, d_kps_ptr, d_desc_ptr, cap,
                             d_nout_ptr, wait_event=None, done_event=None):
        check(self.L.orbgpu_pipeline_extract_device(self.h, d_gray_ptr, batch, width, height, stride, frame_stride, d_kps_ptr,
                                                    d_desc_ptr, cap, d_nout_ptr, C.c_void_p(wait_event or 0),
                                                    C.c_void_p(done_event or 0)))

    def wait(self, stream=0):
        check(self.L.orbgpu_pipeline_wait(self.h, C.c_void_p(stream or 0)))


# --------------------------------------------------------------------------------------------
# ORBmatcher (reference include/ORBmatcher.h:41-106)
# --------------------------------------------------------------------------------------------
def assign_features_to_grid(kp_x, kp_y, min_x, min_y, inv_w, inv_h):
    kp_x = np.ascontiguousarray(kp_x, np.float32)
    kp_y = np.ascontiguousarray(kp_y, np.float32)
    cs = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    items = np.zeros(max(1, len(kp_x)), np.int32)
    check(lib().orbgpu_assign_features_to_grid(len(kp_x), _p(kp_x), _p(kp_y), min_x, min_y, inv_w, inv_h, _p(cs),
                                               _p(items)))
    return cs, items[:cs[-1]].copy()


def make_camera(fx, fy, cx, cy, mbf, width, height, dist=(), device_id=0):
    """orbgpu_camera with the image bounds of Frame::ComputeImageBounds (Frame.cc:436-468)."""
    cam = Camera()
    cam.fx, cam.fy, cam.cx, cam.cy, cam.mbf = fx, fy, cx, cy, mbf
    for i in range(5):
        cam.dist[i] = float(dist[i]) if i < len(dist) else 0.0
    cam.min_x, cam.max_x, cam.min_y, cam.max_y = 0.0, float(width), 0.0, float(height)
    if cam.dist[0] != 0.0:
        c = undistort_points(np.array([[0, 0], [width, 0], [0, height], [width, height]], np.float32), cam, device_id)
        cam.min_x, cam.max_x = min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0])
        cam.min_y, cam.max_y = min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])
    return cam


def undistort_points(xy, cam, device_id=0):
    """cv::undistortPoints(xy, xy, mK, mDistCoef, Mat(), mK) (orbgpu_undistort_points)."""
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    check(lib().orbgpu_undistort_points(len(xy), _p(xy), C.byref(cam), _p(out), device_id))
    return out


def frame_glue_batch_device(batch, cap, d_kps, d_n, d_depth, depth_stride, depth_frame_stride, cam, d_kps_un,
                            d_u_right, d_kp_depth, d_cell_start, d_cell_items, stream=0, device_id=0):
    """UndistortKeyPoints + ComputeStereoFromRGBD + AssignFeaturesToGrid on the device
    (orbgpu_frame_glue_batch_device). cam: Camera (make_camera)."""
    check(lib().orbgpu_frame_glue_batch_device(device_id, batch, cap, d_kps, d_n, d_depth, depth_stride,
                                               depth_frame_stride, C.byref(cam), d_kps_un, d_u_right, d_kp_depth,
                                               d_cell_start, d_cell_items, stream))


def search_local_points_device(frame_view, table, Tcw, fx, fy, cx, cy, mbf, log_sf, th, nnratio, d_kp_to_mp, d_counts,
                               track=None, cos_limit=0.5, stream=0, device_id=0):
    """Tracking::SearchLocalPoints on device-resident data (orbgpu_search_local_points_device).
    frame_view: DeviceFrameView, table: DeviceMapPointTable, track: TrackScratch or None."""
    T = np.ascontiguousarray(Tcw, np.float32)
    check(lib().orbgpu_search_local_points_device(C.byref(frame_view), C.byref(table), _p(T), fx, fy, cx, cy, mbf, log_sf,
                                                  cos_limit, th, nnratio, d_kp_to_mp, d_counts,
                                                  C.byref(track) if track is not None else None, device_id, stream))


def search_local_points_batch_device(problems, cos_limit, th, nnratio, stream=0, device_id=0):
    """n independent Tracking::SearchLocalPoints problems in one call (orbgpu_search_local_points_batch_device).
    problems: list of dicts with frame (DeviceFrameView), table (DeviceMapPointTable), Tcw (4x4), fx, fy, cx, cy, mbf,
    log_sf, d_kp_to_mp, d_counts (device pointers)."""
    n = len(problems)
    arr = (LocalPointsProblem * max(n, 1))()
    keep = []
    for k, p in enumerate(problems):
        T = np.ascontiguousarray(p["Tcw"], np.float32)
        keep.append(T)
        arr[k].frame = C.addressof(p["frame"])
        arr[k].table = C.addressof(p["table"])
        arr[k].Tcw = T.ctypes.data
        arr[k].fx, arr[k].fy, arr[k].cx, arr[k].cy, arr[k].mbf = p["fx"], p["fy"], p["cx"], p["cy"], p["mbf"]
        arr[k].log_scale_factor = p["log_sf"]
        arr[k].d_kp_to_mp, arr[k].d_counts, arr[k].d_track = p["d_kp_to_mp"], p["d_counts"], None
    L = lib()
    L.orbgpu_search_local_points_batch_device.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int32,
                                                          C.c_void_p]
    check(L.orbgpu_search_local_points_batch_device(n, arr, cos_limit, th, nnratio, device_id, stream))


def search_by_projection_last_device(cur_view, cur_Tcw, last_view, last_Tcw, fx, fy, cx, cy, mbf, mb, th, mono, check_ori,
                                     d_kp_to_mp, d_counts, stream=0, device_id=0):
    """TrackWithMotionModel's matcher on device-resident frames (orbgpu_search_by_projection_last_device).
    cur_view: DeviceFrameView, last_view: DeviceLastFrameView; poses are host 4x4 arrays."""
    Tc = np.ascontiguousarray(cur_Tcw, np.float32)
    Tl = np.ascontiguousarray(last_Tcw, np.float32)
    L = lib()
    L.orbgpu_search_by_projection_last_device.argtypes = [
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
        C.c_float, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    check(L.orbgpu_search_by_projection_last_device(C.byref(cur_view), _p(Tc), C.byref(last_view), _p(Tl), fx, fy, cx, cy,
                                                    mbf, mb, th, int(mono), int(check_ori), d_kp_to_mp, d_counts,
                                                    device_id, stream))


class MapPointTable:
    """Device-resident MapPoint table keyed by mnId (orbgpu_mappoint_table_*)."""

    def __init__(self, initial_rows=0, device_id=0):
        self.L = lib()
        self.h = C.c_void_p()
        check(self.L.orbgpu_mappoint_table_create(device_id, int(initial_rows), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.orbgpu_mappoint_table_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def rows(self):
        v = C.c_int32()
        check(self.L.orbgpu_mappoint_table_rows(self.h, C.byref(v)))
        return v.value

    def upsert(self, ids, world_pos=None, normal=None, min_dist=None, max_dist=None, desc=None, n_obs=None):
        ids = np.ascontiguousarray(ids, np.int64)
        keep = [ids]

        def arr(a, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            keep.append(a)
            return _p(a)
        self.L.orbgpu_mappoint_table_upsert.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7
        check(self.L.orbgpu_mappoint_table_upsert(self.h, len(ids), _p(ids), arr(world_pos, np.float32), arr(normal, np.float32),
                                                  arr(min_dist, np.float32), arr(max_dist, np.float32), arr(desc, np.uint8),
                                                  arr(n_obs, np.int32)))

    def set_bad(self, ids):
        ids = np.ascontiguousarray(ids, np.int64)
        k = C.c_int32()
        self.L.orbgpu_mappoint_table_set_bad.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_mappoint_table_set_bad(self.h, len(ids), _p(ids), C.byref(k)))
        return k.value

    def set_observations(self, ids, n_obs):
        ids = np.ascontiguousarray(ids, np.int64)
        n_obs = np.ascontiguousarray(n_obs, np.int32)
        k = C.c_int32()
        self.L.orbgpu_mappoint_table_set_observations.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_mappoint_table_set_observations(self.h, len(ids), _p(ids), _p(n_obs), C.byref(k)))
        return k.value

    def retain(self, ids):
        """Keeps only the listed ids (rows renumbered densely, capacity shrunk); returns the number of rows released."""
        ids = np.ascontiguousarray(ids, np.int64)
        d = C.c_int32()
        self.L.orbgpu_mappoint_table_retain.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_mappoint_table_retain(self.h, len(ids), _p(ids), C.byref(d)))
        return d.value

    def last_unknown(self):
        """(list ids, key-point ids) of the last search call over the table that the table did not know."""
        a, b = C.c_int32(), C.c_int32()
        self.L.orbgpu_mappoint_table_last_unknown.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_mappoint_table_last_unknown(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def read(self, id_):
        wp, nr, ds = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(32, np.uint8)
        mn, mx, ob, bad = C.c_float(), C.c_float(), C.c_int32(), C.c_int32()
        self.L.orbgpu_mappoint_table_read.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7
        check(self.L.orbgpu_mappoint_table_read(self.h, int(id_), _p(wp), _p(nr), C.byref(mn), C.byref(mx), _p(ds), C.byref(ob),
                                                C.byref(bad)))
        return {"world_pos": wp, "normal": nr, "min_dist": mn.value, "max_dist": mx.value, "desc": ds,
                "has_observations": ob.value, "bad": bad.value}


class DeviceFrame:
    """A Frame's matcher-side members uploaded once and kept on the device (orbgpu_frame_*)."""

    def __init__(self, device_id=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.n = 0
        check(self.L.orbgpu_frame_create(device_id, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.orbgpu_frame_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, frame):
        """frame: lib.Frame (host SoA)."""
        v = frame.view()
        self.L.orbgpu_frame_upload.argtypes = [C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_frame_upload(self.h, C.byref(v)))
        self.n = frame.n
        return self

    def device_view(self):
        v, n = DeviceFrameView(), C.c_int32()
        self.L.orbgpu_frame_device_view.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        check(self.L.orbgpu_frame_device_view(self.h, C.byref(v), C.byref(n)))
        return v, n.value


def search_local_points_table(dframe, table, ids, Tcw, fx, fy, cx, cy, mbf, log_sf, th, nnratio, skip=None, scratch=None,
                              kp_ids=None, want_track=False, cos_limit=0.5):
    """Tracking::SearchLocalPoints over the MapPoint table (orbgpu_search_local_points_table).
    scratch: dict with in_view, level, view_cos, proj_x, proj_y, proj_xr (the mTrack* members filled on the host) or None
    (isInFrustum on the device).  Returns (nmatches, kp_to_mp[, track dict])."""
    L = lib()
    ids = np.ascontiguousarray(ids, np.int64)
    m, n = len(ids), dframe.n
    keep = []

    def arr(a, dt):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dt)
        keep.append(a)
        return _p(a)
    mv = None
    if scratch is not None:
        mv = MapPointView()
        mv.m = m
        mv.in_view, mv.level, mv.view_cos = arr(scratch["in_view"], np.uint8), arr(scratch["level"], np.int32), arr(scratch["view_cos"], np.float32)
        mv.proj_x, mv.proj_y, mv.proj_xr = (arr(scratch["proj_x"], np.float32), arr(scratch["proj_y"], np.float32),
                                            arr(scratch["proj_xr"], np.float32))
    T = np.ascontiguousarray(Tcw, np.float32) if Tcw is not None else None
    out = np.zeros(max(n, 1), np.int32)
    nm = C.c_int32()
    trk, ts = None, None
    if want_track:
        trk = {"in_view": np.zeros(m, np.uint8), "proj_x": np.zeros(m, np.float32), "proj_y": np.zeros(m, np.float32),
               "proj_xr": np.zeros(m, np.float32), "view_cos": np.zeros(m, np.float32), "level": np.zeros(m, np.int32)}
        ts = TrackScratch()
        for k in trk:
            setattr(ts, k, _p(trk[k]))
    L.orbgpu_search_local_points_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                   C.c_void_p] + [C.c_float] * 9 + [C.c_void_p] * 4
    check(L.orbgpu_search_local_points_table(dframe.h, table.h, m, _p(ids), arr(skip, np.uint8),
                                             C.byref(mv) if mv is not None else None, _p(T) if T is not None else None,
                                             fx, fy, cx, cy, mbf, log_sf, cos_limit, th, nnratio, arr(kp_ids, np.int64), _p(out),
                                             C.byref(nm), C.byref(ts) if ts is not None else None))
    return (nm.value, out[:n], trk) if want_track else (nm.value, out[:n])


def search_by_projection_last_table(cur, cur_Tcw, last, last_Tcw, table, last_ids, fx, fy, cx, cy, mbf, mb, th, mono,
                                    check_ori, last_outlier=None, cur_kp_ids=None):
    """SearchByProjection(CurrentFrame, LastFrame, th, bMono) over the MapPoint table; cur / last: DeviceFrame."""
    L = lib()
    last_ids = np.ascontiguousarray(last_ids, np.int64)
    lo = np.ascontiguousarray(last_outlier, np.uint8) if last_outlier is not None else None
    ck = np.ascontiguousarray(cur_kp_ids, np.int64) if cur_kp_ids is not None else None
    Tc, Tl = np.ascontiguousarray(cur_Tcw, np.float32), np.ascontiguousarray(last_Tcw, np.float32)
    out = np.zeros(max(cur.n, 1), np.int32)
    nm = C.c_int32()
    L.orbgpu_search_by_projection_last_table.argtypes = [C.c_void_p] * 8 + [C.c_float] * 7 + [C.c_int32, C.c_int32, C.c_void_p,
                                                                                              C.c_void_p]
    check(L.orbgpu_search_by_projection_last_table(cur.h, _p(Tc), last.h, _p(Tl), table.h, _p(last_ids),
                                                   _p(lo) if lo is not None else None, _p(ck) if ck is not None else None,
                                                   fx, fy, cx, cy, mbf, mb, th, int(mono), int(check_ori), _p(out), C.byref(nm)))
    return nm.value, out[:cur.n]


def distinctive_descriptors(groups, device_id=0):
    """MapPoint::ComputeDistinctiveDescriptors for a list of (n_i, 32) descriptor arrays -> best index per group."""
    off = np.zeros(len(groups) + 1, np.int32)
    for i, g in enumerate(groups):
        off[i + 1] = off[i] + len(g)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(g, np.uint8).reshape(-1, 32) for g in groups])
                                if len(groups) else np.zeros((0, 32), np.uint8))
    out = np.zeros(max(1, len(groups)), np.int32)
    check(lib().orbgpu_distinctive_descriptors(len(groups), _p(off), _p(flat) if len(flat) else None, _p(out), device_id))
    return out[:len(groups)]


def projection_last_sweeps():
    """(claim sweeps, re-walked rows) of this thread's most recent projection match."""
    a, b = C.c_int32(), C.c_int32()
    check(lib().orbgpu_projection_last_sweeps(C.byref(a), C.byref(b)))
    return a.value, b.value


class Frame:
    """SoA view of the Frame members the matcher reads (reference include/Frame.h:100-190)."""

    def __init__(self, kp_x, kp_y, octave, angle, u_right, desc, width, height, scale_factors):
        f32 = np.float32
        self.kp_x = np.ascontiguousarray(kp_x, f32)
        self.kp_y = np.ascontiguousarray(kp_y, f32)
        self.octave = np.ascontiguousarray(octave, np.int32)
        self.angle = np.ascontiguousarray(angle, f32)
        self.u_right = np.ascontiguousarray(u_right, f32)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        self.n = len(self.kp_x)
        self.min_x, self.max_x, self.min_y, self.max_y = f32(0), f32(width), f32(0), f32(height)
        self.inv_w = f32(f32(GRID_COLS) / f32(self.max_x - self.min_x))  # Frame.cc:155
        self.inv_h = f32(f32(GRID_ROWS) / f32(self.max_y - self.min_y))  # Frame.cc:156
        self.scale_factors = np.ascontiguousarray(scale_factors, f32)
        self.cell_start, self.cell_items = assign_features_to_grid(self.kp_x, self.kp_y, self.min_x, self.min_y,
                                                                   self.inv_w, self.inv_h)
        if len(self.cell_items) == 0:
            self.cell_items = np.zeros(1, np.int32)

    def view(self):
        v = FrameView()
        v.n = self.n
        v.kp_x, v.kp_y, v.kp_octave, v.kp_angle = _p(self.kp_x), _p(self.kp_y), _p(self.octave), _p(self.angle)
        v.u_right, v.desc = _p(self.u_right), _p(self.desc)
        v.min_x, v.max_x, v.min_y, v.max_y = self.min_x, self.max_x, self.min_y, self.max_y
        v.grid_inv_w, v.grid_inv_h = self.inv_w, self.inv_h
        v.scale_factors, v.nlevels = _p(self.scale_factors), len(self.scale_factors)
        v.cell_start, v.cell_items = _p(self.cell_start), _p(self.cell_items)
        return v


class ORBmatcher:
    """Mirror of ORB_SLAM2::ORBmatcher(nnratio, checkOri) (ORBmatcher.h:41-106)."""

    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio=0.6, checkOri=True, device_id=0):
        self.L = lib()
        self.nnratio, self.check_ori, self.device_id = float(nnratio), bool(checkOri), device_id

    @staticmethod
    def DescriptorDistance(a, b, device_id=0):
        """ORBmatcher::DescriptorDistance (ORBmatcher.h:44); a, b: [n,32] or [32] uint8."""
        a = np.ascontiguousarray(a, np.uint8).reshape(-1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(-1, 32)
        out = np.zeros(len(a), np.int32)
        check(lib().orbgpu_hamming256(_p(a), _p(b), len(a), _p(out), device_id))
        return out

    def MatchBruteForce(self, desc_a, angle_a, desc_b, angle_b, valid_a=None, th_low=TH_LOW):
        """SearchByBoW(KF,F) with one vocabulary node (ORBmatcher.cc:159-288)."""
        desc_a = np.ascontiguousarray(desc_a, np.uint8).reshape(-1, 32)
        desc_b = np.ascontiguousarray(desc_b, np.uint8).reshape(-1, 32)
        angle_a = np.ascontiguousarray(angle_a, np.float32)
        angle_b = np.ascontiguousarray(angle_b, np.float32)
        va = None if valid_a is None else np.ascontiguousarray(valid_a, np.uint8)
        out = np.zeros(max(1, len(desc_b)), np.int32)
        n = C.c_int32()
        check(self.L.orbgpu_match_bf(_p(desc_a), _p(angle_a), _p(va), len(desc_a), _p(desc_b), _p(angle_b), len(desc_b),
                                     th_low, self.nnratio, int(self.check_ori), _p(out), C.byref(n), self.device_id))
        return n.value, out[:len(desc_b)].copy()

    def SearchByProjection(self, frame, mp, th, kp_to_mp):
        """SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:45-129).
        mp: dict of arrays in_view,bad,obs_pos,level,view_cos,proj_x,proj_y,proj_xr,desc."""
        keep = {k: np.ascontiguousarray(mp[k], dt) for k, dt in
                (("in_view", np.uint8), ("bad", np.uint8), ("obs_pos", np.uint8), ("level", np.int32),
                 ("view_cos", np.float32), ("proj_x", np.float32), ("proj_y", np.float32),
                 ("proj_xr", np.float32), ("desc", np.uint8))}
        v = MapPointView()
        v.m = len(keep["level"])
        for k in keep:
            setattr(v, k, _p(keep[k]))
        fv = frame.view()
        out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
        n = C.c_int32()
        check(self.L.orbgpu_search_by_projection(C.byref(fv), C.byref(v), th, self.nnratio, _p(out), C.byref(n),
                                                 self.device_id))
        return n.value, out

    def SearchByProjectionLast(self, cur, cur_Tcw, fx, fy, cx, cy, mbf, mb, last, th, mono, kp_to_mp):
        """SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1328-1470)."""
        keep = {k: np.ascontiguousarray(last[k], dt) for k, dt in
                (("has_mp", np.uint8), ("outlier", np.uint8), ("obs_pos", np.uint8), ("world_pos", np.float32),
                 ("desc", np.uint8), ("kp_octave", np.int32), ("kp_angle", np.float32), ("Tcw", np.float32))}
        v = LastFrameView()
        v.n = len(keep["kp_octave"])
        for k in keep:
            setattr(v, k, _p(keep[k]))
        fv = cur.view()
        T = np.ascontiguousarray(cur_Tcw, np.float32)
        out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
        n = C.c_int32()
        check(self.L.orbgpu_search_by_projection_last(C.byref(fv), _p(T), fx, fy, cx, cy, mbf, mb, C.byref(v), th,
                                                      int(mono), int(self.check_ori), _p(out), C.byref(n),
                                                      self.device_id))
        return n.value, out


    def SearchByProjectionKeyFrame(self, cur, cur_Tcw, fx, fy, cx, cy, log_sf, kf, th, orb_dist, kp_to_mp):
        """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1472-1599).
        kf: dict of arrays has_mp,bad,already_found,world_pos,min_dist_inv,max_dist_inv,max_dist,desc,kp_angle."""
        keep = {k: np.ascontiguousarray(kf[k], dt) for k, dt in
                (("has_mp", np.uint8), ("bad", np.uint8), ("already_found", np.uint8), ("world_pos", np.float32),
                 ("min_dist_inv", np.float32), ("max_dist_inv", np.float32), ("max_dist", np.float32),
                 ("desc", np.uint8), ("kp_angle", np.float32))}
        v = KeyFrameView()
        v.n = len(keep["has_mp"])
        for k in keep:
            setattr(v, k, _p(keep[k]))
        fv = cur.view()
        T = np.ascontiguousarray(cur_Tcw, np.float32)
        out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
        n = C.c_int32()
        check(self.L.orbgpu_search_by_projection_keyframe(C.byref(fv), _p(T), fx, fy, cx, cy, log_sf, C.byref(v), th,
                                                          int(orb_dist), int(self.check_ori), _p(out), C.byref(n),
                                                          self.device_id))
        return n.value, out


def search_by_projection_sim3(kf_frame, Scw, fx, fy, cx, cy, log_sf, pts, th, kp_to_mp, device_id=0):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:290-403).
    pts: dict of arrays bad, world_pos, normal, min_dist, max_dist, desc."""
    keep = {k: np.ascontiguousarray(pts[k], dt) for k, dt in
            (("bad", np.uint8), ("world_pos", np.float32), ("normal", np.float32), ("min_dist", np.float32),
             ("max_dist", np.float32), ("desc", np.uint8))}
    v = PointsView()
    v.m = len(keep["bad"])
    for k in keep:
        setattr(v, k, _p(keep[k]))
    fv = kf_frame.view()
    S = np.ascontiguousarray(Scw, np.float32)
    out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
    n = C.c_int32()
    check(lib().orbgpu_search_by_projection_sim3(C.byref(fv), _p(S), fx, fy, cx, cy, log_sf, C.byref(v), int(th), _p(out),
                                                 C.byref(n), device_id))
    return n.value, out


# ---- M6: background-thread matchers ---------------------------------------------------------------------------
_PTS_FIELDS = (("bad", np.uint8), ("world_pos", np.float32), ("normal", np.float32), ("min_dist", np.float32),
               ("max_dist", np.float32), ("desc", np.uint8))


def _points_view(cls, pts):
    keep = {k: np.ascontiguousarray(pts[k], dt) for k, dt in _PTS_FIELDS}
    v = cls()
    v.m = len(keep["bad"])
    for k in keep:
        setattr(v, k, _p(keep[k]))
    return v, keep


def fuse(kf_frame, Tcw, fx, fy, cx, cy, bf, log_sf, pts, th, inv_level_sigma2, device_id=0):
    """Candidate phase of ORBmatcher::Fuse(pKF, vpMapPoints, th) (ORBmatcher.cc:825-975): best_idx per point."""
    v, keep = _points_view(PointsView, pts)
    fv = kf_frame.view()
    T = np.ascontiguousarray(Tcw, np.float32)
    inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
    out = np.full(max(v.m, 1), -1, np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_fuse.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                                                            C.c_void_p, C.c_int32]
    check(L.orbgpu_fuse(C.byref(fv), _p(T), fx, fy, cx, cy, bf, log_sf, C.byref(v), th, _p(inv), _p(out), C.byref(n),
                        device_id))
    return n.value, out[:v.m]


def fuse_sim3(kf_frame, Scw, fx, fy, cx, cy, log_sf, pts, th, device_id=0):
    """Candidate phase of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (ORBmatcher.cc:977-1100)."""
    v, keep = _points_view(PointsView, pts)
    fv = kf_frame.view()
    S = np.ascontiguousarray(Scw, np.float32)
    out = np.full(max(v.m, 1), -1, np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_fuse_sim3.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 5 + [C.c_void_p, C.c_float, C.c_void_p,
                                                                                 C.c_void_p, C.c_int32]
    check(L.orbgpu_fuse_sim3(C.byref(fv), _p(S), fx, fy, cx, cy, log_sf, C.byref(v), th, _p(out), C.byref(n), device_id))
    return n.value, out[:v.m]


def search_by_sim3(kf1, kf2, T1w, T2w, s12, R12, t12, fx, fy, cx, cy, log_sf1, log_sf2, pts1, already1, pts2, already2,
                   th, device_id=0):
    """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1102-1326): match12 per key point of key frame 1."""
    v1, k1 = _points_view(PointsView, pts1)
    v2, k2 = _points_view(PointsView, pts2)
    f1, f2 = kf1.view(), kf2.view()
    A = [np.ascontiguousarray(a, np.float32) for a in (T1w, T2w, R12, t12)]
    a1 = None if already1 is None else np.ascontiguousarray(already1, np.uint8)
    a2 = None if already2 is None else np.ascontiguousarray(already2, np.uint8)
    out = np.full(max(v1.m, 1), -1, np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_search_by_sim3.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p] + [C.c_float] * 6 + \
        [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p, C.c_int32]
    check(L.orbgpu_search_by_sim3(C.byref(f1), C.byref(f2), _p(A[0]), _p(A[1]), s12, _p(A[2]), _p(A[3]), fx, fy, cx, cy,
                                  log_sf1, log_sf2, C.byref(v1), _p(a1), C.byref(v2), _p(a2), th, _p(out), C.byref(n),
                                  device_id))
    return n.value, out[:v1.m]


def search_for_triangulation(kf1, has_mp1, node1, kf2, has_mp2, node2, F12, ex, ey, level_sigma2_2, only_stereo,
                             check_ori, device_id=0):
    """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:657-823): match12 per key point of key frame 1."""
    f1, f2 = kf1.view(), kf2.view()
    h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
    n1, n2 = np.ascontiguousarray(node1, np.int32), np.ascontiguousarray(node2, np.int32)
    F = np.ascontiguousarray(F12, np.float32)
    sg = np.ascontiguousarray(level_sigma2_2, np.float32)
    out = np.full(max(kf1.n, 1), -1, np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_search_for_triangulation.argtypes = [C.c_void_p] * 7 + [C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                                                     C.c_void_p, C.c_void_p, C.c_int32]
    check(L.orbgpu_search_for_triangulation(C.byref(f1), _p(h1), _p(n1), C.byref(f2), _p(h2), _p(n2), _p(F), ex, ey,
                                            _p(sg), int(only_stereo), int(check_ori), _p(out), C.byref(n), device_id))
    return n.value, out[:kf1.n]


def search_for_initialization(f1, f2, prev_matched, window_size, nnratio=0.9, check_ori=True, device_id=0):
    """ORBmatcher::SearchForInitialization (ORBmatcher.cc:405-520): (nmatches, matches12, updated prev_matched)."""
    v1, v2 = f1.view(), f2.view()
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    out = np.full(max(f1.n, 1), -1, np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32,
                                                   C.c_void_p, C.c_void_p, C.c_int32]
    check(L.orbgpu_search_for_initialization(C.byref(v1), C.byref(v2), _p(pm), int(window_size), nnratio, int(check_ori),
                                             _p(out), C.byref(n), device_id))
    return n.value, out[:f1.n], pm


def search_by_bow_keyframes(desc1, angle1, valid1, node1, desc2, angle2, valid2, node2, nnratio=0.75, check_ori=True,
                            device_id=0):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (ORBmatcher.cc:522-655): match12 per key point of pKF1."""
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    a1, a2 = np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(angle2, np.float32)
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    nd1, nd2 = np.ascontiguousarray(node1, np.int32), np.ascontiguousarray(node2, np.int32)
    out = np.full(max(len(d1), 1), -1, np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_search_by_bow_keyframes.argtypes = [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_float,
                                                                                                   C.c_int32, C.c_void_p,
                                                                                                   C.c_void_p, C.c_int32]
    check(L.orbgpu_search_by_bow_keyframes(_p(d1), _p(a1), _p(v1), _p(nd1), len(d1), _p(d2), _p(a2), _p(v2), _p(nd2),
                                           len(d2), nnratio, int(check_ori), _p(out), C.byref(n), device_id))
    return n.value, out[:len(d1)]


class BatchMatcher:
    """Device-resident batched brute-force matcher (orbgpu_match_bf_batch_device)."""

    def __init__(self, max_pairs, cap, device_id=0):
        self.L = lib()
        h = C.c_void_p()
        check(self.L.orbgpu_matcher_create(device_id, max_pairs, cap, C.byref(h)))
        self.h, self.max_pairs, self.cap = h, max_pairs, cap

    def close(self):
        if getattr(self, "h", None):
            self.L.orbgpu_matcher_destroy(self.h)
            self.h = None

    __del__ = close

    def match(self, pairs, cap, d_desc_a, d_angle_a, d_valid_a, d_na, d_desc_b, d_angle_b, d_nb, angle_stride,
              th_low, nnratio, check_ori, d_match_b, d_nmatches, stream=0):
        check(self.L.orbgpu_match_bf_batch_device(self.h, pairs, cap, d_desc_a, d_angle_a, d_valid_a, d_na, d_desc_b,
                                                  d_angle_b, d_nb, angle_stride, th_low, nnratio, int(check_ori),
                                                  d_match_b, d_nmatches, stream))

    def last_sweeps(self, pairs):
        out = np.zeros(pairs, np.int32)
        check(self.L.orbgpu_matcher_last_sweeps(self.h, _p(out)))
        return out


# --------------------------------------------------------------------------------------------
# On-disk formats (Map::Save records, binary PCD)
# --------------------------------------------------------------------------------------------
def keyframe_record(kf_id, timestamp, Tcw, keys, desc, mappoint_index):
    """Bytes of Map::_WriteKeyFrame (Map.cc:133-183). keys: KEYPOINT_DTYPE array, mappoint_index: uint64 (2**64-1 = none)."""
    L = lib()
    L.orbgpu_keyframe_record_bytes.restype = C.c_size_t
    L.orbgpu_keyframe_record_bytes.argtypes = [C.c_int32]
    L.orbgpu_write_keyframe_record.argtypes = [C.c_uint64, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    keys = np.ascontiguousarray(keys, KEYPOINT_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    idx = np.ascontiguousarray(mappoint_index, np.uint64)
    T = np.ascontiguousarray(Tcw, np.float32)
    n = len(keys)
    out = np.zeros(L.orbgpu_keyframe_record_bytes(n), np.uint8)
    w = C.c_size_t()
    check(L.orbgpu_write_keyframe_record(kf_id, timestamp, _p(T), n, _p(keys) if n else None, _p(desc) if n else None,
                                         _p(idx) if n else None, _p(out), len(out), C.byref(w)))
    return out[:w.value].tobytes()


def mappoint_record(mp_id, world_pos):
    L = lib()
    L.orbgpu_write_mappoint_record.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    p = np.ascontiguousarray(world_pos, np.float32)
    out = np.zeros(20, np.uint8)
    w = C.c_size_t()
    check(L.orbgpu_write_mappoint_record(mp_id, _p(p), _p(out), 20, C.byref(w)))
    return out[:w.value].tobytes()


def pcd_binary_header(n):
    L = lib()
    L.orbgpu_pcd_binary_header.argtypes = [C.c_int64, C.c_char_p, C.c_size_t, C.c_void_p]
    buf = C.create_string_buffer(512)
    w = C.c_size_t()
    check(L.orbgpu_pcd_binary_header(n, buf, 512, C.byref(w)))
    return buf.raw[:w.value]


# --------------------------------------------------------------------------------------------
# ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, reference include/ORBVocabulary.h)
# --------------------------------------------------------------------------------------------
TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3
L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = range(6)


class ORBVocabulary:
    """The vocabulary tree on the device.  parent / is_leaf / desc / weight: one row per node in the order
    TemplatedVocabulary::loadFromTextFile creates them (TemplatedVocabulary.h:1348-1437)."""

    def __init__(self, k, L, parent, is_leaf, desc, weight, weighting=TF_IDF, scoring=L1_NORM, device_id=0):
        self.L_ = lib()
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        self.k, self.L, self.device_id = k, L, device_id
        h = C.c_void_p()
        fn = self.L_.orbgpu_vocabulary_create
        fn.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                       C.c_int32, C.c_int32, C.c_void_p]
        check(fn(k, L, len(self.parent), _p(self.parent), _p(self.is_leaf), _p(self.desc), _p(self.weight), weighting,
                 scoring, device_id, C.byref(h)))
        self.h = h
        self.L_.orbgpu_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 10
        self.L_.orbgpu_bow_transform_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.L_.orbgpu_vocabulary_destroy.argtypes = [C.c_void_p]
        self.L_.orbgpu_vocabulary_size.argtypes = [C.c_void_p, C.c_void_p]

    def close(self):
        if getattr(self, "h", None):
            self.L_.orbgpu_vocabulary_destroy(self.h)
            self.h = None

    __del__ = close

    def size(self):
        n = C.c_int32()
        check(self.L_.orbgpu_vocabulary_size(self.h, C.byref(n)))
        return n.value

    def transform(self, desc, levelsup=4):
        """Frame::ComputeBoW: dict with word_id, weight, node_id per feature, bow (ids, values) and the
        FeatureVector as (nodes, start, items)."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        wid, nid = np.zeros(m, np.int32), np.zeros(m, np.int32)
        wgt = np.zeros(m, np.float64)
        bid, bval = np.zeros(m, np.int32), np.zeros(m, np.float64)
        fvn, fvs, fvi = np.zeros(m, np.int32), np.zeros(m + 1, np.int32), np.zeros(m, np.int32)
        nb, nf = C.c_int32(), C.c_int32()
        check(self.L_.orbgpu_bow_transform(self.h, _p(desc), n, levelsup, _p(wid), _p(wgt), _p(nid), _p(bid), _p(bval),
                                           C.byref(nb), _p(fvn), _p(fvs), _p(fvi), C.byref(nf)))
        return {"word_id": wid[:n], "weight": wgt[:n], "node_id": nid[:n], "bow_ids": bid[:nb.value],
                "bow_vals": bval[:nb.value], "fv_nodes": fvn[:nf.value], "fv_start": fvs[:nf.value + 1],
                "fv_items": fvi[:fvs[nf.value]]}

    def transform_batch_device(self, d_desc, batch, cap, d_n, levelsup, d_word, d_weight, d_node, stream=0):
        check(self.L_.orbgpu_bow_transform_batch_device(self.h, d_desc, batch, cap, d_n, levelsup, d_word, d_weight,
                                                        d_node, stream))


def search_by_bow(desc_kf, angle_kf, valid_kf, node_kf, desc_f, angle_f, node_f, th_low=TH_LOW, nnratio=0.7,
                  check_ori=True, device_id=0):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) on per-feature node ids (orbgpu_search_by_bow)."""
    desc_kf = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32)
    desc_f = np.ascontiguousarray(desc_f, np.uint8).reshape(-1, 32)
    akf, af = np.ascontiguousarray(angle_kf, np.float32), np.ascontiguousarray(angle_f, np.float32)
    nkf, nf_ = np.ascontiguousarray(node_kf, np.int32), np.ascontiguousarray(node_f, np.int32)
    va = None if valid_kf is None else np.ascontiguousarray(valid_kf, np.uint8)
    out = np.zeros(max(1, len(desc_f)), np.int32)
    n = C.c_int32()
    L = lib()
    L.orbgpu_search_by_bow.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_int32]
    check(L.orbgpu_search_by_bow(_p(desc_kf), _p(akf), _p(va), _p(nkf), len(desc_kf), _p(desc_f), _p(af), _p(nf_),
                                 len(desc_f), th_low, nnratio, int(check_ori), _p(out), C.byref(n), device_id))
    return n.value, out[:len(desc_f)].copy()


# --------------------------------------------------------------------------------------------
# PointCloudMapping (reference include/PointCloudMap.h:41-88)
# --------------------------------------------------------------------------------------------
class PointCloudMapping:
    """Arithmetic of ORB_SLAM2::PointCloudMapping: insertKeyFrame -> back-project, transform,
    append, voxel-filter the global map (PointCloudMap.cc:204-262). Thread/condvar protocol and
    viewer stay on the host side of the C++ shim."""

    def __init__(self, resolution, device_id=0):
        self.L = lib()
        h = C.c_void_p()
        check(self.L.orbgpu_cloud_create(float(resolution), device_id, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.orbgpu_cloud_destroy(self.h)
            self.h = None

    __del__ = close

    def insertKeyFrame(self, depth, rgb, fx, fy, cx, cy, Tcw):
        depth = np.ascontiguousarray(depth, np.float32)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        T = np.ascontiguousarray(Tcw, np.float32)
        h, w = depth.shape
        check(self.L.orbgpu_cloud_insert(self.h, _p(depth), depth.strides[0] // 4, _p(rgb), rgb.strides[0], w, h, fx, fy,
                                         cx, cy, _p(T)))

    def insertKeyFrameDevice(self, d_depth, depth_stride, d_rgb, rgb_stride, w, h, fx, fy, cx, cy, Tcw):
        """Device-resident key frame (pointers into HBM, strides in floats / bytes)."""
        T = np.ascontiguousarray(Tcw, np.float32)
        check(self.L.orbgpu_cloud_insert_device(self.h, d_depth, depth_stride, d_rgb, rgb_stride, w, h, fx, fy, cx, cy,
                                                _p(T)))

    def set_profiling(self, on):
        check(self.L.orbgpu_cloud_set_profiling(self.h, int(on)))

    def last_insert_ms(self):
        v = C.c_float()
        check(self.L.orbgpu_cloud_last_insert_ms(self.h, C.byref(v)))
        return v.value

    def clear(self):
        self.L.orbgpu_cloud_clear.argtypes = [C.c_void_p]
        check(self.L.orbgpu_cloud_clear(self.h))

    def appendFiltered(self, depth, rgb, fx, fy, cx, cy, Tcw):
        """One iteration of the shutdown loop (PointCloudMap.cc:272-282)."""
        depth = np.ascontiguousarray(depth, np.float32)
        rgb = np.ascontiguousarray(rgb, np.uint8)
        T = np.ascontiguousarray(Tcw, np.float32)
        h, w = depth.shape
        self.L.orbgpu_cloud_append_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int32,
                                                        C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        check(self.L.orbgpu_cloud_append_filtered(self.h, _p(depth), depth.strides[0] // 4, _p(rgb), rgb.strides[0], w, h,
                                                  fx, fy, cx, cy, _p(T)))

    def remove_outliers(self, mean_k=50, stddev_mul=1.0):
        """sor.filter of the global map (PointCloudMap.cc:283-285); returns the number of points removed."""
        r = C.c_int64()
        check(self.L.orbgpu_cloud_remove_outliers(self.h, mean_k, float(stddev_mul), C.byref(r)))
        return r.value

    def save_pcd(self, path):
        self.L.orbgpu_cloud_save_pcd.argtypes = [C.c_void_p, C.c_char_p]
        check(self.L.orbgpu_cloud_save_pcd(self.h, path.encode()))

    def last_path(self):
        v = C.c_int32()
        check(self.L.orbgpu_cloud_last_path(self.h, C.byref(v)))
        return v.value

    def rebuild(self, depths, rgbs, fx, fy, cx, cy, Tcws):
        depths = [np.ascontiguousarray(d, np.float32) for d in depths]
        rgbs = [np.ascontiguousarray(r, np.uint8) for r in rgbs]
        Ts = [np.ascontiguousarray(t, np.float32) for t in Tcws]
        n = len(depths)
        h, w = depths[0].shape
        dp = (C.c_void_p * n)(*[d.ctypes.data for d in depths])
        rp = (C.c_void_p * n)(*[r.ctypes.data for r in rgbs])
        tp = (C.c_void_p * n)(*[t.ctypes.data for t in Ts])
        check(self.L.orbgpu_cloud_rebuild(self.h, n, dp, depths[0].strides[0] // 4, rp, rgbs[0].strides[0], w, h, fx, fy,
                                          cx, cy, tp))

    def size(self):
        n = C.c_int64()
        check(self.L.orbgpu_cloud_size(self.h, C.byref(n)))
        return n.value

    def download(self):
        n = self.size()
        out = np.zeros(max(1, n), POINT_DTYPE)
        got = C.c_int64()
        check(self.L.orbgpu_cloud_download(self.h, _p(out), max(1, n), C.byref(got)))
        return out[:got.value].copy()

    def last_overflow(self):
        o = C.c_int32()
        check(self.L.orbgpu_cloud_last_overflow(self.h, C.byref(o)))
        return bool(o.value)


def backproject(depth, rgb, fx, fy, cx, cy, Tcw=None, device_id=0):
    depth = np.ascontiguousarray(depth, np.float32)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = depth.shape
    cap = ((h + 2) // 3) * ((w + 2) // 3)
    out = np.zeros(max(1, cap), POINT_DTYPE)
    n = C.c_int64()
    T = None if Tcw is None else np.ascontiguousarray(Tcw, np.float32)
    check(lib().orbgpu_backproject(_p(depth), depth.strides[0] // 4, _p(rgb), rgb.strides[0], w, h, fx, fy, cx, cy,
                                   _p(T), _p(out), max(1, cap), C.byref(n), device_id))
    return out[:n.value].copy()


def statistical_outlier_removal(points, mean_k=50, stddev_mul=1.0, device_id=0):
    """pcl::StatisticalOutlierRemoval::filter: (kept points, mean neighbour distance of every input point)."""
    points = np.ascontiguousarray(points, POINT_DTYPE)
    out = np.zeros(max(1, len(points)), POINT_DTYPE)
    md = np.zeros(max(1, len(points)), np.float32)
    n = C.c_int64()
    check(lib().orbgpu_statistical_outlier_removal(_p(points), len(points), mean_k, float(stddev_mul), _p(out),
                                                   max(1, len(points)), C.byref(n), _p(md), device_id))
    return out[:n.value].copy(), md[:len(points)].copy()


def voxel_filter(points, resolution, device_id=0):
    points = np.ascontiguousarray(points, POINT_DTYPE)
    out = np.zeros(max(1, len(points)), POINT_DTYPE)
    n = C.c_int64()
    ov = C.c_int32()
    check(lib().orbgpu_voxel_filter(_p(points), len(points), float(resolution), _p(out), max(1, len(points)),
                                    C.byref(n), C.byref(ov), device_id))
    return out[:n.value].copy(), bool(ov.value)
