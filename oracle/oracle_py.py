"""ctypes binding of the CPU oracle (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (orb_slam2_map_amd) never imports this module.
Parity status of the oracle itself: see oracle/orb_oracle.h ("parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_LEVELS = 16
GRID_COLS, GRID_ROWS = 64, 48

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")])
CORNER_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])
POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int), ("kp_x", C.c_void_p), ("kp_y", C.c_void_p), ("kp_octave", C.c_void_p),
                ("kp_angle", C.c_void_p), ("u_right", C.c_void_p), ("desc", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("scale_factors", C.c_void_p),
                ("nlevels", C.c_int), ("cell_start", C.c_void_p), ("cell_items", C.c_void_p)]


class MapPointView(C.Structure):
    _fields_ = [("m", C.c_int), ("in_view", C.c_void_p), ("bad", C.c_void_p), ("obs_pos", C.c_void_p),
                ("level", C.c_void_p), ("view_cos", C.c_void_p), ("proj_x", C.c_void_p),
                ("proj_y", C.c_void_p), ("proj_xr", C.c_void_p), ("desc", C.c_void_p)]


class KeyFrameView(C.Structure):
    _fields_ = [("n", C.c_int), ("has_mp", C.c_void_p), ("bad", C.c_void_p), ("already_found", C.c_void_p),
                ("world_pos", C.c_void_p), ("min_dist_inv", C.c_void_p), ("max_dist_inv", C.c_void_p),
                ("max_dist", C.c_void_p), ("desc", C.c_void_p), ("kp_angle", C.c_void_p)]


class LastFrameView(C.Structure):
    _fields_ = [("n", C.c_int), ("has_mp", C.c_void_p), ("outlier", C.c_void_p), ("obs_pos", C.c_void_p),
                ("world_pos", C.c_void_p), ("desc", C.c_void_p), ("kp_octave", C.c_void_p),
                ("kp_angle", C.c_void_p), ("Tcw", C.c_void_p)]


def build(out=None, extra_cflags=None):
    """Compile the oracle with gcc (building the checker is not using it)."""
    env = dict(os.environ)
    args = ["make", "-C", _HERE]
    if out:
        args.append("OUT=%s" % out)
    if extra_cflags:
        args.append("CFLAGS=%s" % extra_cflags)
    subprocess.run(args, check=True, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return out or os.path.join(_HERE, "liborb_oracle.so")


def lib(path=None):
    global _LIB
    if path is None and _LIB is not None:
        return _LIB
    p = path or os.path.join(_HERE, "liborb_oracle.so")
    if not os.path.exists(p):
        build()
    L = C.CDLL(p)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.ora_extractor_create.restype = vp
    L.ora_extractor_create.argtypes = [ci, cf, ci, ci, ci]
    L.ora_extractor_destroy.argtypes = [vp]
    for name in ("ora_get_scale_factors", "ora_get_inv_scale_factors", "ora_get_sigma2", "ora_get_inv_sigma2",
                 "ora_get_quotas", "ora_get_umax"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    L.ora_get_pattern.restype = vp
    L.ora_get_levels.argtypes = [vp]
    L.ora_extract.argtypes = [vp, vp, ci, ci, C.c_size_t, vp, vp, ci]
    L.ora_extractor_stage_seconds.argtypes = [vp, vp, ci]
    L.ora_extractor_stage_seconds.restype = None
    L.ora_pyramid_level.restype = vp
    L.ora_pyramid_level.argtypes = [vp, ci, vp, vp, vp]
    L.ora_blurred_level.restype = vp
    L.ora_blurred_level.argtypes = [vp, ci, vp, vp]
    L.ora_level_candidates.argtypes = [vp, ci, vp]
    L.ora_level_selected.argtypes = [vp, ci, vp]
    L.ora_resize_linear_u8.argtypes = [vp, ci, ci, C.c_size_t, vp, ci, ci, C.c_size_t]
    L.ora_border_reflect101_u8.argtypes = [vp, ci, ci, C.c_size_t, vp, C.c_size_t, ci]
    L.ora_gauss7_u8.argtypes = [vp, ci, ci, C.c_size_t, vp, C.c_size_t]
    L.ora_fast9_16.argtypes = [vp, ci, ci, C.c_size_t, ci, vp, ci]
    L.ora_distribute_octtree.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, ci]
    L.ora_ic_angle.restype = cf
    L.ora_ic_angle.argtypes = [vp, ci, vp]
    L.ora_fast_atan2.restype = cf
    L.ora_fast_atan2.argtypes = [cf, cf]
    L.ora_orb_descriptor.argtypes = [vp, ci, cf, vp]
    L.ora_set_trig_mode.argtypes = [ci]
    L.ora_set_trig_mode.restype = None
    L.ora_descriptor_trig.argtypes = [cf, vp, vp]
    L.ora_descriptor_trig.restype = None
    L.ora_descriptor_trig_array.argtypes = [vp, ci, vp, vp]
    L.ora_descriptor_trig_array.restype = None
    L.ora_descriptor_distance.argtypes = [vp, vp]
    L.ora_match_bf.argtypes = [vp, vp, vp, ci, vp, vp, ci, ci, cf, ci, vp]
    L.ora_assign_features_to_grid.argtypes = [ci, vp, vp, cf, cf, cf, cf, vp, vp]
    L.ora_get_features_in_area.argtypes = [vp, cf, cf, cf, ci, ci, vp]
    L.ora_search_by_projection.argtypes = [vp, vp, cf, cf, vp]
    L.ora_search_by_projection_last.argtypes = [vp, vp, cf, cf, cf, cf, cf, cf, vp, cf, ci, ci, vp]
    L.ora_search_by_projection_keyframe.argtypes = [vp, vp, cf, cf, cf, cf, cf, vp, cf, ci, ci, vp]
    L.ora_is_in_frustum.argtypes = [vp, cf, cf, cf, cf, cf, cf, cf, cf, cf, vp, vp, cf, cf, cf, ci, cf,
                                    vp, vp, vp, vp, vp]
    L.ora_compute_stereo_from_rgbd.argtypes = [ci, vp, vp, vp, vp, C.c_size_t, cf, vp, vp]
    L.ora_backproject.argtypes = [vp, C.c_size_t, vp, C.c_size_t, ci, ci, cf, cf, cf, cf, vp]
    L.ora_pose_inverse.argtypes = [vp, vp, vp]
    L.ora_transform_points.argtypes = [vp, ci, vp, vp, vp]
    L.ora_voxel_filter.argtypes = [vp, ci, cf, vp, vp]
    L.ora_statistical_outlier_removal.argtypes = [vp, ci, ci, C.c_double, vp, vp]
    L.ora_statistical_outlier_removal.restype = ci
    L.ora_vocabulary_create.restype = vp
    L.ora_vocabulary_create.argtypes = [ci, ci, ci, vp, vp, vp, vp, ci, ci]
    L.ora_vocabulary_destroy.argtypes = [vp]
    L.ora_vocabulary_words.argtypes = [vp]
    L.ora_bow_transform.argtypes = [vp, vp, ci, ci] + [vp] * 10
    L.ora_search_by_bow.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, ci, cf, ci, vp]
    if path is None:
        _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _arr(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Extractor:
    """Oracle ORBextractor (reference include/ORBextractor.h:46-110)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, libpath=None):
        self.L = lib(libpath)
        self.h = self.L.ora_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad extractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ora_extractor_destroy(self.h)
            self.h = None

    def scale_factors(self):
        return _arr(self.L.ora_get_scale_factors(self.h), self.nlevels, "<f4")

    def inv_scale_factors(self):
        return _arr(self.L.ora_get_inv_scale_factors(self.h), self.nlevels, "<f4")

    def sigma2(self):
        return _arr(self.L.ora_get_sigma2(self.h), self.nlevels, "<f4")

    def inv_sigma2(self):
        return _arr(self.L.ora_get_inv_sigma2(self.h), self.nlevels, "<f4")

    def quotas(self):
        return _arr(self.L.ora_get_quotas(self.h), self.nlevels, "<i4")

    def umax(self):
        return _arr(self.L.ora_get_umax(self.h), 16, "<i4")

    STAGES = ("pyramid", "fast", "quadtree", "orient", "blur", "describe")

    def stage_seconds(self, reset=True):
        """Seconds spent per stage since creation / the last reset (CPU-baseline aid)."""
        out = np.zeros(len(self.STAGES), np.float64)
        self.L.ora_extractor_stage_seconds(self.h, _p(out), int(reset))
        return dict(zip(self.STAGES, out.tolist()))

    def extract(self, gray):
        gray = np.ascontiguousarray(gray, dtype=np.uint8)
        h, w = gray.shape
        cap = self.nfeatures + 4 * self.nlevels + 64
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.ora_extract(self.h, _p(gray), w, h, gray.strides[0], _p(kps), _p(desc), cap)
        if n < 0:
            raise RuntimeError("oracle capacity exceeded")
        return kps[:n].copy(), desc[:n].copy()

    def pyramid_level(self, level):
        w, h, pitch = C.c_int(), C.c_int(), C.c_int()
        ptr = self.L.ora_pyramid_level(self.h, level, C.byref(w), C.byref(h), C.byref(pitch))
        a = _arr(ptr, pitch.value * (h.value + 38), np.uint8)
        return a.reshape(h.value + 38, pitch.value)

    def blurred_level(self, level):
        w, h = C.c_int(), C.c_int()
        ptr = self.L.ora_blurred_level(self.h, level, C.byref(w), C.byref(h))
        if not ptr:
            return None
        return _arr(ptr, w.value * h.value, np.uint8).reshape(h.value, w.value)

    def level_candidates(self, level):
        out = C.c_void_p()
        n = self.L.ora_level_candidates(self.h, level, C.byref(out))
        return _arr(out.value, n, CORNER_DTYPE)

    def level_selected(self, level):
        out = C.c_void_p()
        n = self.L.ora_level_selected(self.h, level, C.byref(out))
        return _arr(out.value, n, CORNER_DTYPE)


def pattern():
    return _arr(lib().ora_get_pattern(), 1024, np.int8)


TRIG_LIBM_FLOAT, TRIG_ROUNDED_DOUBLE = 0, 1


def set_trig_mode(mode):
    """ORBextractor.cc:112-113: TRIG_LIBM_FLOAT (default) = this host's cosf / sinf, what `using namespace std` makes the
    reference call; TRIG_ROUNDED_DOUBLE = (float)cos((double)angle)."""
    lib().ora_set_trig_mode(int(mode))


def descriptor_trig(angle_rad):
    a, b = C.c_float(), C.c_float()
    lib().ora_descriptor_trig(float(angle_rad), C.byref(a), C.byref(b))
    return np.float32(a.value), np.float32(b.value)


def descriptor_trig_array(angle_rad):
    x = np.ascontiguousarray(angle_rad, np.float32)
    a, b = np.zeros(len(x), np.float32), np.zeros(len(x), np.float32)
    lib().ora_descriptor_trig_array(_p(x), len(x), _p(a), _p(b))
    return a, b


def orb_descriptor(plane, x, y, angle_deg):
    """computeOrbDescriptor on the pixel (x, y) of a 2-D uint8 array (the caller guarantees the 18-px margin)."""
    plane = np.ascontiguousarray(plane, np.uint8)
    out = np.zeros(32, np.uint8)
    lib().ora_orb_descriptor(plane.ctypes.data + int(y) * plane.strides[0] + int(x), plane.strides[0], float(angle_deg), _p(out))
    return out


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().ora_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def border101(src, border):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * border, w + 2 * border), np.uint8)
    lib().ora_border_reflect101_u8(_p(src), w, h, src.strides[0], _p(dst), dst.strides[0], border)
    return dst


def gauss7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().ora_gauss7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    cap = max(1, w * h)
    out = np.zeros(cap, CORNER_DTYPE)
    n = lib().ora_fast9_16(_p(img), w, h, img.strides[0], threshold, _p(out), cap)
    return out[:n].copy()


def distribute(keys, min_x, max_x, min_y, max_y, n):
    keys = np.ascontiguousarray(keys, CORNER_DTYPE)
    cap = max(1, len(keys))
    out = np.zeros(cap, CORNER_DTYPE)
    r = lib().ora_distribute_octtree(_p(keys), len(keys), min_x, max_x, min_y, max_y, n, _p(out), cap)
    if r < 0:
        raise ValueError("invalid region")
    return out[:r].copy()


def descriptor_distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().ora_descriptor_distance(_p(a), _p(b))


def match_bf(desc_a, angle_a, desc_b, angle_b, valid_a=None, th_low=50, nnratio=0.7, check_orientation=True):
    desc_a = np.ascontiguousarray(desc_a, np.uint8)
    desc_b = np.ascontiguousarray(desc_b, np.uint8)
    angle_a = np.ascontiguousarray(angle_a, np.float32)
    angle_b = np.ascontiguousarray(angle_b, np.float32)
    va = None if valid_a is None else np.ascontiguousarray(valid_a, np.uint8)
    out = np.zeros(max(1, len(desc_b)), np.int32)
    n = lib().ora_match_bf(_p(desc_a), _p(angle_a), _p(va), len(desc_a), _p(desc_b), _p(angle_b), len(desc_b),
                           th_low, nnratio, int(check_orientation), _p(out))
    return n, out[:len(desc_b)].copy()


def assign_grid(kp_x, kp_y, min_x, min_y, inv_w, inv_h):
    kp_x = np.ascontiguousarray(kp_x, np.float32)
    kp_y = np.ascontiguousarray(kp_y, np.float32)
    cs = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32)
    items = np.zeros(max(1, len(kp_x)), np.int32)
    lib().ora_assign_features_to_grid(len(kp_x), _p(kp_x), _p(kp_y), min_x, min_y, inv_w, inv_h, _p(cs), _p(items))
    return cs, items[:cs[-1]].copy()


class Frame:
    """SoA view of the Frame members the matcher reads (reference include/Frame.h)."""

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
        # Frame.cc:155-156
        self.inv_w = f32(f32(GRID_COLS) / f32(self.max_x - self.min_x))
        self.inv_h = f32(f32(GRID_ROWS) / f32(self.max_y - self.min_y))
        self.scale_factors = np.ascontiguousarray(scale_factors, f32)
        self.cell_start, self.cell_items = assign_grid(self.kp_x, self.kp_y, self.min_x, self.min_y, self.inv_w,
                                                       self.inv_h)
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

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(1, self.n), np.int32)
        v = self.view()
        n = lib().ora_get_features_in_area(C.byref(v), x, y, r, min_level, max_level, _p(out))
        return out[:n].copy()


def search_by_projection(frame, mp, th, nnratio, kp_to_mp):
    """mp: dict of arrays in_view,bad,obs_pos,level,view_cos,proj_x,proj_y,proj_xr,desc."""
    keep = {k: np.ascontiguousarray(mp[k], dt) for k, dt in
            (("in_view", np.uint8), ("bad", np.uint8), ("obs_pos", np.uint8), ("level", np.int32),
             ("view_cos", np.float32), ("proj_x", np.float32), ("proj_y", np.float32), ("proj_xr", np.float32),
             ("desc", np.uint8))}
    v = MapPointView()
    v.m = len(keep["level"])
    for k in keep:
        setattr(v, k, _p(keep[k]))
    fv = frame.view()
    out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
    n = lib().ora_search_by_projection(C.byref(fv), C.byref(v), th, nnratio, _p(out))
    return n, out


def search_local_points(frame, Tcw, fx, fy, cx, cy, mbf, table, log_sf, th=3.0, nnratio=0.8, cos_limit=0.5, kp_to_mp=None,
                        libpath=None):
    """Tracking::SearchLocalPoints for a fresh frame (Tracking.cc:1447-1497).  table: dict with world_pos, normal, min_dist,
    max_dist, desc, skip, obs_pos.  Returns (nmatches, kp_to_mp, in_view, levels_out_of_range)."""
    L = lib(libpath)
    L.ora_search_local_points.argtypes = [C.c_void_p] * 2 + [C.c_float] * 5 + [C.c_int] + [C.c_void_p] * 7 + \
        [C.c_float] * 4 + [C.c_void_p] * 8
    keep = {k: np.ascontiguousarray(table[k], dt) for k, dt in
            (("world_pos", np.float32), ("normal", np.float32), ("min_dist", np.float32), ("max_dist", np.float32),
             ("skip", np.uint8), ("obs_pos", np.uint8), ("desc", np.uint8))}
    m = len(keep["min_dist"])
    in_view = np.zeros(m, np.uint8)
    px, py, pxr, vc = (np.zeros(m, np.float32) for _ in range(4))
    lvl = np.zeros(m, np.int32)
    out = np.full(frame.n, -1, np.int32) if kp_to_mp is None else np.ascontiguousarray(kp_to_mp, np.int32).copy()
    fv = frame.view()
    T = np.ascontiguousarray(Tcw, np.float32)
    nlo = C.c_int()
    n = L.ora_search_local_points(C.addressof(fv), _p(T), fx, fy, cx, cy, mbf, m, _p(keep["world_pos"]), _p(keep["normal"]),
                                  _p(keep["min_dist"]), _p(keep["max_dist"]), _p(keep["skip"]), _p(keep["obs_pos"]),
                                  _p(keep["desc"]), log_sf, cos_limit, th, nnratio, _p(in_view), _p(px), _p(py), _p(pxr),
                                  _p(lvl), _p(vc), _p(out), C.addressof(nlo))
    return n, out, in_view, nlo.value


def search_by_projection_last(cur, cur_Tcw, fx, fy, cx, cy, mbf, mb, last, th, mono, check_ori, kp_to_mp):
    """last: dict of arrays has_mp,outlier,obs_pos,world_pos,desc,kp_octave,kp_angle,Tcw."""
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
    n = lib().ora_search_by_projection_last(C.byref(fv), _p(T), fx, fy, cx, cy, mbf, mb, C.byref(v), th, int(mono),
                                            int(check_ori), _p(out))
    return n, out


KF_FIELDS = (("has_mp", np.uint8), ("bad", np.uint8), ("already_found", np.uint8), ("world_pos", np.float32),
             ("min_dist_inv", np.float32), ("max_dist_inv", np.float32), ("max_dist", np.float32), ("desc", np.uint8),
             ("kp_angle", np.float32))


def search_by_projection_keyframe(cur, cur_Tcw, fx, fy, cx, cy, log_sf, kf, th, orb_dist, check_ori, kp_to_mp):
    """kf: dict of arrays has_mp,bad,already_found,world_pos,min_dist_inv,max_dist_inv,max_dist,desc,kp_angle."""
    keep = {k: np.ascontiguousarray(kf[k], dt) for k, dt in KF_FIELDS}
    v = KeyFrameView()
    v.n = len(keep["has_mp"])
    for k in keep:
        setattr(v, k, _p(keep[k]))
    fv = cur.view()
    T = np.ascontiguousarray(cur_Tcw, np.float32)
    out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
    n = lib().ora_search_by_projection_keyframe(C.byref(fv), _p(T), fx, fy, cx, cy, log_sf, C.byref(v), th, int(orb_dist),
                                                int(check_ori), _p(out))
    return n, out


class _PointsView(C.Structure):
    _fields_ = [("m", C.c_int32), ("bad", C.c_void_p), ("world_pos", C.c_void_p), ("normal", C.c_void_p),
                ("min_dist", C.c_void_p), ("max_dist", C.c_void_p), ("desc", C.c_void_p)]


def search_by_projection_sim3(kf_frame, Scw, fx, fy, cx, cy, log_sf, pts, th, kp_to_mp):
    keep = {k: np.ascontiguousarray(pts[k], dt) for k, dt in
            (("bad", np.uint8), ("world_pos", np.float32), ("normal", np.float32), ("min_dist", np.float32),
             ("max_dist", np.float32), ("desc", np.uint8))}
    v = _PointsView()
    v.m = len(keep["bad"])
    for k in keep:
        setattr(v, k, _p(keep[k]))
    fv = kf_frame.view()
    S = np.ascontiguousarray(Scw, np.float32)
    out = np.ascontiguousarray(kp_to_mp, np.int32).copy()
    L = lib()
    L.ora_search_by_projection_sim3.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                                C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    L.ora_search_by_projection_sim3.restype = C.c_int
    n = L.ora_search_by_projection_sim3(C.byref(fv), _p(S), fx, fy, cx, cy, log_sf, C.byref(v), int(th), _p(out))
    return n, out


def is_in_frustum(Tcw, fx, fy, cx, cy, mbf, width, height, P, normal, min_dist, max_dist, log_sf, cos_limit=0.5):
    T = np.ascontiguousarray(Tcw, np.float32)
    P = np.ascontiguousarray(P, np.float32)
    nrm = np.ascontiguousarray(normal, np.float32)
    px, py, pxr, vc = C.c_float(), C.c_float(), C.c_float(), C.c_float()
    lvl = C.c_int32()
    ok = lib().ora_is_in_frustum(_p(T), fx, fy, cx, cy, mbf, 0.0, float(width), 0.0, float(height), _p(P), _p(nrm),
                                 min_dist, max_dist, log_sf, 0, cos_limit, C.byref(px), C.byref(py), C.byref(pxr),
                                 C.byref(lvl), C.byref(vc))
    return bool(ok), px.value, py.value, pxr.value, lvl.value, vc.value


def distinctive_descriptor(desc):
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    L = lib()
    L.ora_distinctive_descriptor.argtypes = [C.c_int, C.c_void_p]
    L.ora_distinctive_descriptor.restype = C.c_int
    return int(L.ora_distinctive_descriptor(len(desc), _p(desc)))


def undistort_points(xy, fx, fy, cx, cy, dist):
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    d = np.zeros(5, np.float32)
    d[:len(dist)] = dist
    out = np.zeros_like(xy)
    L = lib()
    L.ora_undistort_points.argtypes = [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                       C.c_void_p]
    L.ora_undistort_points.restype = None
    L.ora_undistort_points(len(xy), _p(xy), fx, fy, cx, cy, _p(d), _p(out))
    return out


def compute_stereo_from_rgbd(kp_x, kp_y, kpun_x, depth, mbf):
    kp_x = np.ascontiguousarray(kp_x, np.float32)
    kp_y = np.ascontiguousarray(kp_y, np.float32)
    kpun_x = np.ascontiguousarray(kpun_x, np.float32)
    depth = np.ascontiguousarray(depth, np.float32)
    n = len(kp_x)
    ur = np.zeros(max(1, n), np.float32)
    d = np.zeros(max(1, n), np.float32)
    lib().ora_compute_stereo_from_rgbd(n, _p(kp_x), _p(kp_y), _p(kpun_x), _p(depth), depth.strides[0] // 4, mbf,
                                       _p(ur), _p(d))
    return ur[:n], d[:n]


def backproject(depth, rgb, fx, fy, cx, cy):
    depth = np.ascontiguousarray(depth, np.float32)
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w = depth.shape
    cap = ((h + 2) // 3) * ((w + 2) // 3)
    out = np.zeros(max(1, cap), POINT_DTYPE)
    n = lib().ora_backproject(_p(depth), depth.strides[0] // 4, _p(rgb), rgb.strides[0], w, h, fx, fy, cx, cy, _p(out))
    return out[:n].copy()


def pose_inverse(Tcw):
    T = np.ascontiguousarray(Tcw, np.float32)
    R = np.zeros(9, np.float64)
    t = np.zeros(3, np.float64)
    lib().ora_pose_inverse(_p(T), _p(R), _p(t))
    return R, t


def transform_points(pts, R, t):
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    out = np.zeros(max(1, len(pts)), POINT_DTYPE)
    R = np.ascontiguousarray(R, np.float64)
    t = np.ascontiguousarray(t, np.float64)
    lib().ora_transform_points(_p(pts), len(pts), _p(R), _p(t), _p(out))
    return out[:len(pts)].copy()


def voxel_filter(pts, leaf):
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    out = np.zeros(max(1, len(pts)), POINT_DTYPE)
    ov = C.c_int(0)
    n = lib().ora_voxel_filter(_p(pts), len(pts), leaf, _p(out), C.byref(ov))
    return out[:n].copy(), bool(ov.value)


def statistical_outlier_removal(pts, mean_k=50, stddev_mul=1.0):
    """pcl::StatisticalOutlierRemoval (brute-force neighbours): (kept points, mean neighbour distance of every point)."""
    pts = np.ascontiguousarray(pts, POINT_DTYPE)
    out = np.zeros(max(1, len(pts)), POINT_DTYPE)
    md = np.zeros(max(1, len(pts)), np.float32)
    n = lib().ora_statistical_outlier_removal(_p(pts), len(pts), mean_k, stddev_mul, _p(out), _p(md))
    if n < 0:
        raise ValueError("statistical_outlier_removal: needs more than mean_k finite points")
    return out[:n].copy(), md[:len(pts)].copy()


class Vocabulary:
    """Oracle vocabulary tree (DBoW2 TemplatedVocabulary, TemplatedVocabulary.h)."""

    def __init__(self, k, L, parent, is_leaf, desc, weight, weighting=0, scoring=0):
        self.lib = lib()
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        self.h = self.lib.ora_vocabulary_create(k, L, len(self.parent), _p(self.parent), _p(self.is_leaf), _p(self.desc),
                                                _p(self.weight), weighting, scoring)
        if not self.h:
            raise ValueError("bad vocabulary")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ora_vocabulary_destroy(self.h)
            self.h = None

    def size(self):
        return self.lib.ora_vocabulary_words(self.h)

    def transform(self, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(desc)
        m = max(n, 1)
        wid, nid = np.zeros(m, np.int32), np.zeros(m, np.int32)
        wgt = np.zeros(m, np.float64)
        bid, bval = np.zeros(m, np.int32), np.zeros(m, np.float64)
        fvn, fvs, fvi = np.zeros(m, np.int32), np.zeros(m + 1, np.int32), np.zeros(m, np.int32)
        nb, nf = C.c_int(), C.c_int()
        self.lib.ora_bow_transform(self.h, _p(desc), n, levelsup, _p(wid), _p(wgt), _p(nid), _p(bid), _p(bval),
                                   C.byref(nb), _p(fvn), _p(fvs), _p(fvi), C.byref(nf))
        return {"word_id": wid[:n], "weight": wgt[:n], "node_id": nid[:n], "bow_ids": bid[:nb.value],
                "bow_vals": bval[:nb.value], "fv_nodes": fvn[:nf.value], "fv_start": fvs[:nf.value + 1],
                "fv_items": fvi[:fvs[nf.value]]}


def search_by_bow(desc_kf, angle_kf, valid_kf, fv_kf, desc_f, angle_f, fv_f, th_low=50, nnratio=0.7, check_ori=True):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...); fv_*: dicts from Vocabulary.transform."""
    desc_kf = np.ascontiguousarray(desc_kf, np.uint8).reshape(-1, 32)
    desc_f = np.ascontiguousarray(desc_f, np.uint8).reshape(-1, 32)
    akf, af = np.ascontiguousarray(angle_kf, np.float32), np.ascontiguousarray(angle_f, np.float32)
    va = None if valid_kf is None else np.ascontiguousarray(valid_kf, np.uint8)
    out = np.zeros(max(1, len(desc_f)), np.int32)
    a = [np.ascontiguousarray(fv_kf[k], np.int32) for k in ("fv_nodes", "fv_start", "fv_items")]
    b = [np.ascontiguousarray(fv_f[k], np.int32) for k in ("fv_nodes", "fv_start", "fv_items")]
    n = lib().ora_search_by_bow(_p(desc_kf), _p(akf), _p(va), len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), _p(desc_f), _p(af),
                                len(desc_f), len(b[0]), _p(b[0]), _p(b[1]), _p(b[2]), th_low, nnratio, int(check_ori),
                                _p(out))
    return n, out[:len(desc_f)].copy()


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


def fuse(kf_frame, Tcw, fx, fy, cx, cy, bf, log_sf, pts, th, inv_level_sigma2):
    v, keep = _points_view(_PointsView, pts)
    fv = kf_frame.view()
    T = np.ascontiguousarray(Tcw, np.float32)
    inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
    out = np.full(max(v.m, 1), -1, np.int32)
    L = lib()
    L.ora_fuse.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    n = L.ora_fuse(C.byref(fv), _p(T), fx, fy, cx, cy, bf, log_sf, C.byref(v), th, _p(inv), _p(out))
    return n, out[:v.m]


def fuse_sim3(kf_frame, Scw, fx, fy, cx, cy, log_sf, pts, th):
    v, keep = _points_view(_PointsView, pts)
    fv = kf_frame.view()
    S = np.ascontiguousarray(Scw, np.float32)
    out = np.full(max(v.m, 1), -1, np.int32)
    L = lib()
    L.ora_fuse_sim3.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 5 + [C.c_void_p, C.c_float, C.c_void_p]
    n = L.ora_fuse_sim3(C.byref(fv), _p(S), fx, fy, cx, cy, log_sf, C.byref(v), th, _p(out))
    return n, out[:v.m]


def search_by_sim3(kf1, kf2, T1w, T2w, s12, R12, t12, fx, fy, cx, cy, log_sf1, log_sf2, pts1, already1, pts2, already2, th):
    v1, k1 = _points_view(_PointsView, pts1)
    v2, k2 = _points_view(_PointsView, pts2)
    f1, f2 = kf1.view(), kf2.view()
    A = [np.ascontiguousarray(a, np.float32) for a in (T1w, T2w, R12, t12)]
    a1 = None if already1 is None else np.ascontiguousarray(already1, np.uint8)
    a2 = None if already2 is None else np.ascontiguousarray(already2, np.uint8)
    out = np.full(max(v1.m, 1), -1, np.int32)
    L = lib()
    L.ora_search_by_sim3.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p] + [C.c_float] * 6 + \
        [C.c_void_p] * 4 + [C.c_float, C.c_void_p]
    n = L.ora_search_by_sim3(C.byref(f1), C.byref(f2), _p(A[0]), _p(A[1]), s12, _p(A[2]), _p(A[3]), fx, fy, cx, cy, log_sf1,
                             log_sf2, C.byref(v1), _p(a1), C.byref(v2), _p(a2), th, _p(out))
    return n, out[:v1.m]


def _fv(t):
    return [np.ascontiguousarray(t[k], np.int32) for k in ("fv_nodes", "fv_start", "fv_items")]


def search_for_triangulation(kf1, has_mp1, fv1, kf2, has_mp2, fv2, F12, ex, ey, level_sigma2_2, only_stereo, check_ori):
    f1, f2 = kf1.view(), kf2.view()
    h1, h2 = np.ascontiguousarray(has_mp1, np.uint8), np.ascontiguousarray(has_mp2, np.uint8)
    a, b = _fv(fv1), _fv(fv2)
    F = np.ascontiguousarray(F12, np.float32)
    sg = np.ascontiguousarray(level_sigma2_2, np.float32)
    out = np.full(max(kf1.n, 1), -1, np.int32)
    L = lib()
    L.ora_search_for_triangulation.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    n = L.ora_search_for_triangulation(C.byref(f1), _p(h1), len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), C.byref(f2), _p(h2),
                                       len(b[0]), _p(b[0]), _p(b[1]), _p(b[2]), _p(F), ex, ey, _p(sg), int(only_stereo),
                                       int(check_ori), _p(out))
    return n, out[:kf1.n]


def search_by_bow_keyframes(desc1, angle1, valid1, fv1, desc2, angle2, valid2, fv2, nnratio=0.75, check_ori=True):
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    a1, a2 = np.ascontiguousarray(angle1, np.float32), np.ascontiguousarray(angle2, np.float32)
    v1 = None if valid1 is None else np.ascontiguousarray(valid1, np.uint8)
    v2 = None if valid2 is None else np.ascontiguousarray(valid2, np.uint8)
    a, b = _fv(fv1), _fv(fv2)
    out = np.full(max(len(d1), 1), -1, np.int32)
    L = lib()
    L.ora_search_by_bow_kf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    n = L.ora_search_by_bow_kf(_p(d1), _p(a1), _p(v1), len(d1), len(a[0]), _p(a[0]), _p(a[1]), _p(a[2]), _p(d2), _p(a2),
                               _p(v2), len(d2), len(b[0]), _p(b[0]), _p(b[1]), _p(b[2]), nnratio, int(check_ori), _p(out))
    return n, out[:len(d1)]


def search_for_initialization(f1, f2, prev_matched, window_size, nnratio=0.9, check_ori=True):
    v1, v2 = f1.view(), f2.view()
    pm = np.ascontiguousarray(prev_matched, np.float32).copy()
    out = np.full(max(f1.n, 1), -1, np.int32)
    L = lib()
    L.ora_search_for_initialization.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    n = L.ora_search_for_initialization(C.byref(v1), C.byref(v2), _p(pm), int(window_size), nnratio, int(check_ori), _p(out))
    return n, out[:f1.n], pm
