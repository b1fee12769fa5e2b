"""ctypes loader for the CPU oracle (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT: import only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  PARITY UNPINNED (oracle/orb_constants.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborb_oracle.so")
MAX_LEVELS = 16


class Params(C.Structure):
    _fields_ = [("n_features", C.c_int), ("scale_factor", C.c_float), ("n_levels", C.c_int),
                ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int),
                ("lapping_x0", C.c_int), ("lapping_x1", C.c_int), ("steer_fma", C.c_int)]


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int)]


class Point(C.Structure):
    _fields_ = [("x", C.c_int), ("y", C.c_int), ("response", C.c_int)]


class Geometry(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS),
                ("scale", C.c_float * MAX_LEVELS), ("inv_scale", C.c_float * MAX_LEVELS),
                ("quota", C.c_int * MAX_LEVELS), ("umax", C.c_int * 16)]


class SortItem(C.Structure):
    _fields_ = [("size", C.c_int), ("ulx", C.c_int), ("id", C.c_int)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])
PT_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("response", "<i4")])

_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_fast_atan2.restype = C.c_float
        _lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        _lib.orc_ic_angle.restype = C.c_float
    return _lib


def _u8p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def default_params(**kw) -> Params:
    p = Params()
    lib().orc_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def geometry(p: Params, w: int, h: int) -> Geometry:
    g = Geometry()
    rc = lib().orc_geometry_init(C.byref(g), C.byref(p), w, h)
    if rc != 0:
        raise ValueError(f"orc_geometry_init failed: {rc}")
    return g


def gray(src: np.ndarray, rgb: int) -> np.ndarray:
    h, w, c = src.shape
    src = np.ascontiguousarray(src)
    out = np.empty((h, w), np.uint8)
    lib().orc_gray(_u8p(src), w, h, c, w * c, int(rgb), _u8p(out))
    return out


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    out = np.empty((dh, dw), np.uint8)
    lib().orc_resize_linear(_u8p(src), src.shape[1], src.shape[0], _u8p(out), dw, dh)
    return out


def pyramid(img: np.ndarray, p: Params):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    g = geometry(p, w, h)
    levels = [np.empty((g.h[l], g.w[l]), np.uint8) for l in range(g.n_levels)]
    ptrs = (C.POINTER(C.c_uint8) * g.n_levels)(*[_u8p(a) for a in levels])
    lib().orc_pyramid(_u8p(img), w, C.byref(g), ptrs)
    return levels


def fast_score_map(img: np.ndarray, threshold: int) -> np.ndarray:
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_fast_score_map(_u8p(img), img.shape[1], img.shape[0], int(threshold), _u8p(out))
    return out


def candidates(img: np.ndarray, ini_th: int, min_th: int) -> np.ndarray:
    img = np.ascontiguousarray(img)
    h, w = img.shape
    cap = w * h // 2 + 16
    out = np.empty(cap, PT_DTYPE)
    n = lib().orc_candidates(_u8p(img), w, h, int(ini_th), int(min_th),
                             out.ctypes.data_as(C.POINTER(Point)), cap)
    if n < 0:
        raise RuntimeError("orc_candidates overflow")
    return out[:n].copy()


def distribute(cand: np.ndarray, w: int, h: int, n_wanted: int) -> np.ndarray:
    """cand: PT_DTYPE relative to the (16,16) border origin; w, h: level size."""
    cand = np.ascontiguousarray(cand)
    cap = max(n_wanted + 8, 16) + 4 * 8
    out = np.empty(cap, PT_DTYPE)
    n = lib().orc_distribute(cand.ctypes.data_as(C.POINTER(Point)), len(cand), 16, w - 16, 16,
                             h - 16, int(n_wanted), out.ctypes.data_as(C.POINTER(Point)), cap)
    if n < 0:
        raise RuntimeError(f"orc_distribute failed: {n}")
    return out[:n].copy()


def fast_atan2(y: float, x: float) -> float:
    return float(lib().orc_fast_atan2(C.c_float(y), C.c_float(x)))


def ic_angle(img: np.ndarray, x: int, y: int, umax) -> float:
    img = np.ascontiguousarray(img)
    um = (C.c_int * 16)(*list(umax))
    return float(lib().orc_ic_angle(_u8p(img), img.shape[1], int(x), int(y), um))


def blur(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().orc_blur(_u8p(img), img.shape[1], img.shape[0], _u8p(out))
    return out


def descriptor(blurred: np.ndarray, x: int, y: int, angle_deg: float, steer_fma: int = 0) -> np.ndarray:
    blurred = np.ascontiguousarray(blurred)
    out = np.empty(32, np.uint8)
    lib().orc_descriptor_ex(_u8p(blurred), blurred.shape[1], int(x), int(y), C.c_float(angle_deg), int(steer_fma),
                            _u8p(out))
    return out


def extract(img: np.ndarray, p: Params):
    """-> (keypoints KP_DTYPE[n], descriptors u8[n,32], level_counts int[n_levels])"""
    img = np.ascontiguousarray(img)
    h, w = img.shape
    cap = 4 * p.n_features + 64 * p.n_levels  # a level may return up to 4 x its quota (orb_oracle.c, orc_extract)
    kps = np.empty(cap, KP_DTYPE)
    desc = np.empty((cap, 32), np.uint8)
    counts = (C.c_int * MAX_LEVELS)()
    n = lib().orc_extract(_u8p(img), w, h, w, C.byref(p), kps.ctypes.data_as(C.POINTER(Keypoint)),
                          _u8p(desc), cap, counts)
    if n < 0:
        raise RuntimeError(f"orc_extract failed: {n}")
    return kps[:n].copy(), desc[:n].copy(), np.array(list(counts)[:p.n_levels])


def match(q: np.ndarray, t: np.ndarray, th: int = 50, ratio_num: int = 9, ratio_den: int = 10,
          exclude_self: bool = False):
    q, t = np.ascontiguousarray(q), np.ascontiguousarray(t)
    nq, nt = len(q), len(t)
    idx = np.empty(nq, np.int32)
    d1 = np.empty(nq, np.uint16)
    d2 = np.empty(nq, np.uint16)
    lib().orc_match(_u8p(q), nq, _u8p(t), nt, int(th), int(ratio_num), int(ratio_den),
                    int(exclude_self), idx.ctypes.data_as(C.POINTER(C.c_int32)),
                    d1.ctypes.data_as(C.POINTER(C.c_uint16)), d2.ctypes.data_as(C.POINTER(C.c_uint16)))
    return idx, d1, d2


def std_sort(size: np.ndarray, ulx: np.ndarray) -> np.ndarray:
    """Returns the permutation (ids) libstdc++ std::sort leaves for compareNodes."""
    n = len(size)
    items = (SortItem * n)()
    for i in range(n):
        items[i].size, items[i].ulx, items[i].id = int(size[i]), int(ulx[i]), i
    lib().orc_std_sort(items, n)
    return np.array([items[i].id for i in range(n)], np.int32)


def pnp_pose_only(pts3d, obs, inv_sigma2, fx, fy, cx, cy, R0=None, t0=None):
    """Pose-only optimisation (downstream of the path; test infrastructure).  -> (R 3x3, t 3, inlier mask)"""
    pts3d = np.ascontiguousarray(pts3d, np.float64)
    obs = np.ascontiguousarray(obs, np.float64)
    inv_sigma2 = np.ascontiguousarray(inv_sigma2, np.float64)
    n = len(pts3d)
    R = np.ascontiguousarray(np.eye(3) if R0 is None else R0, np.float64).copy()
    t = np.ascontiguousarray(np.zeros(3) if t0 is None else t0, np.float64).copy()
    inl = np.zeros(n, np.uint8)
    f = lib().orc_pnp_pose_only
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double,
                  C.c_void_p, C.c_void_p, C.c_void_p]
    rc = f(n, pts3d.ctypes.data, obs.ctypes.data, inv_sigma2.ctypes.data, fx, fy, cx, cy, R.ctypes.data, t.ctypes.data,
           inl.ctypes.data)
    if rc < 0:
        raise RuntimeError(f"orc_pnp_pose_only failed: {rc}")
    return R, t, inl.astype(bool)
