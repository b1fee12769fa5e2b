"""ctypes binding of libsendslam_orb.so (include/sendslam_orb.h).

This is plumbing over the C ABI: it owns no arithmetic.  There is no fallback of any kind:
if the shared library is missing or exports the wrong ABI, importing the symbols raises, and
without a HIP device `OrbContext()` raises `OrbError(SS_ERR_NO_DEVICE)`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libsendslam_orb.so")
SS_MAX_LEVELS = 16
EXPANDED_ROW_BYTES = 128  # one FP4 value (+1 / -1) per descriptor bit
ABI_VERSION = 5
SS_TRACK_DESC_STAYS_VALID = 1

SS_OK = 0
SS_ERR_INVALID_ARG, SS_ERR_NO_DEVICE, SS_ERR_HIP, SS_ERR_TOO_SMALL = -1, -2, -3, -4
SS_ERR_OVERFLOW, SS_ERR_NOT_CALIBRATED, SS_ERR_BAD_FRAME, SS_ERR_NO_MEMORY, SS_ERR_STATE = -5, -6, -7, -8, -9
SS_ERR_BUSY = -10

EXPORTS = ["ss_abi_version", "ss_orb_params_default", "ss_create", "ss_destroy", "ss_last_error",
           "ss_set_calibration", "ss_extract", "ss_extract_batch_device", "ss_get_batch_view", "ss_fetch_frame", "ss_match",
           "ss_match_device", "ss_match_batch_device", "ss_track", "ss_track_reset", "ss_synchronize", "ss_get_stream",
           "ss_profile_enable", "ss_profile_reset", "ss_stats", "ss_debug_fetch", "ss_debug_sort",
           "ss_match_pairs_device", "ss_expand_descriptors_device", "ss_match_expanded_device",
           "ss_match_partial_expanded_device", "ss_track_features", "ss_track_features_matched", "ss_match_partial_device", "ss_match_fold_device", "ss_wait_stream",
           "ss_pipe_create", "ss_pipe_destroy", "ss_pipe_last_error", "ss_pipe_acquire", "ss_pipe_submit",
           "ss_pipe_submit_frames", "ss_pipe_wait", "ss_pipe_poll", "ss_pipe_release", "ss_pipe_in_flight",
           "ss_match_fold_strided_device", "ss_xchg_create", "ss_xchg_destroy", "ss_xchg_last_error", "ss_xchg_status",
           "ss_xchg_allgather", "ss_xchg_broadcast", "ss_pipe_debug_inject_failure", "ss_stereo_exchange_match"]


class OrbParams(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("lapping_x0", C.c_int32),
                ("lapping_x1", C.c_int32), ("max_batch", C.c_int32), ("steer_fma", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("type", C.c_char * 16), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("k1", C.c_double), ("k2", C.c_double), ("p1", C.c_double),
                ("p2", C.c_double), ("width", C.c_int32), ("height", C.c_int32), ("fps", C.c_double),
                ("rgb", C.c_int32), ("th_depth", C.c_double), ("baseline", C.c_double),
                ("depth_map_factor", C.c_double)]


class FrameResult(C.Structure):
    _fields_ = [("n_keypoints", C.c_int32), ("camera_id", C.c_int32), ("timestamp", C.c_double),
                ("keypoints", C.c_void_p), ("descriptors", C.c_void_p),
                ("level_counts", C.c_int32 * SS_MAX_LEVELS)]


class BatchView(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("kp_capacity", C.c_int32), ("keypoints", C.c_void_p),
                ("descriptors", C.c_void_p), ("n_keypoints", C.c_void_p), ("level_counts", C.c_void_p),
                ("frame_error", C.c_void_p)]


class StageStats(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("mean_ms", C.c_double), ("median_ms", C.c_double), ("algorithmic_bytes", C.c_int64)]


class Pose(C.Structure):
    _fields_ = [("tracking_state", C.c_int32), ("camera_id", C.c_int32), ("timestamp", C.c_double),
                ("position", C.c_double * 3), ("quaternion", C.c_double * 4), ("n_keypoints", C.c_int32),
                ("n_matches", C.c_int32), ("n_inliers", C.c_int32), ("n_map_points", C.c_int32)]


class PipeConfig(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("batch", C.c_int32),
                ("depth", C.c_int32), ("match_mode", C.c_int32), ("match_th", C.c_int32), ("ratio_num", C.c_int32),
                ("ratio_den", C.c_int32), ("copy_threads", C.c_int32)]


class PipeSlot(C.Structure):
    _fields_ = [("slot", C.c_int32), ("pixels", C.c_void_p), ("row_stride", C.c_int64), ("frame_stride", C.c_int64)]


class PipeResult(C.Structure):
    _fields_ = [("slot", C.c_int32), ("n_frames", C.c_int32), ("kp_capacity", C.c_int32), ("sequence", C.c_uint64),
                ("status", C.c_void_p), ("camera_id", C.c_void_p), ("timestamp", C.c_void_p),
                ("n_keypoints", C.c_void_p), ("level_counts", C.c_void_p), ("keypoints", C.c_void_p),
                ("descriptors", C.c_void_p), ("match_idx", C.c_void_p), ("match_d1", C.c_void_p),
                ("match_d2", C.c_void_p), ("d_descriptors", C.c_void_p)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4")])


class OrbError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libsendslam_orb: {message} (status {code})")
        self.code = code
        self.message = message


_lib = None


def load():
    """Loads the shared library or raises; never substitutes anything for it."""
    global _lib, LIB_PATH
    if _lib is not None:
        return _lib
    if os.environ.get("SENDSLAM_LIB"):  # A/B builds of the same ABI (profiles/tools/*.sh); never a fallback
        LIB_PATH = os.environ["SENDSLAM_LIB"]
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `make -C send-slam_amd` "
                          f"(or __graft_entry__.build()); there is no CPU fallback")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7 and a
    # second copy (from /opt/rocm) cannot initialise the device after the first has.  Loading
    # torch first makes the dynamic loader resolve our NEEDED libamdhip64.so.7 to torch's copy,
    # so device pointers, streams and RCCL buffers are shared.  Without torch installed the
    # library simply uses /opt/rocm's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ImportError(f"{LIB_PATH} does not export {name}")
    lib.ss_last_error.restype = C.c_char_p
    lib.ss_last_error.argtypes = [C.c_void_p]
    lib.ss_create.argtypes = [C.c_int, C.POINTER(OrbParams), C.POINTER(C.c_void_p)]
    lib.ss_destroy.argtypes = [C.c_void_p]
    lib.ss_set_calibration.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera)]
    lib.ss_extract.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                               C.c_double, C.POINTER(FrameResult)]
    lib.ss_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int64, C.c_int64]
    lib.ss_get_batch_view.argtypes = [C.c_void_p, C.POINTER(BatchView)]
    lib.ss_fetch_frame.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameResult)]
    lib.ss_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ss_match_device.argtypes = lib.ss_match.argtypes
    lib.ss_match_batch_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    lib.ss_track.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                             C.POINTER(Pose)]
    lib.ss_track_reset.argtypes = [C.c_void_p]
    lib.ss_synchronize.argtypes = [C.c_void_p]
    lib.ss_get_stream.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.ss_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.ss_profile_reset.argtypes = [C.c_void_p]
    lib.ss_stats.argtypes = [C.c_void_p, C.POINTER(StageStats), C.c_int]
    lib.ss_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64]
    lib.ss_debug_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.ss_match_pairs_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ss_expand_descriptors_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.ss_match_expanded_device.argtypes = lib.ss_match.argtypes
    lib.ss_match_partial_expanded_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    lib.ss_track_features.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Pose)]
    lib.ss_track_features_matched.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                              C.POINTER(Pose)]
    lib.ss_match_partial_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
    lib.ss_match_fold_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ss_wait_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.ss_match_fold_strided_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_void_p, C.c_void_p, C.c_void_p]
    lib.ss_xchg_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.ss_xchg_destroy.argtypes = [C.c_void_p]
    lib.ss_xchg_last_error.restype = C.c_char_p
    lib.ss_xchg_last_error.argtypes = [C.c_void_p]
    lib.ss_xchg_status.argtypes = [C.c_void_p]
    lib.ss_xchg_allgather.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.ss_xchg_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.ss_stereo_exchange_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ss_pipe_create.argtypes = [C.c_int, C.POINTER(OrbParams), C.POINTER(Camera), C.POINTER(PipeConfig),
                                   C.POINTER(C.c_void_p)]
    lib.ss_pipe_destroy.argtypes = [C.c_void_p]
    lib.ss_pipe_last_error.restype = C.c_char_p
    lib.ss_pipe_last_error.argtypes = [C.c_void_p]
    lib.ss_pipe_acquire.argtypes = [C.c_void_p, C.POINTER(PipeSlot)]
    lib.ss_pipe_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ss_pipe_submit_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]
    lib.ss_pipe_wait.argtypes = [C.c_void_p, C.POINTER(PipeResult)]
    lib.ss_pipe_poll.argtypes = [C.c_void_p, C.POINTER(PipeResult)]
    lib.ss_pipe_release.argtypes = [C.c_void_p, C.c_int]
    lib.ss_pipe_in_flight.argtypes = [C.c_void_p]
    lib.ss_pipe_debug_inject_failure.argtypes = [C.c_void_p, C.c_int]
    if lib.ss_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI {lib.ss_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def default_params(**kw) -> OrbParams:
    p = OrbParams()
    load().ss_orb_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class OrbContext:
    """One extraction context = one HIP stream on one device (ss_create / ss_destroy)."""

    def __init__(self, device: int = 0, **params):
        self._lib = load()
        self.params = default_params(**params)
        h = C.c_void_p()
        rc = self._lib.ss_create(int(device), C.byref(self.params), C.byref(h))
        if rc != SS_OK:
            raise OrbError(rc, (self._lib.ss_last_error(None) or b"").decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ss_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc < 0:
            raise OrbError(rc, (self._lib.ss_last_error(self._h) or b"").decode())
        return rc

    def last_error(self) -> str:
        return (self._lib.ss_last_error(self._h) or b"").decode()

    # ---- calibration (the "calibration" message of the wire protocol) ----
    def set_calibration(self, camera_id: int, cam: Camera):
        self._check(self._lib.ss_set_calibration(self._h, int(camera_id), C.byref(cam)))

    # ---- host in / host out ----
    def extract(self, img: np.ndarray, camera_id: int = 1, timestamp: float = 0.0):
        """img: (H, W) or (H, W, C) uint8.  -> (keypoints KP_DTYPE[n], desc u8[n,32], level_counts)"""
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        res = FrameResult()
        self._check(self._lib.ss_extract(self._h, int(camera_id), img.ctypes.data, w, h, ch, w * ch,
                                         float(timestamp), C.byref(res)))
        n = res.n_keypoints
        kps = np.empty(n, KP_DTYPE)
        desc = np.empty((n, 32), np.uint8)
        if n:
            C.memmove(kps.ctypes.data, res.keypoints, n * KP_DTYPE.itemsize)
            C.memmove(desc.ctypes.data, res.descriptors, n * 32)
        return kps, desc, np.array(list(res.level_counts)[:self.params.n_levels])

    def match(self, q: np.ndarray, t: np.ndarray, th: int = 50, ratio_num: int = 9, ratio_den: int = 10,
              exclude_self: bool = False):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        nq, nt = len(q), len(t)
        idx = np.empty(nq, np.int32)
        d1 = np.empty(nq, np.uint16)
        d2 = np.empty(nq, np.uint16)
        self._check(self._lib.ss_match(self._h, q.ctypes.data, nq, t.ctypes.data if nt else None, nt, int(th),
                                       int(ratio_num), int(ratio_den), int(exclude_self), idx.ctypes.data,
                                       d1.ctypes.data, d2.ctypes.data))
        return idx, d1, d2

    # ---- pose (bounded monocular front-end; needs set_calibration) ----
    def track(self, img: np.ndarray, camera_id: int = 1, timestamp: float = 0.0) -> dict:
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape[:2]
        ch = 1 if img.ndim == 2 else img.shape[2]
        po = Pose()
        self._check(self._lib.ss_track(self._h, int(camera_id), img.ctypes.data, w, h, ch, w * ch, float(timestamp),
                                       C.byref(po)))
        return {"state": po.tracking_state, "camera_id": po.camera_id, "timestamp": po.timestamp,
                "position": np.array(list(po.position)), "quaternion": np.array(list(po.quaternion)),
                "n_keypoints": po.n_keypoints, "n_matches": po.n_matches, "n_inliers": po.n_inliers,
                "n_map_points": po.n_map_points}

    def track_features(self, d_desc: int, kps: np.ndarray, camera_id: int = 1, timestamp: float = 0.0) -> dict:
        """Pose step alone: descriptors already on the device (n x 32 B at d_desc), keypoints KP_DTYPE[n] on the host."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        po = Pose()
        self._check(self._lib.ss_track_features(self._h, int(camera_id), float(timestamp), C.c_void_p(d_desc),
                                                kps.ctypes.data, len(kps), C.byref(po)))
        return {"state": po.tracking_state, "camera_id": po.camera_id, "timestamp": po.timestamp,
                "position": np.array(list(po.position)), "quaternion": np.array(list(po.quaternion)),
                "n_keypoints": po.n_keypoints, "n_matches": po.n_matches, "n_inliers": po.n_inliers,
                "n_map_points": po.n_map_points}

    def track_features_matched(self, d_desc: int, kps: np.ndarray, match_idx=None, match_d1=None, desc_stays_valid: bool = False,
                               camera_id: int = 1, timestamp: float = 0.0) -> dict:
        """track_features with this frame's matches against the previous call's frame handed in (int32[n], uint16[n] or None)."""
        kps = np.ascontiguousarray(kps, KP_DTYPE)
        mi = None if match_idx is None else np.ascontiguousarray(match_idx, np.int32)
        md = None if match_d1 is None else np.ascontiguousarray(match_d1, np.uint16)
        if mi is not None and (len(mi) < len(kps) or md is None or len(md) < len(kps)):
            raise ValueError("match arrays shorter than the keypoints")
        po = Pose()
        self._check(self._lib.ss_track_features_matched(self._h, int(camera_id), float(timestamp), C.c_void_p(d_desc), kps.ctypes.data, len(kps),
                                                        None if mi is None else mi.ctypes.data, None if md is None else md.ctypes.data,
                                                        SS_TRACK_DESC_STAYS_VALID if desc_stays_valid else 0, C.byref(po)))
        return {"state": po.tracking_state, "camera_id": po.camera_id, "timestamp": po.timestamp,
                "position": np.array(list(po.position)), "quaternion": np.array(list(po.quaternion)),
                "n_keypoints": po.n_keypoints, "n_matches": po.n_matches, "n_inliers": po.n_inliers,
                "n_map_points": po.n_map_points}

    def track_reset(self):
        self._check(self._lib.ss_track_reset(self._h))

    # ---- device in / device out (pointers are raw device addresses, e.g. tensor.data_ptr()) ----
    def extract_batch_device(self, d_ptr: int, n_frames: int, width: int, height: int, channels: int = 1,
                             row_stride: Optional[int] = None, frame_stride: Optional[int] = None):
        row_stride = width * channels if row_stride is None else row_stride
        frame_stride = row_stride * height if frame_stride is None else frame_stride
        self._check(self._lib.ss_extract_batch_device(self._h, C.c_void_p(d_ptr), n_frames, width, height,
                                                      channels, row_stride, frame_stride))

    def batch_view(self) -> BatchView:
        v = BatchView()
        self._check(self._lib.ss_get_batch_view(self._h, C.byref(v)))
        return v

    def fetch_frame(self, frame: int):
        """Host copy of one frame of the last batch -> (keypoints, desc, level_counts)."""
        res = FrameResult()
        self._check(self._lib.ss_fetch_frame(self._h, int(frame), C.byref(res)))
        n = res.n_keypoints
        kps = np.empty(n, KP_DTYPE)
        desc = np.empty((n, 32), np.uint8)
        if n:
            C.memmove(kps.ctypes.data, res.keypoints, n * KP_DTYPE.itemsize)
            C.memmove(desc.ctypes.data, res.descriptors, n * 32)
        return kps, desc, np.array(list(res.level_counts)[:self.params.n_levels])

    def match_device(self, d_q: int, nq: int, d_t: int, nt: int, d_idx: int, d_d1: int, d_d2: int, th: int = 50,
                     ratio_num: int = 9, ratio_den: int = 10, exclude_self: bool = False):
        self._check(self._lib.ss_match_device(self._h, C.c_void_p(d_q), nq, C.c_void_p(d_t), nt, th, ratio_num,
                                              ratio_den, int(exclude_self), C.c_void_p(d_idx), C.c_void_p(d_d1),
                                              C.c_void_p(d_d2)))

    def match_batch_device(self, mode: int, d_idx: int, d_d1: int, d_d2: int, th: int = 50, ratio_num: int = 9,
                           ratio_den: int = 10):
        self._check(self._lib.ss_match_batch_device(self._h, mode, th, ratio_num, ratio_den, C.c_void_p(d_idx),
                                                    C.c_void_p(d_d1), C.c_void_p(d_d2)))

    def match_pairs_device(self, d_q: int, d_nq: int, d_t: int, d_nt: int, n_frames: int, rows_per_frame: int, d_idx: int,
                           d_d1: int, d_d2: int, th: int = 50, ratio_num: int = 9, ratio_den: int = 10):
        """n_frames independent (query frame, train frame) pairs: [n_frames][rows_per_frame][32] both sides."""
        self._check(self._lib.ss_match_pairs_device(self._h, C.c_void_p(d_q), C.c_void_p(d_nq), C.c_void_p(d_t), C.c_void_p(d_nt),
                                                    n_frames, rows_per_frame, th, ratio_num, ratio_den, C.c_void_p(d_idx),
                                                    C.c_void_p(d_d1), C.c_void_p(d_d2)))

    @staticmethod
    def expanded_bytes(n: int) -> int:
        """bytes of n descriptors in the matrix-core matcher's operand format (128 B per row, rows rounded up to 32)"""
        return ((n + 31) & ~31) * EXPANDED_ROW_BYTES

    def expand_descriptors_device(self, d_packed: int, n: int, d_expanded: int):
        self._check(self._lib.ss_expand_descriptors_device(self._h, C.c_void_p(d_packed), n, C.c_void_p(d_expanded)))

    def match_expanded_device(self, d_qx: int, nq: int, d_tx: int, nt: int, d_idx: int, d_d1: int, d_d2: int, th: int = 50,
                              ratio_num: int = 9, ratio_den: int = 10, exclude_self: bool = False):
        self._check(self._lib.ss_match_expanded_device(self._h, C.c_void_p(d_qx), nq, C.c_void_p(d_tx), nt, th, ratio_num,
                                                       ratio_den, int(exclude_self), C.c_void_p(d_idx), C.c_void_p(d_d1),
                                                       C.c_void_p(d_d2)))

    def match_partial_expanded_device(self, d_qx: int, nq: int, d_tx: int, nt: int, row_offset: int, d_part: int):
        self._check(self._lib.ss_match_partial_expanded_device(self._h, C.c_void_p(d_qx), nq, C.c_void_p(d_tx), nt,
                                                               int(row_offset), C.c_void_p(d_part)))

    def match_partial_device(self, d_q: int, nq: int, d_t: int, nt: int, row_offset: int, d_part: int):
        """Raw local match of a database shard -> nq 8-byte ss_match_part records (global rows) at d_part."""
        self._check(self._lib.ss_match_partial_device(self._h, C.c_void_p(d_q), nq, C.c_void_p(d_t), nt, int(row_offset),
                                                      C.c_void_p(d_part)))

    def match_fold_device(self, d_parts: int, n_parts: int, nq: int, d_idx: int, d_d1: int, d_d2: int, th: int = 50,
                          ratio_num: int = 9, ratio_den: int = 10):
        self._check(self._lib.ss_match_fold_device(self._h, C.c_void_p(d_parts), n_parts, nq, th, ratio_num, ratio_den,
                                                   C.c_void_p(d_idx), C.c_void_p(d_d1), C.c_void_p(d_d2)))

    def match_fold_strided_device(self, d_parts: int, n_parts: int, part_stride_bytes: int, nq: int, d_idx: int, d_d1: int,
                                  d_d2: int, th: int = 50, ratio_num: int = 9, ratio_den: int = 10):
        """the fold on parts part_stride_bytes apart: the layout Exchange.allgather leaves"""
        self._check(self._lib.ss_match_fold_strided_device(self._h, C.c_void_p(d_parts), n_parts, part_stride_bytes, nq, th, ratio_num,
                                                           ratio_den, C.c_void_p(d_idx), C.c_void_p(d_d1), C.c_void_p(d_d2)))

    def wait_stream(self, hip_stream: int):
        """Orders this context's stream after everything enqueued so far on another stream of the device."""
        self._check(self._lib.ss_wait_stream(self._h, C.c_void_p(hip_stream)))

    def synchronize(self):
        self._check(self._lib.ss_synchronize(self._h))

    def stream(self) -> int:
        s = C.c_void_p()
        self._check(self._lib.ss_get_stream(self._h, C.byref(s)))
        return s.value or 0

    def profile(self, on: bool):
        self._check(self._lib.ss_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._lib.ss_profile_reset(self._h))

    def stats(self):
        arr = (StageStats * 32)()
        n = self._check(self._lib.ss_stats(self._h, arr, 32))
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     mean_ms=arr[i].mean_ms, median_ms=arr[i].median_ms,
                     algorithmic_bytes=arr[i].algorithmic_bytes) for i in range(min(n, 32))]

    def debug_fetch(self, what: int, frame: int, level: int, shape, dtype=np.uint8) -> np.ndarray:
        out = np.empty(shape, dtype)
        n = self._check(self._lib.ss_debug_fetch(self._h, what, frame, level, out.ctypes.data, out.nbytes))
        return out.reshape(-1)[: n // out.itemsize]


class Exchange:
    """ss_xchg_*: the C ABI's own all-gather / broadcast between the ranks of one node (one process per GPU): every rank
    stores its block straight into every peer's IPC-mapped slab and raises a flag there.  Creation and destruction are
    collective; every rank issues the same sequence of messages; consumers of a gathered block are enqueued on the same
    context's stream."""

    def __init__(self, device: int, rank: int, world: int, max_bytes: int, rendezvous: str, timeout_ms: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.ss_xchg_create(int(device), int(rank), int(world), int(max_bytes), rendezvous.encode(), int(timeout_ms), C.byref(h))
        if rc != SS_OK:
            raise OrbError(rc, (self._lib.ss_xchg_last_error(None) or b"").decode())
        self._h = h
        self.rank, self.world = rank, world

    def _check(self, rc: int):
        if rc < 0:
            raise OrbError(rc, (self._lib.ss_xchg_last_error(self._h) or b"").decode())
        return rc

    def allgather(self, ctx: "OrbContext", segments):
        """segments: [(device pointer, bytes), ...] (<= 4), the same sizes on every rank.  Returns (base, stride): rank r's
        block is at base + r * stride in LOCAL device memory, the segments back to back (each padded to 16 bytes); valid
        until the second-next message."""
        n = len(segments)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p, _ in segments])
        sizes = (C.c_int64 * n)(*[int(b) for _, b in segments])
        base, stride = C.c_void_p(), C.c_int64()
        self._check(self._lib.ss_xchg_allgather(self._h, ctx._h, ptrs, sizes, n, C.byref(base), C.byref(stride)))
        return base.value, stride.value

    def stereo_match(self, ctx: "OrbContext", peer: int, kp_capacity: int, th: int = 50, ratio_num: int = 9, ratio_den: int = 10):
        """ss_stereo_exchange_match: the frame ctx extracted last against the peer rank's -> (idx, d1, d2 over this eye's
        keypoints, peer's keypoint count)"""
        idx = np.empty(kp_capacity, np.int32)
        d1 = np.empty(kp_capacity, np.uint16)
        d2 = np.empty(kp_capacity, np.uint16)
        n_own, n_peer = C.c_int32(), C.c_int32()
        rc = self._lib.ss_stereo_exchange_match(ctx._h, self._h, int(peer), th, ratio_num, ratio_den, idx.ctypes.data, d1.ctypes.data,
                                                d2.ctypes.data, C.byref(n_own), C.byref(n_peer))
        if rc < 0:
            raise OrbError(rc, (self._lib.ss_last_error(ctx._h) or b"").decode())
        n = n_own.value
        return idx[:n], d1[:n], d2[:n], n_peer.value

    def broadcast(self, ctx: "OrbContext", root: int, d_buf: int, nbytes: int):
        self._check(self._lib.ss_xchg_broadcast(self._h, ctx._h, int(root), C.c_void_p(d_buf), int(nbytes)))

    def status(self):
        """raises if a peer's message did not arrive in time (valid once the stream has been synchronised)"""
        self._check(self._lib.ss_xchg_status(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ss_xchg_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Pipe:
    """ss_pipe_*: ring of pinned slots; host frames in, host keypoints / descriptors / matches out, copies and
    kernels of different batches overlapped.  Arrays of a result are views of the slot's pinned memory: valid until
    release(slot)."""

    def __init__(self, device: int, width: int, height: int, channels: int = 1, batch: int = 64, depth: int = 4,
                 match_mode: int = 0, copy_threads: int = 0, cam: Optional[Camera] = None, match_th: int = 0, ratio_num: int = 0,
                 ratio_den: int = 0, **params):
        self._lib = load()
        self.params = default_params(**params)
        self.cfg = PipeConfig(width=width, height=height, channels=channels, batch=batch, depth=depth,
                              match_mode=match_mode, match_th=match_th, ratio_num=ratio_num, ratio_den=ratio_den, copy_threads=copy_threads)
        h = C.c_void_p()
        rc = self._lib.ss_pipe_create(int(device), C.byref(self.params), C.byref(cam) if cam is not None else None,
                                      C.byref(self.cfg), C.byref(h))
        if rc != SS_OK:
            raise OrbError(rc, (self._lib.ss_pipe_last_error(None) or b"").decode())
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.ss_pipe_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc < 0:
            raise OrbError(rc, (self._lib.ss_pipe_last_error(self._h) or b"").decode())
        return rc

    def acquire(self):
        """-> (slot id, uint8 view [batch, height, row_stride] of the slot's pinned pixels) or None when busy"""
        sl = PipeSlot()
        rc = self._lib.ss_pipe_acquire(self._h, C.byref(sl))
        if rc == SS_ERR_BUSY:
            return None
        self._check(rc)
        n = self.cfg.batch * sl.frame_stride
        buf = (C.c_uint8 * n).from_address(sl.pixels)
        return sl.slot, np.frombuffer(buf, np.uint8).reshape(self.cfg.batch, self.cfg.height, sl.row_stride)

    def submit(self, slot: int, n_frames: int, camera_ids=None, timestamps=None):
        ci = None if camera_ids is None else np.ascontiguousarray(camera_ids, np.int32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
        self._check(self._lib.ss_pipe_submit(self._h, slot, n_frames, None if ci is None else ci.ctypes.data,
                                             None if ts is None else ts.ctypes.data))

    def submit_frames(self, frames, camera_ids=None, timestamps=None, row_stride: Optional[int] = None) -> bool:
        """frames: sequence of uint8 arrays of the pipe's shape (None = a bad frame).  False when no slot is free."""
        keep = [None if f is None else np.ascontiguousarray(f, np.uint8) for f in frames]
        rs = self.cfg.width * self.cfg.channels if row_stride is None else row_stride
        need = (self.cfg.height - 1) * rs + self.cfg.width * self.cfg.channels  # what the library's copy threads read per frame
        for i, f in enumerate(keep):
            if f is not None and f.size < need:
                raise ValueError(f"frame {i}: {f.size} bytes, the pipe's shape needs {need} (height {self.cfg.height}, row stride {rs})")
        ptrs = (C.c_void_p * len(keep))(*[None if f is None else f.ctypes.data for f in keep])
        ci = None if camera_ids is None else np.ascontiguousarray(camera_ids, np.int32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
        rc = self._lib.ss_pipe_submit_frames(self._h, ptrs, len(keep), rs, None if ci is None else ci.ctypes.data,
                                             None if ts is None else ts.ctypes.data)
        if rc == SS_ERR_BUSY:
            return False
        self._check(rc)
        return True

    def submit_batch_array(self, batch: np.ndarray, camera_ids=None, timestamps=None) -> bool:
        """batch: one contiguous uint8 array [n, height, width(, channels)]: frame pointers without per-frame Python work"""
        assert batch.dtype == np.uint8 and batch.flags.c_contiguous
        want = (self.cfg.height, self.cfg.width) if self.cfg.channels == 1 and batch.ndim == 3 else (self.cfg.height, self.cfg.width, self.cfg.channels)
        if tuple(batch.shape[1:]) != want:
            raise ValueError(f"batch frames have shape {tuple(batch.shape[1:])}, the pipe's shape is {want}")
        n = batch.shape[0]
        fs = batch.strides[0]
        base = batch.ctypes.data
        ptrs = (C.c_void_p * n)(*[base + i * fs for i in range(n)])
        ci = None if camera_ids is None else np.ascontiguousarray(camera_ids, np.int32)
        ts = None if timestamps is None else np.ascontiguousarray(timestamps, np.float64)
        rc = self._lib.ss_pipe_submit_frames(self._h, ptrs, n, self.cfg.width * self.cfg.channels,
                                             None if ci is None else ci.ctypes.data, None if ts is None else ts.ctypes.data)
        if rc == SS_ERR_BUSY:
            return False
        self._check(rc)
        return True

    def _result(self, r: PipeResult) -> dict:
        n, k = r.n_frames, r.kp_capacity

        def view(ptr, dtype, shape):
            if not ptr:
                return None
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype).reshape(shape)
        return {"slot": r.slot, "n_frames": n, "kp_capacity": k, "sequence": r.sequence,
                "status": view(r.status, np.int32, (n,)), "camera_id": view(r.camera_id, np.int32, (n,)),
                "timestamp": view(r.timestamp, np.float64, (n,)), "n_keypoints": view(r.n_keypoints, np.int32, (n,)),
                "level_counts": view(r.level_counts, np.int32, (n, SS_MAX_LEVELS)),
                "keypoints": view(r.keypoints, KP_DTYPE, (n, k)), "descriptors": view(r.descriptors, np.uint8, (n, k, 32)),
                "match_idx": view(r.match_idx, np.int32, (n, k)), "match_d1": view(r.match_d1, np.uint16, (n, k)),
                "match_d2": view(r.match_d2, np.uint16, (n, k)), "d_descriptors": r.d_descriptors}

    def wait(self) -> dict:
        r = PipeResult()
        self._check(self._lib.ss_pipe_wait(self._h, C.byref(r)))
        return self._result(r)

    def poll(self) -> Optional[dict]:
        r = PipeResult()
        rc = self._check(self._lib.ss_pipe_poll(self._h, C.byref(r)))
        return self._result(r) if rc == 1 else None

    def release(self, slot: int):
        self._check(self._lib.ss_pipe_release(self._h, slot))

    def debug_inject_failure(self, after_operations: int):
        """test hook: the next submission fails after that many of its enqueues"""
        self._check(self._lib.ss_pipe_debug_inject_failure(self._h, int(after_operations)))

    def in_flight(self) -> int:
        return self._check(self._lib.ss_pipe_in_flight(self._h))


def _debug_sort(self, size, ulx):
    """Permutation (ids) the device's std::sort restatement leaves for compareNodes keys."""
    n = len(size)
    items = (np.asarray(size, np.uint64) << np.uint64(32)) | (np.asarray(ulx, np.uint64) << np.uint64(20)) | np.arange(n, dtype=np.uint64)
    items = np.ascontiguousarray(items)
    self._check(self._lib.ss_debug_sort(self._h, items.ctypes.data, n))
    return (items & np.uint64(0xFFFFF)).astype(np.int32)


OrbContext.debug_sort = _debug_sort
