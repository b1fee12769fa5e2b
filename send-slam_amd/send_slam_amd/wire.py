"""The SEND-SLAM host <-> backend wire protocol, host half, in Python.

Mirrors the functions of send_slam/lib/send_slam/slam_handler.ex that touch the wire (same
names, argument meaning and packet shapes) so a Python process can play the Elixir host
against the front-door binary, and so tests pin the bytes:

    encode_payload/1            :283-291   u32 big-endian length + one MessagePack map
    extract_packets/2           :114-123   peel complete frames off a receive buffer
    build_frame_packet/3        :140-156   type "frame", PPM/PGM image as msgpack bin
    build_calibration_packet/3  :189-230   type "calibration", calibration -> camera map
    handle_incoming_packet/1    :125-137   only %{"type" => "pose"} is acted on
    encode_to_ppm/1             :275-281   Evision.imencode(".ppm", mat)

Backend half (what the shim sends): SendPosePacket, orbslam3_mono_networked.cc:225-282.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple

import msgpack
import numpy as np

MAX_MESSAGE = 50 * 1024 * 1024  # shim's kMaxMessageSize, orbslam3_mono_networked.cc:412


def encode_payload(payload: dict) -> bytes:
    packed = msgpack.packb(payload, use_bin_type=True)
    return struct.pack(">I", len(packed)) + packed


def extract_packets(buffer: bytes) -> Tuple[List[bytes], bytes]:
    out = []
    while len(buffer) >= 4:
        (n,) = struct.unpack(">I", buffer[:4])
        if len(buffer) - 4 < n:
            break
        out.append(buffer[4:4 + n])
        buffer = buffer[4 + n:]
    return out, buffer


def encode_to_ppm(mat: np.ndarray) -> bytes:
    """What Evision.imencode(".ppm", mat) writes: binary P6 for 3-channel BGR Mats (bytes
    stored R,G,B), binary P5 for 1-channel Mats."""
    mat = np.ascontiguousarray(mat, np.uint8)
    if mat.ndim == 2:
        h, w = mat.shape
        return b"P5\n%d %d\n255\n" % (w, h) + mat.tobytes()
    h, w, c = mat.shape
    if c != 3:
        raise ValueError("ppm_encode_failed: 1 or 3 channels")
    return b"P6\n%d %d\n255\n" % (w, h) + mat[:, :, ::-1].tobytes()


def build_frame_packet(image_file: bytes, dims: Dict[str, int], camera_id: int = 1,
                       timestamp: float = 0.0) -> bytes:
    return encode_payload({"type": "frame", "camera_id": camera_id, "encoding": "ppm",
                           "timestamp": float(timestamp), "width": dims["width"], "height": dims["height"],
                           "channels": dims["channels"], "frame": image_file})


def calibration_camera_payload(camera_matrix, distortion_coeffs, dims: Dict[str, int], fps=30) -> dict:
    m = [float(v) for v in np.asarray(camera_matrix, np.float64).reshape(-1)[:9]]
    fx, _, cx, _, fy, cy = m[:6]
    d = [float(v) for v in np.asarray(distortion_coeffs, np.float64).reshape(-1)] + [0.0] * 4
    return {"type": "PinHole", "fx": fx, "fy": fy, "cx": cx, "cy": cy, "k1": d[0], "k2": d[1], "p1": d[2],
            "p2": d[3], "width": dims["width"], "height": dims["height"], "fps": fps, "rgb": 1,
            "th_depth": 40.0, "baseline": 0.0, "depth_map_factor": 1000.0}


def build_calibration_packet(camera_matrix, distortion_coeffs, dims: Dict[str, int], camera_id: int = 1,
                             fps=30) -> bytes:
    camera = calibration_camera_payload(camera_matrix, distortion_coeffs, dims, fps)
    return encode_payload({"type": "calibration", "camera_id": camera_id, "calibration": {"camera": camera}})


def build_terminate_packet() -> bytes:
    return encode_payload({"type": "terminate"})


def handle_incoming_packet(payload: bytes) -> Optional[dict]:
    """Returns the string-keyed pose map that would be dispatched on PoseRegistry, or None."""
    try:
        decoded = msgpack.unpackb(payload, raw=False)
    except Exception:
        return None
    if isinstance(decoded, dict) and decoded.get("type") == "pose":
        return decoded
    return None


def pose_packet(timestamp: float, camera_id: int, tracking_state: int, position, orientation_xyzw) -> bytes:
    """Backend half: SendPosePacket's six-key map, doubles, world-from-camera."""
    x, y, z = (float(v) for v in position)
    qx, qy, qz, qw = (float(v) for v in orientation_xyzw)
    return encode_payload({"type": "pose", "timestamp": float(timestamp), "camera_id": int(camera_id),
                           "tracking_state": int(tracking_state), "position": {"x": x, "y": y, "z": z},
                           "orientation": {"x": qx, "y": qy, "z": qz, "w": qw}})
