"""ReplayProducer -- synthetic / replay frame source.

Python mirror of send-slam_amd/nif/replay_producer.ex, which is modelled on the image-sequence mode
of the reference's VideoProducer (send_slam/lib/send_slam/video_producer.ex:212-245 printf-style
`%06d` patterns, :303-326 read / pace / loop / warm-up; usage sketch application.ex:60-72).  Same
observable behaviour:

  * `pattern` with one `%d` / `%0Nd` / `%Nd` field names the files; the sequence starts at index 0,
    or at 1 when file 0 does not exist (cv::CAP_IMAGES accepts both, :219-226);
  * every frame is delivered as the keyword list VideoProducer broadcasts (:345-357):
    frame, calibration, timestamp, fps, camera_id;
  * `warmup_ms`: the first frame is re-delivered every frame interval until the warm-up time has
    passed (:310-343), so a backend that is still starting up does not miss the sequence start;
  * `fps` paces delivery (sleep of one frame interval after each frame); `loop` rewinds at the end,
    otherwise the producer stops with reason "eof".

Frames are PGM (P5) / PPM (P6) files -- what SlamHandler itself puts on the wire
(slam_handler.ex:275-281) -- read without any imaging library.  `write_sequence` makes such a
directory from numpy frames (e.g. send_slam_amd.synth).
"""
from __future__ import annotations

import os
import re
import time
from typing import Callable, Dict, Iterator, List, Optional

import numpy as np

_FIELD = re.compile(r"%0?(\d*)d")


def format_sequence_filename(pattern: str, index: int) -> str:
    """`frame_%06d.pgm`, 7 -> `frame_000007.pgm`; `%d` and `%3d` are zero-padded to their width too."""
    m = _FIELD.search(pattern)
    if not m:
        raise ValueError(f"{pattern!r} holds no %d field")
    width = int(m.group(1)) if m.group(1) else 0
    return pattern[:m.start()] + str(index).rjust(width, "0") + pattern[m.end():]


def looks_like_image_sequence(path: str) -> bool:
    return _FIELD.search(path) is not None


def write_pnm(path: str, img: np.ndarray) -> None:
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        head = b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0])
    elif img.ndim == 3 and img.shape[2] == 3:
        head = b"P6\n%d %d\n255\n" % (img.shape[1], img.shape[0])
    else:
        raise ValueError("PGM / PPM hold 1 or 3 channels")
    with open(path, "wb") as f:
        f.write(head + img.tobytes())


def read_pnm(path: str) -> np.ndarray:
    data = open(path, "rb").read()
    m = re.match(rb"(P[56])\s+(?:#[^\n]*\n\s*)*(\d+)\s+(\d+)\s+(\d+)\s", data)
    if not m or int(m.group(4)) != 255:
        raise ValueError(f"{path}: not an 8-bit binary PGM / PPM")
    w, h, ch = int(m.group(2)), int(m.group(3)), 1 if m.group(1) == b"P5" else 3
    body = np.frombuffer(data, np.uint8, w * h * ch, m.end())
    return body.reshape((h, w) if ch == 1 else (h, w, 3)).copy()


def write_sequence(directory: str, frames, pattern: str = "frame_%06d.pgm", start: int = 0) -> str:
    os.makedirs(directory, exist_ok=True)
    for i, f in enumerate(frames):
        write_pnm(os.path.join(directory, format_sequence_filename(pattern, start + i)), f)
    return os.path.join(directory, pattern)


class ReplayProducer:
    def __init__(self, video_path: str, fps: float = 30, loop: bool = False, warmup_ms: int = 0, camera_id: int = 1,
                 calibration: Optional[dict] = None, clock: Callable[[], float] = time.monotonic,
                 sleep: Callable[[float], None] = time.sleep):
        if not looks_like_image_sequence(video_path):
            raise ValueError("ReplayProducer replays image sequences: the path needs a %d field")
        first = [i for i in (0, 1) if os.path.exists(format_sequence_filename(video_path, i))]
        if not first:
            raise FileNotFoundError(f"no frame 0 or 1 for {video_path}")
        self.pattern, self.first_index = video_path, first[0]
        self.fps, self.loop, self.warmup_ms, self.camera_id = fps, loop, warmup_ms, camera_id
        self.calibration = calibration
        self.interval_s = 1.0 / fps if fps and fps > 0 else 0.0
        self._clock, self._sleep = clock, sleep
        self.stop_reason: Optional[str] = None
        self.frames_sent = 0

    def set_calibration(self, calibration: dict) -> None:
        """the {:broadcast_message, {:calibration, data}} message (video_producer.ex:79-83)"""
        self.calibration = calibration

    def _event(self, mat: np.ndarray) -> Dict:
        self.frames_sent += 1
        return {"frame": mat, "calibration": self.calibration, "timestamp": self._clock(), "fps": self.fps,
                "camera_id": self.camera_id}

    def events(self, max_frames: Optional[int] = None) -> Iterator[Dict]:
        """Yields the frame events in delivery order, sleeping one frame interval after each."""
        index, first = self.first_index, True
        while max_frames is None or self.frames_sent < max_frames:
            path = format_sequence_filename(self.pattern, index)
            if not os.path.exists(path):
                if self.loop and index != self.first_index:
                    index, first = self.first_index, True
                    continue
                self.stop_reason = "eof"
                return
            mat = read_pnm(path)
            if first and self.warmup_ms > 0:
                t0, tick = self._clock(), self.interval_s if self.interval_s > 0 else 0.010
                while (self._clock() - t0) * 1000.0 < self.warmup_ms and (max_frames is None or self.frames_sent < max_frames):
                    yield self._event(mat)
                    self._sleep(tick)
                if max_frames is not None and self.frames_sent >= max_frames:
                    return
            yield self._event(mat)
            if self.interval_s > 0:
                self._sleep(self.interval_s)
            index, first = index + 1, False


def drive_frontdoor(send: Callable[[bytes], None], producer: ReplayProducer, max_frames: Optional[int] = None) -> List[float]:
    """What SlamHandler does with the producer's events (slam_handler.ex:59-88): PNM-encode the Mat and send a
    `frame` packet; returns the timestamps sent."""
    from . import wire
    stamps = []
    for ev in producer.events(max_frames):
        mat = ev["frame"]
        h, w = mat.shape[:2]
        dims = {"width": w, "height": h, "channels": 1 if mat.ndim == 2 else mat.shape[2]}
        send(wire.build_frame_packet(wire.encode_to_ppm(mat), dims, camera_id=ev["camera_id"], timestamp=ev["timestamp"]))
        stamps.append(ev["timestamp"])
    return stamps
