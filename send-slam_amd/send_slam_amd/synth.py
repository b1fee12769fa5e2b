"""Seeded synthetic camera frames for the ORB hot path.

The reference ships no sample frames (send_slam/test/ holds two files, neither an image;
its replay hook is a commented-out VideoProducer, send_slam/lib/send_slam/application.ex:60-72),
so benches and parity tests run on frames made here: 1-channel u8, row-major, stride = width.

A frame is the sum of
  (i)   low-frequency value noise (64-px lattice, integer bilinear),
  (ii)  N filled rectangles / triangles with uniform random gray levels (corner-rich),
  (iii) +-3 uniform pixel noise,
clipped to [0, 255].  Frame t of a sequence is the seed's scene translated by (3t, -2t) px
plus fresh pixel noise, so consecutive-frame matching is meaningful.

Everything is integer arithmetic on numpy.random.Generator(PCG64(seed)) draws, so a
(seed, width, height, t) tuple names the same bytes on every machine.
"""
from __future__ import annotations

import numpy as np

_MARGIN = 256  # scene canvas margin so a sequence can translate without running out


def _value_noise(rng: np.random.Generator, h: int, w: int, cell: int = 64) -> np.ndarray:
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.integers(40, 216, size=(gh, gw), dtype=np.int64)
    ys, xs = np.arange(h, dtype=np.int64), np.arange(w, dtype=np.int64)
    iy, fy = ys // cell, (ys % cell)[:, None]
    ix, fx = xs // cell, (xs % cell)[None, :]
    g00 = g[iy][:, ix]
    g01 = g[iy][:, ix + 1]
    g10 = g[iy + 1][:, ix]
    g11 = g[iy + 1][:, ix + 1]
    top = g00 * (cell - fx) + g01 * fx
    bot = g10 * (cell - fx) + g11 * fx
    return (top * (cell - fy) + bot * fy) // (cell * cell)


def _draw_shapes(rng: np.random.Generator, img: np.ndarray, n: int) -> None:
    h, w = img.shape
    for _ in range(n):
        kind = int(rng.integers(0, 2))
        cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
        sw, sh = int(rng.integers(10, 121)), int(rng.integers(10, 121))
        level = int(rng.integers(0, 256))
        x0, x1 = max(cx - sw // 2, 0), min(cx + sw // 2, w)
        y0, y1 = max(cy - sh // 2, 0), min(cy + sh // 2, h)
        if x1 <= x0 or y1 <= y0:
            rng.integers(0, 1 << 30, size=6)  # keep the stream aligned
            continue
        if kind == 0:
            rng.integers(0, 1 << 30, size=6)
            img[y0:y1, x0:x1] = level
        else:
            p = rng.integers(0, 1 << 30, size=6)
            vx = x0 + (p[0::2] % max(x1 - x0, 1))
            vy = y0 + (p[1::2] % max(y1 - y0, 1))
            yy, xx = np.mgrid[y0:y1, x0:x1]

            def edge(ax, ay, bx, by):
                return (bx - ax) * (yy - ay) - (by - ay) * (xx - ax)

            e0 = edge(vx[0], vy[0], vx[1], vy[1])
            e1 = edge(vx[1], vy[1], vx[2], vy[2])
            e2 = edge(vx[2], vy[2], vx[0], vy[0])
            inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
            sub = img[y0:y1, x0:x1]
            sub[inside] = level


def scene(seed: int, width: int, height: int) -> np.ndarray:
    """The static scene of a seed on a canvas (height + 2M) x (width + 2M), int64."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ch, cw = height + 2 * _MARGIN, width + 2 * _MARGIN
    img = _value_noise(rng, ch, cw)
    n_shapes = max(40, (400 * cw * ch) // (1280 * 720))
    _draw_shapes(rng, img, n_shapes)
    return img


def frame_from_scene(sc: np.ndarray, seed: int, width: int, height: int, t: int = 0) -> np.ndarray:
    # scene content moves by (+3t, -2t) px: sample the canvas at the opposite offset
    ox = _MARGIN - (3 * t) % _MARGIN
    oy = _MARGIN + (2 * t) % _MARGIN
    crop = sc[oy:oy + height, ox:ox + width]
    nrng = np.random.Generator(np.random.PCG64([seed, t, 0x5EED]))
    noise = nrng.integers(-3, 4, size=(height, width), dtype=np.int64)
    return np.clip(crop + noise, 0, 255).astype(np.uint8)


def frame(seed: int, width: int, height: int, t: int = 0) -> np.ndarray:
    """Frame t of the sequence of `seed`: (height, width) uint8, C-contiguous."""
    return np.ascontiguousarray(frame_from_scene(scene(seed, width, height), seed, width, height, t))


def batch(seeds, width: int, height: int, t: int = 0) -> np.ndarray:
    """(len(seeds), height, width) uint8."""
    return np.stack([frame(s, width, height, t) for s in seeds])


PARALLAX_DISPARITIES = (2, 3, 4, 5, 6, 8)


def parallax_frame(seed: int, width: int, height: int, t: int = 0, disparities=PARALLAX_DISPARITIES,
                   sc: np.ndarray | None = None) -> np.ndarray:
    """Frame t of a sideways-translating camera in front of a staircase of fronto-parallel strips.

    The image is cut into len(disparities) horizontal bands; band b shows the seed's scene shifted by
    disparities[b] * t px in +x, i.e. a rigid scene whose band b lies at depth f * B / disparities[b]
    seen by a pinhole camera that has moved t * B along -x (no rotation).  It gives the pose stage
    (ss_track) a sequence with real parallax and a known answer: translation along one axis, band depths
    in the ratio 1 / disparity.  max(disparities) * t must stay below the canvas margin (256 px).
    """
    if max(disparities) * t >= _MARGIN or t < 0:
        raise ValueError("parallax sequence runs off the canvas")
    if sc is None:
        sc = scene(seed, width, height)
    out = np.empty((height, width), np.int64)
    nb = len(disparities)
    for b, d in enumerate(disparities):
        y0, y1 = height * b // nb, height * (b + 1) // nb
        ox = _MARGIN - d * t
        out[y0:y1] = sc[_MARGIN + y0:_MARGIN + y1, ox:ox + width]
    nrng = np.random.Generator(np.random.PCG64([seed, t, 0x9A7A]))
    noise = nrng.integers(-3, 4, size=(height, width), dtype=np.int64)
    return np.ascontiguousarray(np.clip(out + noise, 0, 255).astype(np.uint8))


def color_frame(seed: int, width: int, height: int, t: int = 0) -> np.ndarray:
    """(height, width, 3) uint8: three decorrelated planes of the same scene geometry."""
    g = frame(seed, width, height, t).astype(np.int64)
    rng = np.random.Generator(np.random.PCG64([seed, t, 0xC0102]))
    off = rng.integers(-20, 21, size=(height, width, 3), dtype=np.int64)
    return np.clip(g[:, :, None] + off, 0, 255).astype(np.uint8)
