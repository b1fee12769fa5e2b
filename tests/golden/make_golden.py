"""Regenerates tests/golden/*.npz from the CPU oracle on seeded synthetic frames.

The reference holds no fixture for this path; these vectors pin
the oracle itself against drift and give the GPU suite inputs + expected outputs that do
not need the oracle at run time.  They are data only: frames, parameters, expected
keypoints / descriptors / matches and SHA-256 digests of the intermediate images.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "send-slam_amd"))

from oracle import orb_oracle as O  # noqa: E402
from send_slam_amd import synth  # noqa: E402

CASES = [  # name, seed, t, width, height, n_features, lapping
    ("g0_320x240_n500", 0, 0, 320, 240, 500, (0, 1000)),
    ("g0t1_320x240_n500", 0, 1, 320, 240, 500, (0, 1000)),
    ("g1_320x240_n150", 1, 0, 320, 240, 150, (0, 1000)),
    ("g2_400x260_n300_lap", 2, 0, 400, 260, 300, (100, 250)),
]


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def main():
    prev = {}
    for name, seed, t, w, h, nf, lap in CASES:
        img = synth.frame(seed, w, h, t)
        p = O.default_params(n_features=nf, lapping_x0=lap[0], lapping_x1=lap[1])
        g = O.geometry(p, w, h)
        levels = O.pyramid(img, p)
        out = dict(frame=img, n_features=nf, lapping=np.array(lap), seed=seed, t=t)
        out["level_sha"] = np.stack([digest(l) for l in levels])
        out["blur_sha"] = np.stack([digest(O.blur(l)) for l in levels])
        out["score7_sha"] = np.stack([digest(O.fast_score_map(l, 7)) for l in levels])
        for l, lv in enumerate(levels):
            c = O.candidates(lv, p.ini_th_fast, p.min_th_fast)
            out[f"cand{l}"] = c
            out[f"sel{l}"] = O.distribute(c, g.w[l], g.h[l], g.quota[l])
        kps, desc, counts = O.extract(img, p)
        out.update(kps=kps, desc=desc, level_counts=counts)
        idx, d1, d2 = O.match(desc, desc, 50, 9, 10, exclude_self=True)
        out.update(self_idx=idx, self_d1=d1, self_d2=d2)
        if (seed, w, h, nf) in prev:
            pd = prev[(seed, w, h, nf)]
            idx, d1, d2 = O.match(pd, desc, 50, 9, 10)
            out.update(prev_idx=idx, prev_d1=d1, prev_d2=d2)
        prev[(seed, w, h, nf)] = desc
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "kps", len(kps), "levels", list(counts), "cands", [len(out[f"cand{l}"]) for l in range(8)],
              "self-accepted", int((out["self_idx"] >= 0).sum()))


if __name__ == "__main__":
    main()
