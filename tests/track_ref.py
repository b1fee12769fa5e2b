"""All-CPU statement of the ss_track pipeline for the tests: C oracle extraction + match
(oracle/orb_oracle.c) feeding the numpy tracker (oracle/vo_oracle.py)."""
import numpy as np

from oracle import vo_oracle as vo


def run(oracle, frames, cam: "vo.Camera", n_features: int, gray=None):
    """frames: list of (H, W) u8 (or (H, W, 3) with gray= a converter).  -> list of tracker outputs"""
    params = oracle.default_params(n_features=n_features)
    tr = vo.Tracker(cam, 1.2)
    stored = {vo.KEEP_AS_REF: None, vo.KEEP_AS_PREV: None}
    outs = []
    for img in frames:
        if img.ndim == 3:
            img = gray(img)
        kps, desc, _ = oracle.extract(img, params)
        wm = tr.want_match()
        idx = d1 = None
        if wm != vo.MATCH_NONE:
            if len(desc):
                idx, d1, _ = oracle.match(desc, stored[wm], 50, 9, 10, False)
            else:
                idx, d1 = np.zeros(0, np.int32), np.zeros(0, np.uint16)
        o, keep = tr.step(np.stack([kps["x"], kps["y"]], axis=1), kps["octave"], idx, d1)
        if keep != vo.KEEP_NONE:
            stored[keep] = desc
        outs.append(o)
    return outs
