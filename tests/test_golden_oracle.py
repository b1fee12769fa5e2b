"""CPU: the oracle reproduces every committed golden vector (tests/golden/*.npz, made by
tests/golden/make_golden.py).  The reference holds no fixture for this path, so these vectors pin
the oracle itself against drift; the GPU suite checks the HIP path against the same files."""
import glob
import hashlib
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


@pytest.mark.parametrize("path", GOLDEN)
def test_oracle_reproduces_golden(path, oracle):
    z = np.load(path)
    img, lap = z["frame"], z["lapping"]
    p = oracle.default_params(n_features=int(z["n_features"]), lapping_x0=int(lap[0]), lapping_x1=int(lap[1]))
    h, w = img.shape
    g = oracle.geometry(p, w, h)
    levels = oracle.pyramid(img, p)
    for l, lv in enumerate(levels):
        assert np.array_equal(sha(lv), z["level_sha"][l])
        assert np.array_equal(sha(oracle.blur(lv)), z["blur_sha"][l])
        assert np.array_equal(sha(oracle.fast_score_map(lv, 7)), z["score7_sha"][l])
        c = oracle.candidates(lv, p.ini_th_fast, p.min_th_fast)
        assert c.tobytes() == z[f"cand{l}"].tobytes()
        assert oracle.distribute(c, g.w[l], g.h[l], g.quota[l]).tobytes() == z[f"sel{l}"].tobytes()
    kps, desc, counts = oracle.extract(img, p)
    assert kps.tobytes() == z["kps"].tobytes() and np.array_equal(desc, z["desc"]) and np.array_equal(counts, z["level_counts"])
    idx, d1, d2 = oracle.match(desc, desc, exclude_self=True)
    assert np.array_equal(idx, z["self_idx"]) and np.array_equal(d1, z["self_d1"]) and np.array_equal(d2, z["self_d2"])


def test_golden_set_is_present_and_small():
    assert len(GOLDEN) == 4
    assert sum(os.path.getsize(p) for p in GOLDEN) < 1 << 20
