"""Pins that are possible in this image without the reference's third-party code:

  (a) the recalled rBRIEF table against the copy of OpenCV's bit_pattern_31_ that scikit-image ships as a DATA
      file (read as text, nothing imported); skipped where that file is absent;
  (b) the 7x7 sigma-2 Gaussian taps derived from exp(-d^2 / 8) and OpenCV's error-diffusion rule for the 8-bit
      fixed-point kernel instead of trusting the literal {18, 34, 48, 56, ...};
  (c) the loader of tests/golden/ref_dump/ (README there): the day a maintainer drops a dump of the real
      ORB-SLAM3 extractor, the oracle (here) and the HIP path (-m gpu) are compared with it.  The loader itself
      is exercised on a dump written from the oracle into a temporary directory, so an empty ref_dump/ does not
      mean an untested loader.  Until a real dump exists parity with ORB-SLAM3 stays UNPINNED.
"""
import glob
import os

import numpy as np
import pytest

import pyref
import ref_dump
from send_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIMAGE_PATTERN = [p for pat in ("/opt/conda/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt",
                                 "/usr/lib/python3*/site-packages/skimage/feature/orb_descriptor_positions.txt",
                                 "/usr/local/lib/python3*/*-packages/skimage/feature/orb_descriptor_positions.txt")
                   for p in glob.glob(pat)]


@pytest.mark.skipif(not SKIMAGE_PATTERN, reason="scikit-image's orb_descriptor_positions.txt is not on this image")
def test_bit_pattern_equals_the_opencv_table_scikit_image_ships():
    want = np.loadtxt(SKIMAGE_PATTERN[0]).astype(np.int64)
    assert want.shape == (256, 4)
    orc = pyref.parse_c_int_table(os.path.join(ROOT, "oracle/orb_constants.h"), "ORC_BIT_PATTERN_31")
    assert orc == want.reshape(-1).tolist()
    # the product's table (csrc/ss_constants.h) is kept equal to the oracle's by tests/test_abi.py; check it here
    # against the data file too, so neither side depends on the other for this pin
    prod = open(os.path.join(ROOT, "send-slam_amd/csrc/ss_constants.h")).read()
    import re
    body = prod[prod.index("SS_BIT_PATTERN_31_VALUES") + 24:].replace("\\\n", " ").split("#endif")[0]
    assert [int(v) for v in re.findall(r"-?\d+", body)] == want.reshape(-1).tolist()


def gaussian_taps_u8(ksize, sigma):
    """cv::getGaussianKernel + the fixed-point conversion GaussianBlur's 8-bit path applies
    (8 fractional bits, rounding error carried to the next tap, centre = 256 - the rest)."""
    d = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2
    k = np.exp(-(d * d) / (2.0 * sigma * sigma))
    k /= k.sum()
    n2 = ksize // 2
    out = [0] * ksize
    err = 0.0
    for i in range(n2):
        adj = k[i] * 256.0 + err
        v = int(np.rint(adj))
        err = adj - v
        out[i] = out[ksize - 1 - i] = v
    out[n2] = 256 - 2 * sum(out[:n2])
    return out


def test_gaussian_taps_follow_from_sigma_2():
    taps = gaussian_taps_u8(7, 2.0)
    assert taps == [18, 34, 48, 56, 48, 34, 18] and sum(taps) == 256
    assert taps == [int(v) for v in pyref.GAUSS]
    import re
    orc = open(os.path.join(ROOT, "oracle/orb_constants.h")).read()
    assert [int(v) for v in re.findall(r"\d+", re.search(r"#define\s+ORC_GAUSS_TAPS\s+(.+)", orc).group(1))] == taps
    prod = open(os.path.join(ROOT, "send-slam_amd/csrc/ss_constants.h")).read()
    assert [int(re.search(rf"#define\s+SS_GAUSS_K{i}\s+(\d+)", prod).group(1)) for i in range(4)] == taps[:4]


def test_ref_dump_loader_round_trip(tmp_path, oracle):
    """A dump in the documented format, written from the ORACLE's own output, loads back to the same arrays and
    compares equal; a perturbed one is reported with the first differing keypoint."""
    img = synth.frame(2, 320, 240)
    p = oracle.default_params(n_features=300)
    kps, desc, _ = oracle.extract(img, p)
    d = str(tmp_path / "self")
    ref_dump.write(d, img, kps, desc, n_features=300, scale_factor=1.2, n_levels=8, ini_th_fast=20, min_th_fast=7,
                   source="oracle (loader self-test, NOT a reference dump)")
    dump = ref_dump.load(d)
    assert np.array_equal(dump.image, img) and dump.params["n_features"] == 300
    assert ref_dump.compare(dump, kps, desc) == []
    bad = kps.copy()
    bad["angle"][5] += 1.0
    desc2 = desc.copy()
    desc2[7, 3] ^= 4
    diffs = ref_dump.compare(dump, bad, desc2)
    assert any("keypoint 5" in s and "angle" in s for s in diffs) and any("descriptor 7" in s for s in diffs)
    # fewer / more keypoints is a difference, not an exception
    assert ref_dump.compare(dump, kps[:-1], desc[:-1])


@pytest.mark.parametrize("path", ref_dump.committed() or [None])
def test_oracle_against_committed_reference_dumps(path, oracle):
    if path is None:
        pytest.skip("tests/golden/ref_dump/ holds no dump of the real ORB-SLAM3 extractor: parity stays UNPINNED")
    dump = ref_dump.load(path)
    p = oracle.default_params(**dump.oracle_params())
    kps, desc, _ = oracle.extract(dump.image, p)
    diffs = ref_dump.compare(dump, kps, desc)
    assert not diffs, "\n".join(diffs[:20])


@pytest.mark.gpu
@pytest.mark.parametrize("path", ref_dump.committed() or [None])
def test_hip_path_against_committed_reference_dumps(path):
    if path is None:
        pytest.skip("tests/golden/ref_dump/ holds no dump of the real ORB-SLAM3 extractor: parity stays UNPINNED")
    from send_slam_amd import binding
    dump = ref_dump.load(path)
    with binding.OrbContext(0, **dump.oracle_params()) as ctx:
        kps, desc, _ = ctx.extract(dump.image)
    diffs = ref_dump.compare(dump, kps, desc)
    assert not diffs, "\n".join(diffs[:20])
