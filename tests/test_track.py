"""CPU tests of the pose stage: the product's host geometry
(send-slam_amd/csrc/ss_track.cpp, compiled here with g++ behind tests/native/track_shim.cpp) against
the numpy restatement oracle/vo_oracle.py and against the ground truth of synthetic scenes.
The GPU half (ss_track through the C ABI on a parallax sequence) is in test_gpu_parity.py."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from oracle import vo_oracle as vo
from send_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-6  # relative; Jacobi eigen-solver vs LAPACK on well-conditioned inputs

CAM = vo.Camera(520.0, 515.0, 318.0, 242.0)
CAM_DIST = vo.Camera(520.0, 515.0, 318.0, 242.0, -0.28, 0.07, 1e-3, -5e-4)


@pytest.fixture(scope="session")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("track") / "libtrack_shim.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                           os.path.join(ROOT, "tests/native/track_shim.cpp"),
                           os.path.join(ROOT, "send-slam_amd/csrc/ss_track.cpp")])
    lib = C.CDLL(so)
    lib.shim_tracker_new.restype = C.c_void_p
    lib.shim_tracker_new.argtypes = [C.c_void_p, C.c_double]
    lib.shim_tracker_free.argtypes = [C.c_void_p]
    lib.shim_tracker_want.argtypes = [C.c_void_p]
    lib.shim_tracker_n_train.argtypes = [C.c_void_p]
    lib.shim_tracker_set_hist_cap.argtypes = [C.c_void_p, C.c_int]
    lib.shim_tracker_hist_len.argtypes = [C.c_void_p]
    lib.shim_tracker_step.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6
    lib.shim_triangulate.argtypes = [C.c_void_p] * 7 + [C.c_double, C.c_double, C.c_void_p]
    return lib


def cam8(cam):
    return np.array([cam.fx, cam.fy, cam.cx, cam.cy, cam.k1, cam.k2, cam.p1, cam.p2], np.float64)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def rot(axis, deg):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    R, _ = vo._se3_exp(np.concatenate([axis * math.radians(deg), np.zeros(3)]))
    return R


def project(cam, R, t, X):
    Y = X @ R.T + t
    return np.stack([cam.fx * Y[:, 0] / Y[:, 2] + cam.cx, cam.fy * Y[:, 1] / Y[:, 2] + cam.cy], axis=1)


def two_view_case(seed, n=300, noise=0.3, outliers=0.1):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(4, 9, n)], axis=1)
    R = rot(rng.normal(size=3), rng.uniform(1, 6))
    t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * 0.6
    x1 = project(CAM, np.eye(3), np.zeros(3), X) + rng.normal(scale=noise, size=(n, 2))
    x2 = project(CAM, R, t, X) + rng.normal(scale=noise, size=(n, 2))
    bad = rng.random(n) < outliers
    x2[bad] += rng.uniform(-60, 60, size=(int(bad.sum()), 2))
    return X, R, t, np.ascontiguousarray(x1), np.ascontiguousarray(x2), bad


def shim_two_view(shim, cam, x1, x2):
    n = len(x1)
    R, t = np.zeros(9), np.zeros(3)
    tri, p3d = np.zeros(n, np.uint8), np.zeros((n, 3))
    k = cam8(cam)
    r = shim.shim_two_view(p(k), n, p(x1), p(x2), p(R), p(t), p(tri), p(p3d))
    return r, R.reshape(3, 3), t, tri.astype(bool), p3d


def test_undistort_matches_oracle_and_inverts_the_model(shim):
    rng = np.random.default_rng(5)
    xy = np.ascontiguousarray(rng.uniform([20, 20], [620, 460], size=(500, 2)).astype(np.float32))
    for cam in (CAM, CAM_DIST):
        out = np.zeros((500, 2))
        k = cam8(cam)
        shim.shim_undistort(p(k), 500, p(xy), p(out))
        ref = vo.undistort(cam, xy)
        assert np.allclose(out, ref, rtol=0, atol=1e-9)
    assert np.array_equal(vo.undistort(CAM, xy), xy.astype(np.float64))
    # forward-distort the undistorted points: back at the input to < 0.2 px (5 iterations, k1 = -0.28, corners)
    u = vo.undistort(CAM_DIST, xy)
    x, y = (u[:, 0] - CAM_DIST.cx) / CAM_DIST.fx, (u[:, 1] - CAM_DIST.cy) / CAM_DIST.fy
    r2 = x * x + y * y
    rad = 1 + CAM_DIST.k1 * r2 + CAM_DIST.k2 * r2 * r2
    xd = x * rad + 2 * CAM_DIST.p1 * x * y + CAM_DIST.p2 * (r2 + 2 * x * x)
    yd = y * rad + CAM_DIST.p1 * (r2 + 2 * y * y) + 2 * CAM_DIST.p2 * x * y
    back = np.stack([xd * CAM_DIST.fx + CAM_DIST.cx, yd * CAM_DIST.fy + CAM_DIST.cy], axis=1)
    assert np.abs(back - xy).max() < 0.2


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_two_view_matches_oracle_and_ground_truth(shim, seed):
    X, R_gt, t_gt, x1, x2, bad = two_view_case(seed)
    n_tri, R, t, tri, p3d = shim_two_view(shim, CAM, x1, x2)
    ref = vo.two_view(CAM, x1, x2)
    assert ref is not None and n_tri > 0
    R_o, t_o, tri_o, p3d_o = ref
    assert np.array_equal(tri, tri_o) and n_tri == int(tri_o.sum())
    assert np.allclose(R, R_o, rtol=0, atol=TOL) and np.allclose(t, t_o, rtol=0, atol=TOL)
    assert np.allclose(p3d[tri], p3d_o[tri], rtol=TOL, atol=TOL)
    # ground truth: rotation within 0.5 degree, translation direction within 3 degrees, no outlier triangulated
    dR = R @ R_gt.T
    assert math.degrees(math.acos(min(1.0, (np.trace(dR) - 1) / 2))) < 0.5
    assert math.degrees(math.acos(min(1.0, float(t @ t_gt) / np.linalg.norm(t_gt)))) < 3.0
    assert n_tri >= 0.8 * (~bad).sum()
    scale = np.linalg.norm(t_gt)
    ok = tri & ~bad
    assert np.median(np.linalg.norm(p3d[ok] * scale - X[ok], axis=1) / X[ok, 2]) < 0.05


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_two_view_ba_matches_oracle_and_improves_the_pose(shim, seed):
    X, R_gt, t_gt, x1, x2, bad = two_view_case(seed, noise=0.5)
    ref = vo.two_view(CAM, x1, x2)
    R0, t0, tri, p3d = ref
    o1, o2 = np.ascontiguousarray(x1[tri]), np.ascontiguousarray(x2[tri])
    rng = np.random.default_rng(seed)
    w1 = np.ascontiguousarray(1.0 / 1.2 ** (2.0 * rng.integers(0, 4, len(o1))))
    w2 = np.ascontiguousarray(1.0 / 1.2 ** (2.0 * rng.integers(0, 4, len(o1))))
    R, t, Xp = R0.copy().ravel(), t0.copy(), np.ascontiguousarray(p3d[tri].copy())
    k = cam8(CAM)
    acc = shim.shim_two_view_ba(p(k), len(o1), p(o1), p(o2), p(w1), p(w2), p(R), p(t), p(Xp), 20)
    R_o, t_o, X_o, acc_o = vo.two_view_ba(CAM, o1, o2, w1, w2, R0, t0, p3d[tri], 20)
    assert acc == acc_o and acc >= 3
    assert np.allclose(R.reshape(3, 3), R_o, rtol=0, atol=TOL) and np.allclose(t, t_o, rtol=0, atol=TOL)
    assert np.allclose(Xp, X_o, rtol=1e-5, atol=1e-6)
    cost0 = vo._ba_cost(CAM, o1, o2, w1, w2, R0, t0, p3d[tri])
    cost1 = vo._ba_cost(CAM, o1, o2, w1, w2, R_o, t_o, X_o)
    assert cost1 < cost0

    def ang(Ra, ta):
        dR = Ra @ R_gt.T
        return (math.degrees(math.acos(min(1.0, (np.trace(dR) - 1) / 2))),
                math.degrees(math.acos(min(1.0, float(ta @ t_gt) / (np.linalg.norm(t_gt) * np.linalg.norm(ta))))))

    r0, d0 = ang(R0, t0)
    r1, d1 = ang(R_o, t_o)
    assert r1 <= r0 + 0.02 and d1 <= d0 + 0.2 and r1 < 0.3 and d1 < 2.0


def planar_case(seed, n=300, noise=0.3, outliers=0.08, lateral=True):
    """Points on a plane seen from two cameras.  Two views of a plane admit two physical reconstructions
    (Faugeras-Lustman: the roles of translation and plane normal swap); `lateral` = camera sliding roughly
    parallel to a roughly fronto-parallel plane, where the twin solution puts the plane edge-on and fails
    the positive-depth test, so the pair is decidable."""
    rng = np.random.default_rng(seed)
    tilt = 0.08 if lateral else 0.3
    nrm = np.array([rng.uniform(-tilt, tilt), rng.uniform(-tilt, tilt), -1.0])
    nrm /= np.linalg.norm(nrm)
    d = -6.0
    xy = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n)], axis=1)
    z = (d - xy @ nrm[:2]) / nrm[2]
    X = np.concatenate([xy, z[:, None]], axis=1)
    if lateral:
        R = rot(rng.normal(size=3), rng.uniform(0.3, 1.5))
        t = np.array([rng.normal(), rng.normal(), 0.1 * rng.normal()])
    else:
        R = rot(rng.normal(size=3), rng.uniform(1, 5))
        t = rng.normal(size=3)
    t = t / np.linalg.norm(t) * 0.7
    x1 = project(CAM, np.eye(3), np.zeros(3), X) + rng.normal(scale=noise, size=(n, 2))
    x2 = project(CAM, R, t, X) + rng.normal(scale=noise, size=(n, 2))
    bad = rng.random(n) < outliers
    x2[bad] += rng.uniform(-50, 50, size=(int(bad.sum()), 2))
    return X, R, t, np.ascontiguousarray(x1), np.ascontiguousarray(x2), bad


@pytest.mark.parametrize("seed", [109, 115, 126, 129, 134, 135])
def test_two_view_planar_scene_takes_the_homography_branch(shim, seed):
    """A planar scene is degenerate for the fundamental matrix: RH = SH / (SH + SF) > 0.45 selects the homography
    and ReconstructH (Faugeras-Lustman, 8 hypotheses) must return the true motion.  The seeds are the pairs of a
    scan of 60 in which the twin solution loses more than a quarter of the points (one in eight: see
    test_two_view_ambiguous_planar_pair_is_refused for the rest)."""
    X, R_gt, t_gt, x1, x2, bad = planar_case(seed)
    n_tri, R, t, tri, p3d = shim_two_view(shim, CAM, x1, x2)
    assert shim.shim_two_view_model() == 2
    ref, model = vo.two_view(CAM, x1, x2, return_model=True)
    assert model == 2 and ref is not None and n_tri > 0
    R_o, t_o, tri_o, p3d_o = ref
    assert np.array_equal(tri, tri_o)
    assert np.allclose(R, R_o, rtol=0, atol=TOL) and np.allclose(t, t_o, rtol=0, atol=TOL)
    assert np.allclose(p3d[tri], p3d_o[tri], rtol=1e-5, atol=1e-6)
    dR = R @ R_gt.T
    assert math.degrees(math.acos(min(1.0, (np.trace(dR) - 1) / 2))) < 0.6
    assert math.degrees(math.acos(min(1.0, float(t @ t_gt) / np.linalg.norm(t_gt)))) < 4.0
    ok = tri & ~bad
    assert ok.sum() > 0.8 * (~bad).sum()
    scale = np.linalg.norm(t_gt)
    assert np.median(np.linalg.norm(p3d[ok] * scale - X[ok], axis=1) / X[ok, 2]) < 0.06


def test_two_view_ambiguous_planar_pair_is_refused(shim):
    """A generic motion over a tilted plane leaves BOTH Faugeras-Lustman solutions in front of the cameras:
    ReconstructH's `second best < 0.75 best` rule refuses the pair (a later frame decides), in both statements."""
    X, R_gt, t_gt, x1, x2, bad = planar_case(51, lateral=False)
    n_tri, R, t, tri, p3d = shim_two_view(shim, CAM, x1, x2)
    assert shim.shim_two_view_model() == 2 and n_tri == 0
    ref, model = vo.two_view(CAM, x1, x2, return_model=True)
    assert model == 2 and ref is None


def test_two_view_general_scene_takes_the_fundamental_branch(shim):
    X, R_gt, t_gt, x1, x2, bad = two_view_case(2)
    shim_two_view(shim, CAM, x1, x2)
    assert shim.shim_two_view_model() == 1
    assert vo.two_view(CAM, x1, x2, return_model=True)[1] == 1


def test_tracker_initialises_on_a_planar_image_sequence(shim, oracle):
    """synth.frame sequences are a fronto-parallel plane sliding by (3, -2) px per frame: only the homography
    branch can initialise on them.  The camera then moves along (-3, +2, 0) at constant speed."""
    import track_ref
    w, h, seed = 640, 480, 31
    sc = synth.scene(seed, w, h)
    frames = [synth.frame_from_scene(sc, seed, w, h, t) for t in range(22)]
    cam = vo.Camera(500.0, 500.0, 320.0, 240.0)
    params = oracle.default_params(n_features=1000)

    def extract(img):
        kps, desc, _ = oracle.extract(img, params)
        return kps, desc

    prod, ref = ShimTracker(shim, cam, 1.2), vo.Tracker(cam, 1.2)
    outs = run_sequence(frames, extract, lambda q, t: oracle.match(q, t, 50, 9, 10, False), [prod, ref])
    prod.close()
    states = [o[0]["state"] for o in outs]
    assert 2 in states and states[-1] == 2 and 4 not in states
    first_ok = states.index(2)
    # the twin solution of the plane keeps > 75 % of the points until the baseline has grown to ~6 degrees of
    # parallax: like ORB-SLAM on a poster, initialisation takes a while (frame 16 here)
    assert 8 <= first_ok <= 19
    for a, b in outs:
        assert (a["state"], a["n_matches"], a["n_inliers"], a["n_map_points"]) == \
               (b["state"], b["n_matches"], b["n_inliers"], b["n_map_points"])
        assert np.allclose(a["position"], b["position"], rtol=0, atol=1e-6)
        assert np.allclose(a["quaternion"], b["quaternion"], rtol=0, atol=1e-6)
    pos = np.array([o[0]["position"] for o in outs[first_ok:]])
    steps = np.arange(first_ok, len(outs))
    direction = pos[-1] / np.linalg.norm(pos[-1])
    want = np.array([-3.0, 2.0, 0.0]) / math.sqrt(13.0)
    assert math.degrees(math.acos(min(1.0, float(direction @ want)))) < 6.0
    speed = np.linalg.norm(pos, axis=1) / steps
    assert np.abs(speed / speed.mean() - 1).max() < 0.08
    # median depth 1 and a 3.6-px shift per frame at f = 500: |step| = 3.6 / 500
    assert abs(speed.mean() - math.sqrt(13.0) / 500.0) < 0.1 * math.sqrt(13.0) / 500.0
    assert max(np.abs(o[0]["quaternion"][:3]).max() for o in outs[first_ok:]) < 5e-3


def test_two_view_refuses_degenerate_pairs(shim):
    X, R_gt, t_gt, x1, x2, bad = two_view_case(7, outliers=0.0)
    # no motion at all: no parallax -> no reconstruction
    assert shim_two_view(shim, CAM, x1, np.ascontiguousarray(x1 + 0.01))[0] == 0
    assert vo.two_view(CAM, x1, x1 + 0.01) is None
    # pure rotation: still no parallax
    x2r = np.ascontiguousarray(project(CAM, rot([0, 1, 0], 3), np.zeros(3), X))
    assert shim_two_view(shim, CAM, x1, x2r)[0] == 0
    assert vo.two_view(CAM, x1, x2r) is None
    # fewer than 8 matches
    assert shim_two_view(shim, CAM, x1[:7].copy(), x2[:7].copy())[0] == 0
    assert vo.two_view(CAM, x1[:7], x2[:7]) is None


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_pose_only_matches_oracle_and_ground_truth(shim, seed):
    rng = np.random.default_rng(seed)
    n = 200
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 9, n)], axis=1)
    R_gt, t_gt = rot(rng.normal(size=3), 4.0), rng.normal(size=3) * 0.2
    octv = rng.integers(0, 8, n)
    sig2 = 1.2 ** (2.0 * octv)
    obs = project(CAM, R_gt, t_gt, X) + rng.normal(size=(n, 2)) * np.sqrt(sig2)[:, None] * 0.4
    bad = rng.random(n) < 0.15
    obs[bad] += rng.uniform(-80, 80, size=(int(bad.sum()), 2))
    obs, w = np.ascontiguousarray(obs), np.ascontiguousarray(1.0 / sig2)
    R, t, inl = np.eye(3).ravel().copy(), np.zeros(3), np.zeros(n, np.uint8)
    k = cam8(CAM)
    n_in = shim.shim_pose_only(n, p(X), p(obs), p(w), p(k), p(R), p(t), p(inl))
    n_o, R_o, t_o, inl_o = vo.pose_only(CAM, X, obs, w, np.eye(3), np.zeros(3))
    assert n_in == n_o and np.array_equal(inl.astype(bool), inl_o)
    assert np.allclose(R.reshape(3, 3), R_o, rtol=0, atol=TOL) and np.allclose(t, t_o, rtol=0, atol=TOL)
    assert not inl.astype(bool)[bad].any()
    assert np.abs(R.reshape(3, 3) - R_gt).max() < 5e-3 and np.abs(t - t_gt).max() < 2e-2


def test_pose_only_agrees_with_the_c_oracle(shim, oracle):
    """The plain-C oracle (oracle/orb_oracle.c orc_pnp_pose_only) is a third statement of the same loop."""
    rng = np.random.default_rng(21)
    n = 150
    X = np.stack([rng.uniform(-3, 3, n), rng.uniform(-2, 2, n), rng.uniform(3, 9, n)], axis=1)
    R_gt, t_gt = rot([0.2, 1, 0.1], 3.0), np.array([0.1, -0.05, 0.15])
    obs = np.ascontiguousarray(project(CAM, R_gt, t_gt, X) + rng.normal(size=(n, 2)) * 0.5)
    w = np.ones(n)
    R_c, t_c, inl_c = oracle.pnp_pose_only(X, obs, w, CAM.fx, CAM.fy, CAM.cx, CAM.cy)
    R, t, inl = np.eye(3).ravel().copy(), np.zeros(3), np.zeros(n, np.uint8)
    k = cam8(CAM)
    shim.shim_pose_only(n, p(X), p(obs), p(w), p(k), p(R), p(t), p(inl))
    assert np.array_equal(inl.astype(bool), inl_c)
    assert np.allclose(R.reshape(3, 3), R_c, rtol=0, atol=1e-9) and np.allclose(t, t_c, rtol=0, atol=1e-9)


def test_triangulate_and_checks(shim):
    rng = np.random.default_rng(31)
    R1, t1 = np.eye(3), np.zeros(3)
    R2, t2 = rot([0, 1, 0], 2.0), np.array([-0.5, 0.02, 0.01])
    k = cam8(CAM)
    for trial in range(200):
        X = np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(2, 40)])
        x1 = project(CAM, R1, t1, X[None])[0] + rng.normal(size=2) * 0.5
        x2 = project(CAM, R2, t2, X[None])[0] + rng.normal(size=2) * (0.5 if trial % 5 else 6.0)
        out = np.zeros(3)
        R1c, R2c = np.ascontiguousarray(R1), np.ascontiguousarray(R2)
        ok = shim.shim_triangulate(p(k), p(x1), p(x2), p(R1c), p(t1), p(R2c), p(t2), 1.0, 1.44, p(out))
        ref = vo.triangulate(CAM, x1, x2, R1, t1, R2, t2, 1.0, 1.44)
        assert bool(ok) == (ref is not None)
        if ok:
            assert np.allclose(out, ref, rtol=1e-6, atol=1e-9)


def test_pose_to_twc_every_quaternion_branch(shim):
    rng = np.random.default_rng(41)
    cases = [rot(rng.normal(size=3), a) for a in (1, 30, 100, 170, 179.9)]
    cases += [rot([1, 0, 0], 180), rot([0, 1, 0], 180), rot([0, 0, 1], 180), rot([1, 1, 0], 175), rot([0, 1, 1], 178)]
    for R in cases:
        t = rng.normal(size=3)
        pos, q = np.zeros(3), np.zeros(4)
        Rc = np.ascontiguousarray(R)
        shim.shim_pose_to_twc(p(Rc), p(t), p(pos), p(q))
        pos_o, q_o = vo.pose_to_twc(R, t)
        assert np.allclose(pos, pos_o, atol=1e-12) and np.allclose(q, q_o, atol=1e-12)
        assert abs(np.linalg.norm(q) - 1) < 1e-9
        x, y, z, w = q  # back to a rotation: must be Rwc = R^T
        Rq = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        assert np.allclose(Rq, R.T, atol=1e-9)
        assert np.allclose(R @ pos + t, 0, atol=1e-9)


class ShimTracker:
    def __init__(self, shim, cam, scale):
        self.shim = shim
        self.k = cam8(cam)
        self.h = shim.shim_tracker_new(p(self.k), float(np.float32(scale)))

    def want(self):
        return self.shim.shim_tracker_want(self.h)

    def n_train(self):
        return self.shim.shim_tracker_n_train(self.h)

    def step(self, xy, octave, idx, d1):
        n = len(octave)
        xy = np.ascontiguousarray(xy, np.float32)
        octave = np.ascontiguousarray(octave, np.int32)
        idx = np.ascontiguousarray(idx if idx is not None else np.full(max(n, 1), -1), np.int32)
        d1 = np.ascontiguousarray(d1 if d1 is not None else np.full(max(n, 1), 0xFFFF), np.uint16)
        pose, cnt = np.zeros(7), np.zeros(4, np.int32)
        keep = self.shim.shim_tracker_step(self.h, n, p(xy), p(octave), p(idx), p(d1), p(pose), p(cnt))
        return {"state": int(cnt[0]), "n_matches": int(cnt[1]), "n_inliers": int(cnt[2]), "n_map_points": int(cnt[3]),
                "position": pose[:3].copy(), "quaternion": pose[3:].copy()}, keep

    def close(self):
        self.shim.shim_tracker_free(self.h)


def run_sequence(frames, extract, match, trackers):
    """Drives several trackers (product shim / numpy oracle) over one sequence with shared features."""
    stored = {vo.KEEP_AS_REF: None, vo.KEEP_AS_PREV: None}
    outs = []
    for img in frames:
        kps, desc = extract(img)
        xy = np.stack([kps["x"], kps["y"]], axis=1)
        want = trackers[0].want_match() if hasattr(trackers[0], "want_match") else trackers[0].want()
        idx = d1 = None
        if want != vo.MATCH_NONE:
            idx, d1, _ = match(desc, stored[want])
        row = []
        for tr in trackers:
            w = tr.want_match() if hasattr(tr, "want_match") else tr.want()
            assert w == want
            o, keep = tr.step(xy, kps["octave"], idx, d1)
            row.append((o, keep))
        assert len({k for _, k in row}) == 1
        if row[0][1] != vo.KEEP_NONE:
            stored[row[0][1]] = desc
        outs.append([o for o, _ in row])
    return outs


def test_tracker_keeps_a_long_back_and_forth_sequence(shim, oracle):
    """120 frames of a camera that goes back and forth over 16 positions: the tracker must still be tracking at the end, the
    poses at the turning points must repeat, and every estimated rotation must BE one.  Regression: the starting pose of
    the pose-only step is a product of estimated rotations (constant-velocity prediction) and was never re-orthonormalised;
    the deviation tripled per frame and every sequence was lost 41 frames after its initialisation."""
    w, h, seed = 320, 240, 4000
    sc = synth.scene(seed, w, h)
    base = [synth.parallax_frame(seed, w, h, t, sc=sc) for t in range(16)]
    order = list(range(16)) + list(range(14, 0, -1))
    params = oracle.default_params(n_features=600)
    feats = [oracle.extract(f, params)[:2] for f in base]
    cam = vo.Camera(0.8 * w, 0.8 * w, w / 2, h / 2)
    tr = ShimTracker(shim, cam, 1.2)
    stored = {vo.KEEP_AS_REF: None, vo.KEEP_AS_PREV: None}
    outs = []
    for i in range(120):
        kps, desc = feats[order[i % len(order)]]
        want = tr.want()
        idx = d1 = None
        if want != vo.MATCH_NONE:
            idx, d1, _ = oracle.match(desc, stored[want], 50, 9, 10, False)
        o, keep = tr.step(np.stack([kps["x"], kps["y"]], axis=1), kps["octave"], idx, d1)
        if keep != vo.KEEP_NONE:
            stored[keep] = desc
        outs.append(o)
    tr.close()
    states = [o["state"] for o in outs]
    first_ok = states.index(2)
    assert first_ok <= 6 and all(s == 2 for s in states[first_ok:]), "".join(map(str, states))
    for o in outs[first_ok:]:
        q = o["quaternion"]
        assert abs(float(q @ q) - 1.0) < 1e-9
    # the same camera position every 30 frames: the drift of a bounded front end, not a divergence
    period = len(order)
    for i in range(first_ok + period, 120):
        assert np.linalg.norm(outs[i]["position"] - outs[i - period]["position"]) < 0.05, i


def test_tracker_on_parallax_sequence_product_vs_oracle_vs_truth(shim, oracle):
    w, h, seed = 640, 480, 77
    sc = synth.scene(seed, w, h)
    frames = [synth.parallax_frame(seed, w, h, t, sc=sc) for t in range(12)]
    params = oracle.default_params(n_features=1000)
    cam = vo.Camera(500.0, 500.0, 320.0, 240.0)

    def extract(img):
        kps, desc, _ = oracle.extract(img, params)
        return kps, desc

    def match(q, t):
        return oracle.match(q, t, 50, 9, 10, False)

    # a third tracker drops unreferenced poses of its history after every frame (cap 2 instead of 256): a
    # long-lived front door must not grow with the run, and the poses must not depend on the compaction
    prod, ref, small = ShimTracker(shim, cam, 1.2), vo.Tracker(cam, 1.2), ShimTracker(shim, cam, 1.2)
    shim.shim_tracker_set_hist_cap(small.h, 2)
    outs3 = run_sequence(frames, extract, match, [prod, ref, small])
    assert shim.shim_tracker_hist_len(small.h) < shim.shim_tracker_hist_len(prod.h) == sum(o[0]["state"] == 2 for o in outs3)
    for a, _, c in outs3:
        assert (a["state"], a["n_matches"], a["n_inliers"], a["n_map_points"]) == (c["state"], c["n_matches"], c["n_inliers"], c["n_map_points"])
        assert np.array_equal(a["position"], c["position"]) and np.array_equal(a["quaternion"], c["quaternion"])
    outs = [o[:2] for o in outs3]
    prod.close()
    small.close()
    states = [o[0]["state"] for o in outs]
    assert states[0] == 1 and states[-1] == 2 and 4 not in states
    first_ok = states.index(2)
    assert first_ok <= 3
    for a, b in outs:
        assert (a["state"], a["n_matches"], a["n_inliers"], a["n_map_points"]) == \
               (b["state"], b["n_matches"], b["n_inliers"], b["n_map_points"])
        assert np.allclose(a["position"], b["position"], rtol=0, atol=1e-6)
        assert np.allclose(a["quaternion"], b["quaternion"], rtol=0, atol=1e-6)
    # truth: the camera slides along -x at constant speed without rotating.  Scale is fixed by the median
    # depth, so check direction, straightness and constant velocity.
    pos = np.array([o[0]["position"] for o in outs[first_ok:]])
    quat = np.array([o[0]["quaternion"] for o in outs[first_ok:]])
    assert np.abs(quat[:, :3]).max() < 5e-3
    steps = np.arange(first_ok, len(outs))  # frames since the reference frame 0
    v = pos[:, 0] / steps
    assert (v < 0).all() and np.abs(v / v.mean() - 1).max() < 0.08
    assert np.abs(pos[:, 1:]).max() < 0.12 * np.abs(pos[:, 0]).max()  # direction within ~7 degrees
    for o in outs[first_ok:]:
        assert o[0]["n_inliers"] >= 100


def test_tracker_state_machine_edges(shim):
    cam = vo.Camera(500.0, 500.0, 320.0, 240.0)
    for tr in (ShimTracker(shim, cam, 1.2), vo.Tracker(cam, 1.2)):
        want = (lambda: tr.want_match()) if hasattr(tr, "want_match") else tr.want
        rng = np.random.default_rng(3)
        few = rng.uniform(50, 400, size=(60, 2)).astype(np.float32)
        many = rng.uniform(50, 400, size=(300, 2)).astype(np.float32)
        # <= 100 keypoints: NOT_INITIALIZED without a reference (MonocularInitialization needs > 100)
        o, keep = tr.step(few, np.zeros(60, np.int32), None, None)
        assert (o["state"], keep, want()) == (1, vo.KEEP_NONE, vo.MATCH_NONE)
        o, keep = tr.step(many, np.zeros(300, np.int32), None, None)
        assert (o["state"], keep, want()) == (1, vo.KEEP_AS_REF, vo.MATCH_REF)
        # too few matches to the reference: the reference is dropped
        idx = np.full(300, -1, np.int32)
        idx[:50] = np.arange(50)
        o, keep = tr.step(many, np.zeros(300, np.int32), idx, np.full(300, 10, np.uint16))
        assert (o["state"], o["n_matches"], keep, want()) == (1, 50, vo.KEEP_NONE, vo.MATCH_NONE)
        # new reference; identical frame -> 300 matches but no parallax: reference kept
        o, keep = tr.step(many, np.zeros(300, np.int32), None, None)
        assert keep == vo.KEEP_AS_REF
        o, keep = tr.step(many, np.zeros(300, np.int32), np.arange(300, dtype=np.int32), np.zeros(300, np.uint16))
        assert (o["state"], o["n_matches"], keep, want()) == (1, 300, vo.KEEP_NONE, vo.MATCH_REF)
        # duplicate train rows: one query per train row, smallest distance wins
        idx = np.full(300, 299, np.int32)
        d1 = np.arange(300, 0, -1).astype(np.uint16)
        o, keep = tr.step(many, np.zeros(300, np.int32), idx, d1)
        assert o["n_matches"] == 1
        # ... and SearchForInitialization's 100-px window: the same match displaced by > 100 px is dropped
        far = many.copy()
        far[299] += 150
        o, keep = tr.step(far, np.zeros(300, np.int32), idx, d1)
        assert o["n_matches"] == 0
        if hasattr(tr, "close"):
            tr.close()


def test_parallax_sequence_is_what_it_says():
    w, h, seed = 320, 240, 5
    sc = synth.scene(seed, w, h)
    f0 = synth.parallax_frame(seed, w, h, 0, sc=sc).astype(int)
    f3 = synth.parallax_frame(seed, w, h, 3, sc=sc).astype(int)
    nb = len(synth.PARALLAX_DISPARITIES)
    for b, d in enumerate(synth.PARALLAX_DISPARITIES):
        y0, y1 = h * b // nb, h * (b + 1) // nb
        assert np.abs(f3[y0:y1, 3 * d:] - f0[y0:y1, :w - 3 * d]).max() <= 6  # +-3 noise on each frame
    assert np.array_equal(synth.parallax_frame(seed, w, h, 3), synth.parallax_frame(seed, w, h, 3, sc=sc))
    with pytest.raises(ValueError):
        synth.parallax_frame(seed, w, h, 32)


TRACK_GOLDEN = sorted(__import__("glob").glob(os.path.join(ROOT, "tests", "golden", "track", "*.npz")))


def _golden_frames(z):
    seed, w, h = int(z["seed"]), int(z["width"]), int(z["height"])
    sc = synth.scene(seed, w, h)
    frames = [synth.parallax_frame(seed, w, h, t, sc=sc) for t in range(int(z["n_frames"]))]
    sha = np.frombuffer(__import__("hashlib").sha256(frames[0].tobytes()).digest(), np.uint8)
    assert np.array_equal(sha, z["frame0_sha"]), "synth.parallax_frame changed: regenerate tests/golden/track"
    return frames


@pytest.mark.parametrize("path", TRACK_GOLDEN)
def test_cpu_pipeline_reproduces_the_pose_golden(path, oracle):
    """The committed pose sequences pin the all-CPU pipeline (C oracle + vo_oracle) against drift."""
    import track_ref
    z = np.load(path)
    outs = track_ref.run(oracle, _golden_frames(z), vo.Camera(*z["camera"]), int(z["n_features"]))
    assert [o["state"] for o in outs] == list(z["state"])
    assert np.array_equal(np.array([[o["n_keypoints"], o["n_matches"], o["n_inliers"], o["n_map_points"]] for o in outs]), z["counts"])
    assert np.allclose(np.array([o["position"] for o in outs]), z["position"], rtol=0, atol=1e-9)
    assert np.allclose(np.array([o["quaternion"] for o in outs]), z["quaternion"], rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", TRACK_GOLDEN)
def test_ss_track_reproduces_the_pose_golden(path):
    """ss_track through the C ABI against the committed pose sequences (no oracle at run time)."""
    from send_slam_amd import binding
    z = np.load(path)
    fx, fy, cx, cy, k1, k2, p1, p2 = [float(v) for v in z["camera"]]
    cam = binding.Camera(type=b"PinHole", fx=fx, fy=fy, cx=cx, cy=cy, k1=k1, k2=k2, p1=p1, p2=p2, width=int(z["width"]),
                         height=int(z["height"]), fps=30, rgb=1, th_depth=40.0, baseline=0.0, depth_map_factor=1000.0)
    with binding.OrbContext(0, n_features=int(z["n_features"])) as ctx:
        ctx.set_calibration(1, cam)
        got = [ctx.track(f) for f in _golden_frames(z)]
    assert [g["state"] for g in got] == list(z["state"])
    assert np.array_equal(np.array([[g["n_keypoints"], g["n_matches"], g["n_inliers"], g["n_map_points"]] for g in got]), z["counts"])
    assert np.allclose(np.array([g["position"] for g in got]), z["position"], rtol=0, atol=1e-6)
    assert np.allclose(np.array([g["quaternion"] for g in got]), z["quaternion"], rtol=0, atol=1e-6)
