"""numpy restatement of the pose-from-matches stage.

TEST INFRASTRUCTURE, NOT PRODUCT: import only from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  PARITY UNPINNED: the reference ships no ORB-SLAM3 source, tests or pose fixtures
(SURVEY.md section 8(c)); what follows restates the published ORB-SLAM3 building blocks from their
descriptions, for a bounded monocular front-end (no keyframes, local mapping, loop closing):

  undistort      Frame::UndistortKeyPoints = cv::undistortPoints, 5 fixed-point iterations
  two_view       TwoViewReconstruction: Normalize; ComputeH21 (DLT) + CheckHomography (chi2 5.991) and
                 ComputeF21 (8-point + rank 2) + CheckFundamental (chi2 3.841 / score 5.991) on the same 200
                 RANSAC sets; RH = SH / (SH + SF) > 0.45 -> ReconstructH (Faugeras-Lustman, 8 hypotheses,
                 second best < 0.75 best, > 0.9 N, > 50) else ReconstructF (E = K^T F K, DecomposeE, 4
                 hypotheses, second best < 0.7 best); CheckRT with th2 = 4, minParallax 1 degree
  pose_only      Optimizer::PoseOptimization: 4 rounds x 10 Gauss-Newton steps (lambda 1e-6 damping),
                 Huber delta sqrt(5.991) in the first two rounds, outliers at chi2 > 5.991
  two_view_ba    Optimizer::GlobalBundleAdjustemnt(map, 20) on the two initial keyframes: camera 1 fixed,
                 Huber sqrt(5.991), Levenberg-Marquardt on the Schur-reduced system
  triangulate    linear DLT + positive depth + chi2 5.991 sigma2 + cos(parallax) < 0.9998
  Tracker        Tracking::MonocularInitialization / CreateInitialMapMonocular (median depth 1) and a
                 frame-to-frame tracker (>= 30 inliers as TrackLocalMap asks)

The RANSAC sampler is a fixed 64-bit LCG (ORB-SLAM3 seeds rand() with 0: any fixed stream serves).
The product (send-slam_amd/csrc/ss_track.cpp) uses a Jacobi eigen-solver where this file uses LAPACK;
tests compare at 1e-6 relative on well-conditioned inputs.
"""
from __future__ import annotations

import math

import numpy as np

MASK64 = (1 << 64) - 1
# homography if SH / (SH + SF) exceeds it.  ORB-SLAM2: 0.40; ORB-SLAM3: `if(RH>0.50) // if(RH>0.40)` under the comment
# "(0.40-0.45)"; a planar scene scores ~0.49, so 0.50 would never take the homography: 0.45 (csrc/ss_track.cpp)
RH_THRESHOLD = 0.45


class Camera:
    def __init__(self, fx, fy, cx, cy, k1=0.0, k2=0.0, p1=0.0, p2=0.0):
        self.fx, self.fy, self.cx, self.cy = float(fx), float(fy), float(cx), float(cy)
        self.k1, self.k2, self.p1, self.p2 = float(k1), float(k2), float(p1), float(p2)

    @property
    def K(self):
        return np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1.0]])


def undistort(cam: Camera, xy) -> np.ndarray:
    xy = np.asarray(xy, np.float32).astype(np.float64).reshape(-1, 2)
    if cam.k1 == 0 and cam.k2 == 0 and cam.p1 == 0 and cam.p2 == 0:
        return xy.copy()
    x0 = (xy[:, 0] - cam.cx) / cam.fx
    y0 = (xy[:, 1] - cam.cy) / cam.fy
    x, y = x0.copy(), y0.copy()
    for _ in range(5):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + (cam.k2 * r2 + cam.k1) * r2)
        dx = 2 * cam.p1 * x * y + cam.p2 * (r2 + 2 * x * x)
        dy = cam.p1 * (r2 + 2 * y * y) + 2 * cam.p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.stack([x * cam.fx + cam.cx, y * cam.fy + cam.cy], axis=1)


def _normalize(p):
    mean = p.mean(axis=0)
    c = p - mean
    dev = np.abs(c).mean(axis=0)
    s = 1.0 / dev
    T = np.array([[s[0], 0, -mean[0] * s[0]], [0, s[1], -mean[1] * s[1]], [0, 0, 1.0]])
    return c * s, T


def _compute_f21(p1, p2):
    u1, v1, u2, v2 = p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1]
    A = np.stack([u2 * u1, u2 * v1, u2, v2 * u1, v2 * v1, v2, u1, v1, np.ones_like(u1)], axis=1)
    _, _, vt = np.linalg.svd(A)
    Fpre = vt[8].reshape(3, 3)
    u, w, vt2 = np.linalg.svd(Fpre)
    w[2] = 0
    return u @ np.diag(w) @ vt2


def _check_fundamental(F, x1, x2):
    th, th_score = 3.841, 5.991
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], axis=1)
    h2 = np.concatenate([x2, np.ones((len(x2), 1))], axis=1)
    l2 = h1 @ F.T  # lines in image 2
    num2 = (l2 * h2).sum(axis=1)
    chi1 = num2 * num2 / (l2[:, 0] ** 2 + l2[:, 1] ** 2)
    l1 = h2 @ F    # lines in image 1
    num1 = (l1 * h1).sum(axis=1)
    chi2 = num1 * num1 / (l1[:, 0] ** 2 + l1[:, 1] ** 2)
    ok1, ok2 = chi1 <= th, chi2 <= th
    score = 0.0
    # same summation order as a sequential loop is not needed at the test tolerance; keep pairwise order anyway
    for a, b, c, d in zip(chi1, ok1, chi2, ok2):
        if b:
            score += th_score - a
        if d:
            score += th_score - c
    return score, ok1 & ok2


class _Lcg:
    def __init__(self):
        self.x = 0x9E3779B97F4A7C15

    def next(self, mod):
        self.x = (self.x * 6364136223846793005 + 1442695040888963407) & MASK64
        return (self.x >> 33) % mod


def _triangulate_dlt(K, x1, x2, R1, t1, R2, t2):
    P1 = K @ np.concatenate([R1, t1.reshape(3, 1)], axis=1)
    P2 = K @ np.concatenate([R2, t2.reshape(3, 1)], axis=1)
    A = np.stack([x1[0] * P1[2] - P1[0], x1[1] * P1[2] - P1[1], x2[0] * P2[2] - P2[0], x2[1] * P2[2] - P2[1]])
    _, _, vt = np.linalg.svd(A)
    h = vt[3]
    if h[3] == 0 or not np.isfinite(h[3]):
        return None
    X = h[:3] / h[3]
    return X if np.all(np.isfinite(X)) else None


def _check_rt(cam, R, t, x1, x2, inl, th2):
    K = cam.K
    O2 = -R.T @ t
    n = len(x1)
    p3d = np.zeros((n, 3))
    good = np.zeros(n, bool)
    cosp = []
    n_good = 0
    I, z = np.eye(3), np.zeros(3)
    for i in range(n):
        if not inl[i]:
            continue
        X = _triangulate_dlt(K, x1[i], x2[i], I, z, R, t)
        if X is None:
            continue
        n2 = X - O2
        cp = float(X @ n2) / (np.linalg.norm(X) * np.linalg.norm(n2))
        if X[2] <= 0 and cp < 0.99998:
            continue
        Y = R @ X + t
        if Y[2] <= 0 and cp < 0.99998:
            continue
        e1 = np.array([cam.fx * X[0] / X[2] + cam.cx, cam.fy * X[1] / X[2] + cam.cy]) - x1[i]
        if e1 @ e1 > th2:
            continue
        e2 = np.array([cam.fx * Y[0] / Y[2] + cam.cx, cam.fy * Y[1] / Y[2] + cam.cy]) - x2[i]
        if e2 @ e2 > th2:
            continue
        cosp.append(cp)
        p3d[i] = X
        n_good += 1
        if cp < 0.99998:
            good[i] = True
    parallax = 0.0
    if n_good > 0:
        cosp.sort()
        parallax = math.degrees(math.acos(min(1.0, max(-1.0, cosp[min(50, len(cosp) - 1)]))))  # rounding can give 1 + 2e-16
    return n_good, p3d, good, parallax


def _compute_h21(p1, p2):
    u1, v1, u2, v2 = p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1]
    z, o = np.zeros_like(u1), np.ones_like(u1)
    ra = np.stack([z, z, z, -u1, -v1, -o, v2 * u1, v2 * v1, v2], axis=1)
    rb = np.stack([u1, v1, o, z, z, z, -u2 * u1, -u2 * v1, -u2], axis=1)
    A = np.empty((2 * len(u1), 9))
    A[0::2], A[1::2] = ra, rb
    _, _, vt = np.linalg.svd(A)
    return vt[8].reshape(3, 3)


def _check_homography(H21, H12, x1, x2):
    th = 5.991
    h1 = np.concatenate([x1, np.ones((len(x1), 1))], axis=1)
    h2 = np.concatenate([x2, np.ones((len(x2), 1))], axis=1)
    a = h2 @ H12.T
    a = a[:, :2] / a[:, 2:3]
    chi1 = ((x1 - a) ** 2).sum(axis=1)
    b = h1 @ H21.T
    b = b[:, :2] / b[:, 2:3]
    chi2 = ((x2 - b) ** 2).sum(axis=1)
    ok1, ok2 = chi1 <= th, chi2 <= th
    score = 0.0
    for c1, o1, c2, o2 in zip(chi1, ok1, chi2, ok2):
        if o1:
            score += th - c1
        if o2:
            score += th - c2
    return score, ok1 & ok2


def _pick(cam, hyps, x1, x2, inl, N, homography):
    res = [_check_rt(cam, R, t, x1, x2, inl, 4.0) for R, t in hyps]
    goods = [r[0] for r in res]
    best_good = max(goods)
    if best_good <= 0:
        return None
    k = goods.index(best_good)
    second = max([g for i, g in enumerate(goods) if i != k] + [0])
    parallax = res[k][3]
    if homography:
        if not (second < 0.75 * best_good and parallax >= 1.0 and best_good > 50 and best_good > 0.9 * N):
            return None
    else:
        n_similar = sum(g > 0.7 * best_good for g in goods)
        if best_good < max(int(0.9 * N), 50) or n_similar > 1 or not parallax > 1.0:
            return None
    return hyps[k][0], hyps[k][1], res[k][2], res[k][1]


def two_view(cam: Camera, x1, x2, return_model=False):
    """-> (R, t, triangulated mask, pts3d) or None   [, model: 1 fundamental, 2 homography]"""
    x1 = np.asarray(x1, np.float64).reshape(-1, 2)
    x2 = np.asarray(x2, np.float64).reshape(-1, 2)
    n = len(x1)

    def ret(r, model):
        return (r, model) if return_model else r

    if n < 8:
        return ret(None, 0)
    n1, T1 = _normalize(x1)
    n2, T2 = _normalize(x2)
    T2inv = np.linalg.inv(T2)
    rng = _Lcg()
    best_sf, best_F, inl_f = -1.0, None, None
    best_sh, best_H, inl_h = -1.0, None, None
    for _ in range(200):
        avail = list(range(n))
        idx = []
        for _j in range(8):
            r = rng.next(len(avail))
            idx.append(avail[r])
            avail[r] = avail[-1]
            avail.pop()
        H = T2inv @ _compute_h21(n1[idx], n2[idx]) @ T1
        if abs(np.linalg.det(H)) > 1e-300:
            sh, ih = _check_homography(H, np.linalg.inv(H), x1, x2)
            if sh > best_sh:
                best_sh, best_H, inl_h = sh, H, ih
        F = T2.T @ _compute_f21(n1[idx], n2[idx]) @ T1
        sf, i_f = _check_fundamental(F, x1, x2)
        if sf > best_sf:
            best_sf, best_F, inl_f = sf, F, i_f
    K = cam.K
    if best_sh > 0 and best_sh / (best_sh + max(best_sf, 0.0)) > RH_THRESHOLD:
        N = int(inl_h.sum())
        if N < 8:
            return ret(None, 2)
        A = np.linalg.inv(K) @ best_H @ K
        U, w, Vt = np.linalg.svd(A)
        V = Vt.T
        sgn = np.linalg.det(U) * np.linalg.det(Vt)
        d1, d2, d3 = w
        if d1 / d2 < 1.00001 or d2 / d3 < 1.00001:
            return ret(None, 2)
        aux1 = math.sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3))
        aux3 = math.sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3))
        x1s, x3s = [aux1, aux1, -aux1, -aux1], [aux3, -aux3, aux3, -aux3]
        root = math.sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3))
        st0, ct = root / ((d1 + d3) * d2), (d2 * d2 + d1 * d3) / ((d1 + d3) * d2)
        sp0, cp = root / ((d1 - d3) * d2), (d1 * d3 - d2 * d2) / ((d1 - d3) * d2)
        sts, sps = [st0, -st0, -st0, st0], [sp0, -sp0, -sp0, sp0]
        hyps = []
        for i in range(4):  # d' = +d2
            Rp = np.array([[ct, 0, -sts[i]], [0, 1, 0], [sts[i], 0, ct]])
            tt = U @ (np.array([x1s[i], 0, -x3s[i]]) * (d1 - d3))
            hyps.append((sgn * U @ Rp @ Vt, tt / np.linalg.norm(tt)))
        for i in range(4):  # d' = -d2
            Rp = np.array([[cp, 0, sps[i]], [0, -1, 0], [sps[i], 0, -cp]])
            tt = U @ (np.array([x1s[i], 0, x3s[i]]) * (d1 + d3))
            hyps.append((sgn * U @ Rp @ Vt, tt / np.linalg.norm(tt)))
        _ = V
        return ret(_pick(cam, hyps, x1, x2, inl_h, N, True), 2)
    N = int(inl_f.sum())
    if N < 8:
        return ret(None, 1)
    E = K.T @ best_F @ K
    u, _, vt = np.linalg.svd(E)
    t = u[:, 2] / np.linalg.norm(u[:, 2])
    W = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    R1 = u @ W @ vt
    R2 = u @ W.T @ vt
    if np.linalg.det(R1) < 0:
        R1 = -R1
    if np.linalg.det(R2) < 0:
        R2 = -R2
    # the sign of a singular vector is arbitrary: (R1, R2, t, -t) as a SET is what is defined
    return ret(_pick(cam, [(R1, t), (R2, t), (R1, -t), (R2, -t)], x1, x2, inl_f, N, False), 1)


def _se3_exp(d):
    w, v = d[:3], d[3:]
    th2 = float(w @ w)
    th = math.sqrt(th2)
    if th < 1e-8:
        A, B, C = 1 - th2 / 6, 0.5 - th2 / 24, 1 / 6 - th2 / 120
    else:
        A = math.sin(th) / th
        B = (1 - math.cos(th)) / th2
        C = (1 - A) / th2
    Wx = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    W2 = Wx @ Wx
    return np.eye(3) + A * Wx + B * W2, (np.eye(3) + B * Wx + C * W2) @ v


def _orthonormal(R):
    """The rotation next to R: rows by Gram-Schmidt, the third as a cross product.  The starting pose of pose_only is a
    product of estimated rotations (constant-velocity prediction R_k R_{k-1}^T R_k), the optimiser only ever multiplies it
    by exact exponentials, so a deviation from orthonormality S (R = Q (I + S)) is never removed and comes back about
    three times larger in the next prediction: after 40 frames of tracking the 'rotation' is sheared by 1e-3 and the
    tracker is lost (Sophus / g2o keep unit quaternions; this is the matrix form's equivalent)."""
    r0 = R[0] / np.sqrt(R[0] @ R[0])
    r1 = R[1] - (R[1] @ r0) * r0
    r1 = r1 / np.sqrt(r1 @ r1)
    return np.stack([r0, r1, np.cross(r0, r1)])


def pose_only(cam: Camera, pts3d, obs, inv_sigma2, R, t):
    """-> (n_inliers, R, t, inlier mask); n_inliers < 0 on failure"""
    P = np.asarray(pts3d, np.float64).reshape(-1, 3)
    obs = np.asarray(obs, np.float64).reshape(-1, 2)
    w0 = np.asarray(inv_sigma2, np.float64)
    R, t = _orthonormal(np.array(R, np.float64)), np.array(t, np.float64)
    n = len(P)
    inl = np.ones(n, bool)
    if n < 3:
        return -1, R, t, inl
    delta = math.sqrt(5.991)
    n_in = n
    for rnd in range(4):
        robust = rnd < 2
        for _ in range(10):
            X = P @ R.T + t
            x, y, z = X[:, 0], X[:, 1], X[:, 2]
            use = inl & (z > 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                iz = 1.0 / z
            iz2 = iz * iz
            ex = obs[:, 0] - (cam.fx * x * iz + cam.cx)
            ey = obs[:, 1] - (cam.fy * y * iz + cam.cy)
            e2 = w0 * (ex * ex + ey * ey)
            with np.errstate(divide="ignore", invalid="ignore"):
                w = np.where(robust & (e2 > delta * delta), w0 * delta / np.sqrt(e2), w0)
            zero = np.zeros(n)
            J0 = np.stack([x * y * iz2 * cam.fx, -(1 + x * x * iz2) * cam.fx, y * iz * cam.fx, -iz * cam.fx, zero, x * iz2 * cam.fx], axis=1)
            J1 = np.stack([(1 + y * y * iz2) * cam.fy, -x * y * iz2 * cam.fy, -x * iz * cam.fy, zero, -iz * cam.fy, y * iz2 * cam.fy], axis=1)
            J0, J1, wu, exu, eyu = J0[use], J1[use], w[use], ex[use], ey[use]
            H = (J0 * wu[:, None]).T @ J0 + (J1 * wu[:, None]).T @ J1
            b = -((J0 * (wu * exu)[:, None]).sum(axis=0) + (J1 * (wu * eyu)[:, None]).sum(axis=0))
            H[np.diag_indices(6)] += 1e-6 * (1.0 + np.diag(H))
            try:
                np.linalg.cholesky(H)
            except np.linalg.LinAlgError:
                return -2, R, t, inl
            d = np.linalg.solve(H, b)
            dR, dt = _se3_exp(d)
            R, t = dR @ R, dR @ t + dt
            if np.max(np.abs(d)) < 1e-10:  # the round ends once a step has moved nothing (same rule as sst_pose_only)
                break
        X = P @ R.T + t
        z = X[:, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            ex = obs[:, 0] - (cam.fx * X[:, 0] / z + cam.cx)
            ey = obs[:, 1] - (cam.fy * X[:, 1] / z + cam.cy)
            chi2 = np.where(z > 0, w0 * (ex * ex + ey * ey), 1e30)
        inl = chi2 <= 5.991
        n_in = int(inl.sum())
        if n_in < 3:
            return -3, R, t, inl
    return n_in, R, t, inl


def _ba_terms(cam, Y, obs, w0):
    """per observation: valid mask, residual e (n,2), d e / d Y rows a, b (n,3), Huber weight, robust cost"""
    z = Y[:, 2]
    valid = z > 0
    with np.errstate(divide="ignore", invalid="ignore"):
        iz = 1.0 / z
        iz2 = iz * iz
        e = np.stack([obs[:, 0] - (cam.fx * Y[:, 0] * iz + cam.cx), obs[:, 1] - (cam.fy * Y[:, 1] * iz + cam.cy)], axis=1)
        zero = np.zeros(len(Y))
        a = np.stack([-cam.fx * iz, zero, cam.fx * Y[:, 0] * iz2], axis=1)
        b = np.stack([zero, -cam.fy * iz, cam.fy * Y[:, 1] * iz2], axis=1)
        e2 = w0 * (e * e).sum(axis=1)
        d2 = 5.991
        big = e2 > d2
        se, dl = np.sqrt(e2), math.sqrt(d2)
        w = np.where(big, w0 * dl / se, w0)
        rho = np.where(big, 2 * dl * se - d2, e2)
    return valid, e, a, b, w, rho


def _ba_cost(cam, obs1, obs2, w1, w2, R, t, X):
    v1, _, _, _, _, r1 = _ba_terms(cam, X, obs1, w1)
    v2, _, _, _, _, r2 = _ba_terms(cam, X @ R.T + t, obs2, w2)
    return float(np.where(v1, r1, 1e6).sum() + np.where(v2, r2, 1e6).sum())


def two_view_ba(cam: Camera, obs1, obs2, w1, w2, R, t, X, iterations=20):
    """Two-view bundle adjustment (camera 1 fixed; camera 2 and the points free).  -> (R, t, X, accepted)"""
    obs1, obs2 = np.asarray(obs1, np.float64).reshape(-1, 2), np.asarray(obs2, np.float64).reshape(-1, 2)
    w1, w2 = np.asarray(w1, np.float64), np.asarray(w2, np.float64)
    R, t, X = np.array(R, np.float64), np.array(t, np.float64), np.array(X, np.float64).reshape(-1, 3)
    n = len(X)
    if n < 6:
        return R, t, X, 0
    lam = -1.0
    cost = _ba_cost(cam, obs1, obs2, w1, w2, R, t, X)
    accepted = 0
    for _ in range(iterations):
        v1, e1, a1, b1, ww1, _ = _ba_terms(cam, X, obs1, w1)
        Y = X @ R.T + t
        v2, e2, a2, b2, ww2, _ = _ba_terms(cam, Y, obs2, w2)
        ww1 = np.where(v1, ww1, 0.0)
        ww2 = np.where(v2, ww2, 0.0)
        a1, b1, e1 = np.where(v1[:, None], a1, 0), np.where(v1[:, None], b1, 0), np.where(v1[:, None], e1, 0)
        a2, b2, e2 = np.where(v2[:, None], a2, 0), np.where(v2[:, None], b2, 0), np.where(v2[:, None], e2, 0)
        ax, bx = a2 @ R, b2 @ R                      # d e / d X through camera 2
        aj = np.concatenate([np.cross(Y, a2), a2], axis=1)  # d e / d xi = [Y x a, a]
        bj = np.concatenate([np.cross(Y, b2), b2], axis=1)
        Hll = (ww1[:, None, None] * (a1[:, :, None] * a1[:, None, :] + b1[:, :, None] * b1[:, None, :]) +
               ww2[:, None, None] * (ax[:, :, None] * ax[:, None, :] + bx[:, :, None] * bx[:, None, :]))
        bl = -(ww1[:, None] * (a1 * e1[:, :1] + b1 * e1[:, 1:]) + ww2[:, None] * (ax * e2[:, :1] + bx * e2[:, 1:]))
        Hpl = ww2[:, None, None] * (aj[:, :, None] * ax[:, None, :] + bj[:, :, None] * bx[:, None, :])  # n x 6 x 3
        Hpp = ((ww2[:, None] * aj).T @ aj) + ((ww2[:, None] * bj).T @ bj)
        bp = -((ww2 * e2[:, 0]) @ aj + (ww2 * e2[:, 1]) @ bj)
        if lam < 0:
            lam = 1e-5 * max(np.diag(Hpp).max(), np.einsum("nii->ni", Hll).max())
        ok = True
        try:
            Hinv = np.linalg.inv(Hll + lam * np.eye(3))
            T = Hpl @ Hinv                                   # n x 6 x 3
            S = Hpp + lam * np.eye(6) - np.einsum("nij,nkj->ik", T, Hpl)
            g = bp - np.einsum("nij,nj->i", T, bl)
            np.linalg.cholesky(S)
            dp = np.linalg.solve(S, g)
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            dX = np.einsum("nij,nj->ni", Hinv, bl - np.einsum("nij,i->nj", Hpl, dp))
            dR, dt = _se3_exp(dp)
            Rn, tn, Xn = dR @ R, dR @ t + dt, X + dX
            new_cost = _ba_cost(cam, obs1, obs2, w1, w2, Rn, tn, Xn)
        if ok and new_cost < cost:
            R, t, X, cost = Rn, tn, Xn, new_cost
            lam = max(lam / 3.0, 1e-12)
            accepted += 1
        else:
            lam *= 4.0
    return R, t, X, accepted


def triangulate(cam: Camera, x1, x2, R1, t1, R2, t2, sigma2_1, sigma2_2):
    # LocalMapping::CreateNewMapPoints: only pairs whose viewing rays show parallax (0 < cos < 0.9998) are triangulated
    q1 = R1.T @ np.array([(x1[0] - cam.cx) / cam.fx, (x1[1] - cam.cy) / cam.fy, 1.0])
    q2 = R2.T @ np.array([(x2[0] - cam.cx) / cam.fx, (x2[1] - cam.cy) / cam.fy, 1.0])
    cr = float(q1 @ q2) / (np.sqrt(float(q1 @ q1)) * np.sqrt(float(q2 @ q2)))
    if not (0 < cr < 0.9998):
        return None
    X = _triangulate_dlt(cam.K, x1, x2, R1, t1, R2, t2)
    if X is None:
        return None
    a, b = R1 @ X + t1, R2 @ X + t2
    if a[2] <= 0 or b[2] <= 0:
        return None
    e1 = np.array([cam.fx * a[0] / a[2] + cam.cx, cam.fy * a[1] / a[2] + cam.cy]) - x1
    if e1 @ e1 > 5.991 * sigma2_1:
        return None
    e2 = np.array([cam.fx * b[0] / b[2] + cam.cx, cam.fy * b[1] / b[2] + cam.cy]) - x2
    if e2 @ e2 > 5.991 * sigma2_2:
        return None
    r1, r2 = X - (-R1.T @ t1), X - (-R2.T @ t2)
    cp = float(r1 @ r2) / (np.linalg.norm(r1) * np.linalg.norm(r2))
    return X if cp < 0.9998 else None


def pose_to_twc(R, t):
    """-> (position, quaternion x y z w): Eigen's Quaternion(Matrix3) branch rule"""
    Rwc = np.asarray(R).T
    pos = -Rwc @ np.asarray(t)
    tr = Rwc[0, 0] + Rwc[1, 1] + Rwc[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0)
        w = 0.5 * s
        s = 0.5 / s
        q = [(Rwc[2, 1] - Rwc[1, 2]) * s, (Rwc[0, 2] - Rwc[2, 0]) * s, (Rwc[1, 0] - Rwc[0, 1]) * s, w]
    else:
        i = 0
        if Rwc[1, 1] > Rwc[0, 0]:
            i = 1
        if Rwc[2, 2] > Rwc[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        s = math.sqrt(Rwc[i, i] - Rwc[j, j] - Rwc[k, k] + 1.0)
        v = [0.0, 0.0, 0.0]
        v[i] = 0.5 * s
        s = 0.5 / s
        w = (Rwc[k, j] - Rwc[j, k]) * s
        v[j] = (Rwc[j, i] + Rwc[i, j]) * s
        v[k] = (Rwc[k, i] + Rwc[i, k]) * s
        q = v + [w]
    return pos, np.array(q)


class _Frame:
    def __init__(self, und, octave):
        self.n = len(und)
        self.und = und
        self.octave = np.asarray(octave, np.int64)
        self.has3d = np.zeros(self.n, bool)
        self.p3d = np.zeros((self.n, 3))
        self.R, self.t = np.eye(3), np.zeros(3)
        self.anchor = {}  # keypoint -> (R, t, xy, sigma2) of the first observation of its track


MATCH_NONE, MATCH_REF, MATCH_PREV = 0, 1, 2
KEEP_NONE, KEEP_AS_REF, KEEP_AS_PREV = 0, 1, 2


def unique_matches(idx, d1, n_train):
    m = np.full(len(idx), -1, np.int64)
    owner = np.full(n_train, -1, np.int64)
    for i, j in enumerate(idx):
        if j < 0 or j >= n_train:
            continue
        if owner[j] < 0 or d1[i] < d1[owner[j]]:
            owner[j] = i
    for j in range(n_train):
        if owner[j] >= 0:
            m[owner[j]] = j
    return m


class Tracker:
    def __init__(self, cam: Camera, scale_factor: float = 1.2):
        self.cam = cam
        self.scale_factor = float(np.float32(scale_factor))
        self.reset()

    def reset(self):
        self.state = 0
        self.ref = None
        self.prev = None
        self.vel = None

    def want_match(self):
        if self.state == 2:
            return MATCH_PREV
        if self.state == 1 and self.ref is not None:
            return MATCH_REF
        return MATCH_NONE

    def n_train(self):
        w = self.want_match()
        return self.prev.n if w == MATCH_PREV else self.ref.n if w == MATCH_REF else 0

    def step(self, xy, octave, match_idx=None, d1=None):
        """-> (dict like binding.OrbContext.track, keep)"""
        cam = self.cam
        cur = _Frame(undistort(cam, xy), octave)
        n = cur.n
        out = {"state": 0, "n_matches": 0, "n_inliers": 0, "n_map_points": 0, "position": np.zeros(3),
               "quaternion": np.array([0, 0, 0, 1.0]), "n_keypoints": n}
        want = self.want_match()
        if want != MATCH_NONE:
            m = unique_matches(match_idx, d1, self.n_train())
            out["n_matches"] = int((m >= 0).sum())
        if want == MATCH_NONE:
            self.state = 1
            self.ref = cur if n > 100 else None
            out["state"] = 1
            return out, (KEEP_AS_REF if self.ref is not None else KEEP_NONE)
        if want == MATCH_REF:
            out["state"] = 1
            for i in range(n):  # the 100-px window of SearchForInitialization
                if m[i] >= 0 and not (abs(cur.und[i, 0] - self.ref.und[m[i], 0]) < 100.0 and
                                      abs(cur.und[i, 1] - self.ref.und[m[i], 1]) < 100.0):
                    m[i] = -1
            out["n_matches"] = int((m >= 0).sum())
            if n <= 100 or out["n_matches"] < 100:
                self.ref = None
                return out, KEEP_NONE
            qi = np.nonzero(m >= 0)[0]
            r = two_view(cam, self.ref.und[m[qi]], cur.und[qi])
            if r is None:
                return out, KEEP_NONE
            R, t, tri, p3d = r
            if tri.sum() == 0:
                return out, KEEP_NONE
            w1 = 1.0 / np.power(self.scale_factor, 2.0 * self.ref.octave[m[qi[tri]]])
            w2 = 1.0 / np.power(self.scale_factor, 2.0 * cur.octave[qi[tri]])
            R, t, Xb, _ = two_view_ba(cam, self.ref.und[m[qi[tri]]], cur.und[qi[tri]], w1, w2, R, t, p3d[tri], 20)
            p3d = p3d.copy()
            p3d[tri] = Xb
            depths = np.sort(p3d[tri, 2])
            median = depths[(len(depths) - 1) // 2]
            if not median > 0:
                self.ref = None
                return out, KEEP_NONE
            inv = 1.0 / median
            cur.R, cur.t = R, t * inv
            cur.has3d[qi[tri]] = True
            cur.p3d[qi[tri]] = p3d[tri] * inv
            self.state, self.ref, self.prev, self.vel = 2, None, cur, None
            out.update(state=2, n_inliers=int(tri.sum()), n_map_points=int(tri.sum()))
            out["position"], out["quaternion"] = pose_to_twc(cur.R, cur.t)
            return out, KEEP_AS_PREV
        prev = self.prev
        if self.vel is not None:  # constant-velocity prediction
            R0, t0 = self.vel[0] @ prev.R, self.vel[0] @ prev.t + self.vel[1]
        else:
            R0, t0 = prev.R, prev.t
        th = 15.0
        while True:  # SearchByProjection gate: th * scale^octave around the projected point
            qi = []
            for i in range(n):
                if m[i] < 0 or not prev.has3d[m[i]]:
                    continue
                Y = R0 @ prev.p3d[m[i]] + t0
                if Y[2] <= 0:
                    continue
                r = th * self.scale_factor ** float(cur.octave[i])
                if abs(cam.fx * Y[0] / Y[2] + cam.cx - cur.und[i, 0]) < r and abs(cam.fy * Y[1] / Y[2] + cam.cy - cur.und[i, 1]) < r:
                    qi.append(i)
            if len(qi) >= 20 or th >= 30.0:
                break
            th *= 2.0
        qi = np.array(qi, np.int64)
        P = prev.p3d[m[qi]] if len(qi) else np.zeros((0, 3))
        obs = cur.und[qi] if len(qi) else np.zeros((0, 2))
        w = 1.0 / np.power(self.scale_factor, 2.0 * cur.octave[qi]) if len(qi) else np.zeros(0)
        n_in, R, t, inl = pose_only(cam, P, obs, w, R0, t0)
        out["n_inliers"] = max(n_in, 0)
        if n_in < 30:
            self.vel = None
            self.state, self.prev = 4, None
            out["state"] = 4
            return out, KEEP_NONE
        cur.R, cur.t = R, t
        cur.has3d[qi[inl]] = True
        cur.p3d[qi[inl]] = P[inl]
        # new points: triangulate a track between its first observation (anchor) and this one
        for i in range(n):
            j = m[i]
            if j < 0 or prev.has3d[j]:
                continue
            anc = prev.anchor.get(j)
            if anc is None:
                anc = (prev.R, prev.t, prev.und[j].copy(), self.scale_factor ** (2.0 * prev.octave[j]))
            s2 = self.scale_factor ** (2.0 * cur.octave[i])
            X = triangulate(cam, anc[2], cur.und[i], anc[0], anc[1], cur.R, cur.t, anc[3], s2)
            if X is not None:
                cur.has3d[i] = True
                cur.p3d[i] = X
            else:
                cur.anchor[i] = anc
        out["n_map_points"] = int(cur.has3d.sum())
        Rv = cur.R @ prev.R.T
        self.vel = (Rv, cur.t - Rv @ prev.t)
        self.prev = cur
        out["state"] = 2
        out["position"], out["quaternion"] = pose_to_twc(cur.R, cur.t)
        return out, KEEP_AS_PREV
