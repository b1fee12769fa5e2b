/*
 * ss_track.cpp -- pose from matches: undistortion, two-view reconstruction, pose-only
 * optimisation, triangulation.  See ss_track.h for what each routine restates and why this is
 * host code (small dense double-precision math on <= 2000 points per frame).
 * Linear algebra is a cyclic Jacobi eigen-solver on small symmetric matrices (n <= 9): it yields
 * the null vectors of the DLT systems and the 3x3 SVDs, with no external library.
 */
#include "ss_track.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <thread>

namespace {

/* eigen-decomposition of a symmetric n x n matrix (row-major), ascending eigenvalues; V columns */
void jacobi_eig(int n, const double *A_in, double *eval, double *V)
{
    double A[81];
    memcpy(A, A_in, sizeof(double) * (size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) V[i * n + j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) (i == j ? diag : off) += A[i * n + j] * A[i * n + j];
        if (off <= 1e-30 * (diag > 0 ? diag : 1.0)) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = A[p * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
                const double tt = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < n; k++) {
                    const double akp = A[k * n + p], akq = A[k * n + q];
                    A[k * n + p] = c * akp - s * akq;
                    A[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {
                    const double apk = A[p * n + k], aqk = A[q * n + k];
                    A[p * n + k] = c * apk - s * aqk;
                    A[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; k++) {
                    const double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    int order[9];
    for (int i = 0; i < n; i++) order[i] = i;
    std::sort(order, order + n, [&](int a, int b) { return A[a * n + a] < A[b * n + b]; });
    double Vs[81];
    for (int k = 0; k < n; k++) {
        eval[k] = A[order[k] * n + order[k]];
        for (int i = 0; i < n; i++) Vs[i * n + k] = V[i * n + order[k]];
    }
    memcpy(V, Vs, sizeof(double) * (size_t)n * n);
}

/* Smallest eigenvector of a symmetric positive semi-definite n x n matrix (n <= 9, upper triangle read) by inverse
 * iteration on the slightly shifted matrix (one Cholesky factorisation, two triangular solves per step): the vector the
 * Jacobi solver returns, up to sign and a few ulp, for a fraction of its cost -- the 400 minimal-set models of an
 * initialisation attempt and the triangulation of every new point go through here.  false = no convergence (a repeated
 * smallest eigenvalue: a degenerate minimal set); the caller falls back to the Jacobi solver.  n is a template parameter:
 * the loops of the two sizes in use (4: triangulation, 9: homography / fundamental matrix) unroll. */
template <int n> bool smallest_eigvec(const double *M, double *x)
{
    double tr = 0;
    for (int i = 0; i < n; i++) tr += M[i * n + i];
    const double shift = 1e-13 * tr;
    if (!(shift > 0)) return false;
    double L[n * n] = {0}, id[n];
    for (int j = 0; j < n; j++) { /* M + shift I = L L^T */
        double d = M[j * n + j] + shift;
        for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0)) return false;
        L[j * n + j] = std::sqrt(d);
        const double inv = 1.0 / L[j * n + j];
        id[j] = inv; /* the solves below multiply by the reciprocals: no division inside the iteration */
        for (int i = j + 1; i < n; i++) {
            double v = M[j * n + i];
            for (int k = 0; k < j; k++) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = v * inv;
        }
    }
    double v[n], y[n];
    const double v0 = 1.0 / std::sqrt((double)n);
    for (int i = 0; i < n; i++) v[i] = v0;
    for (int it = 0; it < 60; it++) {
        for (int i = 0; i < n; i++) {
            double t = v[i];
            for (int k = 0; k < i; k++) t -= L[i * n + k] * y[k];
            y[i] = t * id[i];
        }
        for (int i = n - 1; i >= 0; i--) {
            double t = y[i];
            for (int k = i + 1; k < n; k++) t -= L[k * n + i] * y[k];
            y[i] = t * id[i];
        }
        double nrm = 0;
        for (int i = 0; i < n; i++) nrm += y[i] * y[i];
        nrm = std::sqrt(nrm);
        if (!(nrm > 0) || !std::isfinite(nrm)) return false;
        const double inrm = 1.0 / nrm;
        double diff = 0, dot = 0;
        for (int i = 0; i < n; i++) { y[i] *= inrm; dot += y[i] * v[i]; }
        const double sgn = dot < 0 ? -1.0 : 1.0;
        for (int i = 0; i < n; i++) { const double w = sgn * y[i]; diff = std::max(diff, std::fabs(w - v[i])); v[i] = w; }
        if (diff < 1e-15 && it > 0) {
            for (int i = 0; i < n; i++) x[i] = v[i];
            return true;
        }
    }
    return false;
}

/* unit null vector (smallest singular vector) of an m x n system, n <= 9 */
void null_vector(int m, int n, const double *A, double *x)
{
    double AtA[81] = {0}, ev[9], V[81];
    for (int r = 0; r < m; r++)
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) AtA[i * n + j] += A[r * n + i] * A[r * n + j];
    if (n == 4 ? smallest_eigvec<4>(AtA, x) : n == 9 ? smallest_eigvec<9>(AtA, x) : false) return;
    for (int i = 1; i < n; i++)
        for (int j = 0; j < i; j++) AtA[i * n + j] = AtA[j * n + i];
    jacobi_eig(n, AtA, ev, V);
    for (int i = 0; i < n; i++) x[i] = V[i * n + 0];
}

void mat3_mul(const double *a, const double *b, double *c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
void mat3_t(const double *a, double *c)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * j + i];
}
double det3(const double *m)
{
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* M = U diag(s) V^T, s descending; third left vector completed by a cross product when s3 ~ 0 */
void svd3(const double *M, double *U, double *s, double *V)
{
    double MtM[9], Mt[9], ev[3], Ve[9];
    mat3_t(M, Mt);
    mat3_mul(Mt, M, MtM);
    jacobi_eig(3, MtM, ev, Ve);
    for (int k = 0; k < 3; k++) { /* descending */
        const int src = 2 - k;
        s[k] = std::sqrt(ev[src] > 0 ? ev[src] : 0.0);
        for (int i = 0; i < 3; i++) V[3 * i + k] = Ve[3 * i + src];
    }
    for (int k = 0; k < 3; k++) {
        double u[3];
        for (int i = 0; i < 3; i++) u[i] = M[3 * i] * V[k] + M[3 * i + 1] * V[3 + k] + M[3 * i + 2] * V[6 + k];
        const double nrm = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
        if (k == 2 && !(nrm > 1e-9 * s[0])) { /* rank 2: complete the basis */
            U[2] = U[3 * 1 + 0] * U[3 * 2 + 1] - U[3 * 2 + 0] * U[3 * 1 + 1];
            U[5] = U[3 * 2 + 0] * U[3 * 0 + 1] - U[3 * 0 + 0] * U[3 * 2 + 1];
            U[8] = U[3 * 0 + 0] * U[3 * 1 + 1] - U[3 * 1 + 0] * U[3 * 0 + 1];
            break;
        }
        for (int i = 0; i < 3; i++) U[3 * i + k] = nrm > 0 ? u[i] / nrm : (i == k ? 1.0 : 0.0);
    }
}

bool inv3_full(const double *m, double *o)
{
    const double d = det3(m);
    if (!(std::fabs(d) > 1e-300)) return false;
    const double id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

struct lcg { /* deterministic sampler (the reference seeds rand() with 0; any fixed generator serves) */
    uint64_t x;
    uint32_t next(uint32_t mod)
    {
        x = x * 6364136223846793005ull + 1442695040888963407ull;
        return (uint32_t)((x >> 33) % mod);
    }
};

void normalize_pts(int n, const double *p, std::vector<double> &out, double T[9])
{
    double mx = 0, my = 0;
    for (int i = 0; i < n; i++) { mx += p[2 * i]; my += p[2 * i + 1]; }
    mx /= n; my /= n;
    double dx = 0, dy = 0;
    out.resize((size_t)2 * n);
    for (int i = 0; i < n; i++) {
        out[2 * i] = p[2 * i] - mx;
        out[2 * i + 1] = p[2 * i + 1] - my;
        dx += std::fabs(out[2 * i]);
        dy += std::fabs(out[2 * i + 1]);
    }
    dx /= n; dy /= n;
    const double sx = 1.0 / dx, sy = 1.0 / dy;
    for (int i = 0; i < n; i++) { out[2 * i] *= sx; out[2 * i + 1] *= sy; }
    const double Tm[9] = {sx, 0, -mx * sx, 0, sy, -my * sy, 0, 0, 1};
    memcpy(T, Tm, sizeof(Tm));
}

void compute_f21(const double *p1, const double *p2, const int idx[8], double F[9])
{
    double A[8 * 9];
    for (int k = 0; k < 8; k++) {
        const double u1 = p1[2 * idx[k]], v1 = p1[2 * idx[k] + 1], u2 = p2[2 * idx[k]], v2 = p2[2 * idx[k] + 1];
        double *r = A + 9 * k;
        r[0] = u2 * u1; r[1] = u2 * v1; r[2] = u2; r[3] = v2 * u1; r[4] = v2 * v1; r[5] = v2; r[6] = u1; r[7] = v1; r[8] = 1;
    }
    double f[9];
    null_vector(8, 9, A, f);
    double U[9], s[3], V[9];
    svd3(f, U, s, V);
    s[2] = 0; /* rank 2 */
    double US[9], Vt[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) US[3 * i + j] = U[3 * i + j] * s[j];
    mat3_t(V, Vt);
    mat3_mul(US, Vt, F);
}

double check_fundamental(const double *F, int n, const double *x1, const double *x2, std::vector<uint8_t> &inl)
{
    const double th = 3.841, th_score = 5.991, inv_sigma2 = 1.0;
    double score = 0;
    inl.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        bool in = true;
        const double a2 = F[0] * u1 + F[1] * v1 + F[2], b2 = F[3] * u1 + F[4] * v1 + F[5], c2 = F[6] * u1 + F[7] * v1 + F[8];
        const double num2 = a2 * u2 + b2 * v2 + c2;
        const double chi1 = num2 * num2 / (a2 * a2 + b2 * b2) * inv_sigma2;
        if (chi1 > th) in = false; else score += th_score - chi1;
        const double a1 = F[0] * u2 + F[3] * v2 + F[6], b1 = F[1] * u2 + F[4] * v2 + F[7], c1 = F[2] * u2 + F[5] * v2 + F[8];
        const double num1 = a1 * u1 + b1 * v1 + c1;
        const double chi2 = num1 * num1 / (a1 * a1 + b1 * b1) * inv_sigma2;
        if (chi2 > th) in = false; else score += th_score - chi2;
        inl[i] = in;
    }
    return score;
}

void compute_h21(const double *p1, const double *p2, const int idx[8], double H[9])
{
    double A[16 * 9];
    for (int k = 0; k < 8; k++) {
        const double u1 = p1[2 * idx[k]], v1 = p1[2 * idx[k] + 1], u2 = p2[2 * idx[k]], v2 = p2[2 * idx[k] + 1];
        double *r = A + 18 * k;
        r[0] = 0; r[1] = 0; r[2] = 0; r[3] = -u1; r[4] = -v1; r[5] = -1; r[6] = v2 * u1; r[7] = v2 * v1; r[8] = v2;
        r[9] = u1; r[10] = v1; r[11] = 1; r[12] = 0; r[13] = 0; r[14] = 0; r[15] = -u2 * u1; r[16] = -u2 * v1; r[17] = -u2;
    }
    null_vector(16, 9, A, H);
}

double check_homography(const double *H21, const double *H12, int n, const double *x1, const double *x2, std::vector<uint8_t> &inl)
{
    const double th = 5.991, inv_sigma2 = 1.0;
    double score = 0;
    inl.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        const double u1 = x1[2 * i], v1 = x1[2 * i + 1], u2 = x2[2 * i], v2 = x2[2 * i + 1];
        bool in = true;
        const double w21 = 1.0 / (H12[6] * u2 + H12[7] * v2 + H12[8]); /* x2 seen in image 1 */
        const double a = (H12[0] * u2 + H12[1] * v2 + H12[2]) * w21, b = (H12[3] * u2 + H12[4] * v2 + H12[5]) * w21;
        const double chi1 = ((u1 - a) * (u1 - a) + (v1 - b) * (v1 - b)) * inv_sigma2;
        if (chi1 > th) in = false; else score += th - chi1;
        const double w12 = 1.0 / (H21[6] * u1 + H21[7] * v1 + H21[8]); /* x1 seen in image 2 */
        const double cc = (H21[0] * u1 + H21[1] * v1 + H21[2]) * w12, d = (H21[3] * u1 + H21[4] * v1 + H21[5]) * w12;
        const double chi2 = ((u2 - cc) * (u2 - cc) + (v2 - d) * (v2 - d)) * inv_sigma2;
        if (chi2 > th) in = false; else score += th - chi2;
        inl[i] = in;
    }
    return score;
}

/* P = K [R | t], row-major 3 x 4 */
void projection_matrix(const sst_camera &c, const double R[9], const double t[3], double P[12])
{
    const double K[9] = {c.fx, 0, c.cx, 0, c.fy, c.cy, 0, 0, 1};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
            double a = 0;
            for (int k = 0; k < 3; k++) a += K[3 * i + k] * (j < 3 ? R[3 * k + j] : t[k]);
            P[4 * i + j] = a;
        }
}

/* linear triangulation with the two projection matrices; returns false if the homogeneous weight vanishes */
bool triangulate_dlt(const double P1[12], const double P2[12], const double x1[2], const double x2[2], double X[3])
{
    double A[16];
    for (int j = 0; j < 4; j++) {
        A[j] = x1[0] * P1[8 + j] - P1[j];
        A[4 + j] = x1[1] * P1[8 + j] - P1[4 + j];
        A[8 + j] = x2[0] * P2[8 + j] - P2[j];
        A[12 + j] = x2[1] * P2[8 + j] - P2[4 + j];
    }
    double h[4];
    null_vector(4, 4, A, h);
    if (h[3] == 0 || !std::isfinite(h[3])) return false;
    X[0] = h[0] / h[3]; X[1] = h[1] / h[3]; X[2] = h[2] / h[3];
    return std::isfinite(X[0]) && std::isfinite(X[1]) && std::isfinite(X[2]);
}

int check_rt(const sst_camera &c, const double R[9], const double t[3], int n, const double *x1, const double *x2,
             const std::vector<uint8_t> &inl, double th2, std::vector<double> &p3d, std::vector<uint8_t> &good, double &parallax)
{
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
    const double O2[3] = {-(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]), -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]),
                          -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2])};
    p3d.assign((size_t)3 * n, 0.0);
    good.assign((size_t)n, 0);
    std::vector<double> cosp;
    int n_good = 0;
    double P1[12], P2[12]; /* the same two for every point */
    projection_matrix(c, I, z3, P1);
    projection_matrix(c, R, t, P2);
    for (int i = 0; i < n; i++) {
        if (!inl[i]) continue;
        double X[3];
        if (!triangulate_dlt(P1, P2, x1 + 2 * i, x2 + 2 * i, X)) continue;
        const double n2[3] = {X[0] - O2[0], X[1] - O2[1], X[2] - O2[2]};
        const double d1 = std::sqrt(X[0] * X[0] + X[1] * X[1] + X[2] * X[2]);
        const double d2 = std::sqrt(n2[0] * n2[0] + n2[1] * n2[1] + n2[2] * n2[2]);
        const double cp = (X[0] * n2[0] + X[1] * n2[1] + X[2] * n2[2]) / (d1 * d2);
        if (X[2] <= 0 && cp < 0.99998) continue;
        const double Y[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                             R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2]};
        if (Y[2] <= 0 && cp < 0.99998) continue;
        const double e1x = c.fx * X[0] / X[2] + c.cx - x1[2 * i], e1y = c.fy * X[1] / X[2] + c.cy - x1[2 * i + 1];
        if (e1x * e1x + e1y * e1y > th2) continue;
        const double e2x = c.fx * Y[0] / Y[2] + c.cx - x2[2 * i], e2y = c.fy * Y[1] / Y[2] + c.cy - x2[2 * i + 1];
        if (e2x * e2x + e2y * e2y > th2) continue;
        cosp.push_back(cp);
        p3d[3 * i] = X[0]; p3d[3 * i + 1] = X[1]; p3d[3 * i + 2] = X[2];
        n_good++;
        if (cp < 0.99998) good[i] = 1;
    }
    parallax = 0;
    if (n_good > 0) {
        std::sort(cosp.begin(), cosp.end());
        const size_t k = std::min<size_t>(50, cosp.size() - 1);
        /* a cosine of two nearly parallel rays can round to 1 + 2e-16: acos of that is NaN, and a NaN parallax passes no test */
        parallax = std::acos(std::min(1.0, std::max(-1.0, cosp[k]))) * 180.0 / 3.14159265358979323846;
    }
    return n_good;
}

void se3_exp(const double d[6], double R[9], double t[3])
{
    const double wx = d[0], wy = d[1], wz = d[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
    double A, B, Cc;
    if (th < 1e-8) { A = 1.0 - th2 / 6.0; B = 0.5 - th2 / 24.0; Cc = 1.0 / 6.0 - th2 / 120.0; }
    else { A = std::sin(th) / th; B = (1.0 - std::cos(th)) / th2; Cc = (1.0 - A) / th2; }
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9], V[9];
    mat3_mul(W, W, W2);
    for (int i = 0; i < 9; i++) {
        const double I = (i % 4 == 0) ? 1.0 : 0.0;
        R[i] = I + A * W[i] + B * W2[i];
        V[i] = I + B * W[i] + Cc * W2[i];
    }
    for (int i = 0; i < 3; i++) t[i] = V[3 * i] * d[3] + V[3 * i + 1] * d[4] + V[3 * i + 2] * d[5];
}

bool chol6_solve(double H[36], double b[6])
{
    for (int j = 0; j < 6; j++) {
        double s = H[7 * j];
        for (int k = 0; k < j; k++) s -= H[6 * j + k] * H[6 * j + k];
        if (s <= 0) return false;
        H[7 * j] = std::sqrt(s);
        for (int i = j + 1; i < 6; i++) {
            double v = H[6 * i + j];
            for (int k = 0; k < j; k++) v -= H[6 * i + k] * H[6 * j + k];
            H[6 * i + j] = v / H[7 * j];
        }
    }
    for (int i = 0; i < 6; i++) {
        double v = b[i];
        for (int k = 0; k < i; k++) v -= H[6 * i + k] * b[k];
        b[i] = v / H[7 * i];
    }
    for (int i = 5; i >= 0; i--) {
        double v = b[i];
        for (int k = i + 1; k < 6; k++) v -= H[6 * k + i] * b[k];
        b[i] = v / H[7 * i];
    }
    return true;
}

} // namespace

void sst_undistort(const sst_camera &c, int n, const float *xy_in, double *xy_out)
{
    const bool none = c.k1 == 0.0 && c.k2 == 0.0 && c.p1 == 0.0 && c.p2 == 0.0;
    for (int i = 0; i < n; i++) {
        if (none) {
            xy_out[2 * i] = xy_in[2 * i];
            xy_out[2 * i + 1] = xy_in[2 * i + 1];
            continue;
        }
        const double x0 = (xy_in[2 * i] - c.cx) / c.fx, y0 = (xy_in[2 * i + 1] - c.cy) / c.fy;
        double x = x0, y = y0;
        for (int it = 0; it < 5; it++) { /* cv::undistortPoints fixed-point iteration */
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1.0 + (c.k2 * r2 + c.k1) * r2);
            const double dx = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
            const double dy = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
            x = (x0 - dx) * icdist;
            y = (y0 - dy) * icdist;
        }
        xy_out[2 * i] = x * c.fx + c.cx;
        xy_out[2 * i + 1] = y * c.fy + c.cy;
    }
}

namespace {

/* picks the hypothesis CheckRT likes best, with ReconstructF's / ReconstructH's acceptance rule */
int pick_hypothesis(const sst_camera &c, int n_hyp, const double (*Rs)[9], const double (*ts)[3], int n, const double *x1,
                    const double *x2, const std::vector<uint8_t> &inl, int N, bool homography, double R[9], double t[3],
                    std::vector<uint8_t> &triangulated, std::vector<double> &pts3d)
{
    int best = -1, best_good = 0, second_good = 0, n_similar = 0;
    double best_parallax = -1;
    /* the hypotheses are checked side by side (each triangulates every inlier): one thread per hypothesis, then the
     * sequential choice over their results in hypothesis order */
    std::vector<std::vector<double>> p3ds((size_t)n_hyp);
    std::vector<std::vector<uint8_t>> flags((size_t)n_hyp);
    std::vector<int> goods((size_t)n_hyp);
    std::vector<double> parallaxes((size_t)n_hyp, 0.0);
    {
        std::vector<std::thread> th;
        auto run = [&](int k) { goods[(size_t)k] = check_rt(c, Rs[k], ts[k], n, x1, x2, inl, 4.0, p3ds[(size_t)k], flags[(size_t)k], parallaxes[(size_t)k]); };
        for (int k = 1; k < n_hyp; k++) th.emplace_back(run, k);
        run(0);
        for (auto &t_ : th) t_.join();
    }
    for (int k = 0; k < n_hyp; k++) {
        const int g = goods[(size_t)k];
        if (g > best_good) {
            second_good = best_good;
            best_good = g;
            best = k;
            best_parallax = parallaxes[(size_t)k];
        } else if (g > second_good) {
            second_good = g;
        }
    }
    if (best < 0) return 0;
    if (homography) {
        if (!(second_good < 0.75 * best_good && best_parallax >= 1.0 && best_good > 50 && best_good > 0.9 * N)) return 0;
    } else {
        for (int k = 0; k < n_hyp; k++) n_similar += goods[k] > 0.7 * best_good;
        if (best_good < std::max((int)(0.9 * N), 50) || n_similar > 1 || !(best_parallax > 1.0)) return 0;
    }
    memcpy(R, Rs[best], sizeof(double) * 9);
    memcpy(t, ts[best], sizeof(double) * 3);
    triangulated = std::move(flags[(size_t)best]);
    pts3d = std::move(p3ds[(size_t)best]);
    int cnt = 0;
    for (uint8_t b : triangulated) cnt += b;
    return cnt;
}

} // namespace

int sst_two_view(const sst_camera &c, int n, const double *x1, const double *x2, double R[9], double t[3],
                 std::vector<uint8_t> &triangulated, std::vector<double> &pts3d, int *model)
{
    triangulated.assign((size_t)n, 0);
    pts3d.assign((size_t)3 * n, 0.0);
    if (model) *model = 0;
    if (n < 8) return 0;
    std::vector<double> n1, n2;
    double T1[9], T2[9], T2t[9], T2inv[9];
    normalize_pts(n, x1, n1, T1);
    normalize_pts(n, x2, n2, T2);
    mat3_t(T2, T2t);
    if (!inv3_full(T2, T2inv)) return 0;

    /* FindHomography and FindFundamental on the same 200 minimal sets, side by side on threads as ORB-SLAM3 runs them
     * (Initializer: threadH / threadF): the sets are drawn first, so the outcome does not depend on the threads.  An initialisation attempt is by far the most expensive frame
     * of a connection (2 x 200 models scored on every match). */
    lcg rng{0x9E3779B97F4A7C15ull};
    std::vector<int> avail((size_t)n);
    std::vector<std::array<int, 8>> sets(200);
    for (int it = 0; it < 200; it++) {
        for (int i = 0; i < n; i++) avail[i] = i;
        int na = n;
        for (int j = 0; j < 8; j++) {
            const int r = (int)rng.next((uint32_t)na);
            sets[(size_t)it][(size_t)j] = avail[r];
            avail[r] = avail[na - 1];
            na--;
        }
    }
    /* each model's 200 hypotheses are dealt round-robin to SST_RANSAC_THREADS threads; a thread keeps its best (highest score,
     * first iteration on a tie), the bests are folded with the same rule: the winner is the one the sequential loop keeps */
    struct part_best {
        double score = -1;
        int it = 1 << 30;
        double M[9] = {0};
        std::vector<uint8_t> inl;
    };
    constexpr int NT = SST_RANSAC_THREADS;
    part_best ph[NT], pf[NT];
    auto fit_part = [&](bool homography, int k) {
        part_best &b = homography ? ph[k] : pf[k];
        std::vector<uint8_t> inl;
        for (int it = k; it < 200; it += NT) {
            double Mn[9], tmp[9], M[9], Minv[9];
            double sc;
            if (homography) {
                compute_h21(n1.data(), n2.data(), sets[(size_t)it].data(), Mn);
                mat3_mul(T2inv, Mn, tmp);
                mat3_mul(tmp, T1, M);
                if (!inv3_full(M, Minv)) continue;
                sc = check_homography(M, Minv, n, x1, x2, inl);
            } else {
                compute_f21(n1.data(), n2.data(), sets[(size_t)it].data(), Mn);
                mat3_mul(T2t, Mn, tmp);
                mat3_mul(tmp, T1, M);
                sc = check_fundamental(M, n, x1, x2, inl);
            }
            if (sc > b.score) {
                b.score = sc;
                b.it = it;
                b.inl = inl;
                memcpy(b.M, M, sizeof(M));
            }
        }
    };
    {
        std::thread th[2 * NT - 1];
        for (int k = 0; k < NT; k++) th[k] = std::thread(fit_part, true, k);
        for (int k = 1; k < NT; k++) th[NT + k - 1] = std::thread(fit_part, false, k);
        fit_part(false, 0);
        for (auto &t_ : th) t_.join();
    }
    std::vector<uint8_t> best_inl_f, best_inl_h;
    double best_sf = -1, best_sh = -1, best_F[9] = {0}, best_H[9] = {0};
    {
        int wh = -1, wf = -1;
        for (int k = 0; k < NT; k++) {
            if (ph[k].score > -1 && (wh < 0 || ph[k].score > ph[wh].score || (ph[k].score == ph[wh].score && ph[k].it < ph[wh].it))) wh = k;
            if (pf[k].score > -1 && (wf < 0 || pf[k].score > pf[wf].score || (pf[k].score == pf[wf].score && pf[k].it < pf[wf].it))) wf = k;
        }
        if (wh >= 0) {
            best_sh = ph[wh].score;
            best_inl_h = std::move(ph[wh].inl);
            memcpy(best_H, ph[wh].M, sizeof(best_H));
        }
        if (wf >= 0) {
            best_sf = pf[wf].score;
            best_inl_f = std::move(pf[wf].inl);
            memcpy(best_F, pf[wf].M, sizeof(best_F));
        }
    }
    const double K[9] = {c.fx, 0, c.cx, 0, c.fy, c.cy, 0, 0, 1};
    double tmp[9];
    /* Model selection: RH = SH / (SH + SF) above the threshold -> homography.  ORB-SLAM2 uses 0.40; ORB-SLAM3's
     * source reads `if(RH>0.50) // if(RH>0.40)` under the comment "depending on the ratio (0.40-0.45)".  A
     * planar scene scores RH ~ 0.49 (a line distance is never larger than a point distance, so SF >= SH whenever
     * both models fit), i.e. at 0.50 it would be handed to the degenerate fundamental matrix: 0.45 is used. */
    const bool use_h = best_sh > 0 && best_sh / (best_sh + (best_sf > 0 ? best_sf : 0.0)) > SST_RH_THRESHOLD;
    if (use_h) {
        if (model) *model = 2;
        int N = 0;
        for (uint8_t b : best_inl_h) N += b;
        if (N < 8) return 0;
        /* ReconstructH: Faugeras & Lustman, eight hypotheses from the SVD of K^-1 H K */
        double Kinv[9], A[9], U[9], w[3], V[9], Vt[9];
        if (!inv3_full(K, Kinv)) return 0;
        mat3_mul(Kinv, best_H, tmp);
        mat3_mul(tmp, K, A);
        svd3(A, U, w, V);
        mat3_t(V, Vt);
        const double sgn = det3(U) * det3(Vt);
        const double d1 = w[0], d2 = w[1], d3 = w[2];
        if (d1 / d2 < 1.00001 || d2 / d3 < 1.00001) return 0;
        const double aux1 = std::sqrt((d1 * d1 - d2 * d2) / (d1 * d1 - d3 * d3)), aux3 = std::sqrt((d2 * d2 - d3 * d3) / (d1 * d1 - d3 * d3));
        const double x1s[4] = {aux1, aux1, -aux1, -aux1}, x3s[4] = {aux3, -aux3, aux3, -aux3};
        const double aux_st = std::sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 + d3) * d2);
        const double ct = (d2 * d2 + d1 * d3) / ((d1 + d3) * d2);
        const double st[4] = {aux_st, -aux_st, -aux_st, aux_st};
        const double aux_sp = std::sqrt((d1 * d1 - d2 * d2) * (d2 * d2 - d3 * d3)) / ((d1 - d3) * d2);
        const double cp = (d1 * d3 - d2 * d2) / ((d1 - d3) * d2);
        const double sp[4] = {aux_sp, -aux_sp, -aux_sp, aux_sp};
        double Rs[8][9], ts[8][3];
        for (int i = 0; i < 8; i++) {
            const int j = i & 3;
            const bool pos = i < 4; /* d' = +d2, then d' = -d2 */
            const double Rp[9] = {pos ? ct : cp, 0, pos ? -st[j] : sp[j], 0, pos ? 1.0 : -1.0, 0, pos ? st[j] : sp[j], 0, pos ? ct : -cp};
            mat3_mul(U, Rp, tmp);
            mat3_mul(tmp, Vt, Rs[i]);
            for (double &v : Rs[i]) v *= sgn;
            const double scale = pos ? d1 - d3 : d1 + d3;
            const double tp[3] = {x1s[j] * scale, 0, (pos ? -x3s[j] : x3s[j]) * scale};
            double tt[3], nrm = 0;
            for (int r = 0; r < 3; r++) {
                tt[r] = U[3 * r] * tp[0] + U[3 * r + 1] * tp[1] + U[3 * r + 2] * tp[2];
                nrm += tt[r] * tt[r];
            }
            nrm = std::sqrt(nrm);
            for (int r = 0; r < 3; r++) ts[i][r] = tt[r] / nrm;
        }
        return pick_hypothesis(c, 8, Rs, ts, n, x1, x2, best_inl_h, N, true, R, t, triangulated, pts3d);
    }

    if (model) *model = 1;
    int N = 0;
    for (uint8_t b : best_inl_f) N += b;
    if (N < 8) return 0;
    /* ReconstructF */
    double Kt[9], E[9];
    mat3_t(K, Kt);
    mat3_mul(Kt, best_F, tmp);
    mat3_mul(tmp, K, E);
    double U[9], sv[3], V[9], Vt[9];
    svd3(E, U, sv, V);
    mat3_t(V, Vt);
    double tt[3] = {U[2], U[5], U[8]};
    const double tn = std::sqrt(tt[0] * tt[0] + tt[1] * tt[1] + tt[2] * tt[2]);
    for (double &v : tt) v /= tn;
    const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double Rs[4][9], ts[4][3];
    mat3_mul(U, W, tmp);
    mat3_mul(tmp, Vt, Rs[0]);
    mat3_mul(U, Wt, tmp);
    mat3_mul(tmp, Vt, Rs[1]);
    if (det3(Rs[0]) < 0) for (double &v : Rs[0]) v = -v;
    if (det3(Rs[1]) < 0) for (double &v : Rs[1]) v = -v;
    memcpy(Rs[2], Rs[0], sizeof(Rs[0]));
    memcpy(Rs[3], Rs[1], sizeof(Rs[1]));
    for (int r = 0; r < 3; r++) { ts[0][r] = ts[1][r] = tt[r]; ts[2][r] = ts[3][r] = -tt[r]; }
    return pick_hypothesis(c, 4, Rs, ts, n, x1, x2, best_inl_f, N, false, R, t, triangulated, pts3d);
}

namespace {

/* The normal equations of one Gauss-Newton step of sst_pose_only: H (upper triangle, 21 sums) and b (6 sums) over the
 * active points.  Points go through in blocks: first everything that is element-wise (projection, residual, weight, the
 * two Jacobian rows) into arrays, then the 27 sums with four partial sums each, lane l taking points l, l + 4, ... of the
 * block -- loops a compiler turns into 2- or 4-wide vector code without being allowed to reorder a sum.  (As one scalar
 * loop over the points this was 60 % of a tracked frame's host time.)  The order of additions differs from the scalar loop's
 * by that partial-sum split only: differences of a few ulp, against a parity tolerance of 1e-6 (oracle/vo_oracle.py). */
constexpr int PO_BLOCK = 64;
#if defined(__x86_64__) && defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(SST_NO_TARGET_CLONES)
__attribute__((target_clones("avx2", "default")))
#endif
void pose_normal_equations(int n, const int *active, const double *pts3d, const double *obs, const double *inv_sigma2,
                           const sst_camera &c, const double R[9], const double t[3], bool robust, double delta, double H[36], double b[6])
{
    /* rows of the two Jacobians: J0 = {0, 1, 2, 3, -, 5}, J1 = {0, 1, 2, -, 4, 5} (entries 4 and 3 are zero) */
    alignas(32) double j0[5][PO_BLOCK], j1[5][PO_BLOCK], ex[PO_BLOCK], ey[PO_BLOCK], w[PO_BLOCK];
    alignas(32) double acc[27][4];
    for (auto &a : acc) a[0] = a[1] = a[2] = a[3] = 0.0;
    const double fx = c.fx, fy = c.fy, cx = c.cx, cy = c.cy, d2 = delta * delta;
    for (int i0 = 0; i0 < n; i0 += PO_BLOCK) {
        const int m = std::min(PO_BLOCK, n - i0), m4 = (m + 3) & ~3;
        for (int k = 0; k < m; k++) {
            const int i = active[i0 + k];
            const double *P = pts3d + 3 * i;
            const double x = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0];
            const double y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1];
            const double z0 = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
            const bool front = z0 > 0;
            const double z = front ? z0 : 1.0;
            const double iz = 1.0 / z, iz2 = iz * iz;
            const double rx = obs[2 * i] - (fx * x * iz + cx), ry = obs[2 * i + 1] - (fy * y * iz + cy);
            const double w0 = inv_sigma2[i];
            const double e2 = w0 * (rx * rx + ry * ry);
            const double wr = (robust && e2 > d2) ? w0 * delta / std::sqrt(e2) : w0;
            w[k] = front ? wr : 0.0; /* a point behind the camera takes no part */
            ex[k] = rx;
            ey[k] = ry;
            j0[0][k] = x * y * iz2 * fx;  j0[1][k] = -(1 + x * x * iz2) * fx;  j0[2][k] = y * iz * fx;  j0[3][k] = -iz * fx;  j0[4][k] = x * iz2 * fx;
            j1[0][k] = (1 + y * y * iz2) * fy;  j1[1][k] = -x * y * iz2 * fy;  j1[2][k] = -x * iz * fy;  j1[3][k] = -iz * fy;  j1[4][k] = y * iz2 * fy;
        }
        for (int k = m; k < m4; k++) {
            w[k] = ex[k] = ey[k] = 0.0;
            for (int a = 0; a < 5; a++) j0[a][k] = j1[a][k] = 0.0;
        }
        /* index maps: J0 rows {0,1,2,3,5} are j0[0..4], J1 rows {0,1,2,4,5} are j1[0..4] */
        for (int k = 0; k < m4; k += 4)
            for (int l = 0; l < 4; l++) {
                const int q = k + l;
                const double wq = w[q];
                const double a0 = j0[0][q], a1 = j0[1][q], a2 = j0[2][q], a3 = j0[3][q], a5 = j0[4][q];
                const double c0 = j1[0][q], c1 = j1[1][q], c2 = j1[2][q], c4 = j1[3][q], c5 = j1[4][q];
                const double wa0 = wq * a0, wa1 = wq * a1, wa2 = wq * a2, wa3 = wq * a3, wa5 = wq * a5;
                const double wc0 = wq * c0, wc1 = wq * c1, wc2 = wq * c2, wc4 = wq * c4, wc5 = wq * c5;
                acc[0][l] += wa0 * a0 + wc0 * c0;   /* H00 */
                acc[1][l] += wa0 * a1 + wc0 * c1;   /* H01 */
                acc[2][l] += wa0 * a2 + wc0 * c2;   /* H02 */
                acc[3][l] += wa0 * a3;              /* H03 */
                acc[4][l] += wc0 * c4;              /* H04 */
                acc[5][l] += wa0 * a5 + wc0 * c5;   /* H05 */
                acc[6][l] += wa1 * a1 + wc1 * c1;   /* H11 */
                acc[7][l] += wa1 * a2 + wc1 * c2;   /* H12 */
                acc[8][l] += wa1 * a3;              /* H13 */
                acc[9][l] += wc1 * c4;              /* H14 */
                acc[10][l] += wa1 * a5 + wc1 * c5;  /* H15 */
                acc[11][l] += wa2 * a2 + wc2 * c2;  /* H22 */
                acc[12][l] += wa2 * a3;             /* H23 */
                acc[13][l] += wc2 * c4;             /* H24 */
                acc[14][l] += wa2 * a5 + wc2 * c5;  /* H25 */
                acc[15][l] += wa3 * a3;             /* H33 */
                acc[16][l] += wa3 * a5;             /* H35 */
                acc[17][l] += wc4 * c4;             /* H44 */
                acc[18][l] += wc4 * c5;             /* H45 */
                acc[19][l] += wa5 * a5 + wc5 * c5;  /* H55 */
                const double rx = ex[q], ry = ey[q];
                acc[20][l] += wa0 * rx + wc0 * ry;
                acc[21][l] += wa1 * rx + wc1 * ry;
                acc[22][l] += wa2 * rx + wc2 * ry;
                acc[23][l] += wa3 * rx;
                acc[24][l] += wc4 * ry;
                acc[25][l] += wa5 * rx + wc5 * ry;
            }
    }
    double sum[27];
    for (int k = 0; k < 26; k++) sum[k] = (acc[k][0] + acc[k][1]) + (acc[k][2] + acc[k][3]);
    for (int k = 0; k < 36; k++) H[k] = 0.0;
    H[0] = sum[0];  H[1] = sum[1];  H[2] = sum[2];  H[3] = sum[3];  H[4] = sum[4];  H[5] = sum[5];
    H[7] = sum[6];  H[8] = sum[7];  H[9] = sum[8];  H[10] = sum[9];  H[11] = sum[10];
    H[14] = sum[11];  H[15] = sum[12];  H[16] = sum[13];  H[17] = sum[14];
    H[21] = sum[15];  H[23] = sum[16];  /* H34 = 0 */
    H[28] = sum[17];  H[29] = sum[18];
    H[35] = sum[19];
    for (int a = 0; a < 6; a++) b[a] = -sum[20 + a];
}

} // namespace

int sst_pose_only(int n, const double *pts3d, const double *obs, const double *inv_sigma2, const sst_camera &c,
                  double R[9], double t[3], std::vector<uint8_t> &inlier)
{
    const double chi2_th = 5.991, delta = std::sqrt(5.991);
    inlier.assign((size_t)n, 1);
    {
        /* The rotation next to R: rows by Gram-Schmidt, the third as a cross product.  The starting pose is a product of
         * estimated rotations (the constant-velocity prediction R_k R_{k-1}^T R_k) and the steps below only ever multiply it
         * by exact exponentials, so a deviation S from orthonormality (R = Q (I + S)) is never removed and comes back about
         * three times larger in the next prediction: 40 frames after an initialisation the "rotation" was sheared by 1e-3
         * and the tracker lost every sequence there.  (Sophus / g2o carry unit quaternions; this is the matrix form's
         * equivalent.) */
        const double n0 = std::sqrt(R[0] * R[0] + R[1] * R[1] + R[2] * R[2]);
        const double r0[3] = {R[0] / n0, R[1] / n0, R[2] / n0};
        const double d01 = R[3] * r0[0] + R[4] * r0[1] + R[5] * r0[2];
        double r1[3] = {R[3] - d01 * r0[0], R[4] - d01 * r0[1], R[5] - d01 * r0[2]};
        const double n1 = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
        for (double &v : r1) v /= n1;
        const double Ro[9] = {r0[0], r0[1], r0[2], r1[0], r1[1], r1[2],
                              r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
        memcpy(R, Ro, sizeof(Ro));
    }
    if (n < 3) return -1;
    int n_in = n;
    std::vector<int> active((size_t)n);
    for (int round = 0; round < 4; round++) {
        const bool robust = round < 2;
        const double lambda = 1e-6;
        int n_active = 0;
        for (int i = 0; i < n; i++)
            if (inlier[i]) active[(size_t)n_active++] = i;
        for (int it = 0; it < 10; it++) {
            double H[36], b[6];
            pose_normal_equations(n_active, active.data(), pts3d, obs, inv_sigma2, c, R, t, robust, delta, H, b);
            for (int a = 1; a < 6; a++)
                for (int cc = 0; cc < a; cc++) H[6 * a + cc] = H[6 * cc + a];
            for (int a = 0; a < 6; a++) H[7 * a] += lambda * (1.0 + H[7 * a]);
            if (!chol6_solve(H, b)) return -2;
            double dR[9], dt[3], Rn[9];
            se3_exp(b, dR, dt);
            mat3_mul(dR, R, Rn);
            const double tn[3] = {dR[0] * t[0] + dR[1] * t[1] + dR[2] * t[2] + dt[0], dR[3] * t[0] + dR[4] * t[1] + dR[5] * t[2] + dt[1],
                                  dR[6] * t[0] + dR[7] * t[1] + dR[8] * t[2] + dt[2]};
            memcpy(R, Rn, sizeof(Rn));
            memcpy(t, tn, sizeof(tn));
            /* a round of up to ten steps ends once a step has moved nothing (g2o's optimize(10) stops on its own
             * convergence test too): Gauss-Newton is there after three or four, the other six cost 60 % of a frame's pose */
            double step = 0;
            for (int a = 0; a < 6; a++) step = std::max(step, std::fabs(b[a]));
            if (step < SST_POSE_STEP_EPS) break;
        }
        n_in = 0;
        for (int i = 0; i < n; i++) {
            const double *P = pts3d + 3 * i;
            const double x = R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0];
            const double y = R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1];
            const double z = R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2];
            double chi2 = 1e30;
            if (z > 0) {
                const double ex = obs[2 * i] - (c.fx * x / z + c.cx), ey = obs[2 * i + 1] - (c.fy * y / z + c.cy);
                chi2 = inv_sigma2[i] * (ex * ex + ey * ey);
            }
            inlier[i] = chi2 <= chi2_th;
            n_in += inlier[i];
        }
        if (n_in < 3) return -3;
    }
    return n_in;
}

bool sst_triangulate(const sst_camera &c, const double x1[2], const double x2[2], const double R1[9], const double t1[3],
                     const double R2[9], const double t2[3], double sigma2_1, double sigma2_2, double X[3])
{
    {
        /* LocalMapping::CreateNewMapPoints triangulates a pair only when its two VIEWING RAYS (the observations back-projected,
         * in world coordinates) show parallax: 0 < cos < 0.9998.  Most of a frame's untriangulated tracks are a few frames
         * old and fail here, before the linear system is built -- as the first test this took 0.15 of a tracked frame's
         * 0.2 ms of host time away; the test on the triangulated point below stays. */
        const double n1[3] = {(x1[0] - c.cx) / c.fx, (x1[1] - c.cy) / c.fy, 1.0}, n2[3] = {(x2[0] - c.cx) / c.fx, (x2[1] - c.cy) / c.fy, 1.0};
        const double q1[3] = {R1[0] * n1[0] + R1[3] * n1[1] + R1[6] * n1[2], R1[1] * n1[0] + R1[4] * n1[1] + R1[7] * n1[2],
                              R1[2] * n1[0] + R1[5] * n1[1] + R1[8] * n1[2]};
        const double q2[3] = {R2[0] * n2[0] + R2[3] * n2[1] + R2[6] * n2[2], R2[1] * n2[0] + R2[4] * n2[1] + R2[7] * n2[2],
                              R2[2] * n2[0] + R2[5] * n2[1] + R2[8] * n2[2]};
        const double cr = (q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2]) /
                          (std::sqrt(q1[0] * q1[0] + q1[1] * q1[1] + q1[2] * q1[2]) * std::sqrt(q2[0] * q2[0] + q2[1] * q2[1] + q2[2] * q2[2]));
        if (!(cr > 0 && cr < 0.9998)) return false;
    }
    double P1[12], P2[12];
    projection_matrix(c, R1, t1, P1);
    projection_matrix(c, R2, t2, P2);
    if (!triangulate_dlt(P1, P2, x1, x2, X)) return false;
    const double a[3] = {R1[0] * X[0] + R1[1] * X[1] + R1[2] * X[2] + t1[0], R1[3] * X[0] + R1[4] * X[1] + R1[5] * X[2] + t1[1],
                         R1[6] * X[0] + R1[7] * X[1] + R1[8] * X[2] + t1[2]};
    const double b[3] = {R2[0] * X[0] + R2[1] * X[1] + R2[2] * X[2] + t2[0], R2[3] * X[0] + R2[4] * X[1] + R2[5] * X[2] + t2[1],
                         R2[6] * X[0] + R2[7] * X[1] + R2[8] * X[2] + t2[2]};
    if (a[2] <= 0 || b[2] <= 0) return false;
    const double e1x = c.fx * a[0] / a[2] + c.cx - x1[0], e1y = c.fy * a[1] / a[2] + c.cy - x1[1];
    if (e1x * e1x + e1y * e1y > 5.991 * sigma2_1) return false;
    const double e2x = c.fx * b[0] / b[2] + c.cx - x2[0], e2y = c.fy * b[1] / b[2] + c.cy - x2[1];
    if (e2x * e2x + e2y * e2y > 5.991 * sigma2_2) return false;
    /* parallax between the two viewing rays (in world coordinates) */
    const double O1[3] = {-(R1[0] * t1[0] + R1[3] * t1[1] + R1[6] * t1[2]), -(R1[1] * t1[0] + R1[4] * t1[1] + R1[7] * t1[2]),
                          -(R1[2] * t1[0] + R1[5] * t1[1] + R1[8] * t1[2])};
    const double O2[3] = {-(R2[0] * t2[0] + R2[3] * t2[1] + R2[6] * t2[2]), -(R2[1] * t2[0] + R2[4] * t2[1] + R2[7] * t2[2]),
                          -(R2[2] * t2[0] + R2[5] * t2[1] + R2[8] * t2[2])};
    const double r1[3] = {X[0] - O1[0], X[1] - O1[1], X[2] - O1[2]}, r2[3] = {X[0] - O2[0], X[1] - O2[1], X[2] - O2[2]};
    const double d1 = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]), d2 = std::sqrt(r2[0] * r2[0] + r2[1] * r2[1] + r2[2] * r2[2]);
    const double cp = (r1[0] * r2[0] + r1[1] * r2[1] + r1[2] * r2[2]) / (d1 * d2);
    return cp < 0.9998;
}

/* Robust cost and, optionally, the Schur-reduced normal equations of the two-view problem. */
namespace {

struct ba_problem {
    int n;
    const double *obs1, *obs2, *w1, *w2;
    sst_camera c;
};

inline bool inv3(const double *m, double *o)
{
    const double d = det3(m);
    if (!(std::fabs(d) > 1e-300)) return false;
    const double id = 1.0 / d;
    o[0] = (m[4] * m[8] - m[5] * m[7]) * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = (m[5] * m[6] - m[3] * m[8]) * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = (m[3] * m[7] - m[4] * m[6]) * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
    return true;
}

/* one observation: residual e (2), d e / d Y (2x3, as rows a and b), Huber weight; false if behind the camera */
inline bool ba_residual(const sst_camera &c, const double Y[3], const double *obs, double w0, double e[2], double a[3], double b[3],
                        double &w, double &rho)
{
    if (Y[2] <= 0) return false;
    const double iz = 1.0 / Y[2], iz2 = iz * iz;
    e[0] = obs[0] - (c.fx * Y[0] * iz + c.cx);
    e[1] = obs[1] - (c.fy * Y[1] * iz + c.cy);
    a[0] = -c.fx * iz; a[1] = 0; a[2] = c.fx * Y[0] * iz2;
    b[0] = 0; b[1] = -c.fy * iz; b[2] = c.fy * Y[1] * iz2;
    const double e2 = w0 * (e[0] * e[0] + e[1] * e[1]), d2 = 5.991;
    if (e2 > d2) { const double se = std::sqrt(e2), dl = std::sqrt(d2); w = w0 * dl / se; rho = 2 * dl * se - d2; }
    else { w = w0; rho = e2; }
    return true;
}

double ba_cost(const ba_problem &pb, const double R[9], const double t[3], const double *X)
{
    double cost = 0;
    for (int i = 0; i < pb.n; i++) {
        const double *P = X + 3 * i;
        const double Y2[3] = {R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0], R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1],
                              R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2]};
        double e[2], a[3], b[3], w, rho;
        cost += ba_residual(pb.c, P, pb.obs1 + 2 * i, pb.w1[i], e, a, b, w, rho) ? rho : 1e6;
        cost += ba_residual(pb.c, Y2, pb.obs2 + 2 * i, pb.w2[i], e, a, b, w, rho) ? rho : 1e6;
    }
    return cost;
}

} // namespace

int sst_two_view_ba(const sst_camera &c, int n, const double *obs1, const double *obs2, const double *w1, const double *w2,
                    double R[9], double t[3], double *X, int iterations)
{
    if (n < 6) return 0;
    const ba_problem pb{n, obs1, obs2, w1, w2, c};
    std::vector<double> Hll((size_t)9 * n), Hpl((size_t)18 * n), bl((size_t)3 * n), Xn((size_t)3 * n), dl((size_t)3 * n);
    double lambda = -1, cost = ba_cost(pb, R, t, X);
    int accepted = 0;
    for (int it = 0; it < iterations; it++) {
        double Hpp[36] = {0}, bp[6] = {0};
        for (int i = 0; i < n; i++) {
            const double *P = X + 3 * i;
            double *hl = &Hll[9 * i], *hp = &Hpl[18 * i], *g = &bl[3 * i];
            for (int k = 0; k < 9; k++) hl[k] = 0;
            for (int k = 0; k < 18; k++) hp[k] = 0;
            g[0] = g[1] = g[2] = 0;
            double e[2], a[3], b[3], w, rho;
            if (ba_residual(c, P, obs1 + 2 * i, w1[i], e, a, b, w, rho)) { /* camera 1: identity pose, d Y / d X = I */
                for (int r = 0; r < 3; r++) {
                    g[r] -= w * (a[r] * e[0] + b[r] * e[1]);
                    for (int q = 0; q < 3; q++) hl[3 * r + q] += w * (a[r] * a[q] + b[r] * b[q]);
                }
            }
            const double Y[3] = {R[0] * P[0] + R[1] * P[1] + R[2] * P[2] + t[0], R[3] * P[0] + R[4] * P[1] + R[5] * P[2] + t[1],
                                 R[6] * P[0] + R[7] * P[1] + R[8] * P[2] + t[2]};
            if (ba_residual(c, Y, obs2 + 2 * i, w2[i], e, a, b, w, rho)) {
                /* d e / d X = (d e / d Y) R ; d e / d xi = (d e / d Y) [-[Y]x, I] */
                double ax[3], bx[3], aj[6], bj[6];
                for (int q = 0; q < 3; q++) {
                    ax[q] = a[0] * R[q] + a[1] * R[3 + q] + a[2] * R[6 + q];
                    bx[q] = b[0] * R[q] + b[1] * R[3 + q] + b[2] * R[6 + q];
                }
                aj[0] = a[2] * Y[1] - a[1] * Y[2]; aj[1] = a[0] * Y[2] - a[2] * Y[0]; aj[2] = a[1] * Y[0] - a[0] * Y[1];
                bj[0] = b[2] * Y[1] - b[1] * Y[2]; bj[1] = b[0] * Y[2] - b[2] * Y[0]; bj[2] = b[1] * Y[0] - b[0] * Y[1];
                for (int q = 0; q < 3; q++) { aj[3 + q] = a[q]; bj[3 + q] = b[q]; }
                for (int r = 0; r < 3; r++) {
                    g[r] -= w * (ax[r] * e[0] + bx[r] * e[1]);
                    for (int q = 0; q < 3; q++) hl[3 * r + q] += w * (ax[r] * ax[q] + bx[r] * bx[q]);
                }
                for (int r = 0; r < 6; r++) {
                    bp[r] -= w * (aj[r] * e[0] + bj[r] * e[1]);
                    for (int q = 0; q < 6; q++) Hpp[6 * r + q] += w * (aj[r] * aj[q] + bj[r] * bj[q]);
                    for (int q = 0; q < 3; q++) hp[3 * r + q] += w * (aj[r] * ax[q] + bj[r] * bx[q]);
                }
            }
        }
        if (lambda < 0) {
            double mx = 0;
            for (int r = 0; r < 6; r++) mx = std::max(mx, Hpp[7 * r]);
            for (int i = 0; i < n; i++)
                for (int r = 0; r < 3; r++) mx = std::max(mx, Hll[9 * i + 4 * r]);
            lambda = 1e-5 * mx;
        }
        /* Schur complement on the pose block */
        double S[36], g6[6];
        for (int k = 0; k < 36; k++) S[k] = Hpp[k];
        for (int r = 0; r < 6; r++) { S[7 * r] += lambda; g6[r] = bp[r]; }
        bool ok = true;
        std::vector<double> Hinv((size_t)9 * n);
        for (int i = 0; i < n && ok; i++) {
            double m[9];
            for (int k = 0; k < 9; k++) m[k] = Hll[9 * i + k];
            m[0] += lambda; m[4] += lambda; m[8] += lambda;
            ok = inv3(m, &Hinv[9 * i]);
            if (!ok) break;
            const double *hp = &Hpl[18 * i], *hi = &Hinv[9 * i], *g = &bl[3 * i];
            double T[18]; /* Hpl * Hll^-1, 6x3 */
            for (int r = 0; r < 6; r++)
                for (int q = 0; q < 3; q++) T[3 * r + q] = hp[3 * r] * hi[q] + hp[3 * r + 1] * hi[3 + q] + hp[3 * r + 2] * hi[6 + q];
            for (int r = 0; r < 6; r++) {
                g6[r] -= T[3 * r] * g[0] + T[3 * r + 1] * g[1] + T[3 * r + 2] * g[2];
                for (int q = 0; q < 6; q++) S[6 * r + q] -= T[3 * r] * hp[3 * q] + T[3 * r + 1] * hp[3 * q + 1] + T[3 * r + 2] * hp[3 * q + 2];
            }
        }
        if (ok) ok = chol6_solve(S, g6);
        double Rn[9], tn[3], new_cost = 0;
        if (ok) {
            for (int i = 0; i < n; i++) {
                const double *hp = &Hpl[18 * i], *hi = &Hinv[9 * i], *g = &bl[3 * i];
                double r3[3];
                for (int q = 0; q < 3; q++) {
                    double v = g[q];
                    for (int r = 0; r < 6; r++) v -= hp[3 * r + q] * g6[r];
                    r3[q] = v;
                }
                for (int q = 0; q < 3; q++) {
                    dl[3 * i + q] = hi[3 * q] * r3[0] + hi[3 * q + 1] * r3[1] + hi[3 * q + 2] * r3[2];
                    Xn[3 * i + q] = X[3 * i + q] + dl[3 * i + q];
                }
            }
            double dR[9], dt[3];
            se3_exp(g6, dR, dt);
            mat3_mul(dR, R, Rn);
            for (int r = 0; r < 3; r++) tn[r] = dR[3 * r] * t[0] + dR[3 * r + 1] * t[1] + dR[3 * r + 2] * t[2] + dt[r];
            new_cost = ba_cost(pb, Rn, tn, Xn.data());
        }
        if (ok && new_cost < cost) {
            memcpy(R, Rn, sizeof(Rn));
            memcpy(t, tn, sizeof(tn));
            memcpy(X, Xn.data(), sizeof(double) * 3 * (size_t)n);
            cost = new_cost;
            lambda = std::max(lambda / 3.0, 1e-12);
            accepted++;
        } else {
            lambda *= 4.0;
        }
    }
    return accepted;
}

void sst_pose_to_twc(const double R[9], const double t[3], double pos[3], double q[4])
{
    double Rwc[9];
    mat3_t(R, Rwc);
    for (int i = 0; i < 3; i++) pos[i] = -(Rwc[3 * i] * t[0] + Rwc[3 * i + 1] * t[1] + Rwc[3 * i + 2] * t[2]);
    /* Eigen::Quaternion(Matrix3): trace branch, else the largest diagonal element */
    const double tr = Rwc[0] + Rwc[4] + Rwc[8];
    double x, y, z, w;
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0);
        w = 0.5 * s;
        s = 0.5 / s;
        x = (Rwc[7] - Rwc[5]) * s; y = (Rwc[2] - Rwc[6]) * s; z = (Rwc[3] - Rwc[1]) * s;
    } else {
        int i = 0;
        if (Rwc[4] > Rwc[0]) i = 1;
        if (Rwc[8] > Rwc[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(Rwc[4 * i] - Rwc[4 * j] - Rwc[4 * k] + 1.0);
        double v[3];
        v[i] = 0.5 * s;
        s = 0.5 / s;
        w = (Rwc[3 * k + j] - Rwc[3 * j + k]) * s;
        v[j] = (Rwc[3 * j + i] + Rwc[3 * i + j]) * s;
        v[k] = (Rwc[3 * k + i] + Rwc[3 * i + k]) * s;
        x = v[0]; y = v[1]; z = v[2];
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

/* ---------------------------------------------------------------------------------------------
 * frame-to-frame state machine (ss_track.h)
 * ------------------------------------------------------------------------------------------- */
void sst_tracker::reset()
{
    state = 0;
    have_ref = false;
    have_vel = false;
    pose_hist.clear();
    ref = sst_frame();
    prev = sst_frame();
}

int sst_tracker::want_match() const
{
    if (state == 2) return SST_MATCH_PREV;
    if (state == 1 && have_ref) return SST_MATCH_REF;
    return SST_MATCH_NONE;
}

int sst_tracker::n_train() const
{
    const int w = want_match();
    return w == SST_MATCH_PREV ? prev.n : w == SST_MATCH_REF ? ref.n : 0;
}

namespace {

/* one query per train row: the smallest distance wins, ties to the lowest query index */
void unique_matches(int n, int n_train, const int32_t *idx, const uint16_t *d1, std::vector<int32_t> &m)
{
    m.assign((size_t)n, -1);
    std::vector<int32_t> owner((size_t)n_train, -1);
    for (int i = 0; i < n; i++) {
        const int j = idx[i];
        if (j < 0 || j >= n_train) continue;
        if (owner[j] < 0 || d1[i] < d1[owner[j]]) owner[j] = i;
    }
    for (int j = 0; j < n_train; j++)
        if (owner[j] >= 0) m[owner[j]] = j;
}

} // namespace

#ifdef SST_PHASE_TIMING /* profiles/tools/track_timing.cpp: where a tracked frame's host time goes */
#include <chrono>
double sst_phase_ms[8];
#define SST_PHASE(k) do { const auto now_ = std::chrono::steady_clock::now(); \
        sst_phase_ms[k] += std::chrono::duration_cast<std::chrono::duration<double, std::milli>>(now_ - phase_t0_).count(); phase_t0_ = now_; } while (0)
#define SST_PHASE_START auto phase_t0_ = std::chrono::steady_clock::now()
#else
#define SST_PHASE(k) do { } while (0)
#define SST_PHASE_START do { } while (0)
#endif

int sst_tracker::step(int n, const float *xy, const int32_t *octave, const int32_t *match_idx, const uint16_t *d1, sst_pose_out &out)
{
    SST_PHASE_START;
    sst_frame cur;
    cur.n = n;
    cur.und.resize((size_t)2 * n);
    sst_undistort(cam, n, xy, cur.und.data());
    cur.octave.assign(octave, octave + n);
    cur.has3d.assign((size_t)n, 0);
    cur.p3d.assign((size_t)3 * n, 0.0);
    out = sst_pose_out();

    /* scale^o and scale^(2 o) per octave: the values std::pow returns, looked up instead of recomputed per point */
    double s1[64], s2[64];
    for (int o = 0; o < 64; o++) {
        s1[o] = std::pow(scale_factor, (double)o);
        s2[o] = std::pow(scale_factor, 2.0 * o);
    }
    auto pw1 = [&](int o) { return o >= 0 && o < 64 ? s1[o] : std::pow(scale_factor, (double)o); };
    auto pw2 = [&](int o) { return o >= 0 && o < 64 ? s2[o] : std::pow(scale_factor, 2.0 * o); };
    const int want = want_match();
    SST_PHASE(0); /* undistort + frame arrays + tables */
    std::vector<int32_t> m;
    if (want != SST_MATCH_NONE) {
        unique_matches(n, n_train(), match_idx, d1, m);
        for (int i = 0; i < n; i++) out.n_matches += m[i] >= 0;
    }
    SST_PHASE(1); /* unique matches */

    if (want == SST_MATCH_NONE) { /* NO_IMAGES_YET, LOST, or NOT_INITIALIZED without a reference */
        state = 1;
        have_ref = n > 100;
        if (have_ref) ref = cur;
        out.state = state;
        return have_ref ? SST_KEEP_AS_REF : SST_KEEP_NONE;
    }

    if (want == SST_MATCH_REF) {
        out.state = 1;
        /* SearchForInitialization looks for the match inside a 100-px window around the reference
         * keypoint: gate the all-pairs matches the same way */
        for (int i = 0; i < n; i++)
            if (m[i] >= 0 && !(std::fabs(cur.und[2 * i] - ref.und[2 * m[i]]) < 100.0 && std::fabs(cur.und[2 * i + 1] - ref.und[2 * m[i] + 1]) < 100.0)) {
                m[i] = -1;
                out.n_matches--;
            }
        if (n <= 100 || out.n_matches < 100) { /* MonocularInitialization: drop the reference */
            have_ref = false;
            return SST_KEEP_NONE;
        }
        std::vector<double> x1, x2;
        std::vector<int> qi;
        for (int i = 0; i < n; i++)
            if (m[i] >= 0) {
                x1.push_back(ref.und[2 * m[i]]); x1.push_back(ref.und[2 * m[i] + 1]);
                x2.push_back(cur.und[2 * i]); x2.push_back(cur.und[2 * i + 1]);
                qi.push_back(i);
            }
        double R[9], t[3];
        std::vector<uint8_t> tri;
        std::vector<double> p3d;
        const int n_tri = sst_two_view(cam, (int)qi.size(), x1.data(), x2.data(), R, t, tri, p3d);
        if (n_tri <= 0) return SST_KEEP_NONE; /* keep the reference, try the next frame */
        { /* Optimizer::GlobalBundleAdjustemnt(map, 20) of CreateInitialMapMonocular, on the two views */
            std::vector<double> o1, o2, w1, w2, X;
            std::vector<size_t> ks;
            for (size_t k = 0; k < qi.size(); k++)
                if (tri[k]) {
                    o1.push_back(x1[2 * k]); o1.push_back(x1[2 * k + 1]);
                    o2.push_back(x2[2 * k]); o2.push_back(x2[2 * k + 1]);
                    w1.push_back(1.0 / pw2(ref.octave[m[qi[k]]]));
                    w2.push_back(1.0 / pw2(cur.octave[qi[k]]));
                    for (int a = 0; a < 3; a++) X.push_back(p3d[3 * k + a]);
                    ks.push_back(k);
                }
            sst_two_view_ba(cam, (int)ks.size(), o1.data(), o2.data(), w1.data(), w2.data(), R, t, X.data(), 20);
            for (size_t j = 0; j < ks.size(); j++)
                for (int a = 0; a < 3; a++) p3d[3 * ks[j] + a] = X[3 * j + a];
        }
        std::vector<double> depths;
        for (size_t k = 0; k < qi.size(); k++)
            if (tri[k]) depths.push_back(p3d[3 * k + 2]);
        std::sort(depths.begin(), depths.end());
        const double median = depths[(depths.size() - 1) / 2];
        if (!(median > 0)) { have_ref = false; return SST_KEEP_NONE; }
        const double inv = 1.0 / median;
        memcpy(cur.R, R, sizeof(R));
        for (int k = 0; k < 3; k++) cur.t[k] = t[k] * inv;
        for (size_t k = 0; k < qi.size(); k++)
            if (tri[k]) {
                cur.has3d[qi[k]] = 1;
                for (int a = 0; a < 3; a++) cur.p3d[3 * qi[k] + a] = p3d[3 * k + a] * inv;
            }
        state = 2;
        have_ref = false;
        have_vel = false;
        pose_hist.assign(cur.R, cur.R + 9);
        pose_hist.insert(pose_hist.end(), cur.t, cur.t + 3);
        prev = cur;
        out.state = 2;
        out.n_inliers = n_tri;
        out.n_map_points = n_tri;
        sst_pose_to_twc(cur.R, cur.t, out.pos, out.quat);
        return SST_KEEP_AS_PREV;
    }

    /* state OK: pose-only optimisation on the points carried by the previous frame */
    /* predicted pose: constant velocity when there is one (TrackWithMotionModel), else the last pose */
    if (have_vel) {
        mat3_mul(vel_R, prev.R, cur.R);
        for (int r = 0; r < 3; r++) cur.t[r] = vel_R[3 * r] * prev.t[0] + vel_R[3 * r + 1] * prev.t[1] + vel_R[3 * r + 2] * prev.t[2] + vel_t[r];
    } else {
        memcpy(cur.R, prev.R, sizeof(prev.R));
        memcpy(cur.t, prev.t, sizeof(prev.t));
    }
    /* SearchByProjection accepts a map point only within th * scale^octave px of its projection under
     * the predicted pose (th = 15 for monocular, doubled once if fewer than 20 points pass) */
    std::vector<double> P, obs, w;
    std::vector<int> qi;
    for (double th = 15.0; th <= 30.0; th *= 2.0) {
        P.clear(); obs.clear(); w.clear(); qi.clear();
        for (int i = 0; i < n; i++) {
            if (m[i] < 0 || !prev.has3d[m[i]]) continue;
            const double *Q = &prev.p3d[3 * m[i]];
            const double x = cur.R[0] * Q[0] + cur.R[1] * Q[1] + cur.R[2] * Q[2] + cur.t[0];
            const double y = cur.R[3] * Q[0] + cur.R[4] * Q[1] + cur.R[5] * Q[2] + cur.t[1];
            const double z = cur.R[6] * Q[0] + cur.R[7] * Q[1] + cur.R[8] * Q[2] + cur.t[2];
            if (z <= 0) continue;
            const double r = th * pw1(cur.octave[i]);
            if (!(std::fabs(cam.fx * x / z + cam.cx - cur.und[2 * i]) < r && std::fabs(cam.fy * y / z + cam.cy - cur.und[2 * i + 1]) < r)) continue;
            for (int a = 0; a < 3; a++) P.push_back(Q[a]);
            obs.push_back(cur.und[2 * i]); obs.push_back(cur.und[2 * i + 1]);
            w.push_back(1.0 / pw2(cur.octave[i]));
            qi.push_back(i);
        }
        if (qi.size() >= 20) break;
    }
    SST_PHASE(2); /* search-by-projection gate */
    std::vector<uint8_t> inl;
    const int n_in = sst_pose_only((int)qi.size(), P.data(), obs.data(), w.data(), cam, cur.R, cur.t, inl);
    SST_PHASE(3); /* pose-only optimisation */
    out.n_inliers = n_in > 0 ? n_in : 0;
    if (n_in < 30) {
        state = 4;
        out.state = 4;
        prev = sst_frame();
        have_vel = false;
        return SST_KEEP_NONE;
    }
    for (size_t k = 0; k < qi.size(); k++)
        if (inl[k]) {
            cur.has3d[qi[k]] = 1;
            for (int a = 0; a < 3; a++) cur.p3d[3 * qi[k] + a] = P[3 * k + a];
        }
    /* New points from matches that carry none yet.  A track is triangulated between its FIRST
     * observation (the anchor) and this one, so the baseline grows until the parallax test passes --
     * the role keyframes play in LocalMapping::CreateNewMapPoints. */
    if (prev.anchor.empty()) prev.anchor.assign((size_t)prev.n, -1);
    cur.anchor.assign((size_t)n, -1);
    cur.anchor_xy.assign((size_t)2 * n, 0.0);
    cur.anchor_sigma2.assign((size_t)n, 1.0);
    const int prev_pose_id = (int)(pose_hist.size() / 12) - 1; /* pose of prev, pushed when it was tracked */
    for (int i = 0; i < n; i++) {
        const int j = m[i];
        if (j < 0 || prev.has3d[j]) continue;
        int a_pose;
        double a_xy[2], a_s2;
        if (prev.anchor[j] >= 0) {
            a_pose = prev.anchor[j];
            a_xy[0] = prev.anchor_xy[2 * j]; a_xy[1] = prev.anchor_xy[2 * j + 1];
            a_s2 = prev.anchor_sigma2[j];
        } else {
            a_pose = prev_pose_id;
            a_xy[0] = prev.und[2 * j]; a_xy[1] = prev.und[2 * j + 1];
            a_s2 = pw2(prev.octave[j]);
        }
        double X[3];
        const double s2v = pw2(cur.octave[i]);
        const double *aR = &pose_hist[(size_t)12 * a_pose], *at = aR + 9;
        if (sst_triangulate(cam, a_xy, &cur.und[2 * i], aR, at, cur.R, cur.t, a_s2, s2v, X)) {
            cur.has3d[i] = 1;
            for (int a = 0; a < 3; a++) cur.p3d[3 * i + a] = X[a];
        } else {
            cur.anchor[i] = a_pose;
            cur.anchor_xy[2 * i] = a_xy[0]; cur.anchor_xy[2 * i + 1] = a_xy[1];
            cur.anchor_sigma2[i] = a_s2;
        }
    }
    SST_PHASE(4); /* new points */
    pose_hist.insert(pose_hist.end(), cur.R, cur.R + 9);
    pose_hist.insert(pose_hist.end(), cur.t, cur.t + 3);
    /* one long-lived process per camera: keep only the poses an anchored track still refers to (and the newest, which the
     * next frame reads as prev's), so the history does not grow with the length of the run */
    if (pose_hist.size() / 12 > (size_t)std::max(pose_hist_cap, 1)) {
        const size_t n_pose = pose_hist.size() / 12;
        std::vector<int32_t> remap(n_pose, -1);
        for (int i = 0; i < n; i++)
            if (cur.anchor[i] >= 0) remap[(size_t)cur.anchor[i]] = 0;
        remap[n_pose - 1] = 0;
        size_t kept = 0;
        for (size_t k = 0; k < n_pose; k++) {
            if (remap[k] < 0) continue;
            if (kept != k) std::copy(pose_hist.begin() + 12 * k, pose_hist.begin() + 12 * (k + 1), pose_hist.begin() + 12 * kept);
            remap[k] = (int32_t)kept++;
        }
        pose_hist.resize(12 * kept);
        for (int i = 0; i < n; i++)
            if (cur.anchor[i] >= 0) cur.anchor[i] = remap[(size_t)cur.anchor[i]];
    }
    for (int i = 0; i < n; i++) out.n_map_points += cur.has3d[i];
    { /* mVelocity = Tcw * Twc_last */
        double Rpt[9];
        mat3_t(prev.R, Rpt);
        mat3_mul(cur.R, Rpt, vel_R);
        for (int r = 0; r < 3; r++) vel_t[r] = cur.t[r] - (vel_R[3 * r] * prev.t[0] + vel_R[3 * r + 1] * prev.t[1] + vel_R[3 * r + 2] * prev.t[2]);
        have_vel = true;
    }
    out.state = 2;
    sst_pose_to_twc(cur.R, cur.t, out.pos, out.quat);
    prev = std::move(cur);
    SST_PHASE(5); /* history, velocity, hand-over */
    return SST_KEEP_AS_PREV;
}
