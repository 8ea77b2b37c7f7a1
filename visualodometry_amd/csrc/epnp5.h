// The 5-point EPnP minimal solve of cv::solvePnPRansac (reference call site Source/VOClass.cpp:531) on fixed-size arrays,
// callable from host and device code.  Same arithmetic, in the same order, as pnp_host::epnp (pnp_host.cpp) for n = 5:
// compiled for the host it returns that function's bits (tests/test_host_cpu.py), compiled for gfx950 it differs from it
// only through the device's hypot / acos / sin / cos (an ulp here and there).  Nothing is allocated: the two 12 x 12 Jacobi
// work matrices come from the caller (private arrays, or LDS on the GPU), the 10 x 12 system matrix and the small
// factorizations are private.
#ifndef VO_EPNP5_H
#define VO_EPNP5_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>
#include "../../include/vo_abi.h"

#define VO_HD __host__ __device__ inline

namespace epnp5 {

// Hestenes (one-sided Jacobi) iteration on the columns of an m x n matrix, held as n rows of length m in `a`, with the squared
// column norms in `w` and the identity in `v` (n x n) on entry.  On return: w = singular values (descending), row i of a =
// left singular vector i times w[i], row i of v = right singular vector i.
// S: distance in doubles between consecutive elements of a and v (1 on the host and for private arrays; on the GPU the 12 x 12
// matrices of a block's threads are interleaved in LDS, element e of thread t at e * S + t, which is free of bank conflicts)
template <int S = 1>
VO_HD void jacobi_core(double* a, int m, int n, double* v, double* w)
{
    const double eps = DBL_EPSILON * 10;
    const int sweeps = m > 30 ? m : 30;
    for (int sweep = 0; sweep < sweeps; sweep++) {
        bool rotated = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double* ai = a + i * m * S; double* aj = a + j * m * S;
                double aa = w[i], bb = w[j], p = 0;
                for (int k = 0; k < m; k++) p += ai[k * S] * aj[k * S];
                if (fabs(p) <= eps * sqrt(aa * bb)) continue;
                p *= 2;
                const double beta = aa - bb, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                aa = bb = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * ai[k * S] + s * aj[k * S];
                    const double t1 = -s * ai[k * S] + c * aj[k * S];
                    ai[k * S] = t0; aj[k * S] = t1;
                    aa += t0 * t0; bb += t1 * t1;
                }
                w[i] = aa; w[j] = bb;
                rotated = true;
                double* vi = v + i * n * S; double* vj = v + j * n * S;
                for (int k = 0; k < n; k++) {
                    const double t0 = c * vi[k * S] + s * vj[k * S];
                    const double t1 = -s * vi[k * S] + c * vj[k * S];
                    vi[k * S] = t0; vj[k * S] = t1;
                }
            }
        if (!rotated) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const double t = a[(i * m + k) * S]; sd += t * t; }
        w[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (w[j] < w[k]) j = k;
        if (i != j) {
            double t = w[i]; w[i] = w[j]; w[j] = t;
            for (int k = 0; k < m; k++) { t = a[(i * m + k) * S]; a[(i * m + k) * S] = a[(j * m + k) * S]; a[(j * m + k) * S] = t; }
            for (int k = 0; k < n; k++) { t = v[(i * n + k) * S]; v[(i * n + k) * S] = v[(j * n + k) * S]; v[(j * n + k) * S] = t; }
        }
    }
}

// A (m x n row-major, m >= n, both at most MM / NN) = U diag(W) V^T; U m x n, V n x n, row-major
template <int MM, int NN>
VO_HD void svd(const double* A, int m, int n, double* W, double* U, double* V)
{
    double a[NN * MM], v[NN * NN], w[NN];
    for (int i = 0; i < n * n; i++) v[i] = 0.0;
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const double t = A[k * n + i]; a[i * m + k] = t; sd += t * t; }
        w[i] = sd;
        v[i * n + i] = 1.0;
    }
    jacobi_core(a, m, n, v, w);
    for (int i = 0; i < n; i++) {
        W[i] = w[i];
        const double s = w[i] > DBL_MIN ? 1.0 / w[i] : 0.0;
        if (U) for (int k = 0; k < m; k++) U[k * n + i] = a[i * m + k] * s;
        if (V) for (int k = 0; k < n; k++) V[k * n + i] = v[i * n + k];
    }
}

// minimum-norm least squares through the SVD; singular values <= 2 eps sum(w) are dropped
template <int MM, int NN>
VO_HD void lstsq_factored(const double* W, const double* U, const double* V, int m, int n, const double* b, double* x)
{
    double cut = 0;
    for (int i = 0; i < n; i++) cut += W[i];
    cut *= DBL_EPSILON * 2;
    for (int k = 0; k < n; k++) x[k] = 0;
    for (int i = 0; i < n; i++) {
        if (fabs(W[i]) <= cut) continue;
        double s = 0;
        for (int k = 0; k < m; k++) s += U[k * n + i] * b[k];
        s /= W[i];
        for (int k = 0; k < n; k++) x[k] += s * V[k * n + i];
    }
}
template <int MM, int NN>
VO_HD void lstsq(const double* A, int m, int n, const double* b, double* x)
{
    double W[NN], U[MM * NN], V[NN * NN];
    svd<MM, NN>(A, m, n, W, U, V);
    lstsq_factored<MM, NN>(W, U, V, m, n, b, x);
}

VO_HD void uvt3(const double* U, const double* V, double* R)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
}

VO_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
VO_HD double sqdist3(const double* a, const double* b)
{ return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]); }

// rotation vector of the rotation nearest to Rin (cv::Rodrigues, matrix -> vector)
VO_HD void rodrigues_from_matrix(const double Rin[9], double r[3])
{
    double W[3], U[9], V[9], R[9];
    svd<3, 3>(Rin, 3, 3, W, U, V);
    uvt3(U, V, R);
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    const double s = sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) x = y = z = 0;
        else {
            double t = (R[0] + 1) * 0.5; x = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; y = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; z = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(x) < fabs(y) && fabs(x) < fabs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= sqrt(x * x + y * y + z * z);
            x *= theta; y *= theta; z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth; y *= vth; z *= vth;
    }
    r[0] = x; r[1] = y; r[2] = z;
}

// cv::Rodrigues, vector -> matrix
VO_HD void rodrigues_to_matrix(const double r[3], double R[9])
{
    double kx = r[0], ky = r[1], kz = r[2];
    const double theta = sqrt(kx * kx + ky * ky + kz * kz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    kx *= itheta; ky *= itheta; kz *= itheta;
    const double kkt[9] = {kx * kx, kx * ky, kx * kz, kx * ky, ky * ky, ky * kz, kx * kz, ky * kz, kz * kz};
    const double kcross[9] = {0, -kz, ky, kz, 0, -kx, -ky, kx, 0};
    for (int k = 0; k < 9; k++) R[k] = c * ((k % 4 == 0) ? 1. : 0.) + c1 * kkt[k] + s * kcross[k];
}

struct Problem {
    double fu, fv, uc, vc;
    double world[15], pix[10], alpha[20], cam[15];
    double cw[4][3], cc[4][3];
};

VO_HD void pick_control_points(Problem& e)
{
    const int n = 5;
    e.cw[0][0] = e.cw[0][1] = e.cw[0][2] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) e.cw[0][j] += e.world[3 * i + j];
    for (int j = 0; j < 3; j++) e.cw[0][j] /= n;
    double centred[15];
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) centred[3 * i + j] = e.world[3 * i + j] - e.cw[0][j];
    double cov[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += centred[3 * i + r] * centred[3 * i + c];
        cov[r * 3 + c] = s;
    }
    double ev[3], U[9], V[9];
    svd<3, 3>(cov, 3, 3, ev, U, V);
    for (int i = 1; i < 4; i++) {
        const double k = sqrt(ev[i - 1] / n);
        for (int j = 0; j < 3; j++) e.cw[i][j] = e.cw[0][j] + k * U[j * 3 + (i - 1)];
    }
}

VO_HD void barycentric_coordinates(Problem& e)
{
    double basis[9], inv[9];
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) basis[3 * i + j - 1] = e.cw[j][i] - e.cw[0][i];
    {   // inverse column by column through the least-squares solver (one factorisation serves the three right-hand sides)
        double W[3], U[9], V[9];
        svd<3, 3>(basis, 3, 3, W, U, V);
        for (int c = 0; c < 3; c++) {
            double ec[3] = {0, 0, 0}, x[3];
            ec[c] = 1;
            lstsq_factored<3, 3>(W, U, V, 3, 3, ec, x);
            for (int r = 0; r < 3; r++) inv[r * 3 + c] = x[r];
        }
    }
    for (int i = 0; i < 5; i++) {
        const double* p = &e.world[3 * i]; double* a = &e.alpha[4 * i];
        for (int j = 0; j < 3; j++)
            a[1 + j] = inv[3 * j] * (p[0] - e.cw[0][0]) + inv[3 * j + 1] * (p[1] - e.cw[0][1]) + inv[3 * j + 2] * (p[2] - e.cw[0][2]);
        a[0] = 1.0f - a[1] - a[2] - a[3];
    }
}

// 6 x 10 from the four right singular vectors of the smallest singular values (rows 11..8 of vt)
template <int S = 1>
VO_HD void build_L(const double* vt, double* L)
{
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        const double* v = vt + 12 * (11 - i) * S;
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            for (int k = 0; k < 3; k++) dv[i][j][k] = v[(3 * a + k) * S] - v[(3 * b + k) * S];
            if (++b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; i++) {
        double* row = L + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

// the three closed-form starts for beta; columns of L are [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]
VO_HD void beta_start(int which, const double* L, const double* rho, double* beta)
{
    if (which == 1) {
        double l[24], x[4];
        for (int i = 0; i < 6; i++) { l[i * 4] = L[i * 10]; l[i * 4 + 1] = L[i * 10 + 1]; l[i * 4 + 2] = L[i * 10 + 3]; l[i * 4 + 3] = L[i * 10 + 6]; }
        lstsq<6, 5>(l, 6, 4, rho, x);
        if (x[0] < 0) { beta[0] = sqrt(-x[0]); beta[1] = -x[1] / beta[0]; beta[2] = -x[2] / beta[0]; beta[3] = -x[3] / beta[0]; }
        else { beta[0] = sqrt(x[0]); beta[1] = x[1] / beta[0]; beta[2] = x[2] / beta[0]; beta[3] = x[3] / beta[0]; }
    } else if (which == 2) {
        double l[18], x[3];
        for (int i = 0; i < 6; i++) { l[i * 3] = L[i * 10]; l[i * 3 + 1] = L[i * 10 + 1]; l[i * 3 + 2] = L[i * 10 + 2]; }
        lstsq<6, 5>(l, 6, 3, rho, x);
        if (x[0] < 0) { beta[0] = sqrt(-x[0]); beta[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { beta[0] = sqrt(x[0]); beta[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) beta[0] = -beta[0];
        beta[2] = 0.0; beta[3] = 0.0;
    } else {
        double l[30], x[5];
        for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) l[i * 5 + k] = L[i * 10 + k];
        lstsq<6, 5>(l, 6, 5, rho, x);
        if (x[0] < 0) { beta[0] = sqrt(-x[0]); beta[1] = (x[2] < 0) ? sqrt(-x[2]) : 0.0; }
        else { beta[0] = sqrt(x[0]); beta[1] = (x[2] > 0) ? sqrt(x[2]) : 0.0; }
        if (x[1] < 0) beta[0] = -beta[0];
        beta[2] = x[3] / beta[0];
        beta[3] = 0.0;
    }
}

// Householder least squares for the 6 x 4 Gauss-Newton step (column scale taken over rows k..nr-2, as published)
VO_HD void householder_solve(double* A, double* b, double* X, int nr, int nc)
{
    double A1[8], A2[8];
    for (int k = 0; k < nc; k++) {
        double eta = fabs(A[k * nc + k]);
        for (int i = k + 1; i < nr; i++) { const double elt = fabs(A[(i - 1) * nc + k]); if (eta < elt) eta = elt; }
        if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) { A[i * nc + k] *= inv_eta; sum2 += A[i * nc + k] * A[i * nc + k]; }
        double sigma = sqrt(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
            double sum = 0;
            for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
            const double tau = sum / A1[k];
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    for (int j = 0; j < nc; j++) {
        double tau = 0;
        for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double sum = 0;
        for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

VO_HD void refine_beta(const double* L, const double* rho, double beta[4])
{
    for (int it = 0; it < 5; it++) {
        double A[24], b[6], x[4] = {0, 0, 0, 0};
        for (int i = 0; i < 6; i++) {
            const double* l = L + i * 10; double* a = A + i * 4;
            a[0] = 2 * l[0] * beta[0] + l[1] * beta[1] + l[3] * beta[2] + l[6] * beta[3];
            a[1] = l[1] * beta[0] + 2 * l[2] * beta[1] + l[4] * beta[2] + l[7] * beta[3];
            a[2] = l[3] * beta[0] + l[4] * beta[1] + 2 * l[5] * beta[2] + l[8] * beta[3];
            a[3] = l[6] * beta[0] + l[7] * beta[1] + l[8] * beta[2] + 2 * l[9] * beta[3];
            b[i] = rho[i] - (l[0] * beta[0] * beta[0] + l[1] * beta[0] * beta[1] + l[2] * beta[1] * beta[1] +
                             l[3] * beta[0] * beta[2] + l[4] * beta[1] * beta[2] + l[5] * beta[2] * beta[2] +
                             l[6] * beta[0] * beta[3] + l[7] * beta[1] * beta[3] + l[8] * beta[2] * beta[3] +
                             l[9] * beta[3] * beta[3]);
        }
        householder_solve(A, b, x, 6, 4);
        for (int i = 0; i < 4; i++) beta[i] += x[i];
    }
}

// absolute orientation (Horn / SVD) between camera-frame and world points, then the mean reprojection error
template <int S = 1>
VO_HD double pose_from_beta(Problem& e, const double* vt, const double* beta, double R[3][3], double t[3])
{
    const int n = 5;
    for (int i = 0; i < 4; i++) e.cc[i][0] = e.cc[i][1] = e.cc[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
        const double* v = vt + 12 * (11 - i) * S;
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) e.cc[j][k] += beta[i] * v[(3 * j + k) * S];
    }
    for (int i = 0; i < n; i++) {
        const double* a = &e.alpha[4 * i]; double* pc = &e.cam[3 * i];
        for (int j = 0; j < 3; j++) pc[j] = a[0] * e.cc[0][j] + a[1] * e.cc[1][j] + a[2] * e.cc[2][j] + a[3] * e.cc[3][j];
    }
    if (e.cam[2] < 0.0) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) e.cc[i][j] = -e.cc[i][j];
        for (int i = 0; i < 3 * n; i++) e.cam[i] = -e.cam[i];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) { pc0[j] += e.cam[3 * i + j]; pw0[j] += e.world[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const double* pc = &e.cam[3 * i]; const double* pw = &e.world[3 * i];
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    double d[3], U[9], V[9];
    svd<3, 3>(abt, 3, 3, d, U, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        R[i][j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot3(R[0], pw0); t[1] = pc0[1] - dot3(R[1], pw0); t[2] = pc0[2] - dot3(R[2], pw0);
    double acc = 0.0;
    for (int i = 0; i < n; i++) {
        const double* pw = &e.world[3 * i];
        const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
        const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        const double ue = e.uc + e.fu * Xc * inv_Zc, ve = e.vc + e.fv * Yc * inv_Zc;
        const double u = e.pix[2 * i], v = e.pix[2 * i + 1];
        acc += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return acc / n;
}

// solvePnP(SOLVEPNP_EPNP) on exactly five float correspondences; K row-major 3x3.  a and vt: 144 doubles of work space each,
// element e at [e * S] (the 12 x 12 Jacobi matrices; private arrays with S = 1, or a block's interleaved LDS arrays)
template <int S>
VO_HD void solve_ws(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3], double* a, double* vt)
{
    const int n = 5;
    Problem e;
    e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5];
    const double ifu = 1. / K[0], ifv = 1. / K[4];
    for (int i = 0; i < n; i++) {
        e.world[3 * i] = obj[i].x; e.world[3 * i + 1] = obj[i].y; e.world[3 * i + 2] = obj[i].z;
        // the image points pass through an undistort step that stores normalised coordinates as float
        const float xn = (float)(((double)img[i].x - K[2]) * ifu), yn = (float)(((double)img[i].y - K[5]) * ifv);
        e.pix[2 * i] = xn * e.fu + e.uc;
        e.pix[2 * i + 1] = yn * e.fv + e.vc;
    }
    pick_control_points(e);
    barycentric_coordinates(e);
    double M[120];
    for (int p = 0; p < n; p++) {
        const double* as = &e.alpha[4 * p];
        const double u = e.pix[2 * p], v = e.pix[2 * p + 1];
        double* r1 = &M[(2 * p) * 12]; double* r2 = r1 + 12;
        for (int i = 0; i < 4; i++) {
            r1[3 * i] = as[i] * e.fu; r1[3 * i + 1] = 0.0; r1[3 * i + 2] = as[i] * (e.uc - u);
            r2[3 * i] = 0.0; r2[3 * i + 1] = as[i] * e.fv; r2[3 * i + 2] = as[i] * (e.vc - v);
        }
    }
    // M^T M is symmetric: its rows are at once the columns the Jacobi iteration works on, and after the iteration the rows
    // of v, sorted by singular value, are the rows of V^T
    double w[12];
    for (int r = 0; r < 12; r++) {
        for (int c = 0; c < 12; c++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += M[i * 12 + c] * M[i * 12 + r];
            a[(r * 12 + c) * S] = s;
            vt[(r * 12 + c) * S] = 0.0;
        }
        vt[(r * 12 + r) * S] = 1.0;
    }
    for (int r = 0; r < 12; r++) {
        double sd = 0;
        for (int k = 0; k < 12; k++) { const double t = a[(r * 12 + k) * S]; sd += t * t; }
        w[r] = sd;
    }
    jacobi_core<S>(a, 12, 12, vt, w);
    double L[60], rho[6];
    build_L<S>(vt, L);
    rho[0] = sqdist3(e.cw[0], e.cw[1]); rho[1] = sqdist3(e.cw[0], e.cw[2]); rho[2] = sqdist3(e.cw[0], e.cw[3]);
    rho[3] = sqdist3(e.cw[1], e.cw[2]); rho[4] = sqdist3(e.cw[1], e.cw[3]); rho[5] = sqdist3(e.cw[2], e.cw[3]);
    double beta[4][4], rep[4], Rs[4][3][3], ts[4][3];
    for (int k = 1; k <= 3; k++) {
        beta_start(k, L, rho, beta[k]);
        refine_beta(L, rho, beta[k]);
        rep[k] = pose_from_beta<S>(e, vt, beta[k], Rs[k], ts[k]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    double R[9];
    for (int i = 0; i < 3; i++) { tvec[i] = ts[N][i]; for (int j = 0; j < 3; j++) R[i * 3 + j] = Rs[N][i][j]; }
    rodrigues_from_matrix(R, rvec);
}

// the same with private work matrices
VO_HD void solve(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3])
{
    double a[144], vt[144];
    solve_ws<1>(obj, img, K, rvec, tvec, a, vt);
}

} // namespace epnp5
#endif
