// Host-side minimal solvers and RANSAC bookkeeping (see pnp_host.h).  f64, compiled without FP contraction.
#include "pnp_host.h"
#include "small_linalg.h"
#include "epnp5.h"
#include <math.h>
#include <string.h>
#include <array>
#include <vector>
#include <thread>
#include <atomic>
#include <functional>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <stdio.h>
#include <stdlib.h>

namespace pnp_host {

// ------------------------------------------------------------------ sampling
void draw_subset(Rng& rng, int count, int model_points, int* idx)
{
    int filled = 0;
    while (filled < model_points) {
        const int cand = rng.uniform(0, count);
        bool seen = false;
        for (int j = 0; j < filled && !seen; j++) seen = idx[j] == cand;
        if (!seen) idx[filled++] = cand;
    }
}

int update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = std::min(std::max(p, 0.), 1.);
    ep = std::min(std::max(ep, 0.), 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

// ------------------------------------------------------------------ rotations
void rodrigues_to_matrix(const double r[3], double R[9], double* J)
{
    double kx = r[0], ky = r[1], kz = r[2];
    const double theta = std::sqrt(kx * kx + ky * ky + kz * kz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
        if (J) { memset(J, 0, 27 * sizeof(double)); J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
        return;
    }
    const double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    kx *= itheta; ky *= itheta; kz *= itheta;
    const double kkt[9] = {kx * kx, kx * ky, kx * kz, kx * ky, ky * ky, ky * kz, kx * kz, ky * kz, kz * kz};
    const double kcross[9] = {0, -kz, ky, kz, 0, -kx, -ky, kx, 0};
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * eye[k] + c1 * kkt[k] + s * kcross[k];
    if (!J) return;
    const double dkkt[27] = {kx + kx, ky, kz, ky, 0, 0, kz, 0, 0,
                             0, kx, 0, kx, ky + ky, kz, 0, kz, 0,
                             0, 0, kx, 0, 0, ky, kx, ky, kz + kz};
    const double dkcross[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                0, 0, 1, 0, 0, 0, -1, 0, 0,
                                0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) {
        const double ki = i == 0 ? kx : i == 1 ? ky : kz;
        const double a0 = -s * ki, a1 = (s - 2 * c1 * itheta) * ki, a2 = c1 * itheta;
        const double a3 = (c - s * itheta) * ki, a4 = s * itheta;
        for (int k = 0; k < 9; k++)
            J[i * 9 + k] = a0 * eye[k] + a1 * kkt[k] + a2 * dkkt[i * 9 + k] + a3 * kcross[k] + a4 * dkcross[i * 9 + k];
    }
}

void rodrigues_from_matrix(const double Rin[9], double r[3])
{
    sla::SvdWorkspace ws;
    double W[3], U[9], V[9], R[9];
    sla::svd(Rin, 3, 3, W, U, V, ws);
    sla::uvt3(U, V, R); // nearest rotation first
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    const double s = std::sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) x = y = z = 0;
        else {
            double t = (R[0] + 1) * 0.5; x = std::sqrt(std::max(t, 0.));
            t = (R[4] + 1) * 0.5; y = std::sqrt(std::max(t, 0.)) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; z = std::sqrt(std::max(t, 0.)) * (R[2] < 0 ? -1. : 1.);
            if (std::abs(x) < std::abs(y) && std::abs(x) < std::abs(z) && (R[5] > 0) != (y * z > 0)) z = -z;
            theta /= std::sqrt(x * x + y * y + z * z);
            x *= theta; y *= theta; z *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        x *= vth; y *= vth; z *= vth;
    }
    r[0] = x; r[1] = y; r[2] = z;
}

bool inv4x4(const double* A, double* Ainv) { return sla::inv_lu(A, 4, Ainv); }

// ------------------------------------------------------------------ EPnP (Lepetit, Moreno-Noguer, Fua)
namespace {

inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double sqdist3(const double* a, const double* b)
{ return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]); }

struct EpnpProblem {
    double fu, fv, uc, vc;
    int n;
    std::vector<double> world, pix, alpha, cam; // 3n, 2n, 4n, 3n
    double cw[4][3], cc[4][3];                  // control points, world and camera frame
    sla::SvdWorkspace ws;
};

void pick_control_points(EpnpProblem& e)
{
    const int n = e.n;
    e.cw[0][0] = e.cw[0][1] = e.cw[0][2] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) e.cw[0][j] += e.world[3 * i + j];
    for (int j = 0; j < 3; j++) e.cw[0][j] /= n;
    std::vector<double> centred((size_t)n * 3);
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) centred[3 * i + j] = e.world[3 * i + j] - e.cw[0][j];
    double cov[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += centred[3 * i + r] * centred[3 * i + c];
        cov[r * 3 + c] = s;
    }
    double ev[3], U[9], V[9];
    sla::svd(cov, 3, 3, ev, U, V, e.ws);
    for (int i = 1; i < 4; i++) {
        const double k = std::sqrt(ev[i - 1] / n);
        for (int j = 0; j < 3; j++) e.cw[i][j] = e.cw[0][j] + k * U[j * 3 + (i - 1)];
    }
}

void barycentric_coordinates(EpnpProblem& e)
{
    double basis[9], inv[9];
    for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) basis[3 * i + j - 1] = e.cw[j][i] - e.cw[0][i];
    sla::inv3(basis, inv, e.ws);
    for (int i = 0; i < e.n; i++) {
        const double* p = &e.world[3 * i]; double* a = &e.alpha[4 * i];
        for (int j = 0; j < 3; j++)
            a[1 + j] = inv[3 * j] * (p[0] - e.cw[0][0]) + inv[3 * j + 1] * (p[1] - e.cw[0][1]) + inv[3 * j + 2] * (p[2] - e.cw[0][2]);
        a[0] = 1.0f - a[1] - a[2] - a[3];
    }
}

void build_L(const double* vt, double* L) // 6 x 10 from the four smallest right singular vectors (rows 11..8 of vt)
{
    const double* v[4] = {vt + 12 * 11, vt + 12 * 10, vt + 12 * 9, vt + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
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
void beta_start(EpnpProblem& e, int which, const double* L, const double* rho, double* beta)
{
    if (which == 1) { // [B11 B12 B13 B14]
        double l[24], x[4];
        for (int i = 0; i < 6; i++) { l[i * 4] = L[i * 10]; l[i * 4 + 1] = L[i * 10 + 1]; l[i * 4 + 2] = L[i * 10 + 3]; l[i * 4 + 3] = L[i * 10 + 6]; }
        sla::lstsq(l, 6, 4, rho, x, e.ws);
        if (x[0] < 0) { beta[0] = std::sqrt(-x[0]); beta[1] = -x[1] / beta[0]; beta[2] = -x[2] / beta[0]; beta[3] = -x[3] / beta[0]; }
        else { beta[0] = std::sqrt(x[0]); beta[1] = x[1] / beta[0]; beta[2] = x[2] / beta[0]; beta[3] = x[3] / beta[0]; }
    } else if (which == 2) { // [B11 B12 B22]
        double l[18], x[3];
        for (int i = 0; i < 6; i++) { l[i * 3] = L[i * 10]; l[i * 3 + 1] = L[i * 10 + 1]; l[i * 3 + 2] = L[i * 10 + 2]; }
        sla::lstsq(l, 6, 3, rho, x, e.ws);
        if (x[0] < 0) { beta[0] = std::sqrt(-x[0]); beta[1] = (x[2] < 0) ? std::sqrt(-x[2]) : 0.0; }
        else { beta[0] = std::sqrt(x[0]); beta[1] = (x[2] > 0) ? std::sqrt(x[2]) : 0.0; }
        if (x[1] < 0) beta[0] = -beta[0];
        beta[2] = 0.0; beta[3] = 0.0;
    } else { // [B11 B12 B22 B13 B23]
        double l[30], x[5];
        for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) l[i * 5 + k] = L[i * 10 + k];
        sla::lstsq(l, 6, 5, rho, x, e.ws);
        if (x[0] < 0) { beta[0] = std::sqrt(-x[0]); beta[1] = (x[2] < 0) ? std::sqrt(-x[2]) : 0.0; }
        else { beta[0] = std::sqrt(x[0]); beta[1] = (x[2] > 0) ? std::sqrt(x[2]) : 0.0; }
        if (x[1] < 0) beta[0] = -beta[0];
        beta[2] = x[3] / beta[0];
        beta[3] = 0.0;
    }
}

// Householder least squares for the 6x4 Gauss-Newton step (column scale taken over rows k..nr-2, as published)
void householder_solve(double* A, double* b, double* X, int nr, int nc)
{
    double A1[8], A2[8];
    for (int k = 0; k < nc; k++) {
        double eta = std::abs(A[k * nc + k]);
        for (int i = k + 1; i < nr; i++) { const double elt = std::abs(A[(i - 1) * nc + k]); if (eta < elt) eta = elt; }
        if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) { A[i * nc + k] *= inv_eta; sum2 += A[i * nc + k] * A[i * nc + k]; }
        double sigma = std::sqrt(sum2);
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

void refine_beta(const double* L, const double* rho, double beta[4])
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

// absolute orientation (Horn / SVD) between camera-frame and world points, then mean reprojection error
double pose_from_beta(EpnpProblem& e, const double* vt, const double* beta, double R[3][3], double t[3])
{
    const int n = e.n;
    for (int i = 0; i < 4; i++) e.cc[i][0] = e.cc[i][1] = e.cc[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
        const double* v = vt + 12 * (11 - i);
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) e.cc[j][k] += beta[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
        const double* a = &e.alpha[4 * i]; double* pc = &e.cam[3 * i];
        for (int j = 0; j < 3; j++) pc[j] = a[0] * e.cc[0][j] + a[1] * e.cc[1][j] + a[2] * e.cc[2][j] + a[3] * e.cc[3][j];
    }
    if (e.cam[2] < 0.0) { // points must lie in front of the camera
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
    sla::svd(abt, 3, 3, d, U, V, e.ws);
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
        acc += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return acc / n;
}

} // namespace

void epnp(const vo_point3f* obj, const vo_point2f* img, int n, const double K[9], double rvec[3], double tvec[3])
{
    EpnpProblem e;
    e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5]; e.n = n;
    e.world.resize((size_t)3 * n); e.pix.resize((size_t)2 * n); e.alpha.resize((size_t)4 * n); e.cam.resize((size_t)3 * n);
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
    std::vector<double> M((size_t)2 * n * 12);
    for (int p = 0; p < n; p++) {
        const double* as = &e.alpha[4 * p];
        const double u = e.pix[2 * p], v = e.pix[2 * p + 1];
        double* r1 = &M[(size_t)(2 * p) * 12]; double* r2 = r1 + 12;
        for (int i = 0; i < 4; i++) {
            r1[3 * i] = as[i] * e.fu; r1[3 * i + 1] = 0.0; r1[3 * i + 2] = as[i] * (e.uc - u);
            r2[3 * i] = 0.0; r2[3 * i + 1] = as[i] * e.fv; r2[3 * i + 2] = as[i] * (e.vc - v);
        }
    }
    double mtm[144];
    for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) {
        double s = 0;
        for (int i = 0; i < 2 * n; i++) s += M[(size_t)i * 12 + r] * M[(size_t)i * 12 + c];
        mtm[r * 12 + c] = s;
    }
    double sv[12], U[144], V[144], vt[144];
    sla::svd(mtm, 12, 12, sv, U, V, e.ws);
    for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) vt[r * 12 + c] = V[c * 12 + r];
    double L[60], rho[6];
    build_L(vt, L);
    rho[0] = sqdist3(e.cw[0], e.cw[1]); rho[1] = sqdist3(e.cw[0], e.cw[2]); rho[2] = sqdist3(e.cw[0], e.cw[3]);
    rho[3] = sqdist3(e.cw[1], e.cw[2]); rho[4] = sqdist3(e.cw[1], e.cw[3]); rho[5] = sqdist3(e.cw[2], e.cw[3]);
    double beta[4][4], rep[4], Rs[4][3][3], ts[4][3];
    for (int k = 1; k <= 3; k++) {
        beta_start(e, k, L, rho, beta[k]);
        refine_beta(L, rho, beta[k]);
        rep[k] = pose_from_beta(e, vt, beta[k], Rs[k], ts[k]);
    }
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    double R[9];
    for (int i = 0; i < 3; i++) { tvec[i] = ts[N][i]; for (int j = 0; j < 3; j++) R[i * 3 + j] = Rs[N][i][j]; }
    rodrigues_from_matrix(R, rvec);
}

void epnp5_fixed(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3])
{
    epnp5::solve(obj, img, K, rvec, tvec);
}

// ------------------------------------------------------------------ iterative PnP: DLT start + Levenberg-Marquardt
namespace {

struct Projection { double u, v, x, y, zinv; };
inline Projection project(const double R[9], const double t[3], const double K[9], double X, double Y, double Z)
{
    Projection p;
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    p.u = x * K[0] + K[2]; p.v = y * K[4] + K[5];
    p.x = x; p.y = y; p.zinv = z;
    return p;
}

// Plane-to-image homography by the normalised direct linear transform: both point sets are centred and scaled by their
// mean absolute deviation per axis, the 9-vector is the eigenvector of the smallest eigenvalue of L^T L, and the result
// is mapped back and scaled to h[8] = 1.  (The original follows this with a few Levenberg-Marquardt steps on the
// homography; here the 6-parameter pose refinement that follows makes that polish immaterial.)
bool homography_dlt(const double* M /* n x 2 */, const double* m /* n x 2 */, int n, double h[9], sla::SvdWorkspace& ws)
{
    double cM[2] = {0, 0}, cm[2] = {0, 0}, sM[2] = {0, 0}, sm[2] = {0, 0};
    for (int i = 0; i < n; i++) { cM[0] += M[2 * i]; cM[1] += M[2 * i + 1]; cm[0] += m[2 * i]; cm[1] += m[2 * i + 1]; }
    for (int k = 0; k < 2; k++) { cM[k] /= n; cm[k] /= n; }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 2; k++) { sM[k] += std::abs(M[2 * i + k] - cM[k]); sm[k] += std::abs(m[2 * i + k] - cm[k]); }
    for (int k = 0; k < 2; k++) if (std::abs(sM[k]) < DBL_EPSILON || std::abs(sm[k]) < DBL_EPSILON) return false;
    for (int k = 0; k < 2; k++) { sM[k] = n / sM[k]; sm[k] = n / sm[k]; }
    double LtL[81];
    for (int k = 0; k < 81; k++) LtL[k] = 0;
    for (int i = 0; i < n; i++) {
        const double X = (M[2 * i] - cM[0]) * sM[0], Y = (M[2 * i + 1] - cM[1]) * sM[1];
        const double x = (m[2 * i] - cm[0]) * sm[0], y = (m[2 * i + 1] - cm[1]) * sm[1];
        const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x}, Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
        for (int r = 0; r < 9; r++) for (int c = r; c < 9; c++) LtL[r * 9 + c] += Lx[r] * Lx[c] + Ly[r] * Ly[c];
    }
    for (int r = 0; r < 9; r++) for (int c = 0; c < r; c++) LtL[r * 9 + c] = LtL[c * 9 + r];
    double W[9], U[81], V[81], H0[9];
    sla::svd(LtL, 9, 9, W, U, V, ws);
    for (int k = 0; k < 9; k++) H0[k] = V[k * 9 + 8];
    // H = inv(T_image) * H0 * T_object with T(p) = (p - c) * s
    const double invT[9] = {1. / sm[0], 0, cm[0], 0, 1. / sm[1], cm[1], 0, 0, 1};
    const double T2[9] = {sM[0], 0, -cM[0] * sM[0], 0, sM[1], -cM[1] * sM[1], 0, 0, 1};
    double tmp[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tmp[r * 3 + c] = invT[r * 3] * H0[c] + invT[r * 3 + 1] * H0[3 + c] + invT[r * 3 + 2] * H0[6 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) h[r * 3 + c] = tmp[r * 3] * T2[c] + tmp[r * 3 + 1] * T2[3 + c] + tmp[r * 3 + 2] * T2[6 + c];
    if (std::abs(h[8]) < DBL_MIN) return false;
    const double s = 1. / h[8];
    for (int k = 0; k < 9; k++) h[k] *= s;
    return true;
}

} // namespace

// ------------------------------------------------------------------ P3P (Gao, Hou, Tang, Chang 2003) on exactly four points
namespace {

int real_roots_cubic(double a, double b, double c, double d, double x[3])
{
    if (a == 0) {
        if (b == 0) { if (c == 0) return 0; x[0] = -d / c; return 1; }
        const double disc = c * c - 4 * b * d;
        if (disc < 0) return 0;
        const double inv2b = 1. / (2 * b);
        if (disc == 0) { x[0] = x[1] = -c * inv2b; return 1; }
        const double sq = std::sqrt(disc);
        x[0] = (-c + sq) * inv2b; x[1] = (-c - sq) * inv2b;
        return 2;
    }
    const double ia = 1. / a, ba = ia * b, ba2 = ba * ba, ca = ia * c, da = ia * d;
    const double Q = (3 * ca - ba2) / 9, R = (9 * ba * ca - 27 * da - 2 * ba * ba2) / 54;
    const double Q3 = Q * Q * Q, D = Q3 + R * R, shift = (1. / 3.) * ba;
    if (Q == 0) {
        if (R == 0) { x[0] = x[1] = x[2] = -shift; return 3; }
        x[0] = std::pow(2 * R, 1 / 3.0) - shift;
        return 1;
    }
    if (D <= 0) { // three real roots, trigonometric form
        const double theta = std::acos(R / std::sqrt(-Q3)), sq = std::sqrt(-Q);
        x[0] = 2 * sq * std::cos(theta / 3.0) - shift;
        x[1] = 2 * sq * std::cos((theta + 2 * M_PI) / 3.0) - shift;
        x[2] = 2 * sq * std::cos((theta + 4 * M_PI) / 3.0) - shift;
        return 3;
    }
    const double AD = std::pow(std::abs(R) + std::sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    const double BD = AD == 0 ? 0 : -Q / AD;
    x[0] = AD + BD - shift;
    return 1;
}

// real roots of a x^4 + b x^3 + c x^2 + d x + e by Ferrari's resolvent cubic
int real_roots_quartic(double a, double b, double c, double d, double e, double x[4])
{
    if (a == 0) return real_roots_cubic(b, c, d, e, x);
    const double ia = 1. / a;
    b *= ia; c *= ia; d *= ia; e *= ia;
    const double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r[3];
    if (real_roots_cubic(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, r) == 0) return 0;
    const double R2 = 0.25 * b2 - c + r[0];
    if (R2 < 0) return 0;
    const double R = std::sqrt(R2);
    double D2, E2;
    if (R < 10E-12) {
        const double t = r[0] * r[0] - 4 * e;
        if (t < 0) D2 = E2 = -1;
        else { const double st = std::sqrt(t); D2 = 0.75 * b2 - 2 * c + 2 * st; E2 = D2 - 4 * st; }
    } else {
        const double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * (1. / R) * (4 * bc - 8 * d - b3);
        D2 = u + v; E2 = u - v;
    }
    const double b_4 = 0.25 * b, R_2 = 0.5 * R;
    int nr = 0;
    if (D2 >= 0) { const double Dv = std::sqrt(D2); x[0] = R_2 + 0.5 * Dv - b_4; x[1] = x[0] - Dv; nr = 2; }
    if (E2 >= 0) { const double Ev = std::sqrt(E2); x[nr] = -R_2 + 0.5 * Ev - b_4; x[nr + 1] = x[nr] - Ev; nr += 2; }
    return nr;
}

// |PA|, |PB|, |PC| from the side lengths dist = (|BC|, |AC|, |AB|) and the cosines of (BPC, APC, APB); up to 4 solutions
int p3p_lengths(double lengths[4][3], const double dist[3], const double cosv[3])
{
    const double p = cosv[0] * 2, q = cosv[1] * 2, r = cosv[2] * 2;
    const double inv_d22 = 1. / (dist[2] * dist[2]);
    const double a = inv_d22 * (dist[0] * dist[0]), b = inv_d22 * (dist[1] * dist[1]);
    const double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r, pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0; // the projection centre is coplanar with the three points
    const double ab = a * b, a_2 = 2 * a, a_4 = 4 * a;
    const double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    const double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    const double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    const double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    const double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    const double temp = p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr;
    const double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double roots[4];
    const int n = real_roots_quartic(A, B, C, D, E, roots);
    int ns = 0;
    const double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q, inv_b0 = 1. / b0;
    for (int i = 0; i < n; i++) {
        const double x = roots[i];
        if (x <= 0) continue;
        const double x2 = x * x;
        const double b1 = ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        const double y = inv_b0 * b1, v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        const double Z = dist[2] / std::sqrt(v);
        lengths[ns][0] = x * Z; lengths[ns][1] = y * Z; lengths[ns][2] = Z;
        ns++;
    }
    return ns;
}

// rigid motion that takes three world points onto three camera-frame points (least squares through the SVD of the
// cross-covariance; exact for three non-collinear points)
bool align3(const double cam[3][3], const double wld[3][3], double R[9], double t[3], sla::SvdWorkspace& ws)
{
    double cc[3] = {0, 0, 0}, cw[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cc[j] += cam[i][j] / 3; cw[j] += wld[i][j] / 3; }
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; i++) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) S[r * 3 + c] += (cam[i][r] - cc[r]) * (wld[i][c] - cw[c]);
    double w[3], U[9], V[9];
    sla::svd(S, 3, 3, w, U, V, ws);
    if (!(w[1] > 1e-12 * w[0])) return false; // collinear points
    sla::uvt3(U, V, R);
    if (sla::det3(R) < 0) { // reflection: flip the direction of least variance
        for (int r = 0; r < 3; r++) U[r * 3 + 2] = -U[r * 3 + 2];
        sla::uvt3(U, V, R);
    }
    for (int r = 0; r < 3; r++) t[r] = cc[r] - (R[r * 3] * cw[0] + R[r * 3 + 1] * cw[1] + R[r * 3 + 2] * cw[2]);
    return true;
}

} // namespace

// solvePnP(SOLVEPNP_P3P) as cv::solvePnPRansac uses it when it is handed exactly four correspondences: the first three
// points give up to four poses, the one with the smallest reprojection error over the four points is returned.
int p3p4(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3])
{
    sla::SvdWorkspace ws;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double mu[4], mv[4], mk[3], pix[4][2];
    for (int i = 0; i < 4; i++) {
        // the image points pass through an undistort-and-reproject step (identity here) that stores them as float
        pix[i][0] = (double)(float)((((double)img[i].x - cx) / fx) * fx + cx);
        pix[i][1] = (double)(float)((((double)img[i].y - cy) / fy) * fy + cy);
        mu[i] = (pix[i][0] - cx) / fx; mv[i] = (pix[i][1] - cy) / fy;
    }
    for (int i = 0; i < 3; i++) { mk[i] = 1. / std::sqrt(mu[i] * mu[i] + mv[i] * mv[i] + 1); mu[i] *= mk[i]; mv[i] *= mk[i]; }
    double X[4][3];
    for (int i = 0; i < 4; i++) { X[i][0] = obj[i].x; X[i][1] = obj[i].y; X[i][2] = obj[i].z; }
    auto d3 = [&](int a, int b) { return std::sqrt((X[a][0] - X[b][0]) * (X[a][0] - X[b][0]) + (X[a][1] - X[b][1]) * (X[a][1] - X[b][1]) + (X[a][2] - X[b][2]) * (X[a][2] - X[b][2])); };
    const double dist[3] = {d3(1, 2), d3(0, 2), d3(0, 1)};
    const double cosv[3] = {mu[1] * mu[2] + mv[1] * mv[2] + mk[1] * mk[2], mu[0] * mu[2] + mv[0] * mv[2] + mk[0] * mk[2],
                            mu[0] * mu[1] + mv[0] * mv[1] + mk[0] * mk[1]};
    double lengths[4][3];
    const int ns = p3p_lengths(lengths, dist, cosv);
    double best = DBL_MAX, bestR[9], bestT[3];
    bool found = false;
    for (int s = 0; s < ns; s++) {
        double cam[3][3], wld[3][3], R[9], t[3];
        for (int i = 0; i < 3; i++) {
            cam[i][0] = lengths[s][i] * mu[i]; cam[i][1] = lengths[s][i] * mv[i]; cam[i][2] = lengths[s][i] * mk[i];
            for (int j = 0; j < 3; j++) wld[i][j] = X[i][j];
        }
        if (!align3(cam, wld, R, t, ws)) continue;
        double err = 0;
        for (int i = 0; i < 4; i++) {
            const double xc = R[0] * X[i][0] + R[1] * X[i][1] + R[2] * X[i][2] + t[0];
            const double yc = R[3] * X[i][0] + R[4] * X[i][1] + R[5] * X[i][2] + t[1];
            const double zc = R[6] * X[i][0] + R[7] * X[i][1] + R[8] * X[i][2] + t[2];
            const double du = fx * xc / zc + cx - pix[i][0], dv = fy * yc / zc + cy - pix[i][1];
            err += du * du + dv * dv;
        }
        if (!(err < best)) continue; // the first of equally good solutions stays, as a stable sort would keep it
        best = err; found = true;
        memcpy(bestR, R, sizeof(R)); memcpy(bestT, t, sizeof(t));
    }
    if (!found) return 0;
    rodrigues_from_matrix(bestR, rvec);
    for (int i = 0; i < 3; i++) tvec[i] = bestT[i];
    return 1;
}

int iterative(const double* obj, const double* img, int n, const double K[9], double rvec[3], double tvec[3])
{
    const int max_iter = 20;
    sla::SvdWorkspace ws;
    std::vector<double> mn((size_t)2 * n);
    {
        const double ifx = 1. / K[0], ify = 1. / K[4];
        for (int i = 0; i < n; i++) { mn[2 * i] = (img[2 * i] - K[2]) * ifx; mn[2 * i + 1] = (img[2 * i + 1] - K[5]) * ify; }
    }
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) Mc[j] += obj[3 * i + j];
    for (int j = 0; j < 3; j++) Mc[j] /= n;
    double MM[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += (obj[3 * i + r] - Mc[r]) * (obj[3 * i + c] - Mc[c]);
        MM[r * 3 + c] = s;
    }
    double W[3], U3[9], V3[9];
    sla::svd(MM, 3, 3, W, U3, V3, ws);
    double param[6];
    if (W[2] / W[1] < 1e-3) {
        // All object points lie in one plane: the projection is a homography of the plane's own 2-D coordinates.
        // Rotate the object into the frame of its principal axes (third axis = plane normal), estimate the homography
        // to the normalised image points, and read [r1 r2 t] off its columns.
        double Rtf[9], Ttf[3]; // rows of Rtf = principal axes
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Rtf[r * 3 + c] = V3[c * 3 + r];
        if (Rtf[2] * Rtf[2] + Rtf[5] * Rtf[5] < 1e-10) { for (int k = 0; k < 9; k++) Rtf[k] = (k % 4 == 0) ? 1 : 0; }
        if (sla::det3(Rtf) < 0) for (int k = 0; k < 9; k++) Rtf[k] = -Rtf[k];
        for (int r = 0; r < 3; r++) Ttf[r] = -(Rtf[r * 3] * Mc[0] + Rtf[r * 3 + 1] * Mc[1] + Rtf[r * 3 + 2] * Mc[2]);
        std::vector<double> Mxy((size_t)2 * n);
        for (int i = 0; i < n; i++) {
            const double* M = &obj[3 * i];
            Mxy[2 * i] = Rtf[0] * M[0] + Rtf[1] * M[1] + Rtf[2] * M[2] + Ttf[0];
            Mxy[2 * i + 1] = Rtf[3] * M[0] + Rtf[4] * M[1] + Rtf[5] * M[2] + Ttf[1];
        }
        double h[9];
        bool ok = n >= 4 && homography_dlt(Mxy.data(), mn.data(), n, h, ws);
        for (int k = 0; k < 9 && ok; k++) ok = std::isfinite(h[k]);
        double R[9], t[3] = {0, 0, 0};
        if (ok) {
            const double n1 = std::sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), n2 = std::sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
            const double s1 = 1. / std::max(n1, DBL_EPSILON), s2 = 1. / std::max(n2, DBL_EPSILON), s3 = 2. / std::max(n1 + n2, DBL_EPSILON);
            double c1[3] = {h[0] * s1, h[3] * s1, h[6] * s1}, c2[3] = {h[1] * s2, h[4] * s2, h[7] * s2};
            double tt[3] = {h[2] * s3, h[5] * s3, h[8] * s3};
            double c3[3] = {c1[1] * c2[2] - c1[2] * c2[1], c1[2] * c2[0] - c1[0] * c2[2], c1[0] * c2[1] - c1[1] * c2[0]};
            double Hm[9] = {c1[0], c2[0], c3[0], c1[1], c2[1], c3[1], c1[2], c2[2], c3[2]}, rv[3], Rh[9];
            rodrigues_from_matrix(Hm, rv);          // nearest rotation to [c1 c2 c3] ...
            rodrigues_to_matrix(rv, Rh, nullptr);   // ... as a proper rotation matrix
            for (int r = 0; r < 3; r++) t[r] = Rh[r * 3] * Ttf[0] + Rh[r * 3 + 1] * Ttf[1] + Rh[r * 3 + 2] * Ttf[2] + tt[r];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
                R[r * 3 + c] = Rh[r * 3] * Rtf[c] + Rh[r * 3 + 1] * Rtf[3 + c] + Rh[r * 3 + 2] * Rtf[6 + c];
        } else {
            for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1 : 0;
        }
        rodrigues_from_matrix(R, param);
        for (int k = 0; k < 3; k++) param[3 + k] = t[k];
    } else {
        if (n < 6) return -3;
        std::vector<double> L((size_t)2 * n * 12);
        for (int i = 0; i < n; i++) {
            double* l = &L[(size_t)i * 24];
            const double x = -mn[2 * i], y = -mn[2 * i + 1];
            const double* M = &obj[3 * i];
            l[0] = l[16] = M[0]; l[1] = l[17] = M[1]; l[2] = l[18] = M[2]; l[3] = l[19] = 1.;
            l[4] = l[5] = l[6] = l[7] = 0.; l[12] = l[13] = l[14] = l[15] = 0.;
            l[8] = x * M[0]; l[9] = x * M[1]; l[10] = x * M[2]; l[11] = x;
            l[20] = y * M[0]; l[21] = y * M[1]; l[22] = y * M[2]; l[23] = y;
        }
        // L^T L in one pass over the rows: every entry is still the sum of its products in row order, so the result
        // is the same as 144 separate dot products (and symmetric entries are the same products)
        double LL[144];
        for (int k = 0; k < 144; k++) LL[k] = 0;
        for (int i = 0; i < 2 * n; i++) {
            const double* l = &L[(size_t)i * 12];
            for (int r = 0; r < 12; r++) for (int c = r; c < 12; c++) LL[r * 12 + c] += l[r] * l[c];
        }
        for (int r = 0; r < 12; r++) for (int c = 0; c < r; c++) LL[r * 12 + c] = LL[c * 12 + r];
        double LW[12], LU[144], LV[144];
        sla::svd(LL, 12, 12, LW, LU, LV, ws);
        double P[12];
        for (int k = 0; k < 12; k++) P[k] = LV[k * 12 + 11];
        double RR[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
        double tt[3] = {P[3], P[7], P[11]};
        if (sla::det3(RR) < 0) { for (int k = 0; k < 9; k++) RR[k] = -RR[k]; for (int k = 0; k < 3; k++) tt[k] = -tt[k]; }
        double sc = 0;
        for (int k = 0; k < 9; k++) sc += RR[k] * RR[k];
        sc = std::sqrt(sc);
        if (!(std::abs(sc) > DBL_EPSILON)) return -4;
        double w3[3], u3[9], v3[9], R[9];
        sla::svd(RR, 3, 3, w3, u3, v3, ws);
        sla::uvt3(u3, v3, R);
        double nr = 0;
        for (int k = 0; k < 9; k++) nr += R[k] * R[k];
        nr = std::sqrt(nr);
        for (int k = 0; k < 3; k++) param[3 + k] = tt[k] * (nr / sc);
        rodrigues_from_matrix(R, param);
    }
    const int m = 2 * n;
    std::vector<double> J((size_t)m * 6), err(m);
    double prev[6], JtJ[36], JtErr[6];
    double prevErrNorm = DBL_MAX;
    int lambdaLg10 = -3, iters = 0;
    auto evaluate = [&](const double* prm, bool want_J) {
        double R[9], dRdr[27];
        rodrigues_to_matrix(prm, R, want_J ? dRdr : nullptr);
        for (int i = 0; i < n; i++) {
            const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
            const Projection p = project(R, prm + 3, K, X, Y, Z);
            err[2 * i] = p.u - img[2 * i]; err[2 * i + 1] = p.v - img[2 * i + 1];
            if (!want_J) continue;
            double* jx = &J[(size_t)(2 * i) * 6]; double* jy = jx + 6;
            const double x = p.x, y = p.y, z = p.zinv;
            const double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            for (int j = 0; j < 3; j++) { jx[3 + j] = K[0] * dxdt[j]; jy[3 + j] = K[4] * dydt[j]; }
            const double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11], X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            const double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14], X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            const double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17], X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                const double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                const double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                jx[j] = K[0] * dxdr; jy[j] = K[4] * dydr;
            }
        }
    };
    auto err_norm = [&]() { double s = 0; for (int i = 0; i < m; i++) s += err[i] * err[i]; return std::sqrt(s); };
    auto lm_step = [&]() {
        const double lambda = std::exp(lambdaLg10 * std::log(10.));
        double A[36], x[6];
        for (int k = 0; k < 36; k++) A[k] = JtJ[k];
        for (int r = 0; r < 6; r++) A[r * 6 + r] *= 1. + lambda;
        sla::lstsq(A, 6, 6, JtErr, x, ws);
        for (int i = 0; i < 6; i++) param[i] = prev[i] - x[i];
    };
    for (;;) {
        evaluate(param, true);
        // one pass over J; each accumulator adds its products in row order, like a dot product per entry
        for (int k = 0; k < 36; k++) JtJ[k] = 0;
        for (int k = 0; k < 6; k++) JtErr[k] = 0;
        for (int i = 0; i < m; i++) {
            const double* j = &J[(size_t)i * 6];
            const double e = err[i];
            for (int r = 0; r < 6; r++) {
                for (int c = r; c < 6; c++) JtJ[r * 6 + c] += j[r] * j[c];
                JtErr[r] += j[r] * e;
            }
        }
        for (int r = 0; r < 6; r++) for (int c = 0; c < r; c++) JtJ[r * 6 + c] = JtJ[c * 6 + r];
        memcpy(prev, param, sizeof(param));
        if (iters == 0) prevErrNorm = err_norm();
        lm_step();
        bool done = false;
        for (;;) {
            evaluate(param, false);
            const double errNorm = err_norm();
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) { lm_step(); continue; }
            lambdaLg10 = std::max(lambdaLg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { dn += (param[i] - prev[i]) * (param[i] - prev[i]); pn += prev[i] * prev[i]; }
            if (++iters >= max_iter || std::sqrt(dn) / std::sqrt(pn) < FLT_EPSILON) done = true;
            prevErrNorm = errNorm;
            break;
        }
        if (done) break;
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    return 1;
}

// ------------------------------------------------------------------ batched RANSAC over GPU-scored hypotheses
namespace {

// Worker threads that live as long as their context: a batch needs two or three short parallel sections, and starting
// threads for each of them costs more than the sections themselves.
class Pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    const std::function<void(int)>* fn = nullptr;
    int count = 0, active = 0;
    unsigned long long gen = 0;
    std::atomic<int> next{0};
    bool stop = false;

    void loop()
    {
        unsigned long long seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            const std::function<void(int)>* f = fn;
            const int cnt = count;
            lk.unlock();
            for (int i = next.fetch_add(1); i < cnt; i = next.fetch_add(1)) (*f)(i);
            lk.lock();
            if (--active == 0) cv_done.notify_one();
        }
    }

public:
    explicit Pool(int workers) { for (int t = 0; t < workers; t++) th.emplace_back([this] { loop(); }); }
    ~Pool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : th) t.join();
    }
    void run(int cnt, const std::function<void(int)>& f)
    {
        if (th.empty() || cnt <= 1) { for (int i = 0; i < cnt; i++) f(i); return; }
        { std::lock_guard<std::mutex> lk(mu); fn = &f; count = cnt; next.store(0); active = (int)th.size(); gen++; }
        cv_work.notify_all();
        for (int i = next.fetch_add(1); i < cnt; i = next.fetch_add(1)) f(i); // the caller works too
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return active == 0; });
    }
};

// One pool per process, shared by all contexts (a node runs one process per GPU, several contexts per process: a pool per
// context would multiply the host threads by the batches in flight).  A section that finds the pool busy with another
// thread's section runs on its own thread instead of waiting.
std::mutex g_pool_mu;
Pool* g_pool = nullptr;
int g_pool_workers = 0;

// The worker threads must not outlive the library's code: when the library is unloaded (or the process exits) the pool is
// joined and deleted.
__attribute__((destructor)) void destroy_pool()
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    delete g_pool;
    g_pool = nullptr;
    g_pool_workers = 0;
}

void parallel_for(vo_ctx* c, int count, int threads, const std::function<void(int)>& fn)
{
    if (threads <= 1 || count <= 1) { for (int i = 0; i < count; i++) fn(i); return; }
    std::unique_lock<std::mutex> lk(g_pool_mu, std::try_to_lock);
    if (!lk.owns_lock()) {
        // another thread's section holds the pool: this one runs on the calling thread alone.  Counted, so that a caller
        // who drives several contexts from several threads can see it in vo_get_timing (stage "pnp_host_serial").
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < count; i++) fn(i);
        if (c && c->timing) {
            c->stage_ms[VO_ST_PNP_SERIAL] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            c->stage_calls[VO_ST_PNP_SERIAL]++;
        }
        return;
    }
    if (!g_pool || g_pool_workers < threads - 1) {
        delete g_pool;
        g_pool_workers = threads - 1;
        g_pool = new Pool(g_pool_workers);
    }
    g_pool->run(count, fn);
}

struct FrameRansac {
    int k = 0;                 // correspondences
    const vo_point3f* obj = nullptr;
    const vo_point2f* img = nullptr;
    Rng rng;
    int niters = 0, iter = 0, max_good = 0;
    bool active = false, failed = false, direct = false;
    double best_r[3] = {0, 0, 0}, best_t[3] = {0, 0, 0};
    std::vector<uint8_t> best_mask;
    int nh = 0;                // hypotheses in the current round
    std::vector<std::array<double, 3>> hr, ht; // the round's rvec / tvec, `ransac_chunk` of them
    std::vector<std::array<int, 5>> idx;       // ... and the subsets they were solved from
    int best_h = -1;           // hypothesis of this round that became the best model
};

} // namespace

static int estimate_motion_rounds(vo_ctx* c, double* Rt_out, vo_frame_info* info)
{
    const int n = c->n, chunk = c->p.ransac_chunk;
    const size_t F = c->p.max_features;
    const auto host0 = std::chrono::steady_clock::now();
    const int model_points = 5;
    int threads = c->p.host_threads > 0 ? c->p.host_threads : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    const double K[9] = {c->P0[0], c->P0[1], c->P0[2], c->P0[4], c->P0[5], c->P0[6], c->P0[8], c->P0[9], c->P0[10]};
    const float thresh2 = (float)((double)c->p.ransac_reproj * (double)c->p.ransac_reproj);

    // correspondences of every frame to the host (pinned staging): two strided copies, [n][kmax] points each
    int kmax = 0;
    for (int b = 0; b < n; b++) kmax = std::max(kmax, c->h_nobj[b]);
    const size_t obj_bytes = (size_t)n * kmax * sizeof(vo_point3f), img_off = (obj_bytes + 15) & ~(size_t)15;
    if (img_off + (size_t)n * kmax * sizeof(vo_point2f) > c->h_stage_bytes) return vo_fail(c, VO_ERR_CAPACITY, "staging buffer too small");
    if (kmax > 0) {
        VO_HIP(c, hipMemcpy2DAsync(c->h_stage, (size_t)kmax * sizeof(vo_point3f), c->obj, F * sizeof(vo_point3f), (size_t)kmax * sizeof(vo_point3f), n,
                                   hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipMemcpy2DAsync((char*)c->h_stage + img_off, (size_t)kmax * sizeof(vo_point2f), c->obj_img, F * sizeof(vo_point2f),
                                   (size_t)kmax * sizeof(vo_point2f), n, hipMemcpyDeviceToHost, c->stream));
    }
    VO_HIP(c, hipStreamSynchronize(c->stream));
    // own copies: the staging buffer is reused for hypotheses and masks below
    std::vector<std::vector<vo_point3f>> objs(n);
    std::vector<std::vector<vo_point2f>> imgs(n);
    std::vector<FrameRansac> fr(n);
    int status = VO_OK;
    for (int b = 0; b < n; b++) {
        const int k = c->h_nobj[b];
        const vo_point3f* so = (const vo_point3f*)c->h_stage + (size_t)b * kmax;
        const vo_point2f* si = (const vo_point2f*)((const char*)c->h_stage + img_off) + (size_t)b * kmax;
        objs[b].assign(so, so + k);
        imgs[b].assign(si, si + k);
        FrameRansac& f = fr[b];
        f.k = k; f.obj = objs[b].data(); f.img = imgs[b].data();
        f.niters = std::max(c->p.ransac_iters, 1);
        f.hr.resize(chunk); f.ht.resize(chunk); f.idx.resize(chunk);
        if (k < 4) { f.failed = true; status = VO_ERR_NOT_ENOUGH_POINTS; } // the original asserts at least 4 points
        else if (k == 4 || k == model_points) f.direct = true;          // no sampling: one minimal solve on all points
        else f.active = true;
    }

    double* h_hyp = (double*)c->h_stage;                         // [n][chunk][12]
    const bool on_device = c->p.pnp_device != 0;
    double* h_rt6 = (double*)c->h_stage;                         // GPU solves: [n][chunk][6] rvec, tvec (the subsets sit behind them)
    int* h_cnt = (int*)((char*)c->h_stage + (size_t)n * chunk * 12 * sizeof(double)); // [n][chunk]
    // inlier masks of a round's hypotheses, [n][chunk][kmax] bytes: fetched together with the counts (one strided copy, one
    // wait per round) rather than per frame after the counts have been looked at
    const size_t mask_off = ((size_t)n * chunk * (12 * sizeof(double) + sizeof(int)) + 255) & ~(size_t)255;
    const bool masks_fit = mask_off + (size_t)n * chunk * kmax <= c->h_stage_bytes;
    uint8_t* h_mask = (uint8_t*)c->h_stage + mask_off;
    for (;;) {
        bool any = false;
        for (int b = 0; b < n; b++) any = any || (fr[b].active && fr[b].iter < fr[b].niters);
        if (!any) break;
        // minimal solver on the host: the next `chunk` subsets of every frame still iterating.  The subsets come off each
        // frame's generator in order (a cheap serial pass); the solves are independent and are spread over the worker threads
        // one (frame, hypothesis) at a time, so that a batch of one frame uses the threads as well as a batch of forty.
        for (int b = 0; b < n; b++) {
            FrameRansac& f = fr[b];
            f.nh = 0; f.best_h = -1;
            if (!f.active || f.iter >= f.niters) continue;
            f.nh = std::min(chunk, f.niters - f.iter);
            for (int h = 0; h < f.nh; h++) draw_subset(f.rng, f.k, model_points, f.idx[h].data());
        }
        if (on_device) {
            // minimal solves on the GPU: the subsets go up, the kernel reads the correspondences where the triangulation left
            // them, and only the rotation / translation vectors come back (a new best model keeps them)
            int* h_idx = (int*)((char*)c->h_stage + (size_t)n * chunk * 6 * sizeof(double)); // [n][chunk][5], then [n] counts
            int* h_nh = h_idx + (size_t)n * chunk * 5;
            for (int b = 0; b < n; b++) {
                const FrameRansac& f = fr[b];
                h_nh[b] = f.nh;
                for (int h = 0; h < chunk; h++) // slots beyond a frame's hypotheses repeat its first one (scored and ignored)
                    memcpy(h_idx + ((size_t)b * chunk + h) * 5, f.idx[h < f.nh ? h : 0].data(), 5 * sizeof(int));
            }
            VO_HIP(c, hipMemcpyAsync(c->hyp_idx, h_idx, (size_t)n * chunk * 5 * sizeof(int), hipMemcpyHostToDevice, c->stream));
            VO_HIP(c, hipMemcpyAsync(c->hyp_idx + (size_t)c->p.max_batch * chunk * 5, h_nh, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
            const int rs = launch_pnp_solve(c, n);
            if (rs) return rs;
            VO_HIP(c, hipMemcpyAsync(h_rt6, c->hyp_rt, (size_t)n * chunk * 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        } else {
        parallel_for(c, n * chunk, threads, [&](int t) {
            const int b = t / chunk, h = t - b * chunk;
            FrameRansac& f = fr[b];
            double* dst = h_hyp + ((size_t)b * chunk + h) * 12;
            if (h >= f.nh) return;
            vo_point3f so[5]; vo_point2f si[5];
            for (int j = 0; j < 5; j++) { so[j] = f.obj[f.idx[h][j]]; si[j] = f.img[f.idx[h][j]]; }
            epnp(so, si, 5, K, f.hr[h].data(), f.ht[h].data());
            double R[9];
            rodrigues_to_matrix(f.hr[h].data(), R, nullptr);
            memcpy(dst, R, sizeof(R)); dst[9] = f.ht[h][0]; dst[10] = f.ht[h][1]; dst[11] = f.ht[h][2];
        });
        // slots beyond a frame's hypotheses repeat its first one (scored and ignored; an idle frame's slots are never read)
        for (int b = 0; b < n; b++)
            for (int h = fr[b].nh; h < chunk && fr[b].nh > 0; h++)
                memcpy(h_hyp + ((size_t)b * chunk + h) * 12, h_hyp + ((size_t)b * chunk) * 12, 12 * sizeof(double));
        // GPU: score every hypothesis of every frame against all of that frame's correspondences
        VO_HIP(c, hipMemcpyAsync(c->hyp, h_hyp, (size_t)n * chunk * 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        // masks of every hypothesis only when they can ride to the host with the counts; otherwise the scoring pass writes none
        // (a byte per hypothesis and point is most of its memory traffic) and a second pass writes the new best models' masks
        int rc = launch_pnp_score(c, 0, n, chunk, thresh2, masks_fit);
        if (rc) return rc;
        VO_HIP(c, hipMemcpyAsync(h_cnt, c->hyp_count, (size_t)n * chunk * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        if (masks_fit && kmax > 0)
            VO_HIP(c, hipMemcpy2DAsync(h_mask, (size_t)kmax, c->hyp_mask, F, (size_t)kmax, (size_t)n * chunk, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
        // replay the sequential accept rule; hypotheses past the shrinking iteration budget are discarded
        for (int b = 0; b < n; b++) {
            FrameRansac& f = fr[b];
            for (int h = 0; h < f.nh; h++) {
                if (f.iter >= f.niters) break;
                const int good = h_cnt[(size_t)b * chunk + h];
                if (good > std::max(f.max_good, model_points - 1)) {
                    f.max_good = good; f.best_h = h;
                    if (on_device) {
                        const double* rt = h_rt6 + ((size_t)b * chunk + h) * 6;
                        memcpy(f.best_r, rt, sizeof(f.best_r)); memcpy(f.best_t, rt + 3, sizeof(f.best_t));
                    } else { memcpy(f.best_r, f.hr[h].data(), sizeof(f.best_r)); memcpy(f.best_t, f.ht[h].data(), sizeof(f.best_t)); }
                    f.niters = update_num_iters(c->p.ransac_conf, (double)(f.k - good) / f.k, model_points, f.niters);
                }
                f.iter++;
            }
        }
        // inlier mask of each frame's new best model
        bool fetch = false;
        for (int b = 0; b < n; b++) fetch = fetch || (fr[b].best_h >= 0 && !masks_fit);
        if (fetch) {
            int* h_sel = h_cnt; // the counts have been consumed
            for (int b = 0; b < n; b++) h_sel[b] = fr[b].best_h;
            VO_HIP(c, hipMemcpyAsync(c->hyp_sel, h_sel, (size_t)n * sizeof(int), hipMemcpyHostToDevice, c->stream));
            rc = launch_pnp_masks(c, n, thresh2);
            if (rc) return rc;
        }
        for (int b = 0; b < n; b++) {
            FrameRansac& f = fr[b];
            if (f.best_h < 0) continue;
            f.best_mask.resize(f.k);
            if (masks_fit) { memcpy(f.best_mask.data(), h_mask + ((size_t)b * chunk + f.best_h) * kmax, (size_t)f.k); continue; }
            VO_HIP(c, hipMemcpyAsync(f.best_mask.data(), c->hyp_mask + ((size_t)b * chunk + f.best_h) * F, (size_t)f.k, hipMemcpyDeviceToHost, c->stream));
        }
        if (fetch) VO_HIP(c, hipStreamSynchronize(c->stream));
    }

    // final solve on the inliers, Rodrigues, Rt
    const auto host1 = std::chrono::steady_clock::now();
    std::vector<vo_frame_info> li(n);
    parallel_for(c, n, threads, [&](int b) {
        FrameRansac& f = fr[b];
        vo_frame_info& o = li[b];
        memset(&o, 0, sizeof(o));
        o.n_features = c->h_nfeat[b]; o.n_matched = c->h_nmatch[b]; o.n_object_points = f.k;
        double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
        if (f.direct) {
            if (f.k == 4) o.ok = p3p4(f.obj, f.img, K, rvec, tvec);
            else { epnp(f.obj, f.img, f.k, K, rvec, tvec); o.ok = 1; }
            if (!o.ok) rvec[0] = rvec[1] = rvec[2] = tvec[0] = tvec[1] = tvec[2] = 0;
            o.best_inliers = o.ok ? f.k : 0;
        } else if (!f.failed && f.max_good > 0) {
            o.iters_run = f.iter; o.best_inliers = f.max_good;
            std::vector<double> po, pi;
            po.reserve((size_t)f.max_good * 3); pi.reserve((size_t)f.max_good * 2);
            for (int i = 0; i < f.k; i++) if (f.best_mask[i]) {
                po.push_back(f.obj[i].x); po.push_back(f.obj[i].y); po.push_back(f.obj[i].z);
                pi.push_back(f.img[i].x); pi.push_back(f.img[i].y);
            }
            const int n1 = (int)(pi.size() / 2);
            const int res = iterative(po.data(), pi.data(), n1, K, rvec, tvec);
            if (res > 0) { o.ok = 1; o.final_from_refine = 1; }
            else {
                memcpy(rvec, f.best_r, sizeof(rvec)); memcpy(tvec, f.best_t, sizeof(tvec));
                o.ok = (res == -3 && n1 == 5) ? 1 : 0;
            }
        } else o.iters_run = f.iter;
        double R[9];
        rodrigues_to_matrix(rvec, R, nullptr);
        double* Rt = Rt_out + (size_t)b * 16;
        for (int r = 0; r < 3; r++) { for (int cc = 0; cc < 3; cc++) Rt[r * 4 + cc] = R[r * 3 + cc]; Rt[r * 4 + 3] = tvec[r]; }
        Rt[12] = Rt[13] = Rt[14] = 0; Rt[15] = 1;
    });
    if (info) memcpy(info, li.data(), (size_t)n * sizeof(vo_frame_info));
    if (getenv("VO_PNP_TRACE")) {
        const auto host2 = std::chrono::steady_clock::now();
        int it = 0; for (int b = 0; b < n; b++) it = std::max(it, fr[b].iter);
        fprintf(stderr, "[pnp] rounds part %.3f ms (max iters %d), refine part %.3f ms, threads %d\n",
                std::chrono::duration<double, std::milli>(host1 - host0).count(), it,
                std::chrono::duration<double, std::milli>(host2 - host1).count(), threads);
    }
    if (c->timing) {
        c->stage_ms[VO_ST_PNP_HOST] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host0).count();
        c->stage_calls[VO_ST_PNP_HOST]++;
    }
    if (status != VO_OK) return vo_fail(c, status, "a frame of the batch has fewer than 4 3D-2D correspondences");
    return VO_OK;
}

// The RANSAC rounds are short host <-> GPU round trips (a few small copies and one scoring kernel each).  They run on
// the context's high-priority stream so that they do not queue behind the bulk kernels other contexts have in flight.
int estimate_motion_batch(vo_ctx* c, double* Rt_out, vo_frame_info* info)
{
    hipStream_t bulk = c->stream;
    VO_HIP(c, hipStreamSynchronize(bulk));
#ifndef VO_NO_HI_STREAM
    if (c->stream_hi) c->stream = c->stream_hi;
#endif
    const int rc = estimate_motion_rounds(c, Rt_out, info);
    c->stream = bulk;
    return rc;
}

} // namespace pnp_host
