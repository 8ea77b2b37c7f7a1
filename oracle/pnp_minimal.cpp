/* CPU ORACLE (test infrastructure only) — the two branches of cv::solvePnPRansac (reference Source/VOClass.cpp:531) that
 * only degenerate inputs reach:
 *   - exactly four correspondences: solvePnP(SOLVEPNP_P3P) (OpenCV 4.5.4 calib3d solvepnp.cpp, p3p.cpp: three points give
 *     up to four poses, the fourth point picks one);
 *   - a planar inlier set in the final SOLVEPNP_ITERATIVE solve: start from the plane's homography
 *     (calibration.cpp, cvFindExtrinsicCameraParams2) instead of the 12-parameter DLT.
 * Written for checking, not for speed, and on purpose by other numerical routes than a closed-form implementation would
 * take: the three-point problem is solved as the resultant of its two quadrics with an iterative polynomial root finder,
 * the absolute orientation by Horn's quaternion, the homography by a singular value decomposition of the stacked
 * constraint matrix with isotropic point normalisation.  Parity unpinned (see vo_oracle.h).
 */
#include "vo_oracle.h"
#include "linalg.h"
#include <complex>
#include <cstring>

namespace {

typedef std::vector<double> Poly; /* coefficients, lowest degree first */

Poly pmul(const Poly& a, const Poly& b)
{
    Poly r(a.size() + b.size() - 1, 0.0);
    for (size_t i = 0; i < a.size(); i++) for (size_t j = 0; j < b.size(); j++) r[i + j] += a[i] * b[j];
    return r;
}
Poly padd(const Poly& a, const Poly& b, double sb = 1.0)
{
    Poly r(std::max(a.size(), b.size()), 0.0);
    for (size_t i = 0; i < a.size(); i++) r[i] += a[i];
    for (size_t i = 0; i < b.size(); i++) r[i] += sb * b[i];
    return r;
}
double peval(const Poly& p, double x) { double s = 0; for (size_t i = p.size(); i-- > 0;) s = s * x + p[i]; return s; }

/* all complex roots by the Durand-Kerner iteration */
std::vector<std::complex<double>> poly_roots(Poly p)
{
    while (p.size() > 1 && std::abs(p.back()) < 1e-300) p.pop_back();
    const int n = (int)p.size() - 1;
    std::vector<std::complex<double>> z(n);
    if (n < 1) return z;
    for (double& c : p) c /= p[n];
    double radius = 0;
    for (int i = 0; i < n; i++) radius = std::max(radius, std::abs(p[i]));
    radius = 1 + radius;
    for (int i = 0; i < n; i++) z[i] = std::polar(radius * 0.5, 2 * M_PI * i / n + 0.4);
    for (int it = 0; it < 2000; it++) {
        double move = 0;
        for (int i = 0; i < n; i++) {
            std::complex<double> num = 0, den = 1;
            for (int k = n; k >= 0; k--) num = num * z[i] + p[k];
            for (int j = 0; j < n; j++) if (j != i) den *= (z[i] - z[j]);
            const std::complex<double> d = num / den;
            z[i] -= d;
            move = std::max(move, std::abs(d));
        }
        if (move < 1e-15 * radius) break;
    }
    return z;
}

/* quaternion (w, x, y, z) of the rotation that best maps wld -> cam (Horn 1987), then the translation */
bool horn(const double cam[3][3], const double wld[3][3], double R[9], double t[3])
{
    double cc[3] = {0, 0, 0}, cw[3] = {0, 0, 0};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cc[j] += cam[i][j] / 3; cw[j] += wld[i][j] / 3; }
    double S[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}; /* S[a][b] = sum wld_a * cam_b */
    for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) S[a][b] += (wld[i][a] - cw[a]) * (cam[i][b] - cc[b]);
    double N[16] = {
        S[0][0] + S[1][1] + S[2][2], S[1][2] - S[2][1], S[2][0] - S[0][2], S[0][1] - S[1][0],
        S[1][2] - S[2][1], S[0][0] - S[1][1] - S[2][2], S[0][1] + S[1][0], S[2][0] + S[0][2],
        S[2][0] - S[0][2], S[0][1] + S[1][0], -S[0][0] + S[1][1] - S[2][2], S[1][2] + S[2][1],
        S[0][1] - S[1][0], S[2][0] + S[0][2], S[1][2] + S[2][1], -S[0][0] - S[1][1] + S[2][2]};
    /* largest eigenvalue of the symmetric N: shift to positive definite, then the SVD's first singular vector */
    double shift = 0;
    for (int k = 0; k < 16; k++) shift += std::abs(N[k]);
    if (!(shift > 0)) return false;
    for (int k = 0; k < 4; k++) N[k * 4 + k] += shift;
    double W[4], U[16], V[16];
    orc::svd(N, 4, 4, W, U, V);
    if (!(W[0] - W[1] > 1e-12 * W[0])) return false; /* collinear points: the rotation about their line is free */
    const double q0 = V[0], qx = V[4], qy = V[8], qz = V[12];
    const double nq = q0 * q0 + qx * qx + qy * qy + qz * qz;
    R[0] = (q0 * q0 + qx * qx - qy * qy - qz * qz) / nq; R[1] = 2 * (qx * qy - q0 * qz) / nq; R[2] = 2 * (qx * qz + q0 * qy) / nq;
    R[3] = 2 * (qy * qx + q0 * qz) / nq; R[4] = (q0 * q0 - qx * qx + qy * qy - qz * qz) / nq; R[5] = 2 * (qy * qz - q0 * qx) / nq;
    R[6] = 2 * (qz * qx - q0 * qy) / nq; R[7] = 2 * (qz * qy + q0 * qx) / nq; R[8] = (q0 * q0 - qx * qx - qy * qy + qz * qz) / nq;
    for (int r = 0; r < 3; r++) t[r] = cc[r] - (R[r * 3] * cw[0] + R[r * 3 + 1] * cw[1] + R[r * 3 + 2] * cw[2]);
    return true;
}

} // namespace

/* Four correspondences.  With s_i the distances from the projection centre to points 0, 1, 2 and u_i the unit bearings:
 *   s1^2 + s2^2 - 2 s1 s2 c12 = d12^2,  s0^2 + s2^2 - 2 s0 s2 c02 = d02^2,  s0^2 + s1^2 - 2 s0 s1 c01 = d01^2.
 * With x = s0 / s2, y = s1 / s2 and the third equation used to eliminate s2, two quadrics in (x, y) remain;
 * as polynomials in y with coefficients in x their resultant is a quartic in x. */
extern "C" int orc_p3p(const orc_point3f* obj, const orc_point2f* img, const double* K, double rvec[3], double tvec[3])
{
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double X[4][3], pix[4][2], u[3][3];
    for (int i = 0; i < 4; i++) {
        X[i][0] = obj[i].x; X[i][1] = obj[i].y; X[i][2] = obj[i].z;
        /* undistortPoints(..., P = cameraMatrix) keeps the points as float pixels */
        pix[i][0] = (double)(float)((((double)img[i].x - cx) / fx) * fx + cx);
        pix[i][1] = (double)(float)((((double)img[i].y - cy) / fy) * fy + cy);
    }
    for (int i = 0; i < 3; i++) {
        const double a = (pix[i][0] - cx) / fx, b = (pix[i][1] - cy) / fy, n = std::sqrt(a * a + b * b + 1);
        u[i][0] = a / n; u[i][1] = b / n; u[i][2] = 1 / n;
    }
    auto sq = [&](int a, int b) { double s = 0; for (int j = 0; j < 3; j++) s += (X[a][j] - X[b][j]) * (X[a][j] - X[b][j]); return s; };
    auto cs = [&](int a, int b) { return u[a][0] * u[b][0] + u[a][1] * u[b][1] + u[a][2] * u[b][2]; };
    const double d12 = sq(1, 2), d02 = sq(0, 2), d01 = sq(0, 1);
    if (!(d01 > 0) || !(d12 > 0) || !(d02 > 0)) return 0;
    const double c12 = cs(1, 2), c02 = cs(0, 2), c01 = cs(0, 1);
    const double a = d12 / d01, b = d02 / d01;
    /* v = x^2 + y^2 - 2 x y c01 (= d01^2 / s2^2):   y^2 + 1 - 2 y c12 = a v   and   x^2 + 1 - 2 x c02 = b v
     * as quadratics in y:  (1 - a) y^2 + (2 a c01 x - 2 c12) y + (1 - a x^2) = 0
     *                      (-b) y^2 + (2 b c01 x) y + ((1 - b) x^2 - 2 c02 x + 1) = 0 */
    const Poly A2 = {1 - a}, A1 = {-2 * c12, 2 * a * c01}, A0 = {1, 0, -a};
    const Poly B2 = {-b}, B1 = {0, 2 * b * c01}, B0 = {1, -2 * c02, 1 - b};
    /* resultant of two quadratics: (A2 B0 - A0 B2)^2 - (A2 B1 - A1 B2)(A1 B0 - A0 B1) */
    const Poly t0 = padd(pmul(A2, B0), pmul(A0, B2), -1), t1 = padd(pmul(A2, B1), pmul(A1, B2), -1), t2 = padd(pmul(A1, B0), pmul(A0, B1), -1);
    const Poly res = padd(pmul(t0, t0), pmul(t1, t2), -1);
    double scale = 0;
    for (double c : res) scale = std::max(scale, std::abs(c));
    if (!(scale > 0)) return 0;
    double best = DBL_MAX, bestR[9], bestT[3];
    bool found = false;
    std::vector<double> seen;
    for (const std::complex<double>& z : poly_roots(res)) {
        if (std::abs(z.imag()) > 1e-7 * (1 + std::abs(z.real())) || !(z.real() > 0)) continue;
        const double x = z.real();
        bool dup = false;
        for (double s : seen) dup = dup || std::abs(s - x) < 1e-9 * (1 + std::abs(x));
        if (dup) continue;
        seen.push_back(x);
        /* the common root in y: eliminate y^2 between the two quadratics */
        const double den = peval(t1, x);
        if (std::abs(den) < 1e-14) continue;
        const double y = -peval(t0, x) / den;
        if (!(y > 0)) continue;
        const double v = x * x + y * y - 2 * x * y * c01;
        if (!(v > 0)) continue;
        const double s2 = std::sqrt(d01 / v), s[3] = {x * s2, y * s2, s2};
        double cam[3][3], wld[3][3], R[9], t[3];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { cam[i][j] = s[i] * u[i][j]; wld[i][j] = X[i][j]; }
        if (!horn(cam, wld, R, t)) continue;
        double err = 0;
        for (int i = 0; i < 4; i++) {
            const double xc = R[0] * X[i][0] + R[1] * X[i][1] + R[2] * X[i][2] + t[0];
            const double yc = R[3] * X[i][0] + R[4] * X[i][1] + R[5] * X[i][2] + t[1];
            const double zc = R[6] * X[i][0] + R[7] * X[i][1] + R[8] * X[i][2] + t[2];
            const double du = fx * xc / zc + cx - pix[i][0], dv = fy * yc / zc + cy - pix[i][1];
            err += du * du + dv * dv;
        }
        if (err < best) { best = err; found = true; memcpy(bestR, R, sizeof(R)); memcpy(bestT, t, sizeof(t)); }
    }
    if (!found) return 0;
    orc_rodrigues_mat2vec(bestR, rvec);
    for (int i = 0; i < 3; i++) tvec[i] = bestT[i];
    return 1;
}

/* Pose start for a planar object: obj (n x 3, all in one plane), mn (n x 2 normalised image points), Vt = rows are the
 * principal axes of the centred object points (third = plane normal), Mc their centroid.  Returns [rvec | tvec]. */
extern "C" int orc_planar_start(const double* obj, const double* mn, int n, const double* Vt, const double* Mc, double param[6])
{
    double Rt[9];
    memcpy(Rt, Vt, sizeof(Rt));
    if (Rt[2] * Rt[2] + Rt[5] * Rt[5] < 1e-10) { for (int k = 0; k < 9; k++) Rt[k] = (k % 4 == 0) ? 1 : 0; }
    if (orc::det3(Rt) < 0) for (int k = 0; k < 9; k++) Rt[k] = -Rt[k];
    double Tt[3];
    for (int r = 0; r < 3; r++) Tt[r] = -(Rt[r * 3] * Mc[0] + Rt[r * 3 + 1] * Mc[1] + Rt[r * 3 + 2] * Mc[2]);
    /* isotropic normalisation of both point sets (centroid to the origin, mean distance sqrt 2) */
    std::vector<double> P((size_t)2 * n), Q((size_t)2 * n);
    double cP[2] = {0, 0}, cQ[2] = {0, 0}, sP = 0, sQ = 0;
    for (int i = 0; i < n; i++) {
        const double* M = obj + 3 * i;
        P[2 * i] = Rt[0] * M[0] + Rt[1] * M[1] + Rt[2] * M[2] + Tt[0];
        P[2 * i + 1] = Rt[3] * M[0] + Rt[4] * M[1] + Rt[5] * M[2] + Tt[1];
        cP[0] += P[2 * i] / n; cP[1] += P[2 * i + 1] / n; cQ[0] += mn[2 * i] / n; cQ[1] += mn[2 * i + 1] / n;
    }
    for (int i = 0; i < n; i++) { sP += std::hypot(P[2 * i] - cP[0], P[2 * i + 1] - cP[1]) / n; sQ += std::hypot(mn[2 * i] - cQ[0], mn[2 * i + 1] - cQ[1]) / n; }
    bool ok = n >= 4 && sP > 0 && sQ > 0;
    double h[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (ok) {
        sP = std::sqrt(2.0) / sP; sQ = std::sqrt(2.0) / sQ;
        const int rows = std::max(2 * n, 9);
        std::vector<double> L((size_t)rows * 9, 0.0);
        for (int i = 0; i < n; i++) {
            const double Xn = (P[2 * i] - cP[0]) * sP, Yn = (P[2 * i + 1] - cP[1]) * sP, xn = (mn[2 * i] - cQ[0]) * sQ, yn = (mn[2 * i + 1] - cQ[1]) * sQ;
            double* r1 = &L[(size_t)(2 * i) * 9]; double* r2 = r1 + 9;
            r1[0] = Xn; r1[1] = Yn; r1[2] = 1; r1[6] = -xn * Xn; r1[7] = -xn * Yn; r1[8] = -xn;
            r2[3] = Xn; r2[4] = Yn; r2[5] = 1; r2[6] = -yn * Xn; r2[7] = -yn * Yn; r2[8] = -yn;
        }
        std::vector<double> W(9), U((size_t)rows * 9), V(81);
        orc::svd(L.data(), rows, 9, W.data(), U.data(), V.data());
        double Hn[9];
        for (int k = 0; k < 9; k++) Hn[k] = V[k * 9 + 8];
        /* h = inv(Tq) Hn Tp */
        const double iTq[9] = {1 / sQ, 0, cQ[0], 0, 1 / sQ, cQ[1], 0, 0, 1}, Tp[9] = {sP, 0, -sP * cP[0], 0, sP, -sP * cP[1], 0, 0, 1};
        double tmp[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tmp[r * 3 + c] = iTq[r * 3] * Hn[c] + iTq[r * 3 + 1] * Hn[3 + c] + iTq[r * 3 + 2] * Hn[6 + c];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) h[r * 3 + c] = tmp[r * 3] * Tp[c] + tmp[r * 3 + 1] * Tp[3 + c] + tmp[r * 3 + 2] * Tp[6 + c];
        ok = std::abs(h[8]) > DBL_MIN;
        if (ok) { const double s = 1 / h[8]; for (int k = 0; k < 9; k++) { h[k] *= s; ok = ok && std::isfinite(h[k]); } }
    }
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    if (ok) {
        const double n1 = std::sqrt(h[0] * h[0] + h[3] * h[3] + h[6] * h[6]), n2 = std::sqrt(h[1] * h[1] + h[4] * h[4] + h[7] * h[7]);
        double c1[3], c2[3], tt[3];
        for (int r = 0; r < 3; r++) { c1[r] = h[r * 3] / std::max(n1, DBL_EPSILON); c2[r] = h[r * 3 + 1] / std::max(n2, DBL_EPSILON); tt[r] = h[r * 3 + 2] * 2 / std::max(n1 + n2, DBL_EPSILON); }
        const double c3[3] = {c1[1] * c2[2] - c1[2] * c2[1], c1[2] * c2[0] - c1[0] * c2[2], c1[0] * c2[1] - c1[1] * c2[0]};
        const double Hm[9] = {c1[0], c2[0], c3[0], c1[1], c2[1], c3[1], c1[2], c2[2], c3[2]};
        double rv[3], Rh[9];
        orc_rodrigues_mat2vec(Hm, rv);          /* Rodrigues there and back: the nearest rotation */
        orc_rodrigues_vec2mat(rv, Rh, nullptr);
        for (int r = 0; r < 3; r++) t[r] = Rh[r * 3] * Tt[0] + Rh[r * 3 + 1] * Tt[1] + Rh[r * 3 + 2] * Tt[2] + tt[r];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[r * 3 + c] = Rh[r * 3] * Rt[c] + Rh[r * 3 + 1] * Rt[3 + c] + Rh[r * 3 + 2] * Rt[6 + c];
    }
    orc_rodrigues_mat2vec(R, param);
    for (int k = 0; k < 3; k++) param[3 + k] = t[k];
    return ok ? 1 : 0;
}
