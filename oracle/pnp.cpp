/* CPU ORACLE (test infrastructure only) — cv::solvePnPRansac with default arguments, cv::Rodrigues,
 * Mat::inv, as called at reference Source/VOClass.cpp:531,546,587.
 * Restates OpenCV 4.5.4 calib3d: solvepnp.cpp (solvePnPRansac, PnPRansacCallback), ptsetreg.cpp
 * (RANSACPointSetRegistrator, RANSACUpdateNumIters), epnp.cpp (EPnP minimal solver on 5 points),
 * calibration.cpp (cvFindExtrinsicCameraParams2: DLT start + Levenberg-Marquardt, cvProjectPoints2,
 * cvRodrigues2), compat_ptsetreg.cpp (CvLevMarq), core rand.cpp (RNG).
 * Parity unpinned (see vo_oracle.h): OpenCV's own SVD is LAPACK-backed on the reference author's
 * machine, so the basis it returns for the 2-dimensional null space of the 5-point EPnP system is
 * not reproducible anywhere; everything downstream of RANSAC is insensitive to it once the inlier
 * set agrees.
 */
#include "vo_oracle.h"
#include "linalg.h"
#include <cstring>
#include <cstdint>

namespace {

/* ---------------- OpenCV RNG (multiply-with-carry) ---------------- */
struct Rng {
    uint64_t state;
    explicit Rng(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

void draw_subset(Rng& rng, int count, int model_points, int* idx)
{
    for (int i = 0; i < model_points; i++) {
        int v;
        for (;;) {
            v = rng.uniform(0, count);
            bool dup = false;
            for (int j = 0; j < i; j++) if (idx[j] == v) { dup = true; break; }
            if (!dup) break;
        }
        idx[i] = v;
    }
}

int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = std::max(p, 0.); p = std::min(p, 1.);
    ep = std::max(ep, 0.); ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* ---------------- EPnP ---------------- */
struct Epnp {
    double fu, fv, uc, vc;
    int n;
    std::vector<double> pws, us, alphas, pcs;
    double cws[4][3], ccs[4][3];

    static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static double dist2(const double* a, const double* b)
    { return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]); }

    void control_points()
    {
        cws[0][0] = cws[0][1] = cws[0][2] = 0;
        for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
        for (int j = 0; j < 3; j++) cws[0][j] /= n;
        std::vector<double> pw0((size_t)n * 3);
        for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) pw0[3 * i + j] = pws[3 * i + j] - cws[0][j];
        double cov[9] = {0};
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
            double s = 0;
            for (int i = 0; i < n; i++) s += pw0[3 * i + r] * pw0[3 * i + c];
            cov[r * 3 + c] = s;
        }
        double dc[3], U[9], V[9];
        orc::svd(cov, 3, 3, dc, U, V);
        for (int i = 1; i < 4; i++) {
            double k = std::sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * U[j * 3 + (i - 1)];
        }
    }
    void barycentric()
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
        orc::inv3x3_svd(cc, ci);
        for (int i = 0; i < n; i++) {
            const double* pi = &pws[3 * i]; double* a = &alphas[4 * i];
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
            a[0] = 1.0f - a[1] - a[2] - a[3];
        }
    }
    void fill_M(double* M)
    {
        for (int p = 0; p < n; p++) {
            const double* as = &alphas[4 * p]; double u = us[2 * p], v = us[2 * p + 1];
            double* M1 = M + (size_t)(2 * p) * 12; double* M2 = M1 + 12;
            for (int i = 0; i < 4; i++) {
                M1[3 * i] = as[i] * fu; M1[3 * i + 1] = 0.0; M1[3 * i + 2] = as[i] * (uc - u);
                M2[3 * i] = 0.0; M2[3 * i + 1] = as[i] * fv; M2[3 * i + 2] = as[i] * (vc - v);
            }
        }
    }
    void L_6x10(const double* ut, double* L)
    {
        const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                for (int k = 0; k < 3; k++) dv[i][j][k] = v[i][3 * a + k] - v[i][3 * b + k];
                b++;
                if (b > 3) { a++; b = a + 1; }
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
    void rho(double* r)
    {
        r[0] = dist2(cws[0], cws[1]); r[1] = dist2(cws[0], cws[2]); r[2] = dist2(cws[0], cws[3]);
        r[3] = dist2(cws[1], cws[2]); r[4] = dist2(cws[1], cws[3]); r[5] = dist2(cws[2], cws[3]);
    }
    /* betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44] */
    void betas_approx_1(const double* L, const double* r, double* betas)
    {
        double l[24], b4[4];
        for (int i = 0; i < 6; i++) { l[i * 4] = L[i * 10]; l[i * 4 + 1] = L[i * 10 + 1]; l[i * 4 + 2] = L[i * 10 + 3]; l[i * 4 + 3] = L[i * 10 + 6]; }
        orc::svd_solve(l, 6, 4, r, b4);
        if (b4[0] < 0) { betas[0] = std::sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0]; }
        else { betas[0] = std::sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0]; }
    }
    void betas_approx_2(const double* L, const double* r, double* betas)
    {
        double l[18], b3[3];
        for (int i = 0; i < 6; i++) { l[i * 3] = L[i * 10]; l[i * 3 + 1] = L[i * 10 + 1]; l[i * 3 + 2] = L[i * 10 + 2]; }
        orc::svd_solve(l, 6, 3, r, b3);
        if (b3[0] < 0) { betas[0] = std::sqrt(-b3[0]); betas[1] = (b3[2] < 0) ? std::sqrt(-b3[2]) : 0.0; }
        else { betas[0] = std::sqrt(b3[0]); betas[1] = (b3[2] > 0) ? std::sqrt(b3[2]) : 0.0; }
        if (b3[1] < 0) betas[0] = -betas[0];
        betas[2] = 0.0; betas[3] = 0.0;
    }
    void betas_approx_3(const double* L, const double* r, double* betas)
    {
        double l[30], b5[5];
        for (int i = 0; i < 6; i++) for (int k = 0; k < 5; k++) l[i * 5 + k] = L[i * 10 + k];
        orc::svd_solve(l, 6, 5, r, b5);
        if (b5[0] < 0) { betas[0] = std::sqrt(-b5[0]); betas[1] = (b5[2] < 0) ? std::sqrt(-b5[2]) : 0.0; }
        else { betas[0] = std::sqrt(b5[0]); betas[1] = (b5[2] > 0) ? std::sqrt(b5[2]) : 0.0; }
        if (b5[1] < 0) betas[0] = -betas[0];
        betas[2] = b5[3] / betas[0];
        betas[3] = 0.0;
    }
    /* Householder QR least squares, 6x4.  The column-scale search skips the last row, as the original does. */
    static void qr_solve(double* A, double* b, double* X, int nr, int nc)
    {
        std::vector<double> A1(nr), A2(nr);
        for (int k = 0; k < nc; k++) {
            double eta = std::abs(A[k * nc + k]);
            for (int i = k + 1; i < nr; i++) { double elt = std::abs(A[(i - 1) * nc + k]); if (eta < elt) eta = elt; }
            if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
            double sum2 = 0.0, inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) { A[i * nc + k] *= inv_eta; sum2 += A[i * nc + k] * A[i * nc + k]; }
            double sigma = std::sqrt(sum2);
            if (A[k * nc + k] < 0) sigma = -sigma;
            A[k * nc + k] += sigma;
            A1[k] = sigma * A[k * nc + k];
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double sum = 0;
                for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
                double tau = sum / A1[k];
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
    void gauss_newton(const double* L, const double* r, double betas[4])
    {
        for (int it = 0; it < 5; it++) {
            double A[24], b[6], x[4] = {0, 0, 0, 0};
            for (int i = 0; i < 6; i++) {
                const double* rl = L + i * 10; double* ra = A + i * 4;
                ra[0] = 2 * rl[0] * betas[0] + rl[1] * betas[1] + rl[3] * betas[2] + rl[6] * betas[3];
                ra[1] = rl[1] * betas[0] + 2 * rl[2] * betas[1] + rl[4] * betas[2] + rl[7] * betas[3];
                ra[2] = rl[3] * betas[0] + rl[4] * betas[1] + 2 * rl[5] * betas[2] + rl[8] * betas[3];
                ra[3] = rl[6] * betas[0] + rl[7] * betas[1] + rl[8] * betas[2] + 2 * rl[9] * betas[3];
                b[i] = r[i] - (rl[0] * betas[0] * betas[0] + rl[1] * betas[0] * betas[1] + rl[2] * betas[1] * betas[1] +
                               rl[3] * betas[0] * betas[2] + rl[4] * betas[1] * betas[2] + rl[5] * betas[2] * betas[2] +
                               rl[6] * betas[0] * betas[3] + rl[7] * betas[1] * betas[3] + rl[8] * betas[2] * betas[3] +
                               rl[9] * betas[3] * betas[3]);
            }
            qr_solve(A, b, x, 6, 4);
            for (int i = 0; i < 4; i++) betas[i] += x[i];
        }
    }
    double reprojection_error(const double R[3][3], const double t[3])
    {
        double sum2 = 0.0;
        for (int i = 0; i < n; i++) {
            const double* pw = &pws[3 * i];
            double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
            double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
            double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
            double u = us[2 * i], v = us[2 * i + 1];
            sum2 += std::sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n;
    }
    void estimate_R_t(double R[3][3], double t[3])
    {
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
        for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
        double abt[9] = {0};
        for (int i = 0; i < n; i++) {
            const double* pc = &pcs[3 * i]; const double* pw = &pws[3 * i];
            for (int j = 0; j < 3; j++) {
                abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
                abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
                abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
            }
        }
        double d[3], U[9], V[9];
        orc::svd(abt, 3, 3, d, U, V);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            R[i][j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
        const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                           R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
        if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
        t[0] = pc0[0] - dot3(R[0], pw0); t[1] = pc0[1] - dot3(R[1], pw0); t[2] = pc0[2] - dot3(R[2], pw0);
    }
    double R_and_t(const double* ut, const double* betas, double R[3][3], double t[3])
    {
        for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
        for (int i = 0; i < 4; i++) {
            const double* v = ut + 12 * (11 - i);
            for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
        }
        for (int i = 0; i < n; i++) {
            const double* a = &alphas[4 * i]; double* pc = &pcs[3 * i];
            for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
        if (pcs[2] < 0.0) {
            for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
            for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
        }
        estimate_R_t(R, t);
        return reprojection_error(R, t);
    }
    void compute_pose(double Rout[9], double tout[3])
    {
        alphas.resize((size_t)4 * n); pcs.resize((size_t)3 * n);
        control_points();
        barycentric();
        std::vector<double> M((size_t)2 * n * 12);
        fill_M(M.data());
        double mtm[144];
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += M[(size_t)i * 12 + r] * M[(size_t)i * 12 + c];
            mtm[r * 12 + c] = s;
        }
        double d[12], U[144], V[144], ut[144];
        orc::svd(mtm, 12, 12, d, U, V);
        /* rows of ut = singular vectors, descending; taken from the accumulated rotations (orthonormal
         * also for the numerically-zero singular values) */
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) ut[r * 12 + c] = V[c * 12 + r];
        double L[60], r6[6];
        L_6x10(ut, L); rho(r6);
        double Betas[4][4], rep[4], Rs[4][3][3], ts[4][3];
        betas_approx_1(L, r6, Betas[1]); gauss_newton(L, r6, Betas[1]); rep[1] = R_and_t(ut, Betas[1], Rs[1], ts[1]);
        betas_approx_2(L, r6, Betas[2]); gauss_newton(L, r6, Betas[2]); rep[2] = R_and_t(ut, Betas[2], Rs[2], ts[2]);
        betas_approx_3(L, r6, Betas[3]); gauss_newton(L, r6, Betas[3]); rep[3] = R_and_t(ut, Betas[3], Rs[3], ts[3]);
        int N = 1;
        if (rep[2] < rep[1]) N = 2;
        if (rep[3] < rep[N]) N = 3;
        for (int i = 0; i < 3; i++) { tout[i] = ts[N][i]; for (int j = 0; j < 3; j++) Rout[i * 3 + j] = Rs[N][i][j]; }
    }
};

/* undistortPoints with zero distortion, no R/P, float in -> float out: (u - cx) * (1/fx) in f64, stored f32 */
inline float normalize_f32(float u, double c, double f) { double inv = 1. / f; return (float)(((double)u - c) * inv); }

/* cvProjectPoints2 with zero distortion: pixel = (R X + t)_xy * (1/z) * f + c, f64 */
inline void project_f64(const double R[9], const double t[3], const double K[9], double X, double Y, double Z, double* u, double* v,
                        double* xn, double* yn, double* zinv)
{
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    *u = x * K[0] + K[2];
    *v = y * K[4] + K[5];
    if (xn) { *xn = x; *yn = y; *zinv = z; }
}

} // namespace

extern "C" void orc_rodrigues_vec2mat(const double r_in[3], double R[9], double* J)
{
    double rx = r_in[0], ry = r_in[1], rz = r_in[2];
    double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
    if (theta < DBL_EPSILON) {
        for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1 : 0;
        if (J) { memset(J, 0, 27 * sizeof(double)); J[5] = J[15] = J[19] = -1; J[7] = J[11] = J[21] = 1; }
        return;
    }
    double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                             0, 0, 1, 0, 0, 0, -1, 0, 0,
                             0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

extern "C" void orc_rodrigues_mat2vec(const double Rin[9], double r[3])
{
    double W[3], U[9], V[9], R[9];
    orc::svd(Rin, 3, 3, W, U, V);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
    double x = R[7] - R[5], y = R[2] - R[6], z = R[3] - R[1];
    double s = std::sqrt((x * x + y * y + z * z) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = std::acos(c);
    if (s < 1e-5) {
        if (c > 0) { x = y = z = 0; }
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

extern "C" void orc_inv4x4(const double* A, double* Ainv) { orc::inv_lu(A, 4, Ainv); }

extern "C" void orc_rng_subsets(int count, int model_points, int nsubsets, int* idx_out)
{
    Rng rng((uint64_t)-1);
    for (int s = 0; s < nsubsets; s++) draw_subset(rng, count, model_points, idx_out + (size_t)s * model_points);
}

/* solvePnP(..., SOLVEPNP_EPNP) on float points */
extern "C" int orc_epnp(const orc_point3f* obj, const orc_point2f* img, int n, const double* K, double rvec[3], double tvec[3])
{
    Epnp e;
    e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5]; e.n = n;
    e.pws.resize((size_t)3 * n); e.us.resize((size_t)2 * n);
    for (int i = 0; i < n; i++) {
        e.pws[3 * i] = obj[i].x; e.pws[3 * i + 1] = obj[i].y; e.pws[3 * i + 2] = obj[i].z;
        /* image points go through undistortPoints (-> normalised, stored as float) and back to pixels */
        e.us[2 * i] = normalize_f32(img[i].x, K[2], K[0]) * e.fu + e.uc;
        e.us[2 * i + 1] = normalize_f32(img[i].y, K[5], K[4]) * e.fv + e.vc;
    }
    double R[9];
    e.compute_pose(R, tvec);
    orc_rodrigues_mat2vec(R, rvec);
    return 1;
}

extern "C" void orc_pnp_errors(const orc_point3f* obj, const orc_point2f* img, int n, const double* K,
                               const double rvec[3], const double tvec[3], float* err_out)
{
    double R[9];
    orc_rodrigues_vec2mat(rvec, R, nullptr);
    for (int i = 0; i < n; i++) {
        double u, v;
        project_f64(R, tvec, K, obj[i].x, obj[i].y, obj[i].z, &u, &v, nullptr, nullptr, nullptr);
        float pu = (float)u, pv = (float)v;
        float dx = img[i].x - pu, dy = img[i].y - pv;
        err_out[i] = (float)((double)dx * dx + (double)dy * dy);
    }
}

/* cvFindExtrinsicCameraParams2 without an extrinsic guess: DLT start (non-planar) + CvLevMarq (<=20 iterations) */
extern "C" int orc_pnp_iterative(const double* obj, const double* img, int n, const double* K, double rvec[3], double tvec[3])
{
    const int max_iter = 20;
    std::vector<double> mn((size_t)2 * n);
    {
        double ifx = 1. / K[0], ify = 1. / K[4];
        for (int i = 0; i < n; i++) { mn[2 * i] = (img[2 * i] - K[2]) * ifx; mn[2 * i + 1] = (img[2 * i + 1] - K[5]) * ify; }
    }
    double Mc[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) Mc[j] += obj[3 * i + j];
    for (int j = 0; j < 3; j++) Mc[j] /= n;
    double MM[9] = {0};
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        double s = 0;
        for (int i = 0; i < n; i++) s += (obj[3 * i + r] - Mc[r]) * (obj[3 * i + c] - Mc[c]);
        MM[r * 3 + c] = s;
    }
    double W[3], U3[9], V3[9];
    orc::svd(MM, 3, 3, W, U3, V3);
    double param[6];
    if (W[2] / W[1] < 1e-3) { /* planar object: start from the plane's homography (pnp_minimal.cpp) */
        double Vt[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Vt[r * 3 + c] = V3[c * 3 + r];
        orc_planar_start(obj, mn.data(), n, Vt, Mc, param);
    } else {
        if (n < 6) return -3;          /* "DLT algorithm needs at least 6 points" */
        std::vector<double> L((size_t)2 * n * 12);
        for (int i = 0; i < n; i++) {
            double* l = &L[(size_t)i * 24];
            double x = -mn[2 * i], y = -mn[2 * i + 1];
            const double* M = &obj[3 * i];
            l[0] = l[16] = M[0]; l[1] = l[17] = M[1]; l[2] = l[18] = M[2]; l[3] = l[19] = 1.;
            l[4] = l[5] = l[6] = l[7] = 0.; l[12] = l[13] = l[14] = l[15] = 0.;
            l[8] = x * M[0]; l[9] = x * M[1]; l[10] = x * M[2]; l[11] = x;
            l[20] = y * M[0]; l[21] = y * M[1]; l[22] = y * M[2]; l[23] = y;
        }
        double LL[144];
        for (int r = 0; r < 12; r++) for (int c = 0; c < 12; c++) {
            double s = 0;
            for (int i = 0; i < 2 * n; i++) s += L[(size_t)i * 12 + r] * L[(size_t)i * 12 + c];
            LL[r * 12 + c] = s;
        }
        double LW[12], LU[144], LV[144];
        orc::svd(LL, 12, 12, LW, LU, LV);
        double RRt[12];
        for (int k = 0; k < 12; k++) RRt[k] = LV[k * 12 + 11];
        double RR[9] = {RRt[0], RRt[1], RRt[2], RRt[4], RRt[5], RRt[6], RRt[8], RRt[9], RRt[10]};
        double tt[3] = {RRt[3], RRt[7], RRt[11]};
        if (orc::det3(RR) < 0) { for (int k = 0; k < 9; k++) RR[k] = -RR[k]; for (int k = 0; k < 3; k++) tt[k] = -tt[k]; }
        double sc = 0; for (int k = 0; k < 9; k++) sc += RR[k] * RR[k]; sc = std::sqrt(sc);
        if (!(std::abs(sc) > DBL_EPSILON)) return -4;
        double w3[3], u3[9], v3[9], R[9];
        orc::svd(RR, 3, 3, w3, u3, v3);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
            R[i * 3 + j] = u3[i * 3] * v3[j * 3] + u3[i * 3 + 1] * v3[j * 3 + 1] + u3[i * 3 + 2] * v3[j * 3 + 2];
        double nr = 0; for (int k = 0; k < 9; k++) nr += R[k] * R[k]; nr = std::sqrt(nr);
        for (int k = 0; k < 3; k++) param[3 + k] = tt[k] * (nr / sc);
        orc_rodrigues_mat2vec(R, param);
    }
    /* CvLevMarq(6 params, 2n errors, eps = FLT_EPSILON, max 20), update() protocol with analytic Jacobian */
    const int m = 2 * n;
    std::vector<double> J((size_t)m * 6), err(m);
    double prevParam[6], JtJ[36], JtErr[6];
    double prevErrNorm = DBL_MAX, errNorm = 0;
    int lambdaLg10 = -3, iters = 0;
    auto eval = [&](const double* prm, bool want_J) {
        double R[9], dRdr[27];
        orc_rodrigues_vec2mat(prm, R, want_J ? dRdr : nullptr);
        for (int i = 0; i < n; i++) {
            double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2], u, v, x, y, z;
            project_f64(R, prm + 3, K, X, Y, Z, &u, &v, &x, &y, &z);
            err[2 * i] = u - img[2 * i]; err[2 * i + 1] = v - img[2 * i + 1];
            if (want_J) {
                double* jx = &J[(size_t)(2 * i) * 6]; double* jy = jx + 6;
                double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
                for (int j = 0; j < 3; j++) { jx[3 + j] = K[0] * dxdt[j]; jy[3 + j] = K[4] * dydt[j]; }
                double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11], X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
                double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14], X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
                double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17], X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
                for (int j = 0; j < 3; j++) {
                    double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                    double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                    jx[j] = K[0] * dxdr; jy[j] = K[4] * dydr;
                }
            }
        }
    };
    auto norm_err = [&]() { double s = 0; for (int i = 0; i < m; i++) s += err[i] * err[i]; return std::sqrt(s); };
    auto step = [&]() {
        double lambda = std::exp(lambdaLg10 * std::log(10.));
        double A[36], x[6];
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) A[r * 6 + c] = JtJ[r * 6 + c];
        for (int r = 0; r < 6; r++) A[r * 6 + r] *= 1. + lambda;
        orc::svd_solve(A, 6, 6, JtErr, x);
        for (int i = 0; i < 6; i++) param[i] = prevParam[i] - x[i];
    };
    for (;;) {
        /* state CALC_J */
        eval(param, true);
        for (int r = 0; r < 6; r++) {
            for (int c = r; c < 6; c++) {
                double s = 0;
                for (int i = 0; i < m; i++) s += J[(size_t)i * 6 + r] * J[(size_t)i * 6 + c];
                JtJ[r * 6 + c] = JtJ[c * 6 + r] = s;
            }
            double s = 0;
            for (int i = 0; i < m; i++) s += J[(size_t)i * 6 + r] * err[i];
            JtErr[r] = s;
        }
        memcpy(prevParam, param, sizeof(param));
        if (iters == 0) prevErrNorm = norm_err();
        step();
        /* state CHECK_ERR */
        bool done = false;
        for (;;) {
            eval(param, false);
            errNorm = norm_err();
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) { step(); continue; }
            }
            lambdaLg10 = std::max(lambdaLg10 - 1, -16);
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) { dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]); pn += prevParam[i] * prevParam[i]; }
            if (++iters >= max_iter || std::sqrt(dn) / std::sqrt(pn) < FLT_EPSILON) done = true;
            prevErrNorm = errNorm;
            break;
        }
        if (done) break;
    }
    for (int i = 0; i < 3; i++) { rvec[i] = param[i]; tvec[i] = param[3 + i]; }
    return 1;
}

extern "C" int orc_solve_pnp_ransac(const orc_point3f* obj, const orc_point2f* img, int n, const double* K,
                                    const orc_params* p, double rvec[3], double tvec[3], uint8_t* inlier_mask,
                                    orc_pnp_info* info)
{
    orc_pnp_info li; memset(&li, 0, sizeof(li));
    if (info) *info = li;
    if (n < 4) return 0; /* OpenCV asserts npoints >= 4 */
    const int model_points = 5;
    if (n == 4 || n == model_points) { /* no sampling: one minimal solve on all points (P3P on four, EPnP on five) */
        int ok = n == 4 ? orc_p3p(obj, img, K, rvec, tvec) : orc_epnp(obj, img, n, K, rvec, tvec);
        if (!ok) { for (int i = 0; i < 3; i++) rvec[i] = tvec[i] = 0; }
        if (inlier_mask) memset(inlier_mask, 1, n);
        li.ok = ok; li.best_inliers = n; if (info) *info = li;
        return ok;
    }
    Rng rng((uint64_t)-1);
    int niters = std::max(p->ransac_iters, 1), maxGood = 0, iter;
    const float thr = (float)((double)p->ransac_reproj * (double)p->ransac_reproj);
    std::vector<uint8_t> mask(n), bestMask(n, 0);
    std::vector<float> err(n);
    double bestR[3] = {0, 0, 0}, bestT[3] = {0, 0, 0};
    for (iter = 0; iter < niters; iter++) {
        int idx[5];
        draw_subset(rng, n, model_points, idx);
        orc_point3f so[5]; orc_point2f si[5];
        for (int i = 0; i < 5; i++) { so[i] = obj[idx[i]]; si[i] = img[idx[i]]; }
        double r[3], t[3];
        if (orc_epnp(so, si, 5, K, r, t) <= 0) continue;
        orc_pnp_errors(obj, img, n, K, r, t, err.data());
        int good = 0;
        for (int i = 0; i < n; i++) { int f = err[i] <= thr; mask[i] = (uint8_t)f; good += f; }
        if (good > std::max(maxGood, model_points - 1)) {
            mask.swap(bestMask);
            memcpy(bestR, r, sizeof(r)); memcpy(bestT, t, sizeof(t));
            maxGood = good;
            niters = ransac_update_num_iters(p->ransac_conf, (double)(n - good) / n, model_points, niters);
        }
    }
    li.iters_run = iter; li.best_inliers = maxGood;
    if (maxGood <= 0) { if (info) *info = li; return 0; }
    if (inlier_mask) memcpy(inlier_mask, bestMask.data(), n);
    /* final solvePnP(SOLVEPNP_ITERATIVE, no guess) on the inliers as f64 */
    std::vector<double> o, im;
    for (int i = 0; i < n; i++) if (bestMask[i]) {
        o.push_back(obj[i].x); o.push_back(obj[i].y); o.push_back(obj[i].z);
        im.push_back(img[i].x); im.push_back(img[i].y);
    }
    int n1 = (int)(im.size() / 2);
    int res = orc_pnp_iterative(o.data(), im.data(), n1, K, rvec, tvec);
    if (res == -3 && n1 == 5) { /* "DLT needs 6 points" is caught: keep the minimal-sample model */
        memcpy(rvec, bestR, sizeof(bestR)); memcpy(tvec, bestT, sizeof(bestT)); li.ok = 1;
    } else if (res <= 0) {
        memcpy(rvec, bestR, sizeof(bestR)); memcpy(tvec, bestT, sizeof(bestT)); li.ok = 0;
    } else { li.ok = 1; li.final_from_refine = 1; }
    if (info) *info = li;
    return li.ok;
}
