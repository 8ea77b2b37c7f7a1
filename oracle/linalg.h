/* CPU ORACLE (test infrastructure only) — small dense f64 linear algebra used by the PnP restatement.
 * One-sided (Hestenes) Jacobi SVD, SVD back-substitution, LU inverse.  Row-major everywhere.
 */
#ifndef VO_ORACLE_LINALG_H
#define VO_ORACLE_LINALG_H

#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>

namespace orc {

/* A (m x n, m >= n) = U diag(W) V^T.  U: m x n, V: n x n, W descending.
 * Columns of A are rotated pairwise until mutually orthogonal (relative test), V accumulates the rotations. */
inline void svd(const double* A, int m, int n, double* W, double* U, double* V)
{
    std::vector<double> a((size_t)n * m), v((size_t)n * n, 0.0), w(n);
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { double t = A[(size_t)k * n + i]; a[(size_t)i * m + k] = t; sd += t * t; }
        w[i] = sd;
        v[(size_t)i * n + i] = 1.0;
    }
    const double eps = DBL_EPSILON * 10;
    const int max_iter = std::max(m, 30);
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double* ai = &a[(size_t)i * m]; double* aj = &a[(size_t)j * m];
                double aa = w[i], bb = w[j], p = 0;
                for (int k = 0; k < m; k++) p += ai[k] * aj[k];
                if (std::abs(p) <= eps * std::sqrt(aa * bb)) continue;
                p *= 2;
                double beta = aa - bb, gamma = hypot(p, beta), c, s;
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                aa = bb = 0;
                for (int k = 0; k < m; k++) {
                    double t0 = c * ai[k] + s * aj[k];
                    double t1 = -s * ai[k] + c * aj[k];
                    ai[k] = t0; aj[k] = t1;
                    aa += t0 * t0; bb += t1 * t1;
                }
                w[i] = aa; w[j] = bb;
                changed = true;
                double* vi = &v[(size_t)i * n]; double* vj = &v[(size_t)j * n];
                for (int k = 0; k < n; k++) {
                    double t0 = c * vi[k] + s * vj[k];
                    double t1 = -s * vi[k] + c * vj[k];
                    vi[k] = t0; vj[k] = t1;
                }
            }
        if (!changed) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { double t = a[(size_t)i * m + k]; sd += t * t; }
        w[i] = std::sqrt(sd);
    }
    /* selection sort, descending */
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (w[j] < w[k]) j = k;
        if (i != j) {
            std::swap(w[i], w[j]);
            for (int k = 0; k < m; k++) std::swap(a[(size_t)i * m + k], a[(size_t)j * m + k]);
            for (int k = 0; k < n; k++) std::swap(v[(size_t)i * n + k], v[(size_t)j * n + k]);
        }
    }
    for (int i = 0; i < n; i++) {
        W[i] = w[i];
        double s = w[i] > DBL_MIN ? 1.0 / w[i] : 0.0;
        if (U) for (int k = 0; k < m; k++) U[(size_t)k * n + i] = a[(size_t)i * m + k] * s;
        if (V) for (int k = 0; k < n; k++) V[(size_t)k * n + i] = v[(size_t)i * n + k];
    }
}

/* x = V diag(1/w) U^T b with the small-singular-value cut of SVD back-substitution (threshold = 2*eps*sum(w)) */
inline void svd_solve(const double* A, int m, int n, const double* b, double* x)
{
    std::vector<double> W(n), U((size_t)m * n), V((size_t)n * n);
    svd(A, m, n, W.data(), U.data(), V.data());
    double thr = 0;
    for (int i = 0; i < n; i++) thr += W[i];
    thr *= DBL_EPSILON * 2;
    for (int k = 0; k < n; k++) x[k] = 0;
    for (int i = 0; i < n; i++) {
        if (std::abs(W[i]) <= thr) continue;
        double s = 0;
        for (int k = 0; k < m; k++) s += U[(size_t)k * n + i] * b[k];
        s /= W[i];
        for (int k = 0; k < n; k++) x[k] += s * V[(size_t)k * n + i];
    }
}

inline void inv3x3_svd(const double* A, double* Ainv)
{
    for (int c = 0; c < 3; c++) {
        double e[3] = {0, 0, 0}, x[3];
        e[c] = 1;
        svd_solve(A, 3, 3, e, x);
        for (int r = 0; r < 3; r++) Ainv[r * 3 + c] = x[r];
    }
}

inline double det3(const double* M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

/* Gaussian elimination with partial pivoting, n x n, solves A X = I */
inline bool inv_lu(const double* A, int n, double* Ainv)
{
    std::vector<double> a(A, A + (size_t)n * n), b((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) b[(size_t)i * n + i] = 1;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (std::abs(a[(size_t)j * n + i]) > std::abs(a[(size_t)k * n + i])) k = j;
        if (std::abs(a[(size_t)k * n + i]) < DBL_EPSILON) return false;
        if (k != i) for (int j = 0; j < n; j++) { std::swap(a[(size_t)i * n + j], a[(size_t)k * n + j]); std::swap(b[(size_t)i * n + j], b[(size_t)k * n + j]); }
        double d = -1 / a[(size_t)i * n + i];
        for (int j = i + 1; j < n; j++) {
            double alpha = a[(size_t)j * n + i] * d;
            for (int c = i + 1; c < n; c++) a[(size_t)j * n + c] += alpha * a[(size_t)i * n + c];
            for (int c = 0; c < n; c++) b[(size_t)j * n + c] += alpha * b[(size_t)i * n + c];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            double s = b[(size_t)i * n + j];
            for (int k = i + 1; k < n; k++) s -= a[(size_t)i * n + k] * Ainv[(size_t)k * n + j];
            Ainv[(size_t)i * n + j] = s / a[(size_t)i * n + i];
        }
    return true;
}

} // namespace orc
#endif
