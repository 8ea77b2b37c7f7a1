// Dense f64 helpers for the host-side minimal solvers: Hestenes (one-sided Jacobi) SVD, least squares through
// the SVD with a relative singular-value cut, 3x3 inverse, n x n LU inverse.  Row-major.
#ifndef VO_SMALL_LINALG_H
#define VO_SMALL_LINALG_H

#include <cmath>
#include <cfloat>
#include <vector>
#include <algorithm>
#include <utility>

namespace sla {

// A (m x n, m >= n) = U diag(W) V^T, W descending.  U (m x n) and V (n x n) may be null.
// Work arrays are caller-provided through `ws` to keep the solvers allocation-free in their inner loops.
struct SvdWorkspace { std::vector<double> a, v, w; };

inline void svd(const double* A, int m, int n, double* W, double* U, double* V, SvdWorkspace& ws)
{
    ws.a.resize((size_t)n * m); ws.v.assign((size_t)n * n, 0.0); ws.w.resize(n);
    double* a = ws.a.data(); double* v = ws.v.data(); double* w = ws.w.data();
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const double t = A[(size_t)k * n + i]; a[(size_t)i * m + k] = t; sd += t * t; }
        w[i] = sd;
        v[(size_t)i * n + i] = 1.0;
    }
    const double eps = DBL_EPSILON * 10;
    const int sweeps = std::max(m, 30);
    for (int sweep = 0; sweep < sweeps; sweep++) {
        bool rotated = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++) {
                double* ai = a + (size_t)i * m; double* aj = a + (size_t)j * m;
                double aa = w[i], bb = w[j], p = 0;
                for (int k = 0; k < m; k++) p += ai[k] * aj[k];
                if (std::abs(p) <= eps * std::sqrt(aa * bb)) continue;
                p *= 2;
                const double beta = aa - bb, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = std::sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = std::sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                aa = bb = 0;
                for (int k = 0; k < m; k++) {
                    const double t0 = c * ai[k] + s * aj[k];
                    const double t1 = -s * ai[k] + c * aj[k];
                    ai[k] = t0; aj[k] = t1;
                    aa += t0 * t0; bb += t1 * t1;
                }
                w[i] = aa; w[j] = bb;
                rotated = true;
                double* vi = v + (size_t)i * n; double* vj = v + (size_t)j * n;
                for (int k = 0; k < n; k++) {
                    const double t0 = c * vi[k] + s * vj[k];
                    const double t1 = -s * vi[k] + c * vj[k];
                    vi[k] = t0; vj[k] = t1;
                }
            }
        if (!rotated) break;
    }
    for (int i = 0; i < n; i++) {
        double sd = 0;
        for (int k = 0; k < m; k++) { const double t = a[(size_t)i * m + k]; sd += t * t; }
        w[i] = std::sqrt(sd);
    }
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
        const double s = w[i] > DBL_MIN ? 1.0 / w[i] : 0.0;
        if (U) for (int k = 0; k < m; k++) U[(size_t)k * n + i] = a[(size_t)i * m + k] * s;
        if (V) for (int k = 0; k < n; k++) V[(size_t)k * n + i] = v[(size_t)i * n + k];
    }
}

// minimum-norm least squares x = V diag(1/w) U^T b; singular values <= 2*eps*sum(w) are dropped
inline void lstsq(const double* A, int m, int n, const double* b, double* x, SvdWorkspace& ws)
{
    double W[16], U[16 * 16], V[16 * 16]; // m, n <= 16 in every use here
    svd(A, m, n, W, U, V, ws);
    double cut = 0;
    for (int i = 0; i < n; i++) cut += W[i];
    cut *= DBL_EPSILON * 2;
    for (int k = 0; k < n; k++) x[k] = 0;
    for (int i = 0; i < n; i++) {
        if (std::abs(W[i]) <= cut) continue;
        double s = 0;
        for (int k = 0; k < m; k++) s += U[(size_t)k * n + i] * b[k];
        s /= W[i];
        for (int k = 0; k < n; k++) x[k] += s * V[(size_t)k * n + i];
    }
}

inline void inv3(const double* A, double* Ainv, SvdWorkspace& ws)
{
    for (int c = 0; c < 3; c++) {
        double e[3] = {0, 0, 0}, x[3];
        e[c] = 1;
        lstsq(A, 3, 3, e, x, ws);
        for (int r = 0; r < 3; r++) Ainv[r * 3 + c] = x[r];
    }
}

inline double det3(const double* M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// U * V^T for 3x3 factors
inline void uvt3(const double* U, const double* V, double* R)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = U[i * 3] * V[j * 3] + U[i * 3 + 1] * V[j * 3 + 1] + U[i * 3 + 2] * V[j * 3 + 2];
}

// inverse by Gaussian elimination with partial pivoting (n <= 8)
inline bool inv_lu(const double* A, int n, double* Ainv)
{
    double a[64], b[64];
    for (int i = 0; i < n * n; i++) { a[i] = A[i]; b[i] = 0; }
    for (int i = 0; i < n; i++) b[i * n + i] = 1;
    for (int i = 0; i < n; i++) {
        int k = i;
        for (int j = i + 1; j < n; j++) if (std::abs(a[j * n + i]) > std::abs(a[k * n + i])) k = j;
        if (std::abs(a[k * n + i]) < DBL_EPSILON) return false;
        if (k != i) for (int j = 0; j < n; j++) { std::swap(a[i * n + j], a[k * n + j]); std::swap(b[i * n + j], b[k * n + j]); }
        const double d = -1 / a[i * n + i];
        for (int j = i + 1; j < n; j++) {
            const double alpha = a[j * n + i] * d;
            for (int c = i + 1; c < n; c++) a[j * n + c] += alpha * a[i * n + c];
            for (int c = 0; c < n; c++) b[j * n + c] += alpha * b[i * n + c];
        }
    }
    for (int i = n - 1; i >= 0; i--)
        for (int j = 0; j < n; j++) {
            double s = b[i * n + j];
            for (int k = i + 1; k < n; k++) s -= a[i * n + k] * Ainv[k * n + j];
            Ainv[i * n + j] = s / a[i * n + i];
        }
    return true;
}

} // namespace sla
#endif
