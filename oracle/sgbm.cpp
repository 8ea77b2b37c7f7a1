/* CPU ORACLE (test infrastructure only) — StereoSGBM MODE_SGBM_3WAY + medianBlur(3).
 * Restates the OpenCV 4.5.4 calls made at reference Source/VOClass.cpp:258-296
 * (calib3d/src/stereosgbm.cpp: calcPixelCostBT, SGBM3WayMainLoop, computeDisparity3WaySGBM;
 * StereoSGBMImpl::compute applies medianBlur(disp, disp, 3) afterwards).
 * Parity unpinned (see vo_oracle.h).  Items that rest on knowledge of OpenCV internals
 * are isolated in named helpers: clip_tab(), stripe_layout(), wta_pick(), path_init.
 */
#include "vo_oracle.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <climits>
#include <cstdlib>

namespace {

int g_threads = 1;
unsigned g_variant = 0; /* sensitivity switches, see vo_oracle.h */

struct Sgbm {
    int W, H, minD, maxD, D, minX1, maxX1, width1, SW2, SH2, P1, P2, ftzero, disp12MaxDiff, uniq, lanes;
    const uint8_t *L, *R;
};

inline int clip_tab(int v, int ftzero) { return std::min(std::max(v, -ftzero), ftzero) + ftzero; }

/* nstripes is fixed at 4; stripe_sz = ceil(H/4); overlap = (block/2+1) + ceil(0.1*stripe_sz). */
void stripe_layout(int H, int block, int* stripe_sz, int* overlap)
{
    *stripe_sz = (int)std::ceil(H / 4.0);
    *overlap = (block / 2 + 1) + (int)std::ceil(0.1 * (*stripe_sz));
}

inline int16_t sat16(int v) { return (int16_t)std::min(std::max(v, (int)SHRT_MIN), (int)SHRT_MAX); }

/* calcPixelCostBT for one image row: pix[(x-minX1)*D + d], x in [minX1,maxX1), d in [0,D). */
void pixel_cost_row(const Sgbm& s, int y, std::vector<int16_t>& pix,
                    std::vector<uint8_t>& p1, std::vector<uint8_t>& p2,
                    std::vector<uint8_t>& v0b, std::vector<uint8_t>& v1b)
{
    const int W = s.W;
    p1.assign((size_t)2 * W, 0); p2.assign((size_t)2 * W, 0);
    const uint8_t* r1 = s.L + (size_t)y * W; const uint8_t* r2 = s.R + (size_t)y * W;
    const int n = y > 0 ? -W : 0, sdn = y < s.H - 1 ? W : 0;
    /* channel 0: clipped x-Sobel; channel 1: raw intensity.  Columns 0 and W-1 of BOTH channels = tab[0]. */
    for (int c = 0; c < 2; c++) {
        p1[c * W] = p1[c * W + W - 1] = (uint8_t)clip_tab(0, s.ftzero);
        p2[c * W] = p2[c * W + W - 1] = (uint8_t)clip_tab(0, s.ftzero);
    }
    for (int x = 1; x < W - 1; x++) {
        p1[x] = (uint8_t)clip_tab((r1[x + 1] - r1[x - 1]) * 2 + r1[x + n + 1] - r1[x + n - 1] + r1[x + sdn + 1] - r1[x + sdn - 1], s.ftzero);
        p2[x] = (uint8_t)clip_tab((r2[x + 1] - r2[x - 1]) * 2 + r2[x + n + 1] - r2[x + n - 1] + r2[x + sdn + 1] - r2[x + sdn - 1], s.ftzero);
        p1[W + x] = r1[x];
        p2[W + x] = r2[x];
    }
    if (g_variant & ORC_VAR_RAW_BORDER_PIXEL) { p1[W] = r1[0]; p1[2 * W - 1] = r1[W - 1]; p2[W] = r2[0]; p2[2 * W - 1] = r2[W - 1]; }
    pix.assign((size_t)s.width1 * s.D, 0);
    v0b.resize(W); v1b.resize(W);
    for (int c = 0; c < 2; c++) {
        const uint8_t* a = &p1[c * W]; const uint8_t* b = &p2[c * W];
        const int shift = c == 0 ? 0 : 2;
        for (int x = 0; x < W; x++) {
            int v = b[x];
            int vl = x > 0 ? (v + b[x - 1]) / 2 : v;
            int vr = x < W - 1 ? (v + b[x + 1]) / 2 : v;
            v0b[x] = (uint8_t)std::min(std::min(vl, vr), v);
            v1b[x] = (uint8_t)std::max(std::max(vl, vr), v);
        }
        for (int x = s.minX1; x < s.maxX1; x++) {
            int u = a[x];
            int ul = x > 0 ? (u + a[x - 1]) / 2 : u;
            int ur = x < W - 1 ? (u + a[x + 1]) / 2 : u;
            int u0 = std::min(std::min(ul, ur), u), u1 = std::max(std::max(ul, ur), u);
            int16_t* cost = &pix[(size_t)(x - s.minX1) * s.D];
            for (int d = s.minD; d < s.maxD; d++) {
                int xr = x - d;
                int v = b[xr], v0 = v0b[xr], v1 = v1b[xr];
                int c0 = std::max(0, u - v1); c0 = std::max(c0, v0 - u);
                int c1 = std::max(0, v - u1); c1 = std::max(c1, u0 - v);
                cost[d - s.minD] = (int16_t)(cost[d - s.minD] + (std::min(c0, c1) >> shift));
            }
        }
    }
}

/* horizontal window sum of one pixel-cost row, clamped at the cost-column ends */
void hsum_row(const Sgbm& s, const std::vector<int16_t>& pix, std::vector<int16_t>& hs)
{
    const int D = s.D, w1 = s.width1, SW2 = s.SW2;
    hs.assign((size_t)w1 * D, 0);
    for (int d = 0; d < D; d++) {
        int acc = pix[d] * (SW2 + 1);
        for (int x = 1; x <= SW2; x++) acc += pix[(size_t)std::min(x, w1 - 1) * D + d];
        hs[d] = (int16_t)acc;
        for (int x = 1; x < w1; x++) {
            acc += pix[(size_t)std::min(x + SW2, w1 - 1) * D + d] - pix[(size_t)std::max(x - SW2 - 1, 0) * D + d];
            hs[(size_t)x * D + d] = (int16_t)acc;
        }
    }
}

/* Winner-take-all over S[0..D). lanes==0: scalar rule (first minimum).  lanes==8/16: the rule of the
 * vectorised loop: per SIMD lane the LAST d holding that lane's minimum, then over lanes holding the
 * global minimum the SMALLEST such d. */
inline void wta_pick(const int16_t* S, int D, int lanes, int* best_d, int* min_cost)
{
    if (lanes <= 0) {
        int mc = SHRT_MAX, bd = 0;
        for (int d = 0; d < D; d++) if (S[d] < mc) { mc = S[d]; bd = d; }
        *best_d = bd; *min_cost = mc; return;
    }
    int mc = SHRT_MAX;
    for (int d = 0; d < D; d++) mc = std::min(mc, (int)S[d]);
    int best = INT_MAX;
    for (int l = 0; l < lanes; l++) {
        int lane_min = SHRT_MAX, pos = 0;
        for (int d = l; d < D; d += lanes) if (S[d] <= lane_min) { lane_min = S[d]; pos = d; }
        if (lane_min == mc) best = std::min(best, pos);
    }
    *best_d = best; *min_cost = mc;
}

/* One path step: L[d] = C[d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, minLp+P2) - (minLp+P2); returns new min */
inline int path_step(const int16_t* C, const int16_t* Lp, int minLp, int D, int P1, int P2, int16_t* L)
{
    int mP2 = minLp + P2, newMin = SHRT_MAX;
    for (int d = 0; d < D; d++) {
        int lm = d > 0 ? Lp[d - 1] : SHRT_MAX, lp = d < D - 1 ? Lp[d + 1] : SHRT_MAX;
        int v = C[d] + std::min(std::min(lm + P1, lp + P1), std::min((int)Lp[d], mP2)) - mP2;
        L[d] = sat16(v);
        newMin = std::min(newMin, (int)L[d]);
    }
    return newMin;
}

void run_stripe(const Sgbm& s, int n, int stripe_sz, int overlap, int16_t* disp, int16_t* cost_probe, int probe_y)
{
    const int W = s.W, H = s.H, D = s.D, w1 = s.width1, SH2 = s.SH2;
    const int INVALID = (s.minD - 1) * 16;
    const int y_begin = std::max(std::min(n * stripe_sz - overlap, H), 0);
    const int y_end = std::min((n + 1) * stripe_sz, H);
    const int out_begin = n * stripe_sz;
    const int ring_n = 2 * SH2 + 2;
    std::vector<std::vector<int16_t>> ring(ring_n);
    std::vector<int16_t> pix, C((size_t)w1 * D, 0);
    std::vector<uint8_t> t1, t2, t3, t4;
    /* path state: zero cost, zero minimum before the first pixel of every path (path_init) */
    std::vector<int16_t> top((size_t)w1 * D, 0), left((size_t)(w1 + 1) * D, 0), right(D), tmp(D), Ssum((size_t)w1 * D);
    const int startMin = (g_variant & ORC_VAR_PATH_START_C) ? -s.P2 : 0; /* path_init: the minimum "before" a path's first pixel */
    std::vector<int> topMin(w1, startMin);
    std::vector<int16_t> disp2(W), disp2cost(W), row(W);

    auto hs_of = [&](int k) -> std::vector<int16_t>& { return ring[std::min(k, H - 1) % ring_n]; };

    for (int y = y_begin; y < y_end; y++) {
        /* ---- aggregated block cost C(y) ---- */
        if (y == y_begin) {
            for (int k = y_begin; k <= y_begin + SH2; k++) {
                if (k < H) { pixel_cost_row(s, k, pix, t1, t2, t3, t4); hsum_row(s, pix, ring[k % ring_n]); }
                const std::vector<int16_t>& hs = hs_of(k);
                int scale = k == y_begin ? SH2 + 1 : 1;
                for (size_t i = 0; i < C.size(); i++) C[i] = (int16_t)(C[i] + hs[i] * scale);
            }
        } else {
            int k = y + SH2;
            if (k < H) { pixel_cost_row(s, k, pix, t1, t2, t3, t4); hsum_row(s, pix, ring[k % ring_n]); }
            const std::vector<int16_t>& add = hs_of(k);
            const std::vector<int16_t>& sub = ring[std::max(y - SH2 - 1, y_begin) % ring_n];
            for (size_t i = 0; i < C.size(); i++) C[i] = (int16_t)(C[i] + add[i] - sub[i]);
        }
        if (cost_probe && y == probe_y) memcpy(cost_probe, C.data(), C.size() * sizeof(int16_t));

        for (int x = 0; x < W; x++) { row[x] = (int16_t)INVALID; disp2[x] = (int16_t)INVALID; disp2cost[x] = SHRT_MAX; }

        /* ---- left->right and top->bottom paths ---- */
        int leftMin = startMin;
        std::fill(left.begin(), left.begin() + D, (int16_t)0);
        for (int x = 0; x < w1; x++) {
            const int16_t* Cx = &C[(size_t)x * D];
            leftMin = path_step(Cx, &left[(size_t)x * D], leftMin, D, s.P1, s.P2, &left[(size_t)(x + 1) * D]);
            memcpy(tmp.data(), &top[(size_t)x * D], D * sizeof(int16_t));
            topMin[x] = path_step(Cx, tmp.data(), topMin[x], D, s.P1, s.P2, &top[(size_t)x * D]);
        }
        /* ---- right->left path, sum, WTA, disp2, sub-pixel ---- */
        int rightMin = startMin;
        std::fill(right.begin(), right.end(), (int16_t)0);
        for (int x = w1 - 1; x >= 0; x--) {
            const int16_t* Cx = &C[(size_t)x * D];
            memcpy(tmp.data(), right.data(), D * sizeof(int16_t));
            rightMin = path_step(Cx, tmp.data(), rightMin, D, s.P1, s.P2, right.data());
            int16_t* S = &Ssum[(size_t)x * D];
            const int16_t* Lx = &left[(size_t)(x + 1) * D]; const int16_t* Tx = &top[(size_t)x * D];
            for (int d = 0; d < D; d++) S[d] = sat16((int)Lx[d] + right[d] + Tx[d]);
            int bestd, minS;
            wta_pick(S, D, s.lanes, &bestd, &minS);
            if (s.uniq > 0) {
                int d;
                for (d = 0; d < D; d++)
                    if (S[d] * (100 - s.uniq) < minS * 100 && std::abs(d - bestd) > 1) break;
                if (d < D) continue;
            }
            int ximg = x + s.minX1;
            int x2 = ximg - bestd - s.minD;
            if (disp2cost[x2] > minS) { disp2cost[x2] = (int16_t)minS; disp2[x2] = (int16_t)(bestd + s.minD); }
            int d16;
            if (0 < bestd && bestd < D - 1) {
                int denom2 = std::max(S[bestd - 1] + S[bestd + 1] - 2 * S[bestd], 1);
                d16 = bestd * 16 + ((S[bestd - 1] - S[bestd + 1]) * 16 + denom2) / (denom2 * 2);
            } else d16 = bestd * 16;
            row[ximg] = (int16_t)(d16 + s.minD * 16);
        }
        /* ---- left-right consistency ---- */
        for (int x = s.minX1; x < s.maxX1; x++) {
            int d1 = row[x];
            if (d1 == INVALID) continue;
            int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
            int _x = x - _d, x_ = x - d_;
            if (0 <= _x && _x < W && disp2[_x] >= s.minD && std::abs(disp2[_x] - _d) > s.disp12MaxDiff &&
                0 <= x_ && x_ < W && disp2[x_] >= s.minD && std::abs(disp2[x_] - d_) > s.disp12MaxDiff)
                row[x] = (int16_t)INVALID;
        }
        if (y >= out_begin && disp) memcpy(disp + (size_t)y * W, row.data(), W * sizeof(int16_t));
    }
}

bool make_sgbm(const uint8_t* left, const uint8_t* right, const orc_params* p, Sgbm* s)
{
    s->W = p->width; s->H = p->height; s->L = left; s->R = right;
    s->minD = p->min_disparity; s->maxD = s->minD + p->num_disparities; s->D = p->num_disparities;
    s->minX1 = std::max(s->maxD, 0); s->maxX1 = s->W + std::min(s->minD, 0); s->width1 = s->maxX1 - s->minX1;
    s->SW2 = s->SH2 = p->block_size > 0 ? p->block_size / 2 : 1;
    s->P1 = p->p1 > 0 ? p->p1 : 2; s->P2 = std::max(p->p2 > 0 ? p->p2 : 5, s->P1 + 1);
    s->uniq = p->uniqueness_ratio >= 0 ? p->uniqueness_ratio : 10;
    s->disp12MaxDiff = p->disp12_max_diff > 0 ? p->disp12_max_diff : ((g_variant & ORC_VAR_DISP12_ZERO) ? 0 : 1);
    s->ftzero = std::max(p->pre_filter_cap, 15) | 1;
    s->lanes = p->wta_simd_lanes;
    return s->width1 > 0 && s->D > 0 && s->D % 16 == 0 && s->minD == 0;
}

void median3x3_s16(const int16_t* src, int W, int H, int16_t* dst)
{
    for (int y = 0; y < H; y++) {
        int y0 = std::max(y - 1, 0), y2 = std::min(y + 1, H - 1);
        for (int x = 0; x < W; x++) {
            int x0 = std::max(x - 1, 0), x2 = std::min(x + 1, W - 1);
            int16_t v[9] = {src[(size_t)y0 * W + x0], src[(size_t)y0 * W + x], src[(size_t)y0 * W + x2],
                            src[(size_t)y * W + x0],  src[(size_t)y * W + x],  src[(size_t)y * W + x2],
                            src[(size_t)y2 * W + x0], src[(size_t)y2 * W + x], src[(size_t)y2 * W + x2]};
            std::nth_element(v, v + 4, v + 9);
            dst[(size_t)y * W + x] = v[4];
        }
    }
}

} // namespace

extern "C" void orc_set_variant(unsigned flags) { g_variant = flags; }
extern "C" unsigned orc_get_variant(void) { return g_variant; }
extern "C" void orc_set_threads(int n) { g_threads = std::max(1, n); }
extern "C" int orc_get_threads(void) { return g_threads; }

extern "C" int orc_sgbm3way(const uint8_t* left, const uint8_t* right, const orc_params* p,
                            int16_t* disp16_out, int16_t* premedian_out)
{
    Sgbm s;
    if (!make_sgbm(left, right, p, &s)) return -1;
    int stripe_sz, overlap;
    stripe_layout(s.H, p->block_size, &stripe_sz, &overlap);
    std::vector<int16_t> pre((size_t)s.W * s.H, (int16_t)((s.minD - 1) * 16));
    /* stripes are independent by construction (fresh path state per stripe) */
#pragma omp parallel for num_threads(std::min(g_threads, 4)) schedule(static, 1)
    for (int n = 0; n < 4; n++)
        run_stripe(s, n, stripe_sz, overlap, pre.data(), nullptr, -1);
    if (premedian_out) memcpy(premedian_out, pre.data(), pre.size() * sizeof(int16_t));
    median3x3_s16(pre.data(), s.W, s.H, disp16_out);
    return 0;
}

extern "C" int orc_sgbm_cost_row(const uint8_t* left, const uint8_t* right, const orc_params* p,
                                 int stripe, int y, int16_t* cost_out)
{
    Sgbm s;
    if (!make_sgbm(left, right, p, &s)) return -1;
    int stripe_sz, overlap;
    stripe_layout(s.H, p->block_size, &stripe_sz, &overlap);
    run_stripe(s, stripe, stripe_sz, overlap, nullptr, cost_out, y);
    return 0;
}

extern "C" void orc_disp_to_float(const int16_t* disp16, int n, float* out)
{
    for (int i = 0; i < n; i++) out[i] = (float)disp16[i] * (1.0f / 16.0f);
}

/* VOClass::computeDepthMap, reference Source/VOClass.cpp:312-366 */
extern "C" void orc_depth_map(float* disp, int width, int height, double fx, double baseline, double* depth_out)
{
    for (int i = 0; i < width * height; i++) {
        float d = disp[i];
        if (d == 0.0 || d == -1.0) disp[i] = 0.1; /* double literal narrowed to float, as in the reference */
    }
    for (int i = 0; i < width * height; i++) {
        float d = disp[i];
        depth_out[i] = fx * baseline / d;
    }
}
