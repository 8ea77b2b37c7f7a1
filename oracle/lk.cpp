/* CPU ORACLE (test infrastructure only) — pyramidal Lucas–Kanade + circular matching.
 * Restates cv::calcOpticalFlowPyrLK as called at reference Source/VOClassFeatures.cpp:104-111
 * (OpenCV 4.5.4 video/src/lkpyramid.cpp: buildOpticalFlowPyramid, calcSharrDeriv,
 * LKTrackerInvoker; imgproc pyrDown) and the
 * reference's own filtering at Source/VOClassFeatures.cpp:134-201, Source/VOClassUtils.cpp:61-91.
 * Parity unpinned (see vo_oracle.h).
 *
 * Accumulators of the tracker (orc_params.lk_accum).  lkpyramid.cpp declares them as
 *     #if defined __arm__ && !CV_NEON   typedef int64 acctype; typedef int itemtype;
 *     #else                             typedef float acctype; typedef float itemtype;
 * so every mainstream build (x86, and arm64 with NEON, which the reference's author used) sums the structure tensor and the
 * mismatch vector in FLOAT, part of it in 4-lane vector partial sums whose order depends on the instruction set.  Integer
 * sums are what a 32-bit ARM build without NEON computes.  Modes:
 *     0  exact integer sums (order free; what the GPU library computes)
 *     1  float, plain scan order (no vector partial sums)
 *     4  float in the order of the NEON block of lkpyramid.cpp (restated from knowledge of that file, medium confidence):
 *        tensor: 4 lanes step x by 4 (x <= win - 4) over all window rows, each lane adding the float of its exact integer
 *        product; the rest of every row goes to the scalar accumulator; at the end acc += ((n0 + n1) + n2) + n3.
 *        mismatch: 4 lanes step x by 8 (x <= win - 8), lane j adds the float of the INTEGER sum of its two products (pixels
 *        x + j and x + j + 4); rest of the row scalar; at the end acc += (float)(((n0 + n1) + n2) + n3).
 *     8  float in the order of the 128-bit universal-intrinsic block (x86 SSE2; low confidence on the lane permutation):
 *        tensor: 4 lanes step x by 8, lane j adds pixel x + j then pixel x + 4 + j; rest scalar; acc += (q0 + q2) + (q1 + q3).
 *        mismatch: two 4-lane vectors step x by 8; the lanes that feed b1 hold float(int(p[x] + p[x+4])), (p[x+1] + p[x+5]) in one
 *        vector and (p[x+2] + p[x+6]), (p[x+3] + p[x+7]) in the other; acc += (v0.a + v1.a) + (v0.b + v1.b).
 */
#include "vo_oracle.h"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cfloat>

extern "C" int orc_get_threads(void);
namespace {

inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* image level with a REFLECT_101 border of `b` pixels (buildOpticalFlowPyramid keeps winSize) */
struct Level {
    int w, h, b, stride;
    std::vector<uint8_t> data;       /* (h+2b) x (w+2b) */
    std::vector<int16_t> deriv;      /* (h+2b) x (w+2b) x 2, zero border */
    const uint8_t* at(int y, int x) const { return &data[(size_t)(y + b) * stride + (x + b)]; }
    const int16_t* dat(int y, int x) const { return &deriv[((size_t)(y + b) * stride + (x + b)) * 2]; }
};

void make_bordered(const uint8_t* src, int w, int h, int b, Level* L)
{
    L->w = w; L->h = h; L->b = b; L->stride = w + 2 * b;
    L->data.resize((size_t)(h + 2 * b) * L->stride);
    for (int y = -b; y < h + b; y++) {
        int sy = reflect101(y, h);
        for (int x = -b; x < w + b; x++)
            L->data[(size_t)(y + b) * L->stride + (x + b)] = src[(size_t)sy * w + reflect101(x, w)];
    }
}

inline int cv_floor(float v) { int i = (int)v; return i - (v < (float)i); }
inline int cv_round(float v) { return (int)lrintf(v); } /* round-half-even under the default FP mode */
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

} // namespace

extern "C" void orc_pyr_level_size(int width, int height, int level, int* w, int* h)
{
    for (int l = 0; l < level; l++) { width = (width + 1) / 2; height = (height + 1) / 2; }
    *w = width; *h = height;
}

/* buildOpticalFlowPyramid stops when the NEXT level would be <= winSize in either dimension */
extern "C" int orc_pyr_levels(int width, int height, int win, int max_level)
{
    int w = width, h = height;
    for (int level = 0; level <= max_level; level++) {
        w = (w + 1) / 2; h = (h + 1) / 2;
        if (w <= win || h <= win) return level;
    }
    return max_level;
}

extern "C" void orc_pyrdown(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh)
{
    std::vector<int> rowbuf((size_t)sh * dw);
    for (int y = 0; y < sh; y++)
        for (int x = 0; x < dw; x++) {
            const uint8_t* r = src + (size_t)y * sw;
            int s = r[reflect101(2 * x - 2, sw)] + 4 * r[reflect101(2 * x - 1, sw)] + 6 * r[reflect101(2 * x, sw)] +
                    4 * r[reflect101(2 * x + 1, sw)] + r[reflect101(2 * x + 2, sw)];
            rowbuf[(size_t)y * dw + x] = s;
        }
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int s = rowbuf[(size_t)reflect101(2 * y - 2, sh) * dw + x] + 4 * rowbuf[(size_t)reflect101(2 * y - 1, sh) * dw + x] +
                    6 * rowbuf[(size_t)reflect101(2 * y, sh) * dw + x] + 4 * rowbuf[(size_t)reflect101(2 * y + 1, sh) * dw + x] +
                    rowbuf[(size_t)reflect101(2 * y + 2, sh) * dw + x];
            dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
        }
}

extern "C" void orc_scharr(const uint8_t* src, int w, int h, int16_t* dst)
{
    std::vector<int> t0(w + 2), t1(w + 2);
    for (int y = 0; y < h; y++) {
        const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * w;
        const uint8_t* r1 = src + (size_t)y * w;
        const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * w;
        for (int x = 0; x < w; x++) {
            t0[x + 1] = (r0[x] + r2[x]) * 3 + r1[x] * 10;
            t1[x + 1] = r2[x] - r0[x];
        }
        int x0 = w > 1 ? 1 : 0, x1 = w > 1 ? w - 2 : 0;
        t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
        t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
        for (int x = 0; x < w; x++) {
            dst[((size_t)y * w + x) * 2] = (int16_t)(t0[x + 2] - t0[x]);
            dst[((size_t)y * w + x) * 2 + 1] = (int16_t)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
        }
    }
}

extern "C" int orc_get_threads(void);
namespace {

void build_pyramid(const uint8_t* img, int w, int h, int win, int levels, bool with_deriv, std::vector<Level>& pyr)
{
    pyr.resize(levels + 1);
    std::vector<uint8_t> cur(img, img + (size_t)w * h), nxt;
    int cw = w, ch = h;
    for (int l = 0; l <= levels; l++) {
        if (l > 0) {
            int nw = (cw + 1) / 2, nh = (ch + 1) / 2;
            nxt.resize((size_t)nw * nh);
            orc_pyrdown(cur.data(), cw, ch, nxt.data(), nw, nh);
            cur.swap(nxt); cw = nw; ch = nh;
        }
        make_bordered(cur.data(), cw, ch, win, &pyr[l]);
        if (with_deriv) {
            std::vector<int16_t> d((size_t)cw * ch * 2);
            orc_scharr(cur.data(), cw, ch, d.data());
            pyr[l].deriv.assign((size_t)(ch + 2 * win) * pyr[l].stride * 2, 0);
            for (int y = 0; y < ch; y++)
                memcpy(&pyr[l].deriv[((size_t)(y + win) * pyr[l].stride + win) * 2], &d[(size_t)y * cw * 2], (size_t)cw * 2 * sizeof(int16_t));
        }
    }
}

/* statistics for tuning the GPU tracker: window setups that reached the iteration loop, and iterations */
static long long g_lk_levels = 0, g_lk_iters = 0, g_lk_hist[52];
/* optional log of iterations per (call, level, point), in call order; for tuning, single-threaded runs only */
static int* g_lk_log = nullptr;
static long long g_lk_log_pos = 0, g_lk_log_cap = 0;

/* Sum of the win x win integer products `p` (row-major) as the tracker's accumulators form it: see the modes at the top
 * of this file.  mismatch = false: a term of the structure tensor; true: a term of the mismatch vector (the two differ in
 * how the vector blocks walk a row).  Returned as float (acctype) before the 2^-20 scaling. */
float window_sum(const int* p, int win, int accum, bool mismatch)
{
    if (accum == 0) {
        int64_t s = 0;
        for (int i = 0; i < win * win; i++) s += p[i];
        return (float)s; /* acctype int64 -> float in `iA11 * FLT_SCALE` */
    }
    if (accum == 1) {
        float s = 0;
        for (int i = 0; i < win * win; i++) s += (float)p[i];
        return s;
    }
    float acc = 0.f;
    if (accum == 4) {
        float n[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < win; y++) {
            const int* r = p + (size_t)y * win;
            int x = 0;
            if (!mismatch) for (; x <= win - 4; x += 4) for (int j = 0; j < 4; j++) n[j] += (float)r[x + j];
            else for (; x <= win - 8; x += 8) for (int j = 0; j < 4; j++) n[j] += (float)(r[x + j] + r[x + j + 4]);
            for (; x < win; x++) acc += (float)r[x];
        }
        const float lanes = ((n[0] + n[1]) + n[2]) + n[3];
        acc += lanes;
        return acc;
    }
    /* accum == 8 */
    if (!mismatch) {
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        for (int y = 0; y < win; y++) {
            const int* r = p + (size_t)y * win;
            int x = 0;
            for (; x <= win - 8; x += 8) {
                for (int j = 0; j < 4; j++) q[j] += (float)r[x + j];
                for (int j = 0; j < 4; j++) q[j] += (float)r[x + 4 + j];
            }
            for (; x < win; x++) acc += (float)r[x];
        }
        acc += (q[0] + q[2]) + (q[1] + q[3]);
        return acc;
    }
    float v0a = 0.f, v0b = 0.f, v1a = 0.f, v1b = 0.f;
    for (int y = 0; y < win; y++) {
        const int* r = p + (size_t)y * win;
        int x = 0;
        for (; x <= win - 8; x += 8) {
            v0a += (float)(r[x] + r[x + 4]); v0b += (float)(r[x + 1] + r[x + 5]);
            v1a += (float)(r[x + 2] + r[x + 6]); v1b += (float)(r[x + 3] + r[x + 7]);
        }
        for (; x < win; x++) acc += (float)r[x];
    }
    acc += (v0a + v1a) + (v0b + v1b);
    return acc;
}

/* LKTrackerInvoker for one point on one level. */
void track_point(const Level& I, const Level& J, const orc_point2f& prev_pt0, orc_point2f* next_pt_io,
                 uint8_t* status, float* err, int level, int max_level, int win, int max_count, double eps2,
                 float min_eig_thr, int accum)
{
    const float half = (win - 1) * 0.5f;
    orc_point2f prevPt = {prev_pt0.x * (float)(1. / (1 << level)), prev_pt0.y * (float)(1. / (1 << level))};
    orc_point2f nextPt;
    if (level == max_level) nextPt = prevPt;
    else { nextPt.x = next_pt_io->x * 2.f; nextPt.y = next_pt_io->y * 2.f; }
    *next_pt_io = nextPt;

    prevPt.x -= half; prevPt.y -= half;
    int ipx = cv_floor(prevPt.x), ipy = cv_floor(prevPt.y);
    if (ipx < -win || ipx >= I.w || ipy < -win || ipy >= I.h) {
        if (level == 0) { *status = 0; if (err) *err = 0; }
        return;
    }
    float a = prevPt.x - ipx, b = prevPt.y - ipy;
    const int W_BITS = 14;
    const float FLT_SCALE = 1.f / (1 << 20);
    int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    std::vector<int16_t> Iw((size_t)win * win), dIw((size_t)win * win * 2);
    const int stepI = I.stride, dstep = I.stride * 2;
    for (int y = 0; y < win; y++) {
        const uint8_t* src = I.at(y + ipy, ipx);
        const int16_t* dsrc = I.dat(y + ipy, ipx);
        for (int x = 0; x < win; x++, dsrc += 2) {
            int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, W_BITS - 5);
            int ixval = descale(dsrc[0] * iw00 + dsrc[2] * iw01 + dsrc[dstep] * iw10 + dsrc[dstep + 2] * iw11, W_BITS);
            int iyval = descale(dsrc[1] * iw00 + dsrc[3] * iw01 + dsrc[dstep + 1] * iw10 + dsrc[dstep + 3] * iw11, W_BITS);
            Iw[(size_t)y * win + x] = (int16_t)ival;
            dIw[((size_t)y * win + x) * 2] = (int16_t)ixval;
            dIw[((size_t)y * win + x) * 2 + 1] = (int16_t)iyval;
        }
    }
    std::vector<int> prod((size_t)win * win);
    float Asum[3];
    for (int q = 0; q < 3; q++) {
        for (int i = 0; i < win * win; i++) {
            const int ix = dIw[(size_t)i * 2], iy = dIw[(size_t)i * 2 + 1];
            prod[i] = q == 0 ? ix * ix : q == 1 ? ix * iy : iy * iy;
        }
        Asum[q] = window_sum(prod.data(), win, accum, false);
    }
    float A11 = Asum[0] * FLT_SCALE, A12 = Asum[1] * FLT_SCALE, A22 = Asum[2] * FLT_SCALE;
    float Dt = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * win * win);
    if (minEig < min_eig_thr || Dt < FLT_EPSILON) {
        if (level == 0) *status = 0;
        return;
    }
    Dt = 1.f / Dt;
    std::vector<int> prod2((size_t)win * win);

    nextPt.x -= half; nextPt.y -= half;
    orc_point2f prevDelta = {0, 0};
    const int stepJ = J.stride;
    g_lk_levels++;
    int iters_run = 0;
    for (int j = 0; j < max_count; j++) {
        g_lk_iters++; iters_run++;
        int inx = cv_floor(nextPt.x), iny = cv_floor(nextPt.y);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) {
            if (level == 0) *status = 0;
            break;
        }
        a = nextPt.x - inx; b = nextPt.y - iny;
        iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        for (int y = 0; y < win; y++) {
            const uint8_t* Jp = J.at(y + iny, inx);
            for (int x = 0; x < win; x++) {
                int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) -
                           Iw[(size_t)y * win + x];
                prod[(size_t)y * win + x] = diff * dIw[((size_t)y * win + x) * 2];
                prod2[(size_t)y * win + x] = diff * dIw[((size_t)y * win + x) * 2 + 1];
            }
        }
        float b1 = window_sum(prod.data(), win, accum, true) * FLT_SCALE, b2 = window_sum(prod2.data(), win, accum, true) * FLT_SCALE;
        orc_point2f delta = {(float)((A12 * b2 - A22 * b1) * Dt), (float)((A12 * b1 - A11 * b2) * Dt)};
        nextPt.x += delta.x; nextPt.y += delta.y;
        next_pt_io->x = nextPt.x + half; next_pt_io->y = nextPt.y + half;
        if ((double)delta.x * delta.x + (double)delta.y * delta.y <= eps2) break;
        if (j > 0 && std::abs(delta.x + prevDelta.x) < 0.01 && std::abs(delta.y + prevDelta.y) < 0.01) {
            next_pt_io->x -= delta.x * 0.5f; next_pt_io->y -= delta.y * 0.5f;
            break;
        }
        prevDelta = delta;
    }
    g_lk_hist[iters_run < 51 ? iters_run : 51]++;
    /* error block of the original: runs because the caller passes `err`; its bounds test can clear status */
    if (*status && level == 0 && !(orc_get_variant() & ORC_VAR_LK_NO_ERR_STATUS)) {
        orc_point2f np = {next_pt_io->x - half, next_pt_io->y - half};
        int inx = cv_floor(np.x), iny = cv_floor(np.y);
        if (inx < -win || inx >= J.w || iny < -win || iny >= J.h) { *status = 0; return; }
        if (err) {
            float aa = np.x - inx, bb = np.y - iny;
            iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
            iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
            iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
            iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
            float errval = 0.f;
            for (int y = 0; y < win; y++) {
                const uint8_t* Jp = J.at(y + iny, inx);
                for (int x = 0; x < win; x++) {
                    int diff = descale(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, W_BITS - 5) -
                               Iw[(size_t)y * win + x];
                    errval += std::abs((float)diff);
                }
            }
            *err = errval * 1.f / (32 * win * win);
        }
    }
}

struct Pyr { std::vector<Level> lv; int levels; };

void lk_with_pyramids(const Pyr& P, const Pyr& N, const orc_point2f* prev_pts, int n, orc_point2f* next_pts,
                      uint8_t* status, float* err, const orc_params* p)
{
    int max_count = std::min(std::max(p->lk_max_count, 0), 100);
    double eps = std::min(std::max(p->lk_epsilon, 0.), 10.);
    double eps2 = eps * eps;
    int levels = std::min(P.levels, N.levels);
    for (int i = 0; i < n; i++) { status[i] = 1; if (err) err[i] = 0; }
    /* the sensitivity switch of older fixtures (float sums in scan order) maps onto mode 1 */
    const int accum = (p->lk_accum == 0 && (orc_get_variant() & ORC_VAR_LK_FLOAT_ACC)) ? 1 : p->lk_accum;

    for (int level = levels; level >= 0; level--) {
#pragma omp parallel for num_threads(orc_get_threads()) schedule(static)
        for (int i = 0; i < n; i++) {
            const long long before = g_lk_iters;
            track_point(P.lv[level], N.lv[level], prev_pts[i], &next_pts[i], &status[i], err ? &err[i] : nullptr,
                        level, levels, p->lk_win, max_count, eps2, (float)p->lk_min_eig, accum);
            if (g_lk_log && g_lk_log_pos < g_lk_log_cap) g_lk_log[g_lk_log_pos++] = (int)(g_lk_iters - before); /* single-threaded runs only */
        }
    }
}

} // namespace

extern "C" int orc_get_threads(void);

extern "C" int orc_lk(const uint8_t* prev, const uint8_t* next, int width, int height,
                      const orc_point2f* prev_pts, int n, orc_point2f* next_pts, uint8_t* status, float* err,
                      const orc_params* p)
{
    if (n == 0) return 0;
    Pyr P, N;
    P.levels = N.levels = orc_pyr_levels(width, height, p->lk_win, p->lk_max_level);
    build_pyramid(prev, width, height, p->lk_win, P.levels, true, P.lv);
    build_pyramid(next, width, height, p->lk_win, N.levels, false, N.lv);
    lk_with_pyramids(P, N, prev_pts, n, next_pts, status, err, p);
    return 0;
}

/* VOClass::matchFeatureKLT, reference Source/VOClassFeatures.cpp:54-202 */
extern "C" int orc_match_klt(const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                             const orc_params* p, orc_point2f* pts1_inout, int n, orc_point2f* pts2_out,
                             orc_point2f* legs_out, uint8_t* leg_status_out)
{
    const int W = p->width, H = p->height;
    std::vector<orc_point2f> fRT1(n), fRT2(n), fLT2(n), fLT1Re(n);
    std::vector<uint8_t> s0(n), s1(n), s2(n), s3(n);
    std::vector<float> err(n);
    orc_lk(lt1, rt1, W, H, pts1_inout, n, fRT1.data(), s0.data(), err.data(), p);
    orc_lk(rt1, rt2, W, H, fRT1.data(), n, fRT2.data(), s1.data(), err.data(), p);
    orc_lk(rt2, lt2, W, H, fRT2.data(), n, fLT2.data(), s2.data(), err.data(), p);
    orc_lk(lt2, lt1, W, H, fLT2.data(), n, fLT1Re.data(), s3.data(), err.data(), p);
    /* markInvalidFeaturesBounds / isOutOfBounds (Source/VOClassUtils.cpp:61-78): x>W, y>H are out; == is in */
    auto oob = [&](const orc_point2f& q) { return q.x < 0 || q.x > W || q.y < 0 || q.y > H; };
    for (int i = 0; i < n; i++) {
        if (oob(fRT1[i])) s0[i] = 0;
        if (oob(fRT2[i])) s1[i] = 0;
        if (oob(fLT2[i])) s2[i] = 0;
        if (oob(fLT1Re[i])) s3[i] = 0;
    }
    if (legs_out) {
        memcpy(legs_out, fRT1.data(), n * sizeof(orc_point2f));
        memcpy(legs_out + n, fRT2.data(), n * sizeof(orc_point2f));
        memcpy(legs_out + 2 * n, fLT2.data(), n * sizeof(orc_point2f));
        memcpy(legs_out + 3 * n, fLT1Re.data(), n * sizeof(orc_point2f));
    }
    if (leg_status_out) {
        memcpy(leg_status_out, s0.data(), n); memcpy(leg_status_out + n, s1.data(), n);
        memcpy(leg_status_out + 2 * n, s2.data(), n); memcpy(leg_status_out + 3 * n, s3.data(), n);
    }
    int m = 0;
    std::vector<orc_point2f> o1, o2;
    for (int i = 0; i < n; i++) {
        if (!(s0[i] == 1 && s1[i] == 1 && s2[i] == 1 && s3[i] == 1)) continue;
        /* int offset = max(|dx|, |dy|) — float narrowed to int, then compared with 2 */
        int offset = (int)std::max(std::abs(pts1_inout[i].x - fLT1Re[i].x), std::abs(pts1_inout[i].y - fLT1Re[i].y));
        if (offset < p->loop_threshold) { o1.push_back(fLT1Re[i]); o2.push_back(fLT2[i]); m++; }
    }
    for (int i = 0; i < m; i++) { pts1_inout[i] = o1[i]; pts2_out[i] = o2[i]; }
    return m;
}

extern "C" void orc_lk_stats(long long* levels, long long* iters, int reset)
{
    *levels = g_lk_levels; *iters = g_lk_iters;
    if (reset) { g_lk_levels = 0; g_lk_iters = 0; }
}
extern "C" void orc_lk_set_log(int* buf, long long capacity) { g_lk_log = buf; g_lk_log_pos = 0; g_lk_log_cap = capacity; }
extern "C" long long orc_lk_log_size(void) { return g_lk_log_pos; }
extern "C" void orc_lk_hist(long long* out52, int reset)
{
    for (int i = 0; i < 52; i++) { out52[i] = g_lk_hist[i]; if (reset) g_lk_hist[i] = 0; }
}
