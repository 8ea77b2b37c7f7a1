// Disparity-to-3D triangulation and batched PnP hypothesis scoring for gfx950.
// Replaces the per-feature loop of VOClass::estimateMotion (reference Source/VOClass.cpp:409-436) and the
// computeError/findInliers step that cv::solvePnPRansac (Source/VOClass.cpp:531) runs once per RANSAC iteration.
//
//  triangulate_kernel : one block per frame; depth look-up at the truncated feature position (from the f64
//                       depth map when it exists, else from the f32 disparity with the same 0 / -1 -> 0.1f
//                       patch), depth filter, back-projection in f64 narrowed to f32, ordered compaction.
//  score_kernel       : thread per (hypothesis, correspondence): f64 projection, f32 reprojection error,
//                       inlier ballot; one atomic add per wavefront.
#include "vo_ctx.h"
#include <string.h>
#include <algorithm>

namespace {

struct TriArgs { double fx, fy, cx, cy, fxb; float thresh; int W, H, max_features, use_depth; };

__global__ __launch_bounds__(256) void triangulate_kernel(TriArgs a, const vo_point2f* __restrict__ m1, const vo_point2f* __restrict__ m2,
                                                          const int* __restrict__ n_match, const float* __restrict__ disp32,
                                                          const double* __restrict__ depth, vo_point3f* __restrict__ obj,
                                                          vo_point2f* __restrict__ obj_img, int* __restrict__ n_obj)
{
    __shared__ int wcnt[4];
    const int b = blockIdx.x, n = n_match[b];
    const size_t F = a.max_features, px = (size_t)a.W * a.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + threadIdx.x;
        bool keep = false;
        vo_point3f o = {0.f, 0.f, 0.f};
        vo_point2f q = {0.f, 0.f};
        if (i < n) {
            const vo_point2f p = m1[b * F + i];
            const int iu = (int)p.x, iv = (int)p.y;
            if (iu >= 0 && iu < a.W && iv >= 0 && iv < a.H) {
                double dz;
                if (a.use_depth) dz = depth[b * px + (size_t)iv * a.W + iu];
                else {
                    float d = disp32[b * px + (size_t)iv * a.W + iu];
                    if (d == 0.0f || d == -1.0f) d = 0.1f;
                    dz = a.fxb / (double)d;
                }
                const float z = (float)dz;
                if (!(z > a.thresh)) {
                    keep = true;
                    o.x = (float)((double)z * ((double)p.x - a.cx) / a.fx);
                    o.y = (float)((double)z * ((double)p.y - a.cy) / a.fy);
                    o.z = z;
                    q = m2[b * F + i];
                }
            }
        }
        unsigned long long mk = __ballot(keep);
        if (lane == 0) wcnt[wave] = __popcll(mk);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wcnt[w];
        const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (keep) {
            int r = base + before + __popcll(mk & ((1ull << lane) - 1ull));
            obj[b * F + r] = o;
            obj_img[b * F + r] = q;
        }
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_obj[b] = base;
}

struct ScoreArgs { double fx, fy, cx, cy; float thresh2; int max_features, chunk, b0, want_masks; const int* sel; };

__global__ __launch_bounds__(256) void score_kernel(ScoreArgs a, const vo_point3f* __restrict__ obj, const vo_point2f* __restrict__ img,
                                                    const int* __restrict__ n_obj, const double* __restrict__ hyp,
                                                    int* __restrict__ counts, uint8_t* __restrict__ masks)
{
    // One block row (blockIdx.y) per hypothesis; its blocks stride over the frame's correspondences, so the hypothesis is read
    // once per block and the count leaves with one store (a hypothesis owned by one block) or one atomic add per block.
    // A wavefront-level atomic per 64 correspondences, as a thread per (hypothesis, point) would issue, is what bounds a round
    // of 1 024 hypotheses: device-wide atomics are resolved in memory, not in an XCD's L2.
    __shared__ int wsum[4];
    const int b = a.b0 + blockIdx.z;
    int h = blockIdx.y;
    if (a.sel) { // mask pass: only the hypothesis the host selected for this frame, its count is known already
        h = a.sel[b];
        if (h < 0) return;
    }
    const int n = n_obj[b];
    const size_t F = a.max_features;
    const double* R = hyp + ((size_t)b * a.chunk + h) * 12;
    const double r0 = R[0], r1 = R[1], r2 = R[2], r3 = R[3], r4 = R[4], r5 = R[5], r6 = R[6], r7 = R[7], r8 = R[8];
    const double t0 = R[9], t1 = R[10], t2 = R[11];
    int cnt = 0;
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const int i = i0 + threadIdx.x;
        bool in = false;
        if (i < n) {
            const vo_point3f P = obj[b * F + i];
            const vo_point2f m = img[b * F + i];
            const double X = P.x, Y = P.y, Z = P.z;
            double x = r0 * X + r1 * Y + r2 * Z + t0;
            double y = r3 * X + r4 * Y + r5 * Z + t1;
            double z = r6 * X + r7 * Y + r8 * Z + t2;
            z = z != 0.0 ? 1. / z : 1;
            x *= z; y *= z;
            const float pu = (float)(x * a.fx + a.cx), pv = (float)(y * a.fy + a.cy);
            const float dx = m.x - pu, dy = m.y - pv;
            const float e = (float)((double)dx * (double)dx + (double)dy * (double)dy);
            in = e <= a.thresh2;
            if (a.want_masks) masks[((size_t)b * a.chunk + h) * F + i] = (uint8_t)in;
        }
        cnt += __popcll(__ballot(in));
    }
    if (a.sel) return;
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        int* dst = &counts[(size_t)b * a.chunk + h];
        if (gridDim.x == 1) *dst = total;
        else if (total) atomicAdd(dst, total);
    }
}

} // namespace

// blocks per hypothesis: one when the hypotheses alone fill the GPU, else enough to put about 2048 blocks in flight
static int score_blocks_x(int kmax, int hypotheses)
{
    const int most = (kmax + 255) / 256;
    return std::max(1, std::min(most, (2048 + hypotheses - 1) / std::max(hypotheses, 1)));
}

int launch_triangulate(vo_ctx* c)
{
    TriArgs a;
    a.fx = c->P0[0]; a.fy = c->P0[5]; a.cx = c->P0[2]; a.cy = c->P0[6];
    a.fxb = a.fx * (c->P1m[3] / (-1 * a.fx));
    a.thresh = (float)c->p.depth_thresh;
    a.W = c->W; a.H = c->H; a.max_features = c->p.max_features; a.use_depth = c->have_depth ? 1 : 0;
    StageTimer st_(c, VO_ST_TRIANGULATE);
    hipLaunchKernelGGL(triangulate_kernel, dim3(c->n), dim3(256), 0, c->stream, a, c->m1, c->m2, c->n_match, c->disp32, c->depth,
                       c->obj, c->obj_img, c->n_obj);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

int launch_pnp_score(vo_ctx* c, int b0, int nb, int H, float thresh2, bool want_masks)
{
    int kmax = 0;
    for (int b = b0; b < b0 + nb; b++) kmax = std::max(kmax, c->h_nobj[b]);
    const int chunk = c->p.ransac_chunk;
    VO_HIP(c, hipMemsetAsync(c->hyp_count + (size_t)b0 * chunk, 0, (size_t)nb * chunk * sizeof(int), c->stream));
    if (kmax == 0) return VO_OK;
    ScoreArgs a;
    a.fx = c->P0[0]; a.fy = c->P0[5]; a.cx = c->P0[2]; a.cy = c->P0[6];
    a.thresh2 = thresh2; a.max_features = c->p.max_features; a.chunk = chunk; a.b0 = b0; a.want_masks = want_masks ? 1 : 0; a.sel = nullptr;
    StageTimer st_(c, VO_ST_PNP_SCORE);
    hipLaunchKernelGGL(score_kernel, dim3(score_blocks_x(kmax, H * nb), H, nb), dim3(256), 0, c->stream, a, c->obj, c->obj_img, c->n_obj, c->hyp,
                       c->hyp_count, c->hyp_mask);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

// Inlier masks of ONE hypothesis per frame (c->hyp_sel[b], negative: none) of the hypotheses scored last: what a round needs
// when the masks of all its hypotheses would not fit the host staging buffer, so the scoring pass wrote none.
int launch_pnp_masks(vo_ctx* c, int nb, float thresh2)
{
    int kmax = 0;
    for (int b = 0; b < nb; b++) kmax = std::max(kmax, c->h_nobj[b]);
    if (kmax == 0) return VO_OK;
    ScoreArgs a;
    a.fx = c->P0[0]; a.fy = c->P0[5]; a.cx = c->P0[2]; a.cy = c->P0[6];
    a.thresh2 = thresh2; a.max_features = c->p.max_features; a.chunk = c->p.ransac_chunk; a.b0 = 0; a.want_masks = 1; a.sel = c->hyp_sel;
    hipLaunchKernelGGL(score_kernel, dim3(score_blocks_x(kmax, nb), 1, nb), dim3(256), 0, c->stream, a, c->obj, c->obj_img, c->n_obj, c->hyp,
                       c->hyp_count, c->hyp_mask);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}
