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

struct ScoreArgs { double fx, fy, cx, cy; float thresh2; int max_features, chunk, b0, want_masks; };

__global__ __launch_bounds__(256) void score_kernel(ScoreArgs a, const vo_point3f* __restrict__ obj, const vo_point2f* __restrict__ img,
                                                    const int* __restrict__ n_obj, const double* __restrict__ hyp,
                                                    int* __restrict__ counts, uint8_t* __restrict__ masks)
{
    const int b = a.b0 + blockIdx.z, h = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = n_obj[b];
    const size_t F = a.max_features;
    const double* R = hyp + ((size_t)b * a.chunk + h) * 12;
    bool in = false;
    if (i < n) {
        const vo_point3f P = obj[b * F + i];
        const vo_point2f m = img[b * F + i];
        const double X = P.x, Y = P.y, Z = P.z;
        double x = R[0] * X + R[1] * Y + R[2] * Z + R[9];
        double y = R[3] * X + R[4] * Y + R[5] * Z + R[10];
        double z = R[6] * X + R[7] * Y + R[8] * Z + R[11];
        z = z != 0.0 ? 1. / z : 1;
        x *= z; y *= z;
        const float pu = (float)(x * a.fx + a.cx), pv = (float)(y * a.fy + a.cy);
        const float dx = m.x - pu, dy = m.y - pv;
        const float e = (float)((double)dx * (double)dx + (double)dy * (double)dy);
        in = e <= a.thresh2;
        if (a.want_masks) masks[((size_t)b * a.chunk + h) * F + i] = (uint8_t)in;
    }
    unsigned long long mk = __ballot(in);
    if ((threadIdx.x & 63) == 0 && mk) atomicAdd(&counts[(size_t)b * a.chunk + h], __popcll(mk));
}

} // namespace

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
    a.thresh2 = thresh2; a.max_features = c->p.max_features; a.chunk = chunk; a.b0 = b0; a.want_masks = want_masks ? 1 : 0;
    StageTimer st_(c, VO_ST_PNP_SCORE);
    hipLaunchKernelGGL(score_kernel, dim3((kmax + 255) / 256, H, nb), dim3(256), 0, c->stream, a, c->obj, c->obj_img, c->n_obj, c->hyp,
                       c->hyp_count, c->hyp_mask);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}
