// Minimal solves of the RANSAC rounds on the GPU (vo_params.pnp_device = 1): one thread per (frame, hypothesis) runs the
// 5-point EPnP solve of cv::solvePnPRansac (reference call site Source/VOClass.cpp:531) on the correspondences that
// triangulate_kernel left in device memory, picked by the subset indices the host drew from OpenCV's generator.
// The same source (epnp5.h) compiled for the host is pnp_host::epnp's arithmetic; here it runs in f64 without contraction,
// so a hypothesis differs from the host's only through the device's hypot / acos / sin / cos.
#include "vo_ctx.h"
#include "epnp5.h"

namespace {

struct SolveArgs { double K[9]; int max_features, chunk, total; };

__global__ __launch_bounds__(64) void epnp5_kernel(SolveArgs a, const vo_point3f* __restrict__ obj, const vo_point2f* __restrict__ img,
                                                   const int* __restrict__ idx, const int* __restrict__ nh,
                                                   double* __restrict__ hyp, double* __restrict__ rt6)
{
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= a.total) return;
    const int b = t / a.chunk;
    if (nh[b] <= 0) return; // a frame that has stopped iterating: its slots are not read
    const size_t F = a.max_features;
    vo_point3f so[5]; vo_point2f si[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int i = idx[(size_t)t * 5 + j];
        so[j] = obj[b * F + i]; si[j] = img[b * F + i];
    }
    double rvec[3], tvec[3], R[9];
    epnp5::solve(so, si, a.K, rvec, tvec);
    epnp5::rodrigues_to_matrix(rvec, R);
    double* dst = hyp + (size_t)t * 12;
#pragma unroll
    for (int k = 0; k < 9; k++) dst[k] = R[k];
    dst[9] = tvec[0]; dst[10] = tvec[1]; dst[11] = tvec[2];
    double* o = rt6 + (size_t)t * 6;
    o[0] = rvec[0]; o[1] = rvec[1]; o[2] = rvec[2]; o[3] = tvec[0]; o[4] = tvec[1]; o[5] = tvec[2];
}

} // namespace

// idx: [n][chunk][5] subset indices, nh: [n] hypotheses of this round per frame (0: frame idle); both in device memory
int launch_pnp_solve(vo_ctx* c, int n)
{
    SolveArgs a;
    const double K[9] = {c->P0[0], c->P0[1], c->P0[2], c->P0[4], c->P0[5], c->P0[6], c->P0[8], c->P0[9], c->P0[10]};
    for (int k = 0; k < 9; k++) a.K[k] = K[k];
    a.max_features = c->p.max_features; a.chunk = c->p.ransac_chunk; a.total = n * c->p.ransac_chunk;
    StageTimer st_(c, VO_ST_PNP_SOLVE);
    hipLaunchKernelGGL(epnp5_kernel, dim3((a.total + 63) / 64), dim3(64), 0, c->stream, a, c->obj, c->obj_img, c->hyp_idx, c->hyp_idx + (size_t)c->p.max_batch * c->p.ransac_chunk * 5,
                       c->hyp, c->hyp_rt);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}
