// Minimal solves of the RANSAC rounds on the GPU (vo_params.pnp_device = 1): one thread per (frame, hypothesis) runs the
// 5-point EPnP solve of cv::solvePnPRansac (reference call site Source/VOClass.cpp:531) on the correspondences that
// triangulate_kernel left in device memory, picked by the subset indices the host drew from OpenCV's generator.
// The same source (epnp5.h) compiled for the host is pnp_host::epnp's arithmetic; here it runs in f64 without contraction,
// so a hypothesis differs from the host's only through the device's hypot / acos / sin / cos.
#include "vo_ctx.h"
#include "epnp5.h"
#include <atomic>

namespace {

struct SolveArgs { double K[9]; int max_features, chunk, total; };

// threads per block: a thread's two 12 x 12 Jacobi work matrices live in LDS (2 x 144 doubles: 73.7 KB per 32 threads, two
// blocks per CU); with the matrices in private memory instead a round took 1.8 instead of 1.15 ms
constexpr int EP_T = 32;
__global__ __launch_bounds__(EP_T) void epnp5_kernel(SolveArgs a, const vo_point3f* __restrict__ obj, const vo_point2f* __restrict__ img,
                                                   const int* __restrict__ idx, const int* __restrict__ nh,
                                                   double* __restrict__ hyp, double* __restrict__ rt6)
{
    const int t = blockIdx.x * EP_T + threadIdx.x;
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
    // the two 12 x 12 Jacobi matrices of the block's threads interleaved in LDS (element e of thread t at e * EP_T + t): a solve
    // is one thread's serial chain through them, and LDS answers several times faster than private (scratch) memory
    extern __shared__ double ep_lds[];
    epnp5::solve_ws<EP_T>(so, si, a.K, rvec, tvec, ep_lds + threadIdx.x, ep_lds + 144 * EP_T + threadIdx.x);
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
    const size_t lds = (size_t)2 * 144 * EP_T * sizeof(double);
    static std::atomic<int> lds_attr[64];
    if (lds > 48 * 1024 && lds_attr[c->device & 63].load(std::memory_order_acquire) < (int)lds) {
        VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&epnp5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_attr[c->device & 63].store((int)lds, std::memory_order_release);
    }
    hipLaunchKernelGGL(epnp5_kernel, dim3((a.total + EP_T - 1) / EP_T), dim3(EP_T), lds, c->stream, a, c->obj, c->obj_img, c->hyp_idx, c->hyp_idx + (size_t)c->p.max_batch * c->p.ransac_chunk * 5,
                       c->hyp, c->hyp_rt);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}
