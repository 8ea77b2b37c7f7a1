// Dense point cloud of a frame from its depth map (gfx950): the vertices VOClass::writeToPLY lists
// (reference Source/VOClassUtils.cpp:132-148): every pixel whose depth, narrowed to float, is not above the
// threshold, in row-major order, as (row, col, -z, grey).
//
//  cloud_count_kernel : per image row, number of pixels that pass.
//  cloud_scan_kernel  : exclusive scan of the row counts (one block).
//  cloud_write_kernel : per image row, wavefront ballot + prefix popcount gives every vertex its rank, so the order is
//                       the writer's order on every run.
#include "vo_ctx.h"
#include <string.h>
#include <algorithm>

namespace {

__device__ __forceinline__ bool passes(double depth, float thresh, float& z)
{
    z = (float)depth; // `float z = depthMap.at<double>(r, c)`
    return !(z > thresh);
}

__global__ __launch_bounds__(256) void cloud_count_kernel(const double* __restrict__ depth, int W, float thresh, int* __restrict__ row_count)
{
    __shared__ int wsum[4];
    const int y = blockIdx.x;
    const double* d = depth + (size_t)y * W;
    int cnt = 0;
    float z;
    for (int x = threadIdx.x; x < W; x += 256) cnt += passes(d[x], thresh, z);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) row_count[y] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void cloud_scan_kernel(const int* __restrict__ row_count, int H, int* __restrict__ row_off, int* __restrict__ total)
{
    __shared__ int part[256];
    const int per = (H + 255) / 256;
    const int y0 = min((int)threadIdx.x * per, H), y1 = min(y0 + per, H);
    int s = 0;
    for (int y = y0; y < y1; y++) s += row_count[y];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < 256; i++) { int t = part[i]; part[i] = acc; acc += t; }
        *total = acc;
    }
    __syncthreads();
    int acc = part[threadIdx.x];
    for (int y = y0; y < y1; y++) { row_off[y] = acc; acc += row_count[y]; }
}

__global__ __launch_bounds__(256) void cloud_write_kernel(const double* __restrict__ depth, const uint8_t* __restrict__ gray, size_t pitch,
                                                          int W, float thresh, const int* __restrict__ row_off, int cap,
                                                          vo_cloud_point* __restrict__ out)
{
    __shared__ int wcnt[4];
    const int y = blockIdx.x;
    const double* d = depth + (size_t)y * W;
    const uint8_t* g = gray + (size_t)y * pitch;
    int base = row_off[y];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int xs = 0; xs < W; xs += 256) {
        const int x = xs + threadIdx.x;
        float z = 0.f;
        const bool k = x < W && passes(d[x], thresh, z);
        const unsigned long long m = __ballot(k);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wcnt[w];
        const int total = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        if (k) {
            const int rank = base + before + __popcll(m & ((1ull << lane) - 1ull));
            if (rank < cap) {
                vo_cloud_point p;
                p.row = y; p.col = x; p.neg_z = -z; p.gray = g[x]; p.pad[0] = p.pad[1] = p.pad[2] = 0;
                out[rank] = p;
            }
        }
        base += total;
        __syncthreads();
    }
}

} // namespace

// Builds the cloud of frame b in c->cloud (capacity W*H vertices, allocated on first use); *count = vertices that pass.
int launch_depth_cloud(vo_ctx* c, int b, int* count)
{
    const int W = c->W, H = c->H;
    if (!c->cloud) {
        VO_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->cloud), (size_t)W * H * sizeof(vo_cloud_point)));
        VO_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->cloud_rows), (size_t)(2 * H + 1) * sizeof(int)));
    }
    int* row_count = c->cloud_rows, *row_off = c->cloud_rows + H, *total = c->cloud_rows + 2 * H;
    const double* depth = c->depth + (size_t)b * W * H;
    const float thresh = (float)c->p.depth_thresh;
    // frame b's LT1 image: slot b (pairs) or frame b of the left sequence
    const uint8_t* gray = c->img[0] + (size_t)b * c->fstride;
    hipLaunchKernelGGL(cloud_count_kernel, dim3(H), dim3(256), 0, c->stream, depth, W, thresh, row_count);
    hipLaunchKernelGGL(cloud_scan_kernel, dim3(1), dim3(256), 0, c->stream, row_count, H, row_off, total);
    hipLaunchKernelGGL(cloud_write_kernel, dim3(H), dim3(256), 0, c->stream, depth, gray, c->pitch, W, thresh, row_off, W * H, c->cloud);
    VO_HIP(c, hipGetLastError());
    VO_HIP(c, hipMemcpyAsync(c->h_stage, total, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    *count = *static_cast<const int*>(c->h_stage);
    return VO_OK;
}
