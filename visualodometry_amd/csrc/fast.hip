// FAST-9/16 corner detector with 3x3 non-max suppression and row-major ordered compaction (gfx950).
// Replaces cv::FAST(img, keypoints, threshold, true) + KeyPoint::convert as used by
// VOClass::getFeaturesFAST (reference Source/VOClassFeatures.cpp:27-32).
//
//  fast_score_kernel  : 64x16 tile (+3 halo) staged in LDS; a four-pair necessary test on every pixel, then, for the
//                       pixels that pass only: the 16 ring pixels are turned into a bright and a dark 16-bit mask,
//                       "9 contiguous" is a shift-and-AND test, the corner score is the closed form
//                       max(t, max_arc min(d), max_arc min(-d)) - 1.
//  fast_count_kernel  : per image row, the pixels whose score is a strict 3x3 maximum: their number, and their flags as
//                       64-bit ballot words.
//  fast_write_kernel  : per image row, the sum of the counts of the rows above is the row's offset; a prefix sum of the
//                       mask words' popcounts gives every keypoint its rank, so the output order is y-major, x-minor,
//                       identical on every run.
#include "vo_ctx.h"
#include <atomic>
#include <string.h>
#include <algorithm>
#include <mutex>

namespace {

constexpr int TILE_W = 64, TILE_H = 16, HALO = 3;
constexpr int LDS_W = TILE_W + 2 * HALO + 2; // 72
constexpr int LDS_H = TILE_H + 2 * HALO;     // 22

__device__ __forceinline__ int sliding_min9_max(const int d[16])
{
    // max over the 16 circular arcs of length 9 of the arc minimum
    int m2[16], m4[16], m8[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
    for (int i = 0; i < 16; i++) m8[i] = min(m4[i], m4[(i + 4) & 15]);
    int best = -1000;
#pragma unroll
    for (int i = 0; i < 16; i++) best = max(best, min(m8[i], d[(i + 8) & 15]));
    return best;
}

// Two phases per 64 x 16 tile.  Phase 1, every pixel: the cheap necessary condition.  An arc of 9 of the 16 ring pixels holds
// at least one pixel of every antipodal pair, so a corner has, for every pair, a member brighter than v + t (or, for a dark
// arc, a member darker than v - t); four pairs are tested.  The few pixels that pass are appended to a list in LDS (ballot
// ranks, one LDS atomic per wavefront).  Phase 2, the listed pixels only, all lanes busy: the full ring test and the score.
__global__ __launch_bounds__(256) void fast_score_kernel(const uint8_t* __restrict__ img, size_t pitch, size_t fstride,
                                                         int W, int H, int thr, int nms, uint8_t* __restrict__ score)
{
    __shared__ uint8_t tile[LDS_H][LDS_W];
    __shared__ unsigned short cand[TILE_W * TILE_H];
    __shared__ int ncand;
    const int b = blockIdx.z;
    const uint8_t* src = img + (size_t)b * fstride;
    const int x0 = blockIdx.x * TILE_W - HALO, y0 = blockIdx.y * TILE_H - HALO;
    const int tid = threadIdx.y * TILE_W + threadIdx.x;
    if (tid == 0) ncand = 0;
    for (int i = tid; i < LDS_H * (TILE_W + 2 * HALO); i += 256) {
        int ty = i / (TILE_W + 2 * HALO), tx = i - ty * (TILE_W + 2 * HALO);
        int gx = x0 + tx, gy = y0 + ty;
        uint8_t v = 0;
        if (gx >= 0 && gx < W && gy >= 0 && gy < H) v = src[(size_t)gy * pitch + gx];
        tile[ty][tx] = v;
    }
    __syncthreads();
    const int x = blockIdx.x * TILE_W + threadIdx.x;
    uint8_t* dst = score + (size_t)b * H * W;
#pragma unroll
    for (int r = 0; r < TILE_H / 4; r++) {
        const int ly = threadIdx.y + r * 4;
        const int y = blockIdx.y * TILE_H + ly;
        bool pass = false;
        if (x >= 3 && x < W - 3 && y >= 3 && y < H - 3) {
            const int cx = threadIdx.x + HALO, cy = ly + HALO;
            const int v = tile[cy][cx], lo = v - thr, hi = v + thr;
            auto cls = [&](int p) { return (p > hi ? 1 : 0) | (p < lo ? 2 : 0); };
            int m = cls(tile[cy + 3][cx]) | cls(tile[cy - 3][cx]);
            m &= cls(tile[cy][cx + 3]) | cls(tile[cy][cx - 3]);
            m &= cls(tile[cy + 2][cx + 2]) | cls(tile[cy - 2][cx - 2]);
            m &= cls(tile[cy - 2][cx + 2]) | cls(tile[cy + 2][cx - 2]);
            pass = m != 0;
        }
        if (x < W && y < H && !pass) dst[(size_t)y * W + x] = 0;
        const unsigned long long bal = __ballot(pass);
        if (bal) { // uniform per wavefront
            const int lane = tid & 63;
            int base = 0;
            if (lane == 0) base = atomicAdd(&ncand, __popcll(bal));
            base = __shfl(base, 0);
            if (pass) cand[base + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(ly * TILE_W + threadIdx.x);
        }
    }
    __syncthreads();
    const int total = ncand;
    for (int i = tid; i < total; i += 256) {
        const int lx = cand[i] % TILE_W, ly = cand[i] / TILE_W;
        const int cx = lx + HALO, cy = ly + HALO;
        const int v = tile[cy][cx];
        int ring[16];
        ring[0] = tile[cy + 3][cx];      ring[1] = tile[cy + 3][cx + 1];  ring[2] = tile[cy + 2][cx + 2];
        ring[3] = tile[cy + 1][cx + 3];  ring[4] = tile[cy][cx + 3];      ring[5] = tile[cy - 1][cx + 3];
        ring[6] = tile[cy - 2][cx + 2];  ring[7] = tile[cy - 3][cx + 1];  ring[8] = tile[cy - 3][cx];
        ring[9] = tile[cy - 3][cx - 1];  ring[10] = tile[cy - 2][cx - 2]; ring[11] = tile[cy - 1][cx - 3];
        ring[12] = tile[cy][cx - 3];     ring[13] = tile[cy + 1][cx - 3]; ring[14] = tile[cy + 2][cx - 2];
        ring[15] = tile[cy + 3][cx - 1];
        unsigned dark = 0, bright = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            dark |= (unsigned)(ring[k] < v - thr) << k;
            bright |= (unsigned)(ring[k] > v + thr) << k;
        }
        auto arc9 = [](unsigned m) {
            unsigned mm = m | (m << 16);
            unsigned t = mm & (mm >> 1);
            t &= t >> 2;
            t &= t >> 4;
            t &= mm >> 8;
            return (t & 0xffffu) != 0;
        };
        uint8_t out = 0;
        if (arc9(dark) || arc9(bright)) {
            if (nms) {
                int d[16], e[16];
#pragma unroll
                for (int k = 0; k < 16; k++) { d[k] = v - ring[k]; e[k] = -d[k]; }
                int a = sliding_min9_max(d), bm = sliding_min9_max(e);
                out = (uint8_t)(max(thr, max(a, bm)) - 1);
            } else out = 1;
        }
        dst[(size_t)(blockIdx.y * TILE_H + ly) * W + blockIdx.x * TILE_W + lx] = out;
    }
}

__device__ __forceinline__ bool is_keypoint(const uint8_t* __restrict__ s, int W, int x, int y, int nms)
{
    // rows 0 and H-1, columns 0 and W-1 never hold a score, so the 3x3 reads stay inside for every scored pixel
    const uint8_t* p = s + (size_t)y * W + x;
    int v = p[0];
    if (v == 0) return false;
    if (!nms) return true;
    return v > p[-1] && v > p[1] && v > p[-W - 1] && v > p[-W] && v > p[-W + 1] && v > p[W - 1] && v > p[W] && v > p[W + 1];
}

// words of 64 keypoint flags per image row: flag i of a row is column 3 + i (the detection domain starts at column 3)
__host__ __device__ constexpr int fast_mask_words(int W) { return (W - 6 + 63) / 64; }

__global__ __launch_bounds__(256) void fast_count_kernel(const uint8_t* __restrict__ score, int W, int H, int nms, int* __restrict__ row_count,
                                                         unsigned long long* __restrict__ mask)
{
    __shared__ int wsum[4];
    const int y = blockIdx.x, b = blockIdx.y;
    const uint8_t* s = score + (size_t)b * H * W;
    const int MW = fast_mask_words(W);
    unsigned long long* mrow = mask + ((size_t)b * H + y) * MW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool row_in = y >= 3 && y < H - 3;
    int cnt = 0;
    for (int xs = 3, t = 0; xs < W - 3; xs += 256, t++) {
        const int x = xs + threadIdx.x;
        const bool k = row_in && x < W - 3 && is_keypoint(s, W, x, y, nms);
        const unsigned long long m = __ballot(k); // the flags of 64 consecutive columns: one word of the row's mask
        if (lane == 0 && 4 * t + wave < MW) mrow[4 * t + wave] = m;
        cnt += k;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) row_count[b * H + y] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Per image row: its offset in the output is the number of keypoints in the rows above (summed by the block from the row
// counts: at most a few thousand small reads); the first wavefront then turns the row's mask words into keypoints: a
// prefix sum of the words' popcounts gives every set flag its rank, so the output order is y-major, x-minor.
// The block of the last row also publishes the total.
__global__ __launch_bounds__(256) void fast_write_kernel(int W, int H, int cap, const int* __restrict__ row_count,
                                                         const unsigned long long* __restrict__ mask, vo_point2f* __restrict__ feat,
                                                         int* __restrict__ n_feat)
{
    __shared__ int wsum[4];
    const int y = blockIdx.x, b = blockIdx.y;
    const int mine = row_count[b * H + y];
    if (mine == 0 && y != H - 1) return; // uniform per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int above = 0;
    for (int r = threadIdx.x; r < y; r += 256) above += row_count[b * H + r];
    for (int o = 32; o > 0; o >>= 1) above += __shfl_down(above, o);
    if (lane == 0) wsum[wave] = above;
    __syncthreads();
    int base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (y == H - 1 && threadIdx.x == 0) n_feat[b] = min(base + mine, cap);
    if (mine == 0 || wave != 0) return;
    const int MW = fast_mask_words(W);
    const unsigned long long* mrow = mask + ((size_t)b * H + y) * MW;
    vo_point2f* out = feat + (size_t)b * cap;
    for (int w0 = 0; w0 < MW; w0 += 64) {
        unsigned long long m = w0 + lane < MW ? mrow[w0 + lane] : 0ull;
        const int pc = __popcll(m);
        int incl = pc; // inclusive prefix sum over the lanes
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        int rank = base + incl - pc;
        const int xw = 3 + 64 * (w0 + lane);
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (rank < cap) { out[rank].x = (float)(xw + bit); out[rank].y = (float)y; }
            rank++;
        }
        base += __shfl(incl, 63);
    }
}

} // namespace

int launch_fast(vo_ctx* c)
{
    const int W = c->W, H = c->H, n = c->n;
    StageTimer st_(c, VO_ST_FAST);
    dim3 g1((W + TILE_W - 1) / TILE_W, (H + TILE_H - 1) / TILE_H, n), b1(TILE_W, 4);
    hipLaunchKernelGGL(fast_score_kernel, g1, b1, 0, c->stream, c->img[0], c->pitch, c->fstride, W, H,
                       c->p.fast_threshold, c->p.fast_nms, c->score);
    hipLaunchKernelGGL(fast_count_kernel, dim3(H, n), dim3(256), 0, c->stream, c->score, W, H, c->p.fast_nms, c->row_count, c->kp_mask);
    hipLaunchKernelGGL(fast_write_kernel, dim3(H, n), dim3(256), 0, c->stream, W, H, c->p.max_features, c->row_count, c->kp_mask,
                       c->feat, c->n_feat);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

// ---------------------------------------------------------------- detection without any allocation
// The same three kernels on one image, with their scratch (score map, per-row counts, keypoint flag words) in a fixed array
// of the code object instead of a context's hipMalloc'ed buffers.  Every GPU the library runs on holds its own copy of the
// array until the process ends.  In host code the array's name is only a handle for hipGetSymbolAddress: the address the
// kernels get is looked up once per device, with that device current, and kept.  That look-up loads the code object and
// takes a lock, which a stream capture does not allow: it happens in vo_create for the context's device, or in the first
// call of the entry point on a device, so a caller that captures makes one of the two outside the capture first.
constexpr size_t kNoAllocMaxPixels = (size_t)1 << 21;   // 2 Mi pixels (e.g. 1920 x 1080); KITTI frames are 0.47 Mi
constexpr int kNoAllocMaxRows = 4096;
constexpr size_t kNoAllocScoreBytes = kNoAllocMaxPixels;
constexpr size_t kNoAllocCountBytes = (size_t)kNoAllocMaxRows * sizeof(int);
// keypoint flags: at most ceil(W / 64) words per row, i.e. at most pixels / 64 + rows words of 8 bytes
constexpr size_t kNoAllocMaskBytes = (kNoAllocMaxPixels / 64 + kNoAllocMaxRows) * 8;
constexpr size_t kNoAllocBytes = kNoAllocScoreBytes + kNoAllocCountBytes + kNoAllocMaskBytes;
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];

static std::mutex g_scratch_mu;
static std::atomic<unsigned char*> g_scratch_ptr[64]; // per device: address of that device's copy of `scratch`

// Looks up the address of the CURRENT device's copy of the scratch array (dev = its index).  Not for use under stream capture.
int vo_noalloc_prepare(int dev)
{
    if (dev < 0 || dev >= 64) return vo_fail(nullptr, VO_ERR_NO_DEVICE, "device index %d out of range", dev);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (g_scratch_ptr[dev].load(std::memory_order_relaxed)) return VO_OK;
    void* p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p) return vo_fail(nullptr, VO_ERR_HIP, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    g_scratch_ptr[dev].store(static_cast<unsigned char*>(p), std::memory_order_release);
    return VO_OK;
}

extern "C" int vo_detect_fast_noalloc(const uint8_t* img_dev, int width, int height, size_t stride, int threshold, int nms,
                                      vo_point2f* out_dev, int cap, int* n_out_dev, void* stream)
{
    if (!img_dev || !out_dev || !n_out_dev || cap < 1) return vo_fail(nullptr, VO_ERR_BAD_ARG, "null pointer or non-positive capacity");
    if (width < 7 || height < 7 || stride < (size_t)width) return vo_fail(nullptr, VO_ERR_BAD_ARG, "image %dx%d (stride %zu) is not a valid frame", width, height, stride);
    if ((size_t)width * height > kNoAllocMaxPixels || height > kNoAllocMaxRows)
        return vo_fail(nullptr, VO_ERR_CAPACITY, "image %dx%d exceeds the fixed scratch of the allocation-free entry point (%zu pixels, %d rows)",
                       width, height, kNoAllocMaxPixels, kNoAllocMaxRows);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return vo_fail(nullptr, VO_ERR_NO_DEVICE, "no current HIP device");
    unsigned char* base = g_scratch_ptr[dev].load(std::memory_order_acquire);   // set once per device, never changed
    if (!base) {
        const int rc = vo_noalloc_prepare(dev);
        if (rc != VO_OK) return rc;
        base = g_scratch_ptr[dev].load(std::memory_order_acquire);
    }
    uint8_t* score = base;
    int* row_count = reinterpret_cast<int*>(base + kNoAllocScoreBytes);
    unsigned long long* kp_mask = reinterpret_cast<unsigned long long*>(base + kNoAllocScoreBytes + kNoAllocCountBytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    threshold = std::min(std::max(threshold, 0), 255);
    // every word of the scratch that a kernel reads was written by an earlier kernel of this same call (the score kernel
    // writes all width x height scores, the count kernel all `height` counts and every mask word of every row), so nothing has to
    // start at zero and what the previous call left behind is never seen
    dim3 g1((width + TILE_W - 1) / TILE_W, (height + TILE_H - 1) / TILE_H, 1), b1(TILE_W, 4);
    hipLaunchKernelGGL(fast_score_kernel, g1, b1, 0, st, img_dev, stride, (size_t)0, width, height, threshold, nms, score);
    hipLaunchKernelGGL(fast_count_kernel, dim3(height, 1), dim3(256), 0, st, score, width, height, nms, row_count, kp_mask);
    hipLaunchKernelGGL(fast_write_kernel, dim3(height, 1), dim3(256), 0, st, width, height, cap, row_count, kp_mask, out_dev, n_out_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return vo_fail(nullptr, VO_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return VO_OK;
}
