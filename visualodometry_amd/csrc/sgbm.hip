// Semi-global block matching, three-path variant, for gfx950.  Integer end to end (u8 -> s16), bit-exact.
// Replaces cv::StereoSGBM(MODE_SGBM_3WAY)::compute + medianBlur(3) + convertTo(CV_32F, 1/16) as used by
// VOClass::computeDisparity (reference Source/VOClass.cpp:258-296).
//
// Data layout (d fastest): cost volumes are [frame][stripe-row][cost column][D] int16.  The image is cut
// into 4 fixed stripes with a warm-up overlap; each stripe is an independent run of the vertical path, so a
// "stripe row" is (stripe, source row) and overlap rows exist once per stripe that needs them.
//
//  prefilter_kernel : per pixel and image, the two Birchfield-Tomasi channels (clipped x-Sobel, raw) with
//                     their half-pixel min/max, packed into 8 bytes.
//  cost_vpath_kernel: 16 lanes x (D/16) disparities per cost column, 4 (or 2) columns per lane group, 64 columns
//                     per block, marching down the stripe: pixel cost from prefiltered rows staged in LDS, running
//                     vertical window sum, horizontal window sum through LDS -> aggregated cost C, top-down path
//                     L_top.  The minimum over D is a DPP row reduction; d-1 / d+1 neighbours come from DPP row shifts.
//  hpath_ckpt_kernel / hpath_resolve_kernel: 16 lanes per image row, 4 rows per wavefront, serial along x.
//                     First pass: L_lr, keeping its state at segment starts.  Second pass, right to left: L_lr of a
//                     segment replayed into registers, L_rl, S = sum of the three paths, winner-take-all with
//                     OpenCV's tie rule, sub-pixel fit, packed (minS, d16) per pixel.
//  lrmedian_kernel  : per image row, right-view disparity by an order-independent LDS atomic-min scatter
//                     (smallest cost, then largest x) and the two-sided consistency test for the row and its two
//                     neighbours, then the 3x3 median (replicated border) on the s16 map, plus the /16 float map.
#include "vo_ctx.h"
#include <string.h>
#include <algorithm>
#include <atomic>

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int as_int(v2s v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ v2s as_v2s(int v) { return __builtin_bit_cast(v2s, v); }
__device__ __forceinline__ v2s splat(int s) { v2s r; r.x = (short)s; r.y = (short)s; return r; }
__device__ __forceinline__ v2s vmin(v2s a, v2s b) { return __builtin_elementwise_min(a, b); }

// reductions over the 16 lanes of a DPP row; every lane ends with the result
// (every lane of these four patterns has a valid source lane; "bound_ctrl with a zero old value" is merely the operand
// form that lets the DPP move fold into the min / max / add / or itself)
#define DPP_STEP(v, ctrl) __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, true)
__device__ __forceinline__ int row_min(int v)
{
    v = min(v, DPP_STEP(v, 0xB1)); v = min(v, DPP_STEP(v, 0x4E));
    v = min(v, DPP_STEP(v, 0x141)); v = min(v, DPP_STEP(v, 0x140));
    return v;
}
// value of the previous / next lane of the row; `edge` where there is none
__device__ __forceinline__ int row_from_prev(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x111, 0xF, 0xF, false); }
__device__ __forceinline__ int row_from_next(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x101, 0xF, 0xF, false); }

// ---------------------------------------------------------------- prefilter
// A block is 64 threads wide x 4 rows, a thread owns 4 adjacent pixels: it reads columns x0-2 .. x0+5 of three rows once
// and writes 32 contiguous bytes.  Border rule of the original: the clipped gradient and the raw value of the first and
// the last column are the constant ftzero; half-pixel values are averages with the horizontal neighbours.
constexpr int PF_T = 64, PF_R = 4, PF_N = 4;
__global__ __launch_bounds__(256) void prefilter_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right,
                                                        size_t pitch, size_t fstride, int W, int H, int Wp, int ftzero,
                                                        uint4* __restrict__ pre)
{
    // the block's source pixels (its 4 rows and the rows above and below, its 256 columns and two more on each side, all
    // clamped to the image) go through LDS: 6 coalesced byte loads per thread instead of 24 strided ones
    constexpr int PF_COLS = PF_T * PF_N + 4, PF_PITCH = PF_COLS + 4;
    __shared__ __attribute__((aligned(4))) uint8_t tile[(PF_R + 2) * PF_PITCH];
    const int b = blockIdx.z >> 1, which = blockIdx.z & 1;
    const uint8_t* img = (which ? right : left) + (size_t)b * fstride;
    const int bx = blockIdx.x * PF_T * PF_N, by = blockIdx.y * PF_R;
    for (int i = threadIdx.x; i < (PF_R + 2) * PF_COLS; i += 256) {
        const int ry = i / PF_COLS, cx = i - ry * PF_COLS;
        const int gx = min(max(bx - 2 + cx, 0), W - 1), gy = min(max(by - 1 + ry, 0), H - 1);
        tile[ry * PF_PITCH + cx] = img[(size_t)gy * pitch + gx];
    }
    __syncthreads();
    const int tx = threadIdx.x & (PF_T - 1), ty = threadIdx.x / PF_T;
    const int x0 = bx + tx * PF_N, y = by + ty;
    if (x0 >= W || y >= H) return;
    int a0[PF_N + 4], a1[PF_N + 4], a2[PF_N + 4]; // columns x0-2 .. x0+5; a clamped column is never used (border rule)
    {
        const unsigned* q0 = reinterpret_cast<const unsigned*>(tile + ty * PF_PITCH + tx * PF_N);
        const unsigned* q1 = reinterpret_cast<const unsigned*>(tile + (ty + 1) * PF_PITCH + tx * PF_N);
        const unsigned* q2 = reinterpret_cast<const unsigned*>(tile + (ty + 2) * PF_PITCH + tx * PF_N);
#pragma unroll
        for (int i = 0; i < PF_N + 4; i++) {
            a0[i] = (q0[i >> 2] >> (8 * (i & 3))) & 0xff; a1[i] = (q1[i >> 2] >> (8 * (i & 3))) & 0xff; a2[i] = (q2[i >> 2] >> (8 * (i & 3))) & 0xff;
        }
    }
    int g[PF_N + 2], r[PF_N + 2]; // columns x0-1 .. x0+4
#pragma unroll
    for (int k = 0; k < PF_N + 2; k++) {
        const int i = k + 1, x = x0 - 1 + k;
        const int v = (a1[i + 1] - a1[i - 1]) * 2 + a0[i + 1] - a0[i - 1] + a2[i + 1] - a2[i - 1];
        const bool edge = x <= 0 || x >= W - 1;
        g[k] = edge ? ftzero : min(max(v, -ftzero), ftzero) + ftzero;
        r[k] = edge ? ftzero : a1[i];
    }
    unsigned o[2 * PF_N];
#pragma unroll
    for (int j = 0; j < PF_N; j++) {
        const int k = j + 1, x = x0 + j;
        int gl = g[k], gr = g[k], rl = r[k], rr = r[k];
        if (x > 0) { gl = (g[k] + g[k - 1]) >> 1; rl = (r[k] + r[k - 1]) >> 1; }
        if (x < W - 1) { gr = (g[k] + g[k + 1]) >> 1; rr = (r[k] + r[k + 1]) >> 1; }
        const unsigned g0 = min(min(gl, gr), g[k]), g1 = max(max(gl, gr), g[k]);
        const unsigned q0 = min(min(rl, rr), r[k]), q1 = max(max(rl, rr), r[k]);
        o[2 * j] = (unsigned)g[k] | (g0 << 8) | (g1 << 16);
        o[2 * j + 1] = (unsigned)r[k] | (q0 << 8) | (q1 << 16);
    }
    // Wp is W rounded up to 4 pixels: the (at most 3) pixels past W of the last thread land in the row's padding
    uint4* dst = pre + ((((size_t)b * 2 + which) * H + y) * Wp + x0) / 2;
    dst[0] = make_uint4(o[0], o[1], o[2], o[3]);
    dst[1] = make_uint4(o[4], o[5], o[6], o[7]);
}

// ---------------------------------------------------------------- pixel cost + horizontal window
// Birchfield-Tomasi cost of one channel for two adjacent disparities at once (packed 16-bit lanes).
// u, u0, u1: the left pixel's value / half-pixel min / max, splatted.  ra, rb: the right pixels for the
// lower and the upper disparity of the pair, bytes (v, v0, v1, -).
typedef unsigned short v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s sub_sat0(v2s a, v2s b)
{
    return __builtin_bit_cast(v2s, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u, a), __builtin_bit_cast(v2u, b)));
}
__device__ __forceinline__ v2s bt_pair(v2s u, v2s u0, v2s u1, unsigned ra, unsigned rb)
{
    const v2s v = as_v2s((int)__builtin_amdgcn_perm(rb, ra, 0x0c040c00u));
    const v2s v0 = as_v2s((int)__builtin_amdgcn_perm(rb, ra, 0x0c050c01u));
    const v2s v1 = as_v2s((int)__builtin_amdgcn_perm(rb, ra, 0x0c060c02u));
    // all operands are in [0, 255]: max(0, a - b) is an unsigned saturating subtract, and of the two one-sided
    // distances of an interval at most one is non-zero
    // (so their maximum is their bitwise OR, which issues at the full rate where packed 16-bit maxima issue at half)
    const v2s c0 = as_v2s(as_int(sub_sat0(u, v1)) | as_int(sub_sat0(v0, u)));
    const v2s c1 = as_v2s(as_int(sub_sat0(v, u1)) | as_int(sub_sat0(u0, v)));
    return vmin(c0, c1);
}

// ---------------------------------------------------------------- one path step on a lane's NP packed pairs
// L[d] = C[d] + min(Lp[d], Lp[d-1]+P1, Lp[d+1]+P1, m+P2) - (m+P2); returns the minimum of L over the row (all D).
template <int NP>
__device__ __forceinline__ int path_step(const v2s (&C)[NP], v2s (&Lp)[NP], int m, int P1, int P2, int inf_pk)
{
    const v2s p1 = splat(P1), mp2 = splat(m + P2);
    int prev_last = row_from_prev(as_int(Lp[NP - 1]), inf_pk);
    int next_first = row_from_next(as_int(Lp[0]), inf_pk);
    v2s Ln[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) {
        int lo = k == 0 ? prev_last : as_int(Lp[k - 1]);
        int hi = k == NP - 1 ? next_first : as_int(Lp[k + 1]);
        v2s dm1 = as_v2s(__builtin_amdgcn_alignbit(as_int(Lp[k]), lo, 16));
        v2s dp1 = as_v2s(__builtin_amdgcn_alignbit(hi, as_int(Lp[k]), 16));
        v2s t = vmin(dm1, dp1) + p1;
        t = vmin(vmin(t, Lp[k]), mp2);
        Ln[k] = C[k] + (t - mp2);
    }
    v2s mm = Ln[0];
#pragma unroll
    for (int k = 1; k < NP; k++) mm = vmin(mm, Ln[k]);
#pragma unroll
    for (int k = 0; k < NP; k++) Lp[k] = Ln[k];
    return row_min(min((int)mm.x, (int)mm.y));
}

template <int NP>
__device__ __forceinline__ void load_pk(const int16_t* __restrict__ p, v2s (&v)[NP])
{
    const int* q = reinterpret_cast<const int*>(p);
#pragma unroll
    for (int k = 0; k < NP; k++) v[k] = as_v2s(q[k]);
}
template <int NP>
__device__ __forceinline__ void store_pk(int16_t* __restrict__ p, const v2s (&v)[NP])
{
    int* q = reinterpret_cast<int*>(p);
    // streamed out once and read back by later kernels only: keep it from displacing the ring and the image rows in cache
#pragma unroll
    for (int k = 0; k < NP; k++) __builtin_nontemporal_store(as_int(v[k]), q + k);
}

// Buffer addressing: a uniform descriptor (base, size) in scalar registers + a 32-bit lane offset + a uniform scalar offset,
// so the per-row address arithmetic is scalar and an offset at or past the size makes a load return 0 / drops a store.
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
#define VO_BUF_SKIP 0x80000000u // lane offset of a store that must not happen: past any buffer this file describes
// NP packed pairs to byte offset `off` of the buffer, streamed (non-temporal: read back by later kernels only, and must
// not displace the ring and the image rows in cache)
template <int NP, int K = 0, int NT = 2>
__device__ __forceinline__ void store_pk_buf(__amdgpu_buffer_rsrc_t rs, unsigned off, const v2s (&v)[NP], int soff = 0)
{
    if constexpr (NP - K >= 4) {
        v4u32 t = {(unsigned)as_int(v[K]), (unsigned)as_int(v[K + 1]), (unsigned)as_int(v[K + 2]), (unsigned)as_int(v[K + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, (int)off + 4 * K, soff, NT);
        store_pk_buf<NP, K + 4, NT>(rs, off, v, soff);
    } else if constexpr (NP - K == 3) {
        v3u32 t = {(unsigned)as_int(v[K]), (unsigned)as_int(v[K + 1]), (unsigned)as_int(v[K + 2])};
        __builtin_amdgcn_raw_buffer_store_b96(t, rs, (int)off + 4 * K, soff, NT);
    } else if constexpr (NP - K == 2) {
        v2u32 t = {(unsigned)as_int(v[K]), (unsigned)as_int(v[K + 1])};
        __builtin_amdgcn_raw_buffer_store_b64(t, rs, (int)off + 4 * K, soff, NT);
    } else if constexpr (NP - K == 1) {
        __builtin_amdgcn_raw_buffer_store_b32((unsigned)as_int(v[K]), rs, (int)off + 4 * K, soff, NT);
    }
}
// NP packed pairs from lane offset `off` + uniform offset `soff` of the buffer: no address arithmetic in vector registers
template <int NP, int K = 0>
__device__ __forceinline__ void load_pk_buf(__amdgpu_buffer_rsrc_t rs, unsigned off, int soff, v2s (&v)[NP])
{
    if constexpr (NP - K >= 4) {
        const v4u32 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off + 4 * K, soff, 0);
        v[K] = as_v2s((int)t.x); v[K + 1] = as_v2s((int)t.y); v[K + 2] = as_v2s((int)t.z); v[K + 3] = as_v2s((int)t.w);
        load_pk_buf<NP, K + 4>(rs, off, soff, v);
    } else if constexpr (NP - K == 3) {
        const v3u32 t = __builtin_amdgcn_raw_buffer_load_b96(rs, (int)off + 4 * K, soff, 0);
        v[K] = as_v2s((int)t.x); v[K + 1] = as_v2s((int)t.y); v[K + 2] = as_v2s((int)t.z);
    } else if constexpr (NP - K == 2) {
        const v2u32 t = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off + 4 * K, soff, 0);
        v[K] = as_v2s((int)t.x); v[K + 1] = as_v2s((int)t.y);
    } else if constexpr (NP - K == 1) {
        v[K] = as_v2s((int)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)off + 4 * K, soff, 0));
    }
}

// y0: first source row a stripe walks (warm-up included), out0: first row whose results are used, off: stripe-row offset
struct StripeInfo { int y0[4], rows[4], off[4], out0[4]; };

// ---------------------------------------------------------------- pixel cost + 11x11 window + top-down path, fused
// Block = lane groups of 16; a lane group owns NCOL adjacent tile columns (64 per block, SW2 of them on each side are
// halo), a lane 2*NP consecutive disparities of each.  Marching down the stripe:
//   V(x, d)  = running vertical window sum of the pixel cost: the row that enters the window is evaluated and added, the
//              row that leaves it is evaluated AGAIN and subtracted.  No per-thread history of evaluated rows exists,
//              neither in registers nor in LDS nor in memory: keeping one (a ring of byte costs) made the kernel
//              write-bound at KITTI shape and cost the wide disparity ranges their occupancy;
//   C(x, d)  = horizontal window sum of V through LDS (sliding over the group's columns),
//   L_top    = one path step per column; C and L_top are the only HBM traffic (one write each).
// The kernel is written for 128 registers (four wavefronts per SIMD: four 256-thread blocks, or two 512-thread blocks for
// 160 disparities and more, where a lane group takes 2 columns instead of 4):
//   - the prefiltered pixels a tile needs for one image row (64 left, D + 64 right) are staged in LDS by the whole block,
//     one row ahead: a thread requests its two pixels at the top of a row and stores them to LDS at the bottom, so the
//     memory latency hides behind the row's arithmetic, and every lane then reads its 2NP + NCOL - 1 right pixels from LDS
//     (one pad slot per 16 pixels keeps the lanes of a group, 2NP pixels apart, on different banks);
//   - every evaluated cost goes straight into the running sum V, nothing is kept per row;
//   - the horizontal window is summed one register column (k) at a time.
template <int NP, int NCOL_>
struct CostTile {
    static constexpr int NCOL = NCOL_, CV_COLS = 64, THREADS = (CV_COLS / NCOL) * 16; // 512 threads at 2 columns per lane group, 256 at 4
    static constexpr int CS = 16 * NP + 4;                  // dwords per tile column of V
    static constexpr int RPX = CV_COLS + 32 * NP;           // right pixels per tile and row: x0 - D .. x0 + 63
    static constexpr int RSLOTS = RPX + RPX / 16;           // with the pad slots
    static constexpr int EVAL_SLOTS = RSLOTS + CV_COLS;     // + the left pixels
    static constexpr int ENTRIES = RPX + CV_COLS;           // pixels fetched per row evaluation
    // Planar staging (4 columns per lane group): the six prefiltered values of a pixel go to six planes of 16-bit slots, the
    // right pixels of a tile in DESCENDING image order followed by the left pixels in ascending order.  Two adjacent slots
    // of a right plane are then the operands of two adjacent disparities (x - d and x - d - 1) as one packed register, read
    // straight from LDS or formed by one alignbit, and a left value is a half of a register that the packed instructions
    // select by themselves: the cost evaluation needs no byte extraction at all.
    static constexpr bool PLANAR = NCOL_ == 4;
    static constexpr int PSTR = (RPX + CV_COLS) * 2;        // bytes per plane
    static constexpr int EVAL_BYTES = PLANAR ? 6 * PSTR : EVAL_SLOTS * 8;
    static constexpr size_t lds_bytes() { return (size_t)2 * CV_COLS * CS * 4 + (size_t)2 * 2 * EVAL_BYTES; }
};

template <int NP, int SWC, int NCOL_>
__global__ __launch_bounds__((CostTile<NP, NCOL_>::THREADS), 4) void cost_vpath_kernel(const uint2* __restrict__ pre, int W, int H, int Wp, int width1, int D, int SW2,
                                                                 int srows, StripeInfo si, int P1, int P2, int16_t* __restrict__ cost,
                                                                 int16_t* __restrict__ agg, int live_d, int dead_cost)
{
    typedef CostTile<NP, NCOL_> T;
    constexpr int NCOL = T::NCOL, CV_COLS = T::CV_COLS, CS = T::CS;
    constexpr int NR = 2 * NP + NCOL - 1;
    extern __shared__ int vt[];                    // [2][CV_COLS][CS] V tiles, then [2][2][EVAL_SLOTS] staged pixels (8 bytes each)
    v2u32* stage = reinterpret_cast<v2u32*>(vt + 2 * CV_COLS * CS);
    const int lane16 = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int stripe = blockIdx.y, b = blockIdx.z;
    const int y0 = si.y0[stripe], rows = si.rows[stripe];
    if (rows <= 0) return;
    const int tile_out = CV_COLS - 2 * SW2;
    const int xc0 = blockIdx.x * tile_out - SW2;
    const int minX1 = W - width1;
    const int cb = q * NCOL;
    const int d0 = lane16 * 2 * NP;
    const unsigned img_bytes = (unsigned)H * (unsigned)Wp * 8u;
    const unsigned row_pitch = (unsigned)Wp * 8u;
    // one descriptor over both prefiltered images of the frame (left, then right)
    const __amdgpu_buffer_rsrc_t Prs = make_rsrc(pre + (size_t)b * 2 * H * Wp, 2u * img_bytes);

    // ---- staging: entry n of a row's 2 * ENTRIES pixels: evaluation e = n / ENTRIES (0: leaving row, 1: entering row),
    // then right pixels x0 - D + m (m < RPX) and left pixels x0 + m - RPX.  Pixels outside the image are never used.
    const int x0 = minX1 + xc0; // image column of tile column 0
    int st_slot[2]; unsigned st_off[2]; bool st_enter[2];
    char* const stage_b = reinterpret_cast<char*>(stage);
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int n = (int)threadIdx.x + u * T::THREADS;
        const int e = n / T::ENTRIES, mm = n - e * T::ENTRIES;
        const bool right = mm < T::RPX;
        const int x = right ? x0 - D + mm : x0 + (mm - T::RPX);
        st_enter[u] = e == 1;
        if (T::PLANAR) // byte offset of the pixel's slot in the first plane of its evaluation
            st_slot[u] = n < 2 * T::ENTRIES ? e * T::EVAL_BYTES + 2 * (right ? T::RPX - 1 - mm : mm) : -1;
        else
        st_slot[u] = n < 2 * T::ENTRIES ? e * T::EVAL_SLOTS + (right ? mm + (mm >> 4) : T::RSLOTS + (mm - T::RPX)) : -1;
        // byte offset inside the frame's two images; anything not addressable becomes "far out of range" (reads 0)
        st_off[u] = (n < 2 * T::ENTRIES && x >= 0 && x < Wp) ? (right ? img_bytes : 0u) + 8u * (unsigned)x : 0xC0000000u;
    }
    v2u32 pf[2];
    auto prefetch = [&](int leave_row, int enter_row) {
#pragma unroll
        for (int u = 0; u < 2; u++)
            pf[u] = __builtin_amdgcn_raw_buffer_load_b64(Prs, (int)(st_off[u] + (unsigned)(st_enter[u] ? enter_row : leave_row) * row_pitch), 0, 0);
    };
    auto publish = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 2; u++) if (st_slot[u] >= 0) {
            if constexpr (T::PLANAR) {
                // byte stores into the low halves of the 16-bit slots; the high halves were zeroed once and stay zero
                uint8_t* q = reinterpret_cast<uint8_t*>(stage_b + buf * 2 * T::EVAL_BYTES + st_slot[u]);
                const unsigned a = pf[u].x, c = pf[u].y;
                q[0] = (uint8_t)a; q[T::PSTR] = (uint8_t)(a >> 8); q[2 * T::PSTR] = (uint8_t)(a >> 16);
                q[3 * T::PSTR] = (uint8_t)c; q[4 * T::PSTR] = (uint8_t)(c >> 8); q[5 * T::PSTR] = (uint8_t)(c >> 16);
            } else
                stage[buf * 2 * T::EVAL_SLOTS + st_slot[u]] = pf[u];
        }
    };

    v2s V[NCOL][NP], Lp[NCOL][NP];
    // A row evaluation runs in two halves of the lane's disparity pairs, upper half first: pairs k >= KH need the lane's
    // right pixels 0 .. MID, pairs k < KH the pixels 2 (NP - KH) .. NR - 1, so only about half of them are in registers at a time.
    constexpr int KH = NP / 2, MID = 2 * (NP - KH) + NCOL - 2;
    auto add_cost = [&](const v2u32 (&r)[NR], const v2u32 (&l)[NCOL], int scale, int k_begin, int k_end) {
        const v2s sc = splat(scale);
#pragma unroll
        for (int j = 0; j < NCOL; j++) {
            const v2s g = splat(l[j].x & 0xff), g0 = splat((l[j].x >> 8) & 0xff), g1 = splat((l[j].x >> 16) & 0xff);
            const v2s w = splat(l[j].y & 0xff), w0 = splat((l[j].y >> 8) & 0xff), w1 = splat((l[j].y >> 16) & 0xff);
#pragma unroll
            for (int k = 0; k < NP; k++) {
                if (k < k_begin || k >= k_end) continue;
                const int ia = j - 2 * k + 2 * NP - 1; // slot 2k (lower d); slot 2k+1 is ia - 1
                // Both halves of every operand below are small and non-negative (a pixel cost is below 2^7, a window sum of 11
                // of them below 2^11, and a cost that leaves the sum was added to it earlier), so no carry or borrow crosses
                // the halves: plain 32-bit adds, which issue faster than the packed 16-bit ones.
                const int c = as_int(bt_pair(g, g0, g1, r[ia].x, r[ia - 1].x)) + as_int(bt_pair(w, w0, w1, r[ia].y, r[ia - 1].y) >> 2);
                if (scale == 1) V[j][k] = as_v2s(as_int(V[j][k]) + c);
                else if (scale == -1) V[j][k] = as_v2s(as_int(V[j][k]) - c);
                else V[j][k] = as_v2s(c) * sc + V[j][k];
            }
        }
    };
    // V += scale * (pixel cost of image row `row`), operands straight from memory (start of the stripe only)
    const unsigned xl0 = 8u * (unsigned)(x0 + cb);
    const unsigned xr0 = img_bytes + 8u * (unsigned)(x0 + cb - d0 - (2 * NP - 1)); // below zero only for unused halo columns
    auto accum_global = [&](int row, int scale) {
        const int so = row * (int)row_pitch; // uniform
        v2u32 r[NR], l[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; j++) l[j] = __builtin_amdgcn_raw_buffer_load_b64(Prs, (int)(xl0 + 8u * j), so, 0);
#pragma unroll
        for (int i = 0; i <= MID; i++) r[i] = __builtin_amdgcn_raw_buffer_load_b64(Prs, (int)(xr0 + 8u * i), so, 0);
        add_cost(r, l, scale, KH, NP);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = MID + 1; i < NR; i++) r[i] = __builtin_amdgcn_raw_buffer_load_b64(Prs, (int)(xr0 + 8u * i), so, 0);
        add_cost(r, l, scale, 0, KH);
    };
    // the same from the staged pixels of evaluation e in stage buffer `buf`
    const int p0 = cb + 2 * NP * (15 - lane16) + 1;   // index of the lane's first right pixel among the tile's RPX
    const int p0_slot = p0 + (p0 >> 4), p0_wrap = 16 - (p0 & 15); // from its p0_wrap-th pixel on, one pad slot lies between
    // the same from the planes: window of the lane's right pixels p0 .. p0 + NR - 1 (stored descending: pixel p0 + i sits in 16-bit
    // slot NR - 1 - i of the window, which starts on a dword), so the pair (pixel of disparity 2k, pixel of disparity 2k + 1) of
    // column j is slots (o, o + 1) with o = NCOL - 1 + 2k - j: a whole dword when o is even, one alignbit when it is odd
    auto accum_planar = [&](int buf, int e, int sign) {
        constexpr int WN = NP + NCOL / 2, HC = NCOL / 2;
        const char* S = stage_b + (buf * 2 + e) * T::EVAL_BYTES;
        const int win_b = 2 * (T::RPX - 1 - p0 - (NR - 1));
#pragma unroll
        for (int ch = 0; ch < 2; ch++) { // 0: clipped gradient, 1: raw intensity (its cost counts a quarter)
            int Wd[3][WN], L[3][HC], A[3][WN - 1];
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                __builtin_memcpy(Wd[pl], S + (ch * 3 + pl) * T::PSTR + win_b, 4 * WN);
                __builtin_memcpy(L[pl], S + (ch * 3 + pl) * T::PSTR + 2 * (T::RPX + cb), 4 * HC);
#pragma unroll
                for (int t = 0; t + 1 < WN; t++) A[pl][t] = (int)__builtin_amdgcn_alignbit((unsigned)Wd[pl][t + 1], (unsigned)Wd[pl][t], 16);
            }
#pragma unroll
            for (int j = 0; j < NCOL; j++) {
                v2s lu[3];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    const v2s t = as_v2s(L[pl][j >> 1]);
                    lu[pl] = (j & 1) ? __builtin_shufflevector(t, t, 1, 1) : __builtin_shufflevector(t, t, 0, 0);
                }
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const int o = NCOL - 1 + 2 * k - j;
                    v2s rv[3];
#pragma unroll
                    for (int pl = 0; pl < 3; pl++) rv[pl] = as_v2s((o & 1) ? A[pl][o >> 1] : Wd[pl][o >> 1]);
                    const v2s c0 = as_v2s(as_int(sub_sat0(lu[0], rv[2])) | as_int(sub_sat0(rv[1], lu[0])));
                    const v2s c1 = as_v2s(as_int(sub_sat0(rv[0], lu[2])) | as_int(sub_sat0(lu[1], rv[0])));
                    v2s c = vmin(c0, c1);
                    if (ch == 1) c = c >> 2;
                    // no carry or borrow crosses the halves (see add_cost): plain 32-bit adds
                    V[j][k] = as_v2s(sign > 0 ? as_int(V[j][k]) + as_int(c) : as_int(V[j][k]) - as_int(c));
                }
            }
            __builtin_amdgcn_sched_barrier(0); // one channel's operands live at a time
        }
    };
    auto accum_staged = [&](int buf, int e, int scale) {
        if constexpr (T::PLANAR) { accum_planar(buf, e, scale); return; }
        const v2u32* S = stage + (buf * 2 + e) * T::EVAL_SLOTS;
        v2u32 r[NR], l[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; j++) l[j] = S[T::RSLOTS + cb + j];
#pragma unroll
        for (int i = 0; i <= MID; i++) r[i] = S[p0_slot + i + (i >= p0_wrap ? 1 : 0) + (i >= p0_wrap + 16 ? 1 : 0)];
        add_cost(r, l, scale, KH, NP);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = MID + 1; i < NR; i++) r[i] = S[p0_slot + i + (i >= p0_wrap ? 1 : 0) + (i >= p0_wrap + 16 ? 1 : 0)];
        add_cost(r, l, scale, 0, KH);
    };

    int m[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; j++) {
        m[j] = 0;
#pragma unroll
        for (int k = 0; k < NP; k++) { V[j][k] = splat(0); Lp[j][k] = splat(0); }
    }
    if constexpr (T::PLANAR) { // the high halves of the 16-bit slots stay zero for the whole kernel
        for (int i = threadIdx.x; i < 2 * 2 * T::EVAL_BYTES / 4; i += T::THREADS) reinterpret_cast<int*>(stage_b)[i] = 0;
        __syncthreads();
    }
    // rows above the stripe top repeat its first row; past the last image row the last row repeats
    if (rows > 1) prefetch(y0, min(y0 + 1 + SW2, H - 1)); // row 1: row 0 of the stripe leaves (it stood in for the rows above), row 1 + SW2 enters
    accum_global(y0, SW2 + 1);
    for (int dy = 1; dy <= SW2; dy++) accum_global(min(y0 + dy, H - 1), 1);

    const int inf_pk = (32767 - P1) * 0x10001;
    const int cmin = max(-xc0, 0), cmax = min(width1 - 1 - xc0, CV_COLS - 1);
    constexpr int NC = NCOL + 2 * (SWC > 0 ? SWC : 1);
    // The horizontal window clamps at the first / last cost column (tile columns cmin / cmax).  With a compile-time window
    // the clamp is done on the writing side: the owner of an end column also writes its V into the SWC tile columns
    // beyond it, columns outside [cmin, cmax] are not written by their owners, and every read is base + constant.
    const int hbase = (cb - SWC) * CS + lane16;
    const size_t rstride = (size_t)width1 * D;
    const unsigned row_bytes = (unsigned)(rstride * sizeof(int16_t));
    char* crow = reinterpret_cast<char*>(cost + ((size_t)b * srows + si.off[stripe]) * rstride);
    char* arow = reinterpret_cast<char*>(agg + ((size_t)b * srows + si.off[stripe]) * rstride);
    unsigned ocol[NCOL];
#pragma unroll
    for (int j = 0; j < NCOL; j++) {
        const int c = cb + j, xc = xc0 + c;
        ocol[j] = (c >= SW2 && c < CV_COLS - SW2 && xc < width1) ? (unsigned)((xc * D + d0) * (int)sizeof(int16_t)) : VO_BUF_SKIP;
    }

    for (int r = 0; r < rows; r++) {
        const int y = y0 + r;
        if (r > 0) {
            // row r's pixels were published before the previous row's barrier
            accum_staged(r & 1, 0, -1); // the row that leaves the window: y0 + max(r - SW2 - 1, 0)
            accum_staged(r & 1, 1, 1);  // the row that enters it: min(y + SW2, H - 1)
        }
        int* buf = vt + (r & 1) * (CV_COLS * CS);
#pragma unroll
        for (int j = 0; j < NCOL; j++) {
            const int c = cb + j;
            if (SWC == 0 || (c >= cmin && c <= cmax)) {
#pragma unroll
                for (int k = 0; k < NP; k++) buf[c * CS + k * 16 + lane16] = as_int(V[j][k]);
            }
            if (SWC > 0 && (c == cmin || c == cmax)) { // rare: first and last tile of a row only
                for (int e = 1; e <= SWC; e++) {
                    if (c == cmin && c - e >= 0)
                        for (int k = 0; k < NP; k++) buf[(c - e) * CS + k * 16 + lane16] = as_int(V[j][k]);
                    if (c == cmax && c + e < CV_COLS)
                        for (int k = 0; k < NP; k++) buf[(c + e) * CS + k * 16 + lane16] = as_int(V[j][k]);
                }
            }
        }
        // next row's pixels (requested a row ago) go to the other stage buffer, which was last read before the previous barrier
        if (r + 1 < rows) publish((r + 1) & 1);
        __syncthreads(); // one barrier per row: publishes V of this row and the staged pixels of the next
        if (r + 2 < rows) prefetch(y0 + max(r + 2 - SW2 - 1, 0), min(y + 2 + SW2, H - 1));
        v2s C[NCOL][NP];
        if (SWC > 0) {
#pragma unroll
            for (int k = 0; k < NP; k++) {
                v2s col[NC];
#pragma unroll
                for (int i = 0; i < NC; i++) col[i] = as_v2s(buf[hbase + i * CS + k * 16]); // beyond the tile only for unused halo columns
                // 32-bit adds on the packed pairs: every partial sum of window sums stays below 2^15 per half
                int a = as_int(col[0]);
#pragma unroll
                for (int i = 1; i <= 2 * SWC; i++) a += as_int(col[i]);
                C[0][k] = as_v2s(a);
#pragma unroll
                for (int j = 1; j < NCOL; j++) C[j][k] = as_v2s(as_int(C[j - 1][k]) + as_int(col[j + 2 * SWC]) - as_int(col[j - 1]));
                __builtin_amdgcn_sched_barrier(0); // keep the register columns apart: one column's 12 operands live at a time
            }
        } else {
#pragma unroll
            for (int k = 0; k < NP; k++) C[0][k] = splat(0);
            for (int dx = -SW2; dx <= SW2; dx++) {
                const int* p = buf + min(max(cb + dx, cmin), cmax) * CS + lane16;
#pragma unroll
                for (int k = 0; k < NP; k++) C[0][k] += as_v2s(p[k * 16]);
            }
#pragma unroll
            for (int j = 1; j < NCOL; j++) {
                const int* pa = buf + min(max(cb + j + SW2, cmin), cmax) * CS + lane16;
                const int* ps = buf + min(max(cb + j - 1 - SW2, cmin), cmax) * CS + lane16;
#pragma unroll
                for (int k = 0; k < NP; k++) C[j][k] = C[j - 1][k] + as_v2s(pa[k * 16]) - as_v2s(ps[k * 16]);
            }
        }
        // a padded disparity range: the cells beyond it get the cost no path prefers.  The condition is uniform; reading it
        // through readfirstlane keeps it a branch (as a select it would cost every row of every launch 4 NP instructions)
        if (__builtin_amdgcn_readfirstlane(live_d) < D) {
            const v2s dead = splat(dead_cost);
#pragma unroll
            for (int j = 0; j < NCOL; j++)
#pragma unroll
                for (int k = 0; k < NP; k++) C[j][k] = d0 + 2 * k >= live_d ? dead : C[j][k];
        }
#pragma unroll
        for (int j = 0; j < NCOL; j++) m[j] = path_step<NP>(C[j], Lp[j], m[j], P1, P2, inf_pk);
        if (y >= si.out0[stripe]) {
            const __amdgpu_buffer_rsrc_t crs = make_rsrc(crow, row_bytes), ars = make_rsrc(arow, row_bytes);
#pragma unroll
            for (int j = 0; j < NCOL; j++) {
                store_pk_buf<NP>(crs, ocol[j], C[j]);
                store_pk_buf<NP>(ars, ocol[j], Lp[j]);
            }
        }
        crow += rstride * sizeof(int16_t); arow += rstride * sizeof(int16_t);
    }
}

// ---------------------------------------------------------------- horizontal paths
// The left-to-right path L_lr is needed again when the right-to-left pass forms S = L_rl + (L_lr + L_top), but a row of it
// (width1 x D x 2 bytes) is far too large to keep on chip and storing it costs a volume write plus a volume read.
// Instead:
//   hpath_ckpt_kernel    runs L_lr over the row once and keeps only its state (L_lr[D], min) at every HP_SEG-th column;
//   hpath_resolve_kernel walks the segments right to left; per segment it replays L_lr from the checkpoint into registers
//                        (HP_SEG columns x 2NP cells per lane), then runs L_rl back over the same columns with S, the
//                        winner and its neighbours.  Lane u of a row keeps the result of the segment's column u, so the
//                        sub-pixel fit and the store happen once per segment with all lanes busy.
// Traffic per cost cell: C twice and L_top once (6 bytes) instead of 10; only image rows that produce output are
// walked (the warm-up rows of a stripe matter to the vertical path alone).

// stripe row of image row y
__device__ __forceinline__ int stripe_row_of(const StripeInfo& si, int stripe_sz, int y)
{
    const int s = min(y / stripe_sz, 3);
    return si.off[s] + (y - si.y0[s]);
}

constexpr int vo_gcd(int a, int b) { return b == 0 ? a : vo_gcd(b, a % b); }

template <int NP>
__global__ __launch_bounds__(256) void hpath_ckpt_kernel(const int16_t* __restrict__ cost, int width1, int D, int H, int srows, StripeInfo si,
                                                         int stripe_sz, int P1, int P2, int nseg, int* __restrict__ ckpt)
{
    constexpr int HP_SEG = vo_hpath_seg(NP);
    const int lane16 = threadIdx.x & 15;
    const int y = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int b = blockIdx.y;
    // rows past the image repeat the last row: they compute and store exactly what its own lanes do
    const int yy = min(y, H - 1);
    const int sr = stripe_row_of(si, stripe_sz, yy);
    // One descriptor per wavefront at its first row; a lane's offset is its row's distance from there (four consecutive image
    // rows, at most a stripe's warm-up rows apart in the volume) plus its disparities, the column is a uniform scalar offset:
    // a column's address costs no vector instruction and no vector register.
    const int sr0 = __builtin_amdgcn_readfirstlane(sr), yy0 = __builtin_amdgcn_readfirstlane(yy);
    const unsigned col_bytes = (unsigned)D * 2u, row_bytes = (unsigned)width1 * col_bytes;
    const __amdgpu_buffer_rsrc_t Crs = make_rsrc(cost + ((size_t)b * srows + sr0) * width1 * D, 0x7fffffffu);
    const unsigned coff = (unsigned)(sr - sr0) * row_bytes + (unsigned)lane16 * (4u * NP);
    const unsigned seg_bytes = 16u * (NP + 1) * 4u;
    const __amdgpu_buffer_rsrc_t Krs = make_rsrc(ckpt + ((size_t)b * H + yy0) * nseg * 16 * (NP + 1), 0x7fffffffu);
    const unsigned koff = (unsigned)(yy - yy0) * (unsigned)nseg * seg_bytes + (unsigned)lane16 * (4u * (NP + 1));
    const int inf_pk = (32767 - P1) * 0x10001;
    v2s Lp[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) Lp[k] = splat(0);
    int m = 0;
    // Columns are consumed in groups of U, and a group is requested TWO groups before it is consumed: the kernel runs at
    // memory speed, where a request takes several microseconds to come back, longer than one group's arithmetic.  Three
    // register groups take turns.  The body is written out for a whole PERIOD of columns (the checkpoint spacing and the
    // rotation of the groups repeat after it) as ONE straight run, requests clamped to the last column instead of skipped
    // and checkpoints at fixed places: behind a branch the compiler must assume the worst about the requests in flight and
    // waits for all of them, which leaves one group in flight instead of two.
    constexpr int U = 4, PERIOD = HP_SEG * (3 * U) / vo_gcd(HP_SEG, 3 * U), NG = PERIOD / U;
    static_assert(NG % 3 == 0 && HP_SEG % U == 0, "group rotation and checkpoint spacing must divide the period");
    v2s g[3][U][NP];
    auto load_group = [&](int x0, v2s (&gr)[U][NP]) {
#pragma unroll
        for (int u = 0; u < U; u++) load_pk_buf<NP>(Crs, coff, min(x0 + u, width1 - 1) * (int)col_bytes, gr[u]);
    };
    auto checkpoint = [&](int x0) { // state before column x0
        v2s t[NP + 1];
#pragma unroll
        for (int k = 0; k < NP; k++) t[k] = Lp[k];
        t[NP] = as_v2s(m);
        store_pk_buf<NP + 1, 0, 0>(Krs, koff, t, (x0 / HP_SEG) * (int)seg_bytes);
    };
    load_group(0, g[0]);
    load_group(U, g[1]);
    int xb = 0;
    for (; xb + PERIOD <= width1; xb += PERIOD) {
#pragma unroll
        for (int i = 0; i < NG; i++) {
            load_group(xb + (i + 2) * U, g[(i + 2) % 3]);
            if ((i * U) % HP_SEG == 0) checkpoint(xb + i * U);
            __builtin_amdgcn_sched_barrier(0); // the requests stay here, two groups ahead of their use
#pragma unroll
            for (int u = 0; u < U; u++) m = path_step<NP>(g[i % 3][u], Lp, m, P1, P2, inf_pk);
        }
    }
    // the rest of the row (less than a period): the same groups, each behind its own test
#pragma unroll
    for (int i = 0; i < NG; i++) {
        const int x0 = xb + i * U;
        if (x0 >= width1) break; // uniform
        load_group(x0 + 2 * U, g[(i + 2) % 3]);
        if ((i * U) % HP_SEG == 0) checkpoint(x0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (x0 + u < width1) m = path_step<NP>(g[i % 3][u], Lp, m, P1, P2, inf_pk);
    }
}

// 7 and 8 pairs per lane come out just over 256 registers, which is one wavefront per SIMD; held to 256, two fit
template <int NP>
__global__ __launch_bounds__(64, (NP >= 7 ? 2 : 1)) void hpath_resolve_kernel(const int16_t* __restrict__ cost, const int16_t* __restrict__ ltop, int width1,
                                                            int Dp, int W, int H, int srows, StripeInfo si, int stripe_sz, int P1, int P2,
                                                            int tie_lanes, int nseg, const int* __restrict__ ckpt, int32_t* __restrict__ wta, int D)
{
    // Dp: disparity stride of the volumes (a multiple of 32, what NP is instantiated for); D: the disparity range searched
    constexpr int HP_SEG = vo_hpath_seg(NP);
    const int lane16 = threadIdx.x & 15;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 4); // one wavefront per block: four rows
    const int b = blockIdx.y;
    const bool active = y < H;
    const int yy = active ? y : H - 1;
    const int sr = stripe_row_of(si, stripe_sz, yy);
    // One descriptor per volume at the wavefront's first row; a lane's offset is its row's distance from there plus its
    // disparities, the column is a uniform scalar offset: a column's address costs no vector instruction or register.
    const int sr0 = __builtin_amdgcn_readfirstlane(sr), yy0 = __builtin_amdgcn_readfirstlane(yy);
    const unsigned col_bytes = (unsigned)Dp * 2u, row_bytes = (unsigned)width1 * col_bytes;
    const size_t base0 = (((size_t)b * srows + sr0) * width1) * Dp;
    const __amdgpu_buffer_rsrc_t Crs = make_rsrc(cost + base0, 0x7fffffffu), Trs = make_rsrc(ltop + base0, 0x7fffffffu);
    const unsigned voff = (unsigned)(sr - sr0) * row_bytes + (unsigned)lane16 * (4u * NP);
    const unsigned seg_bytes = 16u * (NP + 1) * 4u;
    const __amdgpu_buffer_rsrc_t Krs = make_rsrc(ckpt + ((size_t)b * H + yy0) * nseg * 16 * (NP + 1), 0x7fffffffu);
    const unsigned koff = (unsigned)(yy - yy0) * (unsigned)nseg * seg_bytes + (unsigned)lane16 * (4u * (NP + 1));
    int32_t* out = wta + ((size_t)b * srows + sr) * W + (W - width1);
    const int inf_pk = (32767 - P1) * 0x10001;
    // exchange buffer of this row (16 lanes): [column of the segment][D / 2 packed sums], odd stride against bank conflicts
    extern __shared__ int hp_lds[];
    constexpr int STR = 16 * NP + 1;
    int* trow = hp_lds + (threadIdx.x >> 4) * (HP_SEG * STR);

    v2s LpR[NP];
#pragma unroll
    for (int k = 0; k < NP; k++) LpR[k] = splat(0);
    int mR = 0;

    v2s Cc[HP_SEG][NP], Cn[HP_SEG][NP], Tt[HP_SEG][NP], Lrec[HP_SEG][NP];
    int ckc[NP + 1], ckn[NP + 1];
    auto load_cols = [&](const __amdgpu_buffer_rsrc_t& rs, int s, v2s (&g)[HP_SEG][NP], int u0, int u1) {
#pragma unroll
        for (int u = 0; u < HP_SEG; u++)
            if (u >= u0 && u < u1) load_pk_buf<NP>(rs, voff, min(s * HP_SEG + u, width1 - 1) * (int)col_bytes, g[u]);
    };
    auto load_seg = [&](const __amdgpu_buffer_rsrc_t& rs, int s, v2s (&g)[HP_SEG][NP]) { load_cols(rs, s, g, 0, HP_SEG); };
    auto load_ck = [&](int s, int (&c)[NP + 1]) {
        v2s t[NP + 1];
        load_pk_buf<NP + 1>(Krs, koff, s * (int)seg_bytes, t);
#pragma unroll
        for (int k = 0; k <= NP; k++) c[k] = as_int(t[k]);
    };
    // (the widest ranges have no register to spare for a request that overlaps the winner search: they ask at the top of a segment)
    constexpr bool T_EARLY = NP <= 6;
    load_seg(Crs, nseg - 1, Cc);
    load_ck(nseg - 1, ckc);
    if (T_EARLY) load_seg(Trs, nseg - 1, Tt);
    // Only the last segment of a row can be short: every other one runs with its column count known at compile time, as one
    // straight run of instructions without a test per column.
    auto segment = [&](int s, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int x0 = s * HP_SEG;
        const int cols = FULL ? HP_SEG : min(HP_SEG, width1 - x0);
        if (!T_EARLY) load_seg(Trs, s, Tt);
        // replay L_lr over the segment
        v2s Lp[NP];
#pragma unroll
        for (int k = 0; k < NP; k++) Lp[k] = as_v2s(ckc[k]);
        int m = ckc[NP];
#pragma unroll
        for (int u = 0; u < HP_SEG; u++) {
#pragma unroll
            for (int k = 0; k < NP; k++) Lrec[u][k] = u == 0 ? Lp[k] : Lrec[u - 1][k];
            if (FULL || u < cols) m = path_step<NP>(Cc[u], Lrec[u], m, P1, P2, inf_pk); // steps in place on the column's own registers
        }
        // next segment's C: the half the replay needs first now, the rest once this segment's columns have been consumed
        if (s > 0) { load_cols(Crs, s - 1, Cn, 0, HP_SEG / 2); load_ck(s - 1, ckn); }
        // L_rl and S, right to left.  S of the segment then goes through LDS so that lane u of the row owns all D sums of
        // column u: the winner search runs once per column on one lane instead of 16 lanes co-operating on every column.
#pragma unroll
        for (int u = HP_SEG - 1; u >= 0; u--) {
            if (FULL || u < cols) { // uniform
                mR = path_step<NP>(Cc[u], LpR, mR, P1, P2, inf_pk);
#pragma unroll
                for (int k = 0; k < NP; k++) {
                    const int sum = as_int(__builtin_elementwise_add_sat(LpR[k], Lrec[u][k] + Tt[u][k]));
                    // cells of a padded range can never win (their inner sum may have wrapped: it is not looked at)
                    trow[u * STR + lane16 * NP + k] = lane16 * 2 * NP + 2 * k >= D ? 0x7fff7fff : sum;
                }
            }
            // column u of L_top has had its last use: its registers take the same column of the NEXT segment now, so that the
            // request has the rest of this loop, the winner search and the next replay to come back (a request made at the top
            // of the next segment had the replay alone, which is shorter than a trip to memory at full bandwidth)
            if (T_EARLY && s > 0) load_cols(Trs, s - 1, Tt, u, u + 1);
        }
        if (s > 0) load_cols(Crs, s - 1, Cn, HP_SEG / 2, HP_SEG);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); // same-wavefront exchange: program order in the LDS queue is enough
        {
            const int* col = trow + min(lane16, HP_SEG - 1) * STR;
            const int16_t* col16 = reinterpret_cast<const int16_t*>(col);
            v2s T[16 * NP];
#pragma unroll
            for (int j = 0; j < 16 * NP; j++) T[j] = as_v2s(col[j]);
            v2s mm = T[0];
#pragma unroll
            for (int j = 1; j < 16 * NP; j++) mm = vmin(mm, T[j]);
            const int minS = min((int)mm.x, (int)mm.y);
            int best = 1 << 20, cnt = 0;
#pragma unroll
            for (int j = 16 * NP - 1; j >= 0; j--) {
                if ((int)T[j].y == minS) { best = 2 * j + 1; cnt++; }
                if ((int)T[j].x == minS) { best = 2 * j; cnt++; }
            }
            if (tie_lanes > 0 && cnt > 1) {
                // vectorised-loop rule: per SIMD lane class the last tie, then the smallest of those
                int bb = 1 << 20;
                for (int c = 0; c < tie_lanes; c++) {
                    int last = -1;
                    for (int d = c; d < D; d += tie_lanes) if ((int)col16[d] == minS) last = d;
                    if (last >= 0) bb = min(bb, last);
                }
                best = bb;
            }
            if (active && lane16 < cols) {
                int d16 = best * 16;
                if (best > 0 && best < D - 1) {
                    const int sm = col16[best - 1], sp = col16[best + 1];
                    const int denom2 = max(sm + sp - 2 * minS, 1);
                    d16 += ((sm - sp) * 16 + denom2) / (denom2 * 2);
                }
                out[x0 + lane16] = (int)(((unsigned)minS << 16) | ((unsigned)d16 & 0xffffu));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int u = 0; u < HP_SEG; u++)
#pragma unroll
            for (int k = 0; k < NP; k++) Cc[u][k] = Cn[u][k];
#pragma unroll
        for (int k = 0; k <= NP; k++) ckc[k] = ckn[k];
    };
    segment(nseg - 1, std::false_type());
    for (int s = nseg - 2; s >= 0; s--) segment(s, std::true_type());
}

// ---------------------------------------------------------------- right-view disparity + consistency check + 3x3 median
// One block per LM_ROWS image rows.  The median of row y needs the checked rows y-1, y, y+1 (replicated at the image border),
// so the block runs the consistency check of its rows and of the row above and below them itself (a row's check is a scatter
// and a gather over W packed winners, far cheaper than a launch): per row, the right-view disparity by an order-independent
// LDS atomic-min scatter (smallest cost, then largest x), then the two-sided test.  The checked rows are also the pre-median map.
#define CSWAP(a, b) { int t_ = min(a, b); b = max(a, b); a = t_; }
constexpr int LM_ROWS = 4; // image rows per block: LM_ROWS + 2 rows are checked for them
__global__ __launch_bounds__(256) void lrmedian_kernel(const int32_t* __restrict__ wta, int W, int H, int width1, int srows,
                                                       StripeInfo si, int stripe_sz, int disp12, int16_t* __restrict__ disp_pre,
                                                       int16_t* __restrict__ dst, float* __restrict__ dstf)
{
    extern __shared__ unsigned lm_sm[]; // key[W], then LM_ROWS + 2 rows of checked disparities (int16, W rounded up to 2)
    unsigned* key = lm_sm;
    const int Wr = (W + 1) & ~1;
    int16_t* res = reinterpret_cast<int16_t*>(lm_sm + W);
    const int y0 = blockIdx.x * LM_ROWS, b = blockIdx.y;
    const int minX1 = W - width1;
    for (int t = 0; t < LM_ROWS + 2; t++) {
        if (y0 - 1 + t > H) break; // uniform: past the row below the block's last image row
        const int yy = min(max(y0 - 1 + t, 0), H - 1);
        const int stripe = min(yy / stripe_sz, 3);
        const int sr = si.off[stripe] + (yy - si.y0[stripe]);
        const int32_t* w = wta + ((size_t)b * srows + sr) * W;
        for (int x = threadIdx.x; x < W; x += 256) key[x] = 0xffffffffu;
        __syncthreads();
        for (int x = minX1 + threadIdx.x; x < W; x += 256) {
            const int v = w[x];
            const int minS = v >> 16, d16 = (short)(v & 0xffff);
            const int bestd = (d16 + 7) >> 4;
            if (minS != 32767)
                atomicMin(&key[x - bestd], ((unsigned)(minS + 32768) << 16) | (unsigned)(0xffff - x));
        }
        __syncthreads();
        int16_t* out = res + t * Wr;
        for (int x = threadIdx.x; x < W; x += 256) {
            int r = -16;
            if (x >= minX1) {
                const int d1 = (short)(w[x] & 0xffff);
                const int _d = d1 >> 4, d_ = (d1 + 15) >> 4;
                const int _x = x - _d, x_ = x - d_;
                bool bad = true;
                {
                    const unsigned k = key[_x]; // 0 <= _x: a disparity is below D <= minX1 <= x
                    const int d2 = k == 0xffffffffu ? -16 : (0xffff - (int)(k & 0xffff)) - _x;
                    bad = bad && d2 >= 0 && abs(d2 - _d) > disp12;
                }
                {
                    const bool inr = x_ >= 0;
                    const unsigned k = inr ? key[x_] : 0xffffffffu;
                    const int d2 = k == 0xffffffffu ? -16 : (0xffff - (int)(k & 0xffff)) - x_;
                    bad = bad && inr && d2 >= 0 && abs(d2 - d_) > disp12;
                }
                r = bad ? -16 : d1;
            }
            out[x] = (int16_t)r;
        }
        __syncthreads(); // the key array is reused by the next row; res is read below
    }
    for (int i = threadIdx.x; i < LM_ROWS * W; i += 256) {
        const int ry = i / W, x = i - ry * W;
        if (y0 + ry >= H) break; // rows ascend with i
        const size_t o0 = ((size_t)b * H + y0 + ry) * W;
        const int16_t* r0 = res + ry * Wr;
        const int x0 = max(x - 1, 0), x2 = min(x + 1, W - 1);
        int p0 = r0[x0], p1 = r0[x], p2 = r0[x2];
        int p3 = r0[Wr + x0], p4 = r0[Wr + x], p5 = r0[Wr + x2];
        int p6 = r0[2 * Wr + x0], p7 = r0[2 * Wr + x], p8 = r0[2 * Wr + x2];
        disp_pre[o0 + x] = (int16_t)p4;
        CSWAP(p1, p2); CSWAP(p4, p5); CSWAP(p7, p8); CSWAP(p0, p1); CSWAP(p3, p4); CSWAP(p6, p7);
        CSWAP(p1, p2); CSWAP(p4, p5); CSWAP(p7, p8); CSWAP(p0, p3); CSWAP(p5, p8); CSWAP(p4, p7);
        CSWAP(p3, p6); CSWAP(p1, p4); CSWAP(p2, p5); CSWAP(p4, p7); CSWAP(p4, p2); CSWAP(p6, p4);
        CSWAP(p4, p2);
        dst[o0 + x] = (int16_t)p4;
        dstf[o0 + x] = (float)p4 * (1.0f / 16.0f);
    }
}

__global__ __launch_bounds__(256) void depth_kernel(float* __restrict__ disp, size_t count, double fxb, double* __restrict__ depth)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float d = disp[i];
    if (d == 0.0f || d == -1.0f) { d = 0.1f; disp[i] = d; }
    depth[i] = fxb / (double)d;
}

template <int NP>
int run_sgbm(vo_ctx* c)
{
    const int W = c->W, H = c->H, D = c->Dp, w1 = c->width1, n = c->n, SH2 = c->p.block_size / 2; // D: the padded range (volume stride)
    const int Wp = vo_pre_pitch(c);
    StripeInfo si;
    for (int s = 0; s < 4; s++) { si.y0[s] = c->stripe_y0[s]; si.rows[s] = c->stripe_rows[s]; si.off[s] = c->stripe_row_off[s]; si.out0[s] = std::min(s * c->stripe_sz, H); }
    { StageTimer st_(c, VO_ST_PREFILTER);
    hipLaunchKernelGGL(prefilter_kernel, dim3(((W + PF_N - 1) / PF_N + PF_T - 1) / PF_T, (H + PF_R - 1) / PF_R, 2 * n), dim3(256), 0, c->stream,
                       c->img[0], c->img[1], c->pitch, c->fstride, W, H, Wp, c->ftzero, reinterpret_cast<uint4*>(c->pre));
    }
    { StageTimer st_(c, VO_ST_VPATH);
    // the reference's window (11x11, Source/VOClass.cpp:242) gets the instantiation with a compile-time half window
    const int swc = SH2 == 5 ? 5 : 0;
    {
        // 2 columns per lane group (512-thread blocks) for the wide disparity ranges, 4 (256-thread blocks) below
        constexpr int WN = NP >= 5 ? 2 : 4;
        typedef CostTile<NP, WN> WT;
        constexpr int TILE_COLS = WT::CV_COLS;
        const size_t lds = WT::lds_bytes();
        const void* fn = swc ? reinterpret_cast<const void*>(&cost_vpath_kernel<NP, 5, WN>) : reinterpret_cast<const void*>(&cost_vpath_kernel<NP, 0, WN>);
        static std::atomic<int> wide_attr[64][2];
        if (lds > 48 * 1024 && wide_attr[c->device & 63][swc ? 1 : 0].load(std::memory_order_acquire) < (int)lds) {
            VO_HIP(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            wide_attr[c->device & 63][swc ? 1 : 0].store((int)lds, std::memory_order_release);
        }
        const int tile_out = TILE_COLS - 2 * SH2;
        const dim3 grid((w1 + tile_out - 1) / tile_out, 4, n), blk(WT::THREADS);
        if (swc)
            hipLaunchKernelGGL((cost_vpath_kernel<NP, 5, WN>), grid, blk, lds, c->stream, reinterpret_cast<const uint2*>(c->pre), W, H, Wp, w1, D,
                               SH2, c->srows, si, c->P1, c->P2, c->cost, c->agg, c->D, c->dead_cost);
        else
            hipLaunchKernelGGL((cost_vpath_kernel<NP, 0, WN>), grid, blk, lds, c->stream, reinterpret_cast<const uint2*>(c->pre), W, H, Wp, w1, D,
                               SH2, c->srows, si, c->P1, c->P2, c->cost, c->agg, c->D, c->dead_cost);
    }
    }
    if (c->after_cost) VO_HIP(c, hipEventRecord(c->after_cost, c->stream));
    const int nseg = vo_hpath_segments(c);
    { StageTimer st_(c, VO_ST_HPATH_FWD);
    hipLaunchKernelGGL((hpath_ckpt_kernel<NP>), dim3((H + 15) / 16, n), dim3(256), 0, c->stream, c->cost, w1, D, H, c->srows, si, c->stripe_sz,
                       c->P1, c->P2, nseg, c->ckpt);
    }
    { StageTimer st_(c, VO_ST_HPATH_BWD);
    const size_t hp_lds = (size_t)4 * vo_hpath_seg(NP) * (16 * NP + 1) * sizeof(int);
    static std::atomic<int> hp_attr[64];
    if (hp_lds > 48 * 1024 && hp_attr[c->device & 63].load(std::memory_order_acquire) < (int)hp_lds) {
        VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&hpath_resolve_kernel<NP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)hp_lds));
        hp_attr[c->device & 63].store((int)hp_lds, std::memory_order_release);
    }
    hipLaunchKernelGGL((hpath_resolve_kernel<NP>), dim3((H + 3) / 4, n), dim3(64), hp_lds, c->stream, c->cost, c->agg, w1, D, W, H, c->srows, si,
                       c->stripe_sz, c->P1, c->P2, c->p.wta_simd_lanes, nseg, c->ckpt, c->wta, c->D);
    }
    { StageTimer st_(c, VO_ST_LRCHECK);
    const size_t lm_lds = (size_t)W * 4 + (size_t)(LM_ROWS + 2) * ((W + 1) & ~1) * 2;
    static std::atomic<int> lm_attr[64];
    if (lm_lds > 48 * 1024 && lm_attr[c->device & 63].load(std::memory_order_acquire) < (int)lm_lds) {
        VO_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&lrmedian_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm_lds));
        lm_attr[c->device & 63].store((int)lm_lds, std::memory_order_release);
    }
    hipLaunchKernelGGL(lrmedian_kernel, dim3((H + LM_ROWS - 1) / LM_ROWS, n), dim3(256), lm_lds, c->stream, c->wta, W, H, w1, c->srows, si, c->stripe_sz,
                       c->disp12, c->disp_pre, c->disp16, c->disp32);
    }
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

} // namespace

int launch_sgbm(vo_ctx* c)
{
    switch (c->Dp / 32) {
    case 1: return run_sgbm<1>(c);
    case 2: return run_sgbm<2>(c);
    case 3: return run_sgbm<3>(c);
    case 4: return run_sgbm<4>(c);
    case 5: return run_sgbm<5>(c);
    case 6: return run_sgbm<6>(c);
    case 7: return run_sgbm<7>(c);
    case 8: return run_sgbm<8>(c);
    }
    return vo_fail(c, VO_ERR_BAD_PARAM, "num_disparities must be a multiple of 16 in [16, 256]");
}

int launch_depth_map(vo_ctx* c)
{
    const size_t count = (size_t)c->n * c->W * c->H;
    const double fx = c->P0[0], baseline = c->P1m[3] / (-1 * fx);
    hipLaunchKernelGGL(depth_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, c->stream, c->disp32, count, fx * baseline, c->depth);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}
