// Image pyramids, Scharr derivatives, pyramidal Lucas-Kanade and the circular-match filter for gfx950.
// Replaces the four cv::calcOpticalFlowPyrLK calls and the filtering of VOClass::matchFeatureKLT
// (reference Source/VOClassFeatures.cpp:104-111,134-201; Source/VOClassUtils.cpp:61-78).
//
// Storage: every pyramid level is kept with a REFLECT_101 border of lk_win pixels (what
// buildOpticalFlowPyramid keeps), its Scharr derivative (Ix, Iy int16 pairs) with a zero border of the same
// width, so the tracker never branches on image edges.
//
//  level0_scharr_kernel  : copies an input image into its bordered level-0 slot and writes its 3x3 Scharr
//                          derivative (zero outside the image) in the same pass.
//  pyrdown_scharr_kernel : one launch per further level: 5x5 [1 4 6 4 1]^2 reduction ((sum + 128) >> 8) of the bordered
//                          previous level for a tile of the bordered new level (border pixels evaluate the position
//                          they reflect to), staged in LDS with a one-pixel halo, written with its Scharr derivative.
//  lk_row_kernel   : 16 lanes per feature (4 features per wavefront) run all four legs LT1->RT1->RT2->LT2->LT1
//                    and all pyramid levels; a lane owns one or two window rows, the 2x2 structure tensor and
//                    the mismatch vector are accumulated in integers (order independent) with 16-bit dot
//                    products and reduced with DPP row adds; the 2x2 solve is replicated per lane in f32
//                    without contraction.
//  filter_kernel   : bounds filter per leg, 4-way status AND, loop-closure test, ordered compaction.
#include "vo_ctx.h"
#include <string.h>
#include <algorithm>

namespace {

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

struct LevelArgs { int w, h, pitch, rows; unsigned img_off, der_off; };
// img_per_b: image slots per batch entry, image (b, which) is slot b * img_per_b + which.  4 for independent pairs
// (LT1, RT1, LT2, RT2); 2 for consecutive frames of a sequence, where (b, LT2) is the same image as (b + 1, LT1).
struct PyrArgs { int nlevels, border, img_per_b; LevelArgs lv[VO_MAX_LEVELS]; unsigned long long pyr_bytes, der_elems; };

// Thread layout of the per-pixel kernels below: a block is 64 threads wide x 4 rows, a thread owns 4 adjacent pixels of a
// row (levels are stored with a pitch that is a multiple of 16, so that is one aligned dword of the level).
constexpr int PX_T = 64, PX_R = 4, PX_N = 4;

// Level 0 and its derivative in one pass over the bordered level: a thread owns 4 adjacent pixels of a row, reads the 3 x 6
// source pixels around them (reflected at the image edge, which is what the level's own border holds) and writes 4 image
// bytes and 4 (Ix, Iy) pairs.
__device__ __forceinline__ void scharr4(const int (&c)[3][PX_N + 2], int px0, int border, int w, bool row_inside, int4& o)
{
    int v[PX_N];
#pragma unroll
    for (int j = 0; j < PX_N; j++) {
        const int x = px0 + j - border;
        const int ix = ((c[0][j + 2] + c[2][j + 2]) * 3 + c[1][j + 2] * 10) - ((c[0][j] + c[2][j]) * 3 + c[1][j] * 10);
        const int iy = ((c[2][j] - c[0][j]) + (c[2][j + 2] - c[0][j + 2])) * 3 + (c[2][j + 1] - c[0][j + 1]) * 10;
        v[j] = (row_inside && x >= 0 && x < w) ? ((ix & 0xffff) | (iy << 16)) : 0;
    }
    o = make_int4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void level0_scharr_kernel(const uint8_t* i0, const uint8_t* i1, const uint8_t* i2, const uint8_t* i3,
                                                            size_t pitch, size_t fstride, PyrArgs pa, uint8_t* __restrict__ pyr,
                                                            int16_t* __restrict__ der)
{
    const int px0 = (blockIdx.x * PX_T + (threadIdx.x & (PX_T - 1))) * PX_N, py = blockIdx.y * PX_R + (threadIdx.x / PX_T);
    const int b = blockIdx.z / pa.img_per_b, which = blockIdx.z % pa.img_per_b;
    const LevelArgs L = pa.lv[0];
    if (px0 >= L.pitch || py >= L.rows) return;
    const uint8_t* src = (which == 0 ? i0 : which == 1 ? i1 : which == 2 ? i2 : i3) + (size_t)b * fstride;
    const int y = py - pa.border;
    int c[3][PX_N + 2]; // bordered columns px0-1 .. px0+4 of rows py-1, py, py+1
    const int xl = px0 - 1 - pa.border; // image column of the first of the six
    if (xl >= 0 && xl + PX_N + 1 < L.w && y >= 1 && y + 1 < L.h) { // nothing to reflect: the common case
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint8_t* row = src + (size_t)(y - 1 + r) * pitch + xl;
#pragma unroll
            for (int i = 0; i < PX_N + 2; i++) c[r][i] = row[i];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint8_t* row = src + (size_t)reflect101(y - 1 + r, L.h) * pitch;
#pragma unroll
            for (int i = 0; i < PX_N + 2; i++)
                c[r][i] = row[reflect101(min(px0 - 1 + i, L.w + 2 * pa.border - 1) - pa.border, L.w)];
        }
    }
    const unsigned v = (unsigned)c[1][1] | ((unsigned)c[1][2] << 8) | ((unsigned)c[1][3] << 16) | ((unsigned)c[1][4] << 24);
    *reinterpret_cast<unsigned*>(pyr + (size_t)blockIdx.z * pa.pyr_bytes + L.img_off + (size_t)py * L.pitch + px0) = v;
    int4 o;
    scharr4(c, px0, pa.border, L.w, y >= 0 && y < L.h, o);
    int* out = reinterpret_cast<int*>(der + (size_t)blockIdx.z * pa.der_elems + L.der_off);
    *reinterpret_cast<int4*>(out + (size_t)py * L.pitch + px0) = o;
}

// One pyramid level from the previous one, border and derivative included, in one launch.  A block owns a PD_TX x PD_TY
// tile of the BORDERED level.  The value of a bordered pixel is the 5x5 [1 4 6 4 1] x [1 4 6 4 1] reduction at the position
// it reflects to inside the level ((sum + 128) >> 8; the taps read the bordered previous level, so they need no edge
// cases); the tile is evaluated with a one-pixel halo into LDS, then written out together with its Scharr derivative.
constexpr int PD_TX = 64, PD_TY = 4;

__global__ __launch_bounds__(256) void pyrdown_scharr_kernel(PyrArgs pa, int level, uint8_t* __restrict__ pyr, int16_t* __restrict__ der)
{
    __shared__ int val[PD_TY + 2][PD_TX + 3]; // rows py0-1 .. py0+PD_TY, bordered columns px_tile-1 .. px_tile+PD_TX
    const LevelArgs S = pa.lv[level - 1], Dl = pa.lv[level];
    uint8_t* base = pyr + (size_t)blockIdx.z * pa.pyr_bytes;
    const uint8_t* src = base + S.img_off + (size_t)pa.border * S.pitch + pa.border; // interior origin of the previous level
    const int pxt = blockIdx.x * PD_TX, py0 = blockIdx.y * PD_TY;
    for (int i = threadIdx.x; i < (PD_TY + 2) * (PD_TX + 2); i += 256) {
        const int ry = i / (PD_TX + 2), rx = i - ry * (PD_TX + 2);
        const int x = reflect101(min(pxt - 1 + rx, Dl.w + 2 * pa.border - 1) - pa.border, Dl.w);
        const int y = reflect101(py0 - 1 + ry - pa.border, Dl.h);
        const uint8_t* p = src + (ptrdiff_t)(2 * y - 2) * S.pitch + (2 * x - 2);
        int s = 0;
#pragma unroll
        for (int dy = 0; dy < 5; dy++) {
            const int wy = dy == 0 || dy == 4 ? 1 : dy == 2 ? 6 : 4;
            const uint8_t* q = p + dy * S.pitch;
            s += wy * (q[0] + 4 * q[1] + 6 * q[2] + 4 * q[3] + q[4]);
        }
        val[ry][rx] = (s + 128) >> 8;
    }
    __syncthreads();
    // 64 threads x 4 rows; a thread owns one pixel column of the tile here (the tile is 64 wide): 4 rows x 64 columns
    const int tx = threadIdx.x & (PD_TX - 1), ty = threadIdx.x / PD_TX;
    const int px = pxt + tx, py = py0 + ty;
    if (px >= Dl.pitch || py >= Dl.rows) return;
    base[Dl.img_off + (size_t)py * Dl.pitch + px] = (uint8_t)val[ty + 1][tx + 1];
    const int x = px - pa.border, y = py - pa.border;
    int d = 0;
    if (x >= 0 && x < Dl.w && y >= 0 && y < Dl.h) {
        const int a0 = val[ty][tx], a1 = val[ty][tx + 1], a2 = val[ty][tx + 2];
        const int b0 = val[ty + 1][tx], b2 = val[ty + 1][tx + 2];
        const int c0 = val[ty + 2][tx], c1 = val[ty + 2][tx + 1], c2 = val[ty + 2][tx + 2];
        const int ix = ((a2 + c2) * 3 + b2 * 10) - ((a0 + c0) * 3 + b0 * 10);
        const int iy = ((c0 - a0) + (c2 - a2)) * 3 + (c1 - a1) * 10;
        d = (ix & 0xffff) | (iy << 16);
    }
    reinterpret_cast<int*>(der + (size_t)blockIdx.z * pa.der_elems + Dl.der_off)[(size_t)py * Dl.pitch + px] = d;
}

#define DPP_ADD(v, ctrl) ((v) + __builtin_amdgcn_update_dpp(0, (v), (ctrl), 0xF, 0xF, false))
__device__ __forceinline__ int cv_floor(float v) { return (int)floorf(v); }
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }

struct LkArgs {
    PyrArgs pa;
    int win, max_count, max_features;
    int slots_per_frame, n_frames;  // blocks an XCD is given per frame (one eighth of the frame's feature groups); frames in the batch
    float min_eig;
    double eps2;
};

__device__ __forceinline__ void bilinear_weights(float a, float b, int& w00, int& w01, int& w10, int& w11)
{
    w00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << 14));
    w01 = cv_round(a * (1.f - b) * (float)(1 << 14));
    w10 = cv_round((1.f - a) * b * (float)(1 << 14));
    w11 = (1 << 14) - w00 - w01 - w10;
}

// ---------------------------------------------------------------- tracker, 16 lanes per feature
// One DPP row (16 lanes) per feature, 4 features per wavefront, 8 per block; the four features of a wavefront
// walk legs and levels in lockstep and iterate until the slowest has converged (lanes of finished features are
// masked off).  A lane owns NR consecutive window rows and fetches each with one NDW-dword load; the source row
// below its last one is the next lane's first row and arrives by a DPP row shift, so a group loads every source
// row once.  Pixels are kept as 16-bit pairs (p[x], p[x+1]) and weights as 16-bit pairs, so one bilinear sample
// is two v_dot2_i32_i16; the rounding constant and the template value ride in the accumulator operand:
//     diff = (J00*w00 + J01*w01 + J10*w10 + J11*w11 + 2^8 - (I << 9)) >> 9        (exact: I << 9 is a multiple of 2^9)
// The second product of a sample reads its pixel pairs from the next lane through the DPP form of the accumulating dot product.
// The mismatch vector uses dot2 on pairs of diffs against pairs of Ix (Iy).  Sums are integers, so the order of
// accumulation is free; the 16-lane reduction is 4 DPP row adds on 16-bit halves, recombined exactly in f64.
typedef short lk_v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int dot2(int a, int b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_v2s, a), __builtin_bit_cast(lk_v2s, b), c, false);
}
// first product of a chain whose addend must stay live: the saturating form never saturates here (|sum| < 2^31) but
// it is the three-operand encoding, so the addend needs no copy
__device__ __forceinline__ int dot2_keep(int a, int b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(lk_v2s, a), __builtin_bit_cast(lk_v2s, b), c, true);
}
// acc[i] += dot2(src[i] of the NEXT lane, w) for i in [I0, I0 + 5): the DPP form of the accumulating dot product reads its first
// operand from the neighbouring lane directly (row_shl:1: lane i takes lane i + 1's value, 0 for the last lane of a row of 16,
// which owns no live window row that needs it), which saves the separate cross-lane move per operand.  The compiler does not form this instruction itself, hence the assembly.  A DPP operand written by the vector
// instruction just before it needs two idle issue slots: the leading s_nop provides them whatever was scheduled before the
// block (the hazard logic does not look inside assembly text).
template <int N, int I0>
__device__ __forceinline__ void dot2c_next_lane(int (&acc)[N], const int (&src)[N], int w)
{
    if constexpr (I0 + 5 <= N) {
        asm("s_nop 1\n\t"
            "v_dot2c_i32_i16_dpp %0, %5, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_dot2c_i32_i16_dpp %1, %6, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_dot2c_i32_i16_dpp %2, %7, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_dot2c_i32_i16_dpp %3, %8, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
            "v_dot2c_i32_i16_dpp %4, %9, %10 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "+v"(acc[I0]), "+v"(acc[I0 + 1]), "+v"(acc[I0 + 2]), "+v"(acc[I0 + 3]), "+v"(acc[I0 + 4])
            : "v"(src[I0]), "v"(src[I0 + 1]), "v"(src[I0 + 2]), "v"(src[I0 + 3]), "v"(src[I0 + 4]), "v"(w));
        dot2c_next_lane<N, I0 + 5>(acc, src, w);
    } else if constexpr (I0 < N) {
        asm("s_nop 1\n\t"
            "v_dot2c_i32_i16_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
            : "+v"(acc[I0]) : "v"(src[I0]), "v"(w));
        dot2c_next_lane<N, I0 + 1>(acc, src, w);
    }
}
__device__ __forceinline__ int pack_lo16(int lo, int hi) { return (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }

// (lo >> 9, hi >> 9) as packed 16-bit halves, for |lo|, |hi| < 2^23
__device__ __forceinline__ int pair_shr9(int lo, int hi)
{
    typedef short v2s16 __attribute__((ext_vector_type(2)));
    const v2s16 t = __builtin_bit_cast(v2s16, (int)__builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x06050201u));
    return __builtin_bit_cast(int, t >> (short)1);
}

// hi * 2^16 + lo as the float nearest to that integer, for |hi|, |lo| < 2^24: both convert exactly and the fused
// multiply-add rounds the exact sum once, which is what rounding the 64-bit integer would give
__device__ __forceinline__ float sum_of_halves(int hi, int lo) { return __builtin_fmaf((float)hi, 65536.f, (float)lo); }

// exact sum over the 16 lanes of a group, as the float nearest to the integer (every lane gets it)
__device__ __forceinline__ float group_sum(int v)
{
    int lo = v & 0xffff, hi = v >> 16;
    lo = DPP_ADD(lo, 0xB1); lo = DPP_ADD(lo, 0x4E); lo = DPP_ADD(lo, 0x141); lo = DPP_ADD(lo, 0x140);
    hi = DPP_ADD(hi, 0xB1); hi = DPP_ADD(hi, 0x4E); hi = DPP_ADD(hi, 0x141); hi = DPP_ADD(hi, 0x140);
    return sum_of_halves(hi, lo);
}

// two sums at once: four independent DPP chains, stepped together so no step waits on the previous one
// QUAD32: the per-lane values are small enough (|value| < 2^29) for the first two steps, which add four lanes, to run on
// the whole 32-bit values; only the last two steps then need the split halves
template <bool QUAD32>
__device__ __forceinline__ void group_sum2(int u, int v, float& su, float& sv)
{
    if (QUAD32) {
        int t0 = __builtin_amdgcn_update_dpp(0, u, 0xB1, 0xF, 0xF, false), t1 = __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
        u += t0; v += t1;
        t0 = __builtin_amdgcn_update_dpp(0, u, 0x4E, 0xF, 0xF, false); t1 = __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
        u += t0; v += t1;
    }
    int c[4] = {u & 0xffff, u >> 16, v & 0xffff, v >> 16};
#pragma unroll
    for (int s = QUAD32 ? 2 : 0; s < 4; s++) { // the DPP control must be a literal, hence the chain of selects on the unrolled step
        int t[4];
#pragma unroll
        for (int q = 0; q < 4; q++)
            t[q] = s == 0 ? __builtin_amdgcn_update_dpp(0, c[q], 0xB1, 0xF, 0xF, false)
                 : s == 1 ? __builtin_amdgcn_update_dpp(0, c[q], 0x4E, 0xF, 0xF, false)
                 : s == 2 ? __builtin_amdgcn_update_dpp(0, c[q], 0x141, 0xF, 0xF, false)
                          : __builtin_amdgcn_update_dpp(0, c[q], 0x140, 0xF, 0xF, false);
#pragma unroll
        for (int q = 0; q < 4; q++) c[q] += t[q];
    }
    su = sum_of_halves(c[1], c[0]);
    sv = sum_of_halves(c[3], c[2]);
}

// the 4*NDW-1 pairs (p[x], p[x+1]) of one source row, each as two zero-extended 16-bit halves
template <int NDW>
__device__ __forceinline__ void pixel_pairs(const uint8_t* __restrict__ p, int (&pr)[4 * NDW - 1])
{
    unsigned w[NDW];
    __builtin_memcpy(w, p, 4 * NDW);
#pragma unroll
    for (int px = 0; px < 4 * NDW - 1; px++) {
        const int q = px >> 2, o = px & 3, q1 = q + 1 < NDW ? q + 1 : q;
        pr[px] = (int)__builtin_amdgcn_perm(o == 3 ? w[q1] : w[q], w[q], 0x0c000c00u | (unsigned)o | ((unsigned)(o + 1) << 16));
    }
}
// the same for a derivative row: pairs (Ix[x], Ix[x+1]) and (Iy[x], Iy[x+1])
template <int NDW>
__device__ __forceinline__ void deriv_pairs(const int* __restrict__ p, int (&dx)[4 * NDW - 1], int (&dy)[4 * NDW - 1])
{
    unsigned e[4 * NDW];
    __builtin_memcpy(e, p, 16 * NDW);
#pragma unroll
    for (int px = 0; px < 4 * NDW - 1; px++) {
        dx[px] = (int)__builtin_amdgcn_perm(e[px + 1], e[px], 0x05040100u);
        dy[px] = (int)__builtin_amdgcn_perm(e[px + 1], e[px], 0x07060302u);
    }
}

// Two wavefronts (8 features) per block: the wavefronts of a block share nothing, and a block starts only when every one of
// its wavefronts has a slot, so smaller blocks fill the slots that finishing wavefronts leave sooner (256 threads: 3.08 ms
// per 42 frames, 128: 2.82, 64: 2.83 with more blocks to dispatch).
constexpr int LK_THREADS = 128;
template <int NR, int NDW>
__global__ __launch_bounds__(LK_THREADS) void lk_row_kernel(LkArgs a, const uint8_t* __restrict__ pyr, const int16_t* __restrict__ der,
                                                     const vo_point2f* __restrict__ feat, const int* __restrict__ n_feat,
                                                     vo_point2f* __restrict__ legs, uint8_t* __restrict__ leg_status)
{
    constexpr int NPX = 4 * NDW - 1;   // pixel slots per window row (win <= NPX)
    constexpr int NPP = 2 * NDW;       // slot pairs per window row
    const int lane16 = threadIdx.x & 15;
    // Blocks are dealt round-robin over the 8 XCDs, each with its own L2.  A frame's feature groups (one per block) are cut into runs of
    // LK_RUN consecutive groups and the runs are dealt to the XCDs in turn: neighbouring features (row-major order) read
    // overlapping window rows of the same pyramids, so a run shares its rows in one L2, and every XCD gets the same share of
    // every frame and of every image region, whatever the number of frames and however many features each frame has.
    constexpr int FPB = LK_THREADS / 16, LK_RUN = 64 / FPB; // features per block; a run is 64 features
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3);
    const int b = slot / a.slots_per_frame, t = slot % a.slots_per_frame;
    if (b >= a.n_frames) return;
    const int nf = n_feat[b];
    const int groups = (nf + FPB - 1) / FPB;
    const int fgroup = ((t / LK_RUN) * 8 + xcd) * LK_RUN + t % LK_RUN;
    if (fgroup >= groups) return;
    const int i = fgroup * FPB + (threadIdx.x >> 4);
    if (fgroup * FPB + (int)(threadIdx.x >> 6) * 4 >= nf) return; // uniform per wavefront
    const bool have = i < nf;
    const int win = a.win;
    const float half = (win - 1) * 0.5f;
    const int row0 = lane16 * NR;
    const int leg_prev[4] = {0, 1, 3, 2}, leg_next[4] = {1, 3, 2, 0};
    vo_point2f prev0 = feat[(size_t)b * a.max_features + min(i, nf - 1)];
    const int nl = a.pa.nlevels;
    const float FLT_SCALE = 1.f / (float)(1 << 20);
    const float eps2_lo = (float)(a.eps2 * (1.0 - 1e-6)), eps2_hi = (float)(a.eps2 * (1.0 + 1e-6));

    for (int leg = 0; leg < 4; leg++) {
        const uint8_t* Ibase = pyr + ((size_t)b * a.pa.img_per_b + leg_prev[leg]) * a.pa.pyr_bytes;
        const uint8_t* Jbase = pyr + ((size_t)b * a.pa.img_per_b + leg_next[leg]) * a.pa.pyr_bytes;
        const int16_t* Dbase = der + ((size_t)b * a.pa.img_per_b + leg_prev[leg]) * a.pa.der_elems;
        float nx = 0.f, ny = 0.f;
        int status = 1;
        for (int level = nl; level >= 0; level--) {
            const LevelArgs L = a.pa.lv[level];
            const int P = L.pitch;
            const float sc = 1.f / (float)(1 << level);
            float px = prev0.x * sc, py = prev0.y * sc;
            float qx, qy;
            if (level == nl) { qx = px; qy = py; }
            else { qx = nx * 2.f; qy = ny * 2.f; }
            nx = qx; ny = qy;
            px -= half; py -= half;
            const int ipx = cv_floor(px), ipy = cv_floor(py);
            bool run = have;
            if (ipx < -win || ipx >= L.w || ipy < -win || ipy >= L.h) {
                if (level == 0) status = 0;
                run = false;
            }
            int rowoff[NR];
#pragma unroll
            for (int k = 0; k < NR; k++) rowoff[k] = min(row0 + k, win) * P;
            int C0[NR][NPX], IxP[NR][NPP], IyP[NR][NPP];
            float A11 = 0.f, A12 = 0.f, A22 = 0.f, Dt = 0.f;
            if (run) {
                int w00, w01, w10, w11;
                bilinear_weights(px - ipx, py - ipy, w00, w01, w10, w11);
                const int w0 = (w00 & 0xffff) | (w01 << 16), w1 = (w10 & 0xffff) | (w11 << 16);
                const uint8_t* I = Ibase + L.img_off + (size_t)(ipy + a.pa.border) * P + (ipx + a.pa.border);
                const int* Dv = reinterpret_cast<const int*>(Dbase + L.der_off) + (size_t)(ipy + a.pa.border) * P + (ipx + a.pa.border);
                int pr[NR][NPX], dx[NR][NPX], dy[NR][NPX];
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    pixel_pairs<NDW>(I + rowoff[k], pr[k]);
                    deriv_pairs<NDW>(Dv + rowoff[k], dx[k], dy[k]);
                }
                int s11 = 0, s12 = 0, s22 = 0;
#pragma unroll
                for (int k = 0; k < NR; k++) {
                    const bool live = row0 + k < win;
                    const int wd0 = live ? w0 : 0, wd1 = live ? w1 : 0; // a dead row gets zero derivatives
                    int ivv[NPX], ixv[NPX + 1], iyv[NPX + 1];
#pragma unroll
                    for (int x = 0; x < NPX; x++) {
                        ivv[x] = dot2_keep(pr[k][x], w0, 1 << 8);
                        ixv[x] = dot2_keep(dx[k][x], wd0, 1 << 13);
                        iyv[x] = dot2_keep(dy[k][x], wd0, 1 << 13);
                    }
                    // the source row below a lane's last window row is the next lane's first row
                    if (k + 1 < NR) {
#pragma unroll
                        for (int x = 0; x < NPX; x++) {
                            ivv[x] = dot2(pr[k + 1][x], w1, ivv[x]); ixv[x] = dot2(dx[k + 1][x], wd1, ixv[x]); iyv[x] = dot2(dy[k + 1][x], wd1, iyv[x]);
                        }
                    } else {
                        dot2c_next_lane<NPX, 0>(ivv, pr[0], w1);
                        dot2c_next_lane<NPX, 0>(reinterpret_cast<int (&)[NPX]>(ixv), dx[0], wd1);
                        dot2c_next_lane<NPX, 0>(reinterpret_cast<int (&)[NPX]>(iyv), dy[0], wd1);
                    }
#pragma unroll
                    for (int x = 0; x < NPX; x++) {
                        C0[k][x] = (1 << 8) - ((ivv[x] >> 9) << 9);
                        ixv[x] >>= 14; iyv[x] >>= 14;
                    }
                    ixv[NPX] = 0; iyv[NPX] = 0;
#pragma unroll
                    for (int p = 0; p < NPP; p++) {
                        const int m = 2 * p + 1 < win ? -1 : (2 * p < win ? 0xffff : 0); // slots past the window are zero
                        IxP[k][p] = pack_lo16(ixv[2 * p], ixv[2 * p + 1]) & m;
                        IyP[k][p] = pack_lo16(iyv[2 * p], iyv[2 * p + 1]) & m;
                        s11 = dot2(IxP[k][p], IxP[k][p], s11); s12 = dot2(IxP[k][p], IyP[k][p], s12); s22 = dot2(IyP[k][p], IyP[k][p], s22);
                    }
                }
                A11 = group_sum(s11) * FLT_SCALE; A12 = group_sum(s12) * FLT_SCALE; A22 = group_sum(s22) * FLT_SCALE;
                Dt = A11 * A22 - A12 * A12;
                const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (float)(2 * win * win);
                if (minEig < a.min_eig || Dt < 1.1920929e-07f) {
                    if (level == 0) status = 0;
                    run = false;
                }
                Dt = 1.f / Dt;
            }
            qx -= half; qy -= half;
            float pdx = 0.f, pdy = 0.f;
            const uint8_t* Jl = Jbase + L.img_off + (size_t)a.pa.border * P + a.pa.border;
            bool it = run && a.max_count > 0;
            int j = 0;
            while (__any(it)) {
                if (it) {
                    const int inx = cv_floor(qx), iny = cv_floor(qy);
                    if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) {
                        if (level == 0) status = 0;
                        it = false;
                    } else {
                        int w00, w01, w10, w11;
                        bilinear_weights(qx - inx, qy - iny, w00, w01, w10, w11);
                        const int w0 = (w00 & 0xffff) | (w01 << 16), w1 = (w10 & 0xffff) | (w11 << 16);
                        const uint8_t* J = Jl + (iny * P + inx); // a level is far below 2^31 bytes: 32-bit offset arithmetic
                        int pr[NR][NPX];
#pragma unroll
                        for (int k = 0; k < NR; k++) pixel_pairs<NDW>(J + rowoff[k], pr[k]);
                        int sb1 = 0, sb2 = 0;
#pragma unroll
                        for (int k = 0; k < NR; k++) {
                            int d[NPX + 1];
                            // the source row below a lane's last window row is the next lane's first row
#pragma unroll
                            for (int x = 0; x < NPX; x++) d[x] = dot2_keep(pr[k][x], w0, C0[k][x]);
                            if (k + 1 < NR) {
#pragma unroll
                                for (int x = 0; x < NPX; x++) d[x] = dot2(pr[k + 1][x], w1, d[x]);
                            } else {
                                int (&dd)[NPX] = reinterpret_cast<int (&)[NPX]>(d);
                                dot2c_next_lane<NPX, 0>(dd, pr[0], w1);
                            }
                            d[NPX] = 0;
#pragma unroll
                            for (int p = 0; p < NPP; p++) {
                                // (d >> 9) of two slots as a 16-bit pair: |d| < 2^23, so bits 8..23 hold d >> 8 exactly and one
                                // packed arithmetic shift finishes the job.  The last pair has no upper slot: its Ix/Iy halves
                                // are zero, whatever the upper half holds.
                                const int dp = pair_shr9(d[2 * p], d[2 * p + 1]);
                                sb1 = dot2(dp, IxP[k][p], sb1); sb2 = dot2(dp, IyP[k][p], sb2);
                            }
                        }
                        float b1, b2;
                        // one window row of at most 15 products per lane, each below 2^25: four lanes stay below 2^31
                        group_sum2<NR == 1 && NDW <= 4>(sb1, sb2, b1, b2);
                        b1 *= FLT_SCALE; b2 *= FLT_SCALE;
                        const float dx_ = (A12 * b2 - A22 * b1) * Dt, dy_ = (A12 * b1 - A11 * b2) * Dt;
                        qx += dx_; qy += dy_;
                        nx = qx + half; ny = qy + half;
                        // |delta|^2 <= eps^2 is a double-precision test in the original.  The single-precision value is within
                        // 2^-22 of it, so only a value inside that band around the threshold needs the exact test, and the
                        // wavefront runs it only if one of its lanes is there.
                        const float d2 = __builtin_fmaf(dx_, dx_, dy_ * dy_);
                        bool small = d2 < eps2_lo;
                        const bool unsure = !small && !(d2 > eps2_hi);
                        if (__any(unsure)) small = small || (unsure && (double)dx_ * (double)dx_ + (double)dy_ * (double)dy_ <= a.eps2);
                        // (double)|x| < 0.01 holds exactly for the floats up to 0.01f, the float just below 0.01
                        if (small) it = false;
                        else if (j > 0 && fabsf(dx_ + pdx) <= 0.01f && fabsf(dy_ + pdy) <= 0.01f) {
                            nx -= dx_ * 0.5f; ny -= dy_ * 0.5f;
                            it = false;
                        }
                        pdx = dx_; pdy = dy_;
                    }
                }
                j++;
                if (j >= a.max_count) it = false;
            }
            if (status && level == 0) {
                // the error pass of the original re-tests the final window position and can clear the status
                const int inx = cv_floor(nx - half), iny = cv_floor(ny - half);
                if (inx < -win || inx >= L.w || iny < -win || iny >= L.h) status = 0;
            }
        }
        if (have && lane16 == 0) {
            vo_point2f o; o.x = nx; o.y = ny;
            legs[((size_t)b * 4 + leg) * a.max_features + i] = o;
            leg_status[((size_t)b * 4 + leg) * a.max_features + i] = (uint8_t)status;
        }
        prev0.x = nx; prev0.y = ny;
    }
}

// ---------------------------------------------------------------- circular-match filter
__global__ __launch_bounds__(256) void filter_kernel(const vo_point2f* __restrict__ feat, const int* __restrict__ n_feat, int max_features,
                                                     float fw, float fh, int loop_thr, const vo_point2f* __restrict__ legs,
                                                     uint8_t* __restrict__ leg_status, vo_point2f* __restrict__ m1, vo_point2f* __restrict__ m2,
                                                     int* __restrict__ n_match)
{
    // One 256-thread block per frame (a wider block would wait for a whole CU to drain while other batches' kernels hold
    // the wavefront slots).  A trip takes 1024 features, four per thread, so that their loads are in flight together.
    __shared__ int wcnt[4][4];
    const int b = blockIdx.x, n = n_feat[b];
    const size_t F = max_features;
    const vo_point2f* f0 = feat + (size_t)b * F;
    const vo_point2f* lg = legs + (size_t)b * 4 * F;
    uint8_t* st = leg_status + (size_t)b * 4 * F;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        bool keep[4];
        vo_point2f p3[4], p2[4];
        unsigned long long mk[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = i0 + k * 256 + threadIdx.x;
            keep[k] = false;
            p3[k] = {0.f, 0.f}; p2[k] = {0.f, 0.f};
            if (i < n) {
                int ok = 1;
#pragma unroll
                for (int leg = 0; leg < 4; leg++) {
                    vo_point2f q = lg[(size_t)leg * F + i];
                    int s = st[(size_t)leg * F + i];
                    if (q.x < 0.f || q.x > fw || q.y < 0.f || q.y > fh) { s = 0; st[(size_t)leg * F + i] = 0; }
                    ok &= (s == 1);
                    if (leg == 2) p2[k] = q;
                    if (leg == 3) p3[k] = q;
                }
                vo_point2f o = f0[i];
                int offset = (int)fmaxf(fabsf(o.x - p3[k].x), fabsf(o.y - p3[k].y));
                keep[k] = ok && offset < loop_thr;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            mk[k] = __ballot(keep[k]);
            if (lane == 0) wcnt[k][wave] = __popcll(mk[k]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) { // chunk k holds features i0 + 256 k ..: ranks follow the feature order
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { before += w < wave ? wcnt[k][w] : 0; total += wcnt[k][w]; }
            if (keep[k]) {
                int r = base + before + __popcll(mk[k] & ((1ull << lane) - 1ull));
                m1[(size_t)b * F + r] = p3[k];
                m2[(size_t)b * F + r] = p2[k];
            }
            base += total;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_match[b] = base;
}

} // namespace

static PyrArgs make_pyr_args(const vo_ctx* c)
{
    PyrArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.nlevels = c->nlevels; pa.border = c->border; pa.img_per_b = c->seq_mode ? 2 : 4; pa.pyr_bytes = c->pyr_bytes; pa.der_elems = c->der_elems;
    for (int l = 0; l <= c->nlevels; l++) {
        pa.lv[l].w = c->lv[l].w; pa.lv[l].h = c->lv[l].h; pa.lv[l].pitch = c->lv[l].pitch; pa.lv[l].rows = c->lv[l].rows;
        pa.lv[l].img_off = (unsigned)c->lv[l].img_off; pa.lv[l].der_off = (unsigned)c->lv[l].der_off;
    }
    return pa;
}

int launch_pyramids(vo_ctx* c)
{
    StageTimer st_(c, VO_ST_PYRAMID);
    const PyrArgs pa = make_pyr_args(c);
    const int nimg = c->seq_mode ? 2 * (c->n + 1) : 4 * c->n;
    hipLaunchKernelGGL(level0_scharr_kernel, dim3((pa.lv[0].pitch / PX_N + PX_T - 1) / PX_T, (pa.lv[0].rows + PX_R - 1) / PX_R, nimg), dim3(256), 0,
                       c->stream, c->img[0], c->img[1], c->img[2], c->img[3], c->pitch, c->fstride, pa, c->pyr, c->der);
    for (int l = 1; l <= c->nlevels; l++)
        hipLaunchKernelGGL(pyrdown_scharr_kernel, dim3((pa.lv[l].pitch + PD_TX - 1) / PD_TX, (pa.lv[l].rows + PD_TY - 1) / PD_TY, nimg), dim3(256), 0,
                           c->stream, pa, l, c->pyr, c->der);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

int launch_match_klt(vo_ctx* c, bool counts_on_host)
{
    LkArgs a;
    a.pa = make_pyr_args(c);
    a.win = c->p.lk_win;
    a.max_count = std::min(std::max(c->p.lk_max_count, 0), 100);
    a.max_features = c->p.max_features;
    a.min_eig = (float)c->p.lk_min_eig;
    double eps = std::min(std::max(c->p.lk_epsilon, 0.), 10.);
    a.eps2 = eps * eps;
    // without host-side counts the grid covers the capacity; wavefronts past a frame's feature count exit at once
    int nmax = counts_on_host ? 0 : c->p.max_features;
    if (counts_on_host) for (int b = 0; b < c->n; b++) nmax = std::max(nmax, c->h_nfeat[b]);
    if (nmax > 0) {
        StageTimer st_(c, VO_ST_LK);
        constexpr int fpb = LK_THREADS / 16, run = 64 / fpb; // features per block, blocks per run of 64 features
        a.slots_per_frame = (((nmax + fpb - 1) / fpb + 8 * run - 1) / (8 * run)) * run; a.n_frames = c->n; // runs dealt over 8 XCDs
        dim3 g16(8 * a.slots_per_frame * c->n), blk16(LK_THREADS);
        if (a.win <= 15) hipLaunchKernelGGL((lk_row_kernel<1, 4>), g16, blk16, 0, c->stream, a, c->pyr, c->der, c->feat, c->n_feat, c->legs, c->leg_status);
        else if (a.win <= 23) hipLaunchKernelGGL((lk_row_kernel<2, 6>), g16, blk16, 0, c->stream, a, c->pyr, c->der, c->feat, c->n_feat, c->legs, c->leg_status);
        else hipLaunchKernelGGL((lk_row_kernel<2, 8>), g16, blk16, 0, c->stream, a, c->pyr, c->der, c->feat, c->n_feat, c->legs, c->leg_status);
    }
    StageTimer st2_(c, VO_ST_FILTER);
    hipLaunchKernelGGL(filter_kernel, dim3(c->n), dim3(256), 0, c->stream, c->feat, c->n_feat, c->p.max_features, (float)c->W, (float)c->H,
                       c->p.loop_threshold, c->legs, c->leg_status, c->m1, c->m2, c->n_match);
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}
