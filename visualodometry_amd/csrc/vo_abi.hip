// C ABI entry points (include/vo_abi.h): context life cycle, buffer management, stage sequencing.
#include "vo_ctx.h"
#include "pnp_host.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <chrono>
#include <mutex>

static thread_local std::string g_create_err;

constexpr int kMaxDevices = 64;
static std::mutex g_hi_mu;
static hipStream_t g_hi_stream[kMaxDevices];
static int g_hi_users[kMaxDevices];

int vo_fail(vo_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_err = buf;
    return code;
}

#define CHECK_CTX(c) do { if (!(c)) return VO_ERR_BAD_ARG; } while (0)
#define NOT_PENDING(c) do { if ((c)->step_pending) return vo_fail((c), VO_ERR_STATE, "a submitted step has not been finished: call vo_step_finish first"); } while (0)
#define CHECK_B(c, b) do { if ((b) < 0 || (b) >= (c)->n) return vo_fail((c), VO_ERR_BAD_ARG, "frame index %d outside the current batch of %d", (b), (c)->n); } while (0)

extern "C" int vo_default_params(vo_params* p, int width, int height)
{
    if (!p) return VO_ERR_BAD_ARG;
    memset(p, 0, sizeof(*p));
    p->width = width; p->height = height;
    p->max_batch = 1;
    p->max_features = 16384;
    p->min_disparity = 0; p->num_disparities = 96; p->block_size = 11;
    p->p1 = 8 * 1 * 11 * 11; p->p2 = 32 * 1 * 11 * 11;
    p->disp12_max_diff = 0; p->pre_filter_cap = 0; p->uniqueness_ratio = 0; p->wta_simd_lanes = 8;
    p->fast_threshold = 20; p->fast_nms = 1;
    p->lk_win = 15; p->lk_max_level = 3; p->lk_max_count = 50; p->lk_epsilon = (double)0.03f; p->lk_min_eig = 1e-4;
    p->loop_threshold = 2;
    p->depth_thresh = 3000;
    p->ransac_iters = 100; p->ransac_reproj = 8.0f; p->ransac_conf = 0.99; p->ransac_chunk = 8;
    p->host_threads = 0;
    return VO_OK;
}

static int derive(vo_ctx* c)
{
    const vo_params& p = c->p;
    if (p.width < 16 || p.height < 16 || p.width > 16384 || p.height > 16384) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "frame size %dx%d out of range", p.width, p.height);
    if (p.max_batch < 1 || p.max_batch > 1024) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "max_batch %d out of range", p.max_batch);
    if (p.max_features < 16) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "max_features too small");
    if (p.min_disparity != 0) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "min_disparity must be 0");
    if (p.num_disparities < 16 || p.num_disparities > 256 || p.num_disparities % 16) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "num_disparities must be a multiple of 16 in [16, 256]");
    if (p.block_size < 1 || p.block_size > 13 || !(p.block_size & 1)) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "block_size must be odd and <= 13");
    if (p.uniqueness_ratio != 0) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "uniqueness_ratio must be 0");
    if (p.wta_simd_lanes != 0 && p.wta_simd_lanes != 8 && p.wta_simd_lanes != 16) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "wta_simd_lanes must be 0, 8 or 16");
    if (p.lk_win < 3 || p.lk_win > 31 || !(p.lk_win & 1)) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "lk_win must be odd in [3, 31]");
    if (p.lk_max_level < 0 || p.lk_max_level >= VO_MAX_LEVELS) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "lk_max_level out of range");
    if (p.pnp_device != 0 && p.pnp_device != 1) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "pnp_device must be 0 (host solves) or 1 (GPU solves)");
    if (p.lk_accum != 0) return vo_fail(nullptr, VO_ERR_UNSUPPORTED, "lk_accum %d: the tracker sums in exact integers (0); the float summation orders exist in the CPU oracle only", p.lk_accum);
    c->W = p.width; c->H = p.height; c->D = p.num_disparities; c->Dp = (c->D + 31) & ~31;
    c->width1 = c->W - c->D;
    if (c->width1 < 1) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "width must exceed num_disparities");
    c->P1 = p.p1 > 0 ? p.p1 : 2;
    c->P2 = std::max(p.p2 > 0 ? p.p2 : 5, c->P1 + 1);
    c->disp12 = p.disp12_max_diff > 0 ? p.disp12_max_diff : 1;
    c->ftzero = std::max(p.pre_filter_cap, 15) | 1;
    // packed 16-bit path arithmetic: aggregated cost <= block^2 * 93 and cost + P2 must stay below 32767 - P1
    const int cmax = p.block_size * p.block_size * 93;
    if (cmax > 16383 || c->P2 + cmax + c->P1 > 32767 || c->ftzero > 63) return vo_fail(nullptr, VO_ERR_BAD_PARAM, "P1/P2/block_size exceed the 16-bit cost range");
    // A disparity range that is a multiple of 16 but not of 32 is padded to the next multiple of 32 with dead cells of
    // constant cost.  A path cost L obeys C - P2 <= L <= C, so with dead_cost - P2 + P1 > cmax + P2 a dead neighbour is
    // never the cheapest way into a live cell (the same as the "infinite" cell beyond the range) and never the row minimum.
    c->dead_cost = 32767 - c->P1 - 1;
    if (c->Dp != c->D && !(c->dead_cost - c->P2 + c->P1 > cmax + c->P2))
        return vo_fail(nullptr, VO_ERR_BAD_PARAM, "P1/P2 leave no room for the padding cells of a disparity range that is not a multiple of 32");
    // 4 fixed stripes with a warm-up overlap (the stripe count is fixed so that results do not depend on the machine)
    c->stripe_sz = (int)ceil(c->H / 4.0);
    c->overlap = (p.block_size / 2 + 1) + (int)ceil(0.1 * c->stripe_sz);
    int off = 0;
    for (int s = 0; s < 4; s++) {
        int y0 = std::max(std::min(s * c->stripe_sz - c->overlap, c->H), 0);
        int y1 = std::min((s + 1) * c->stripe_sz, c->H);
        c->stripe_y0[s] = y0; c->stripe_rows[s] = std::max(y1 - y0, 0); c->stripe_row_off[s] = off;
        off += c->stripe_rows[s];
    }
    c->stripe_row_off[4] = off; c->srows = off;
    // pyramid: a level is built only while the next one stays larger than the window in both dimensions
    c->border = p.lk_win;
    {
        int w = c->W, h = c->H, lv = p.lk_max_level;
        for (int level = 0; level <= p.lk_max_level; level++) {
            w = (w + 1) / 2; h = (h + 1) / 2;
            if (w <= p.lk_win || h <= p.lk_win) { lv = level; break; }
        }
        c->nlevels = lv;
    }
    size_t io = 0, dof = 0;
    int w = c->W, h = c->H;
    for (int l = 0; l <= c->nlevels; l++) {
        PyrLevel& L = c->lv[l];
        L.w = w; L.h = h; L.pitch = (w + 2 * c->border + 15) & ~15; L.rows = h + 2 * c->border;
        L.img_off = io; L.der_off = dof;
        io += (size_t)L.pitch * L.rows; dof += (size_t)L.pitch * L.rows * 2;
        w = (w + 1) / 2; h = (h + 1) / 2;
    }
    c->pyr_bytes = (io + 255) & ~(size_t)255; c->der_elems = (dof + 127) & ~(size_t)127;
    return VO_OK;
}

template <class T>
static hipError_t dalloc(T** p, size_t count) { return hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)); }

extern "C" int vo_create(const vo_params* p, int device, vo_ctx** out)
{
    if (!p || !out) return vo_fail(nullptr, VO_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return vo_fail(nullptr, VO_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev || device >= kMaxDevices) return vo_fail(nullptr, VO_ERR_BAD_ARG, "device %d out of range (%d devices)", device, ndev);
    vo_ctx* c = new vo_ctx();
    c->p = *p; c->device = device;
    memset(c->ev0, 0, sizeof(c->ev0)); memset(c->ev1, 0, sizeof(c->ev1)); memset(c->ev_used, 0, sizeof(c->ev_used));
    memset(c->stage_ms, 0, sizeof(c->stage_ms)); memset(c->stage_calls, 0, sizeof(c->stage_calls));
    if (c->p.ransac_chunk < 1) c->p.ransac_chunk = 8;
    if (c->p.ransac_chunk > 1024) c->p.ransac_chunk = 1024;
    c->p.fast_threshold = std::min(std::max(c->p.fast_threshold, 0), 255);
    int rc = derive(c);
    if (rc != VO_OK) { delete c; return rc; }
#define CR(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { vo_fail(nullptr, VO_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); vo_destroy(c); return VO_ERR_HIP; } } while (0)
    CR(hipSetDevice(device));
    if (vo_noalloc_prepare(device) != VO_OK) { vo_destroy(c); return VO_ERR_HIP; } // so that vo_detect_fast_noalloc can be captured on this device
    CR(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        // one high-priority stream per device for all contexts: a process has few hardware queues (4 by default), and
        // every extra stream beyond them shares a queue with, and so serialises behind, some other stream
        std::lock_guard<std::mutex> lk(g_hi_mu);
        if (!g_hi_stream[device]) {
            int least = 0, greatest = 0;
            CR(hipDeviceGetStreamPriorityRange(&least, &greatest));
            CR(hipStreamCreateWithPriority(&g_hi_stream[device], hipStreamNonBlocking, greatest));
        }
        c->stream_hi = g_hi_stream[device];
        g_hi_users[device]++;
    }
    // A second stream can carry the corner / tracking chain of a step beside the disparity chain (see vo_step_submit).  It is
    // created by the first step that uses it: a stream that is never used still takes a share of the process's few hardware
    // queues, and two contexts' bulk streams on one queue serialise their batches.
    c->single_stream = getenv("VO_SINGLE_STREAM") != nullptr;
    c->tracker_at_once = getenv("VO_DUAL_STREAM") != nullptr; // batches too start their tracker without waiting for the cost kernel
    const size_t B = c->p.max_batch, W = c->W, H = c->H, D = c->Dp, w1 = c->width1, F = c->p.max_features;
    c->own_pitch = (W + 63) & ~(size_t)63; c->own_fstride = c->own_pitch * H;
    CR(dalloc(&c->img_own, 4 * B * c->own_fstride));
    CR(dalloc(&c->pre, B * 2 * H * (size_t)vo_pre_pitch(c) * 8 + 4096)); // slack: the wide cost kernel reads a few columns past the last row of halo tiles
    CR(dalloc(&c->cost, B * c->srows * w1 * D));
    CR(dalloc(&c->agg, B * c->srows * w1 * D));
    CR(dalloc(&c->wta, B * c->srows * W));
    CR(dalloc(&c->ckpt, B * H * (size_t)vo_hpath_segments(c) * 16 * (D / 32 + 1)));
    CR(dalloc(&c->disp_pre, B * H * W));
    CR(dalloc(&c->disp16, B * H * W));
    CR(dalloc(&c->disp32, B * H * W));
    CR(dalloc(&c->depth, B * H * W));
    CR(dalloc(&c->score, B * H * W));
    CR(dalloc(&c->row_count, B * H));
    CR(dalloc(&c->kp_mask, B * H * (size_t)((W - 6 + 63) / 64)));
    CR(dalloc(&c->n_feat, 3 * B)); // [3][B]: feature, match and object-point counts, fetched with one copy
    c->n_match = c->n_feat + B; c->n_obj = c->n_feat + 2 * B;
    CR(dalloc(&c->feat, B * F));
    CR(dalloc(&c->pyr, B * 4 * c->pyr_bytes + 256)); // slack: the tracker reads 8 bytes where it needs 5
    CR(dalloc(&c->der, B * 4 * c->der_elems + 64));
    CR(dalloc(&c->legs, B * 4 * F));
    CR(dalloc(&c->leg_status, B * 4 * F));
    CR(dalloc(&c->m1, B * F));
    CR(dalloc(&c->m2, B * F));
    CR(dalloc(&c->obj, B * F));
    CR(dalloc(&c->obj_img, B * F));
    CR(dalloc(&c->hyp, B * c->p.ransac_chunk * 12));
    CR(dalloc(&c->hyp_count, B * c->p.ransac_chunk));
    CR(dalloc(&c->hyp_mask, B * c->p.ransac_chunk * F));
    CR(dalloc(&c->hyp_sel, B));
    if (c->p.pnp_device) {
        CR(dalloc(&c->hyp_idx, B * c->p.ransac_chunk * 5 + B));
        CR(dalloc(&c->hyp_rt, B * c->p.ransac_chunk * 6));
    }
    // pinned staging: the correspondences of a batch ([B][F] points), later reused for a RANSAC round's hypotheses and
    // counts ([B][chunk] x (12 doubles + 1 int)); inlier masks ride along when they fit
    c->h_stage_bytes = std::max<size_t>(std::max<size_t>(B * F * (sizeof(vo_point3f) + sizeof(vo_point2f)) + B * 64,
                                                         B * (size_t)c->p.ransac_chunk * (12 * sizeof(double) + sizeof(int)) + 512), 1 << 20);
    CR(hipHostMalloc(&c->h_stage, c->h_stage_bytes, hipHostMallocDefault));
    CR(hipHostMalloc((void**)&c->h_counts, 3 * B * sizeof(int), hipHostMallocDefault));
    c->h_nfeat.assign(B, 0); c->h_nmatch.assign(B, 0); c->h_nobj.assign(B, 0);
#undef CR
    *out = c;
    return VO_OK;
}

extern "C" void vo_destroy(vo_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void* bufs[] = {c->img_own, c->pre, c->cost, c->agg, c->wta, c->ckpt, c->disp_pre, c->disp16, c->disp32, c->depth, c->score,
                    c->row_count, c->kp_mask, c->n_feat, c->feat, c->pyr, c->der, c->legs, c->leg_status, c->m1, c->m2,
                    c->obj, c->obj_img, c->hyp, c->hyp_count, c->hyp_mask, c->hyp_sel, c->hyp_idx, c->hyp_rt};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (c->cloud) (void)hipFree(c->cloud);
    if (c->cloud_rows) (void)hipFree(c->cloud_rows);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->step_done) (void)hipEventDestroy(c->step_done);
    for (int i = 0; i < VO_NUM_STAGES; i++) { if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]); if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]); }
    if (c->stream_hi) {
        std::lock_guard<std::mutex> lk(g_hi_mu);
        if (--g_hi_users[c->device] == 0) { (void)hipStreamDestroy(g_hi_stream[c->device]); g_hi_stream[c->device] = nullptr; }
    }
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->join2) (void)hipEventDestroy(c->join2);
    if (c->after_cost) (void)hipEventDestroy(c->after_cost);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" const char* vo_last_error(const vo_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" const char* vo_create_error(void) { return g_create_err.c_str(); }
extern "C" void* vo_stream(vo_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int vo_synchronize(vo_ctx* c)
{
    CHECK_CTX(c);
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

extern "C" int vo_set_calib(vo_ctx* c, const double P0[12], const double P1[12])
{
    CHECK_CTX(c);
    if (!P0 || !P1) return vo_fail(c, VO_ERR_BAD_ARG, "null projection matrix");
    memcpy(c->P0, P0, sizeof(c->P0)); memcpy(c->P1m, P1, sizeof(c->P1m));
    c->have_calib = true;
    return VO_OK;
}

static void invalidate(vo_ctx* c)
{
    c->have_disp = c->have_depth = c->have_features = c->have_pyr = c->have_matches = c->have_obj = false;
}

static int check_images(vo_ctx* c, int n, const void* a, const void* b, const void* d, const void* e, size_t stride, size_t fstride)
{
    if (n < 1 || n > c->p.max_batch) return vo_fail(c, VO_ERR_BAD_ARG, "batch size %d outside [1, %d]", n, c->p.max_batch);
    if (!a || !b || !d || !e) return vo_fail(c, VO_ERR_BAD_ARG, "null image pointer");
    if (stride < (size_t)c->W || (n > 1 && fstride < stride * c->H)) return vo_fail(c, VO_ERR_BAD_ARG, "stride smaller than the frame");
    return VO_OK;
}

// Host frames to the context's own image memory.  Tightly packed frames (stride == width, frames back to back) go up with
// one linear copy per image set and keep their layout on the device; anything else is re-pitched by a 2D copy per frame
// (slow from pageable memory: the runtime stages it row by row).
static int upload_frames(vo_ctx* c, uint8_t* dst, const uint8_t* src, int frames, size_t stride, size_t frame_stride, bool packed)
{
    if (packed) {
        VO_HIP(c, hipMemcpyAsync(dst, src, (size_t)frames * c->W * c->H, hipMemcpyHostToDevice, c->stream));
        return VO_OK;
    }
    for (int f = 0; f < frames; f++)
        VO_HIP(c, hipMemcpy2DAsync(dst + (size_t)f * c->own_fstride, c->own_pitch, src + (size_t)f * frame_stride, stride, c->W, c->H,
                                   hipMemcpyHostToDevice, c->stream));
    return VO_OK;
}
static bool frames_packed(const vo_ctx* c, int frames, size_t stride, size_t frame_stride)
{
    return stride == (size_t)c->W && (frames == 1 || frame_stride == (size_t)c->W * c->H);
}

extern "C" int vo_set_images(vo_ctx* c, int n, const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                             size_t stride, size_t frame_stride)
{
    CHECK_CTX(c); NOT_PENDING(c);
    int rc = check_images(c, n, lt1, rt1, lt2, rt2, stride, frame_stride);
    if (rc) return rc;
    VO_HIP(c, hipSetDevice(c->device));
    const uint8_t* src[4] = {lt1, rt1, lt2, rt2};
    const size_t B = c->p.max_batch;
    const bool packed = frames_packed(c, n, stride, frame_stride);
    for (int i = 0; i < 4; i++) {
        uint8_t* dst = c->img_own + (size_t)i * B * c->own_fstride;
        rc = upload_frames(c, dst, src[i], n, stride, frame_stride, packed);
        if (rc) return rc;
        c->img[i] = dst;
    }
    // the copies read pageable host memory: finish them before the caller may reuse its buffers
    VO_HIP(c, hipStreamSynchronize(c->stream));
    c->pitch = packed ? (size_t)c->W : c->own_pitch; c->fstride = packed ? (size_t)c->W * c->H : c->own_fstride;
    c->n = n; c->seq_mode = false;
    invalidate(c);
    return VO_OK;
}

extern "C" int vo_set_images_device(vo_ctx* c, int n, const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                                    size_t stride, size_t frame_stride)
{
    CHECK_CTX(c); NOT_PENDING(c);
    int rc = check_images(c, n, lt1, rt1, lt2, rt2, stride, frame_stride);
    if (rc) return rc;
    c->img[0] = lt1; c->img[1] = rt1; c->img[2] = lt2; c->img[3] = rt2;
    c->pitch = stride; c->fstride = frame_stride; c->n = n; c->seq_mode = false;
    invalidate(c);
    return VO_OK;
}

extern "C" int vo_set_sequence(vo_ctx* c, int n_pairs, const uint8_t* left, const uint8_t* right, size_t stride, size_t frame_stride)
{
    CHECK_CTX(c); NOT_PENDING(c);
    int rc = check_images(c, n_pairs, left, right, left, right, stride, frame_stride);
    if (rc) return rc;
    if (frame_stride < stride * c->H) return vo_fail(c, VO_ERR_BAD_ARG, "stride smaller than the frame");
    VO_HIP(c, hipSetDevice(c->device));
    const uint8_t* src[2] = {left, right};
    const size_t B = c->p.max_batch;
    const bool packed = frames_packed(c, n_pairs + 1, stride, frame_stride);
    const size_t dev_fstride = packed ? (size_t)c->W * c->H : c->own_fstride;
    for (int i = 0; i < 2; i++) {
        uint8_t* dst = c->img_own + (size_t)i * (B + 1) * c->own_fstride; // 2 (B + 1) frames fit the 4 B frames of the pair layout
        rc = upload_frames(c, dst, src[i], n_pairs + 1, stride, frame_stride, packed);
        if (rc) return rc;
        c->img[i] = dst; c->img[i + 2] = dst + dev_fstride;
    }
    VO_HIP(c, hipStreamSynchronize(c->stream));
    c->pitch = packed ? (size_t)c->W : c->own_pitch; c->fstride = dev_fstride; c->n = n_pairs; c->seq_mode = true;
    invalidate(c);
    return VO_OK;
}

extern "C" int vo_set_sequence_device(vo_ctx* c, int n_pairs, const uint8_t* left, const uint8_t* right, size_t stride, size_t frame_stride)
{
    CHECK_CTX(c); NOT_PENDING(c);
    int rc = check_images(c, n_pairs, left, right, left, right, stride, frame_stride);
    if (rc) return rc;
    if (frame_stride < stride * c->H) return vo_fail(c, VO_ERR_BAD_ARG, "stride smaller than the frame");
    c->img[0] = left; c->img[1] = right; c->img[2] = left + frame_stride; c->img[3] = right + frame_stride;
    c->pitch = stride; c->fstride = frame_stride; c->n = n_pairs; c->seq_mode = true;
    invalidate(c);
    return VO_OK;
}

#define NEED_IMAGES(c) do { if ((c)->n < 1 || !(c)->img[0]) return vo_fail((c), VO_ERR_STATE, "no images set"); } while (0)

static int d2h(vo_ctx* c, void* dst, const void* src, size_t bytes)
{
    VO_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

extern "C" int vo_compute_disparity(vo_ctx* c)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c);
    VO_HIP(c, hipSetDevice(c->device));
    int rc = launch_sgbm(c);
    if (rc) return rc;
    c->have_disp = true; c->have_depth = false;
    return VO_OK;
}

extern "C" int vo_get_disparity16(vo_ctx* c, int b, int16_t* out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_disp) return vo_fail(c, VO_ERR_STATE, "disparity not computed");
    return d2h(c, out, c->disp16 + (size_t)b * c->W * c->H, (size_t)c->W * c->H * sizeof(int16_t));
}

extern "C" int vo_get_disparity32(vo_ctx* c, int b, float* out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_disp) return vo_fail(c, VO_ERR_STATE, "disparity not computed");
    return d2h(c, out, c->disp32 + (size_t)b * c->W * c->H, (size_t)c->W * c->H * sizeof(float));
}

extern "C" int vo_compute_depth_map(vo_ctx* c)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c);
    if (!c->have_disp) return vo_fail(c, VO_ERR_STATE, "disparity not computed");
    if (!c->have_calib) return vo_fail(c, VO_ERR_STATE, "calibration not set");
    VO_HIP(c, hipSetDevice(c->device));
    int rc = launch_depth_map(c);
    if (rc) return rc;
    c->have_depth = true;
    return VO_OK;
}

extern "C" int vo_get_depth(vo_ctx* c, int b, double* out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_depth) return vo_fail(c, VO_ERR_STATE, "depth map not computed");
    return d2h(c, out, c->depth + (size_t)b * c->W * c->H, (size_t)c->W * c->H * sizeof(double));
}

extern "C" int vo_set_disparity32(vo_ctx* c, int b, const float* disp)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c); CHECK_B(c, b);
    if (!disp) return vo_fail(c, VO_ERR_BAD_ARG, "null disparity map");
    // the other frames of a batch keep their computed maps; without any they have nothing to fall back on
    if (!c->have_disp && c->n > 1) return vo_fail(c, VO_ERR_STATE, "disparity not computed: a caller-supplied map can replace one frame of a computed batch, or the only frame");
    VO_HIP(c, hipSetDevice(c->device));
    VO_HIP(c, hipMemcpyAsync(c->disp32 + (size_t)b * c->W * c->H, disp, (size_t)c->W * c->H * sizeof(float), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    c->have_disp = true; c->have_depth = false; c->have_obj = false;
    return VO_OK;
}

extern "C" int vo_set_depth(vo_ctx* c, int b, const double* depth)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c); CHECK_B(c, b);
    if (!depth) return vo_fail(c, VO_ERR_BAD_ARG, "null depth map");
    VO_HIP(c, hipSetDevice(c->device));
    if (!c->have_depth && c->n > 1) {
        // the other frames of the batch need their own depth maps before the flag can cover the batch
        if (!c->have_disp || !c->have_calib) return vo_fail(c, VO_ERR_STATE, "depth map not computed: a caller-supplied map can replace one frame of a computed batch, or the only frame");
        int rc = launch_depth_map(c);
        if (rc) return rc;
    }
    VO_HIP(c, hipMemcpyAsync(c->depth + (size_t)b * c->W * c->H, depth, (size_t)c->W * c->H * sizeof(double), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    c->have_depth = true; c->have_obj = false;
    return VO_OK;
}

extern "C" int vo_get_depth_cloud(vo_ctx* c, int b, vo_cloud_point* out, int cap, int* n_out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!n_out || cap < 0) return vo_fail(c, VO_ERR_BAD_ARG, "null count pointer or negative capacity");
    if (!c->have_depth) return vo_fail(c, VO_ERR_STATE, "depth map not computed");
    VO_HIP(c, hipSetDevice(c->device));
    int count = 0;
    const int rc = launch_depth_cloud(c, b, &count);
    if (rc) return rc;
    *n_out = count;
    const int k = std::min(count, cap);
    if (out && k > 0) return d2h(c, out, c->cloud, (size_t)k * sizeof(vo_cloud_point));
    return VO_OK;
}

static int fetch_counts(vo_ctx* c, const int* dev, std::vector<int>& host)
{
    VO_HIP(c, hipMemcpyAsync(c->h_stage, dev, c->n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    memcpy(host.data(), c->h_stage, c->n * sizeof(int));
    return VO_OK;
}

extern "C" int vo_detect_fast(vo_ctx* c)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c);
    VO_HIP(c, hipSetDevice(c->device));
    int rc = launch_fast(c);
    if (rc) return rc;
    rc = fetch_counts(c, c->n_feat, c->h_nfeat);
    if (rc) return rc;
    c->have_features = true; c->have_matches = false; c->have_obj = false;
    return VO_OK;
}

extern "C" int vo_get_features(vo_ctx* c, int b, vo_point2f* out, int cap, int* n_out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_features) return vo_fail(c, VO_ERR_STATE, "features not detected");
    int n = c->h_nfeat[b];
    if (n_out) *n_out = n;
    if (out && cap > 0 && n > 0)
        return d2h(c, out, c->feat + (size_t)b * c->p.max_features, (size_t)std::min(n, cap) * sizeof(vo_point2f));
    return VO_OK;
}

extern "C" int vo_set_features(vo_ctx* c, int b, const vo_point2f* pts, int n)
{
    CHECK_CTX(c); NOT_PENDING(c); CHECK_B(c, b);
    if (n < 0 || n > c->p.max_features || (n > 0 && !pts)) return vo_fail(c, VO_ERR_BAD_ARG, "feature count %d outside [0, %d]", n, c->p.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    if (n > 0) VO_HIP(c, hipMemcpyAsync(c->feat + (size_t)b * c->p.max_features, pts, (size_t)n * sizeof(vo_point2f), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemcpyAsync(c->n_feat + b, &n, sizeof(int), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    c->h_nfeat[b] = n;
    c->have_features = true; c->have_matches = false; c->have_obj = false;
    return VO_OK;
}

extern "C" int vo_match_klt(vo_ctx* c)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c);
    if (!c->have_features) return vo_fail(c, VO_ERR_STATE, "no features: call vo_detect_fast or vo_set_features");
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    if (!c->have_pyr) { rc = launch_pyramids(c); if (rc) return rc; c->have_pyr = true; }
    rc = launch_match_klt(c, true);
    if (rc) return rc;
    rc = fetch_counts(c, c->n_match, c->h_nmatch);
    if (rc) return rc;
    c->have_matches = true; c->have_obj = false;
    return VO_OK;
}

extern "C" int vo_get_matches(vo_ctx* c, int b, vo_point2f* t1, vo_point2f* t2, int cap, int* m_out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_matches) return vo_fail(c, VO_ERR_STATE, "matches not computed");
    int m = c->h_nmatch[b];
    if (m_out) *m_out = m;
    int k = std::min(m, cap);
    if (k > 0 && t1) { int rc = d2h(c, t1, c->m1 + (size_t)b * c->p.max_features, (size_t)k * sizeof(vo_point2f)); if (rc) return rc; }
    if (k > 0 && t2) { int rc = d2h(c, t2, c->m2 + (size_t)b * c->p.max_features, (size_t)k * sizeof(vo_point2f)); if (rc) return rc; }
    return VO_OK;
}

extern "C" int vo_get_klt_legs(vo_ctx* c, int b, vo_point2f* pts, uint8_t* status, int n)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_matches) return vo_fail(c, VO_ERR_STATE, "matches not computed");
    if (n != c->h_nfeat[b]) return vo_fail(c, VO_ERR_BAD_ARG, "n must equal the feature count %d", c->h_nfeat[b]);
    const size_t F = c->p.max_features;
    for (int leg = 0; leg < 4; leg++) {
        if (pts && n > 0) { int rc = d2h(c, pts + (size_t)leg * n, c->legs + ((size_t)b * 4 + leg) * F, (size_t)n * sizeof(vo_point2f)); if (rc) return rc; }
        if (status && n > 0) { int rc = d2h(c, status + (size_t)leg * n, c->leg_status + ((size_t)b * 4 + leg) * F, (size_t)n); if (rc) return rc; }
    }
    return VO_OK;
}

extern "C" int vo_set_matches(vo_ctx* c, int b, const vo_point2f* t1, const vo_point2f* t2, int m)
{
    CHECK_CTX(c); NOT_PENDING(c); CHECK_B(c, b);
    if (m < 0 || m > c->p.max_features || (m > 0 && (!t1 || !t2))) return vo_fail(c, VO_ERR_BAD_ARG, "match count %d outside [0, %d]", m, c->p.max_features);
    VO_HIP(c, hipSetDevice(c->device));
    const size_t F = c->p.max_features;
    if (m > 0) {
        VO_HIP(c, hipMemcpyAsync(c->m1 + (size_t)b * F, t1, (size_t)m * sizeof(vo_point2f), hipMemcpyHostToDevice, c->stream));
        VO_HIP(c, hipMemcpyAsync(c->m2 + (size_t)b * F, t2, (size_t)m * sizeof(vo_point2f), hipMemcpyHostToDevice, c->stream));
    }
    VO_HIP(c, hipMemcpyAsync(c->n_match + b, &m, sizeof(int), hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    c->h_nmatch[b] = m;
    c->have_matches = true; c->have_obj = false;
    return VO_OK;
}

extern "C" int vo_triangulate(vo_ctx* c)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c);
    if (!c->have_matches) return vo_fail(c, VO_ERR_STATE, "matches not computed");
    if (!c->have_disp && !c->have_depth) return vo_fail(c, VO_ERR_STATE, "neither a disparity nor a depth map is present");
    if (!c->have_calib) return vo_fail(c, VO_ERR_STATE, "calibration not set");
    VO_HIP(c, hipSetDevice(c->device));
    int rc = launch_triangulate(c);
    if (rc) return rc;
    rc = fetch_counts(c, c->n_obj, c->h_nobj);
    if (rc) return rc;
    c->have_obj = true;
    return VO_OK;
}

extern "C" int vo_get_object_points(vo_ctx* c, int b, vo_point3f* obj, vo_point2f* img2, int cap, int* k_out)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_obj) return vo_fail(c, VO_ERR_STATE, "object points not computed");
    int k = c->h_nobj[b];
    if (k_out) *k_out = k;
    int kk = std::min(k, cap);
    if (kk > 0 && obj) { int rc = d2h(c, obj, c->obj + (size_t)b * c->p.max_features, (size_t)kk * sizeof(vo_point3f)); if (rc) return rc; }
    if (kk > 0 && img2) { int rc = d2h(c, img2, c->obj_img + (size_t)b * c->p.max_features, (size_t)kk * sizeof(vo_point2f)); if (rc) return rc; }
    return VO_OK;
}

extern "C" int vo_get_object_points_batch(vo_ctx* c, vo_point3f* obj, vo_point2f* img2, int cap, int* counts)
{
    CHECK_CTX(c);
    if (!c->have_obj) return vo_fail(c, VO_ERR_STATE, "object points not computed");
    if (cap < 0) return vo_fail(c, VO_ERR_BAD_ARG, "negative capacity");
    VO_HIP(c, hipSetDevice(c->device));
    int kmax = 0;
    for (int b = 0; b < c->n; b++) { kmax = std::max(kmax, c->h_nobj[b]); if (counts) counts[b] = c->h_nobj[b]; }
    const size_t w = (size_t)std::min(kmax, cap), F = c->p.max_features;
    if (w == 0) return VO_OK;
    if (obj) VO_HIP(c, hipMemcpy2DAsync(obj, (size_t)cap * sizeof(vo_point3f), c->obj, F * sizeof(vo_point3f), w * sizeof(vo_point3f), c->n, hipMemcpyDefault, c->stream));
    if (img2) VO_HIP(c, hipMemcpy2DAsync(img2, (size_t)cap * sizeof(vo_point2f), c->obj_img, F * sizeof(vo_point2f), w * sizeof(vo_point2f), c->n, hipMemcpyDefault, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}

extern "C" int vo_pnp_score(vo_ctx* c, int b, const double* Rt12, int H, float thresh2, int* counts, uint8_t* masks)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!c->have_obj) return vo_fail(c, VO_ERR_STATE, "object points not computed");
    if (!Rt12 || !counts || H < 1) return vo_fail(c, VO_ERR_BAD_ARG, "bad hypothesis arguments");
    VO_HIP(c, hipSetDevice(c->device));
    const int chunk = c->p.ransac_chunk;
    const size_t F = c->p.max_features;
    const int k = c->h_nobj[b];
    for (int h0 = 0; h0 < H; h0 += chunk) {
        int hn = std::min(chunk, H - h0);
        VO_HIP(c, hipMemcpyAsync(c->hyp + (size_t)b * chunk * 12, Rt12 + (size_t)h0 * 12, (size_t)hn * 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        int rc = launch_pnp_score(c, b, 1, hn, thresh2, masks != nullptr);
        if (rc) return rc;
        rc = d2h(c, counts + h0, c->hyp_count + (size_t)b * chunk, (size_t)hn * sizeof(int));
        if (rc) return rc;
        if (masks && k > 0)
            for (int h = 0; h < hn; h++) {
                rc = d2h(c, masks + (size_t)(h0 + h) * k, c->hyp_mask + ((size_t)b * chunk + h) * F, (size_t)k);
                if (rc) return rc;
            }
    }
    return VO_OK;
}

extern "C" int vo_integrate_pose(double pose[16], const double Rt[16])
{
    if (!pose || !Rt) return VO_ERR_BAD_ARG;
    double inv[16], np[16];
    if (!pnp_host::inv4x4(Rt, inv)) return VO_ERR_BAD_ARG;
    for (int r = 0; r < 4; r++) for (int cc = 0; cc < 4; cc++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += pose[r * 4 + k] * inv[k * 4 + cc];
        np[r * 4 + cc] = s;
    }
    memcpy(pose, np, sizeof(np));
    return VO_OK;
}

extern "C" int vo_estimate_motion(vo_ctx* c, double* Rt_out, vo_frame_info* info)
{
    CHECK_CTX(c); NOT_PENDING(c); NEED_IMAGES(c);
    if (!Rt_out) return vo_fail(c, VO_ERR_BAD_ARG, "null output");
    if (!c->have_obj) { int rc = vo_triangulate(c); if (rc) return rc; }
    return pnp_host::estimate_motion_batch(c, Rt_out, info);
}

extern "C" int vo_step_submit(vo_ctx* c)
{
    CHECK_CTX(c); NEED_IMAGES(c);
    if (!c->have_calib) return vo_fail(c, VO_ERR_STATE, "calibration not set");
    if (c->step_pending) return vo_fail(c, VO_ERR_STATE, "a submitted step has not been finished");
    VO_HIP(c, hipSetDevice(c->device));
    c->step_t0 = std::chrono::steady_clock::now();
    int rc;
    const bool dual = !c->single_stream && !c->timing_exclusive && (c->n == 1 || c->tracker_at_once || (c->Dp <= 128 && c->p.lk_win <= 15));
    if (dual && !c->stream2) { // see below
        VO_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        VO_HIP(c, hipEventCreateWithFlags(&c->join2, hipEventDisableTiming));
        VO_HIP(c, hipEventCreateWithFlags(&c->after_cost, hipEventDisableTiming));
    }
    // every GPU stage is enqueued back to back without a host round trip
    if ((rc = launch_sgbm(c))) return rc;
    c->have_disp = true; c->have_depth = false;
    // Disparity and the corner / tracking chain are independent until triangulation, so the second chain runs on a stream of
    // its own.  A single frame pair leaves most of the GPU idle (its kernels are serial along x or y): its tracker starts at
    // once (2.2 -> 1.7 ms per step at KITTI shape).  A batch fills the GPU with every large kernel, and two kernels bound by
    // vector issue (cost kernel, tracker) gain nothing from sharing it: there the tracker waits for the cost kernel and runs
    // beside the horizontal path passes, which are bound by memory and latency (5 850 -> 5 995 frames/s with 8 hardware queues,
    // 5 885 with 4).  That pays where the two are of similar length, as at the reference's parameters; where one dwarfs the
    // other the batch stays on one stream (measured: 4K x 256 disparities 174 -> 165 frames/s, 20 000 corners with a 21 x 21
    // window 1 587 -> 1 531).  VO_SINGLE_STREAM keeps everything on one stream, VO_DUAL_STREAM starts a batch's tracker at once.
    hipStream_t main_stream = c->stream;
    if (dual) c->stream = c->stream2;
    rc = launch_fast(c);
    if (!rc) rc = launch_pyramids(c);
    if (!rc && dual && c->n > 1 && !c->tracker_at_once && hipStreamWaitEvent(c->stream, c->after_cost, 0) != hipSuccess)
        rc = vo_fail(c, VO_ERR_HIP, "hipStreamWaitEvent failed");
    if (!rc) { c->have_pyr = true; rc = launch_match_klt(c, false); }
    if (dual) {
        if (!rc && hipEventRecord(c->join2, c->stream2) != hipSuccess) rc = vo_fail(c, VO_ERR_HIP, "hipEventRecord failed");
        c->stream = main_stream;
        if (!rc && hipStreamWaitEvent(c->stream, c->join2, 0) != hipSuccess) rc = vo_fail(c, VO_ERR_HIP, "hipStreamWaitEvent failed");
    }
    if (rc) return rc;
    if ((rc = launch_triangulate(c))) return rc;
    VO_HIP(c, hipMemcpyAsync(c->h_counts, c->n_feat, 3 * (size_t)c->p.max_batch * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (!c->step_done) VO_HIP(c, hipEventCreateWithFlags(&c->step_done, hipEventDisableTiming));
    VO_HIP(c, hipEventRecord(c->step_done, c->stream));
    c->step_pending = true;
    return VO_OK;
}

extern "C" int vo_step_finish(vo_ctx* c, double* Rt_out, vo_frame_info* info)
{
    CHECK_CTX(c);
    if (!c->step_pending) return vo_fail(c, VO_ERR_STATE, "no submitted step");
    if (!Rt_out) return vo_fail(c, VO_ERR_BAD_ARG, "null output");
    VO_HIP(c, hipSetDevice(c->device));
    c->step_pending = false;
    VO_HIP(c, hipEventSynchronize(c->step_done));
    memcpy(c->h_nfeat.data(), c->h_counts, c->n * sizeof(int));
    memcpy(c->h_nmatch.data(), c->h_counts + c->p.max_batch, c->n * sizeof(int));
    memcpy(c->h_nobj.data(), c->h_counts + 2 * c->p.max_batch, c->n * sizeof(int));
    c->have_features = c->have_matches = c->have_obj = true;
    int rc = pnp_host::estimate_motion_batch(c, Rt_out, info);
    if (c->timing) {
        vo_collect_timing(c);
        c->stage_ms[VO_ST_STEP_WALL] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c->step_t0).count();
        c->stage_calls[VO_ST_STEP_WALL]++;
    }
    return rc;
}

extern "C" int vo_step(vo_ctx* c, double* Rt_out, vo_frame_info* info)
{
    CHECK_CTX(c);
    if (!Rt_out) return vo_fail(c, VO_ERR_BAD_ARG, "null output");
    int rc = vo_step_submit(c);
    if (rc) return rc;
    return vo_step_finish(c, Rt_out, info);
}

static const char* kStageNames[VO_NUM_STAGES] = {"sgbm_prefilter", "", "sgbm_cost_vpath", "sgbm_hpath_ckpt", "sgbm_hpath_resolve",
    "sgbm_lrcheck_median", "pnp_solve", "fast", "pyramid", "lk", "klt_filter", "triangulate", "pnp_score", "pnp_host", "step_wall", "pnp_host_serial"};

extern "C" const char* vo_stage_name(int s) { return s >= 0 && s < VO_NUM_STAGES ? kStageNames[s] : ""; }

extern "C" int vo_set_timing(vo_ctx* c, int enable)
{
    CHECK_CTX(c);
    VO_HIP(c, hipSetDevice(c->device));
    if (enable && !c->ev0[0]) {
        for (int i = 0; i < VO_NUM_STAGES; i++) { VO_HIP(c, hipEventCreate(&c->ev0[i])); VO_HIP(c, hipEventCreate(&c->ev1[i])); }
    }
    c->timing = enable != 0;
    c->timing_exclusive = enable == 2;
    return VO_OK;
}

void vo_collect_timing(vo_ctx* c)
{
    if (!c->timing) return;
    for (int i = 0; i < VO_NUM_STAGES; i++) {
        if (!c->ev_used[i]) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]) == hipSuccess) { c->stage_ms[i] += ms; c->stage_calls[i]++; }
        c->ev_used[i] = false;
    }
}

extern "C" int vo_get_timing(vo_ctx* c, double* ms, long long* calls, int reset)
{
    CHECK_CTX(c);
    for (int i = 0; i < VO_NUM_STAGES; i++) {
        if (ms) ms[i] = c->stage_ms[i];
        if (calls) calls[i] = c->stage_calls[i];
        if (reset) { c->stage_ms[i] = 0; c->stage_calls[i] = 0; }
    }
    return VO_OK;
}

extern "C" int vo_host_epnp(const vo_point3f* obj, const vo_point2f* img, int n, const double K[9], double rvec[3], double tvec[3])
{
    if (!obj || !img || !K || !rvec || !tvec || n < 4) return VO_ERR_BAD_ARG;
    pnp_host::epnp(obj, img, n, K, rvec, tvec);
    return VO_OK;
}

extern "C" int vo_host_epnp5_fixed(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3])
{
    if (!obj || !img || !K || !rvec || !tvec) return VO_ERR_BAD_ARG;
    pnp_host::epnp5_fixed(obj, img, K, rvec, tvec);
    return VO_OK;
}

extern "C" int vo_host_pnp_iterative(const double* obj, const double* img, int n, const double K[9], double rvec[3], double tvec[3])
{
    if (!obj || !img || !K || !rvec || !tvec || n < 1) return VO_ERR_BAD_ARG;
    return pnp_host::iterative(obj, img, n, K, rvec, tvec);
}

extern "C" int vo_host_p3p(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3])
{
    if (!obj || !img || !K || !rvec || !tvec) return VO_ERR_BAD_ARG;
    return pnp_host::p3p4(obj, img, K, rvec, tvec);
}

extern "C" int vo_host_rng_subsets(int count, int nsubsets, int* idx_out)
{
    if (count < 5 || nsubsets < 0 || !idx_out) return VO_ERR_BAD_ARG;
    pnp_host::Rng rng;
    for (int s = 0; s < nsubsets; s++) pnp_host::draw_subset(rng, count, 5, idx_out + (size_t)s * 5);
    return VO_OK;
}

extern "C" int vo_host_rodrigues(const double rvec[3], double R[9])
{
    if (!rvec || !R) return VO_ERR_BAD_ARG;
    pnp_host::rodrigues_to_matrix(rvec, R, nullptr);
    return VO_OK;
}

extern "C" int vo_pyramid_info(vo_ctx* c, int level, int* w, int* h, int* max_level)
{
    CHECK_CTX(c);
    if (max_level) *max_level = c->nlevels;
    if (level < 0 || level > c->nlevels) return vo_fail(c, VO_ERR_BAD_ARG, "level %d outside [0, %d]", level, c->nlevels);
    if (w) *w = c->lv[level].w;
    if (h) *h = c->lv[level].h;
    return VO_OK;
}

extern "C" int vo_read_tap(vo_ctx* c, int tap, int b, int arg0, int arg1, void* out, size_t bytes)
{
    CHECK_CTX(c); CHECK_B(c, b);
    if (!out) return vo_fail(c, VO_ERR_BAD_ARG, "null output");
    const size_t W = c->W, H = c->H;
    switch (tap) {
    case VO_TAP_FAST_SCORE:
        if (!c->have_features || bytes != W * H) return vo_fail(c, VO_ERR_BAD_ARG, "tap size/state");
        return d2h(c, out, c->score + (size_t)b * W * H, bytes);
    case VO_TAP_DISP_PREMEDIAN:
        if (!c->have_disp || bytes != W * H * 2) return vo_fail(c, VO_ERR_BAD_ARG, "tap size/state");
        return d2h(c, out, c->disp_pre + (size_t)b * W * H, bytes);
    case VO_TAP_COST_ROW: {
        if (!c->have_disp || arg0 < 0 || arg0 > 3 || bytes != (size_t)c->width1 * c->D * 2) return vo_fail(c, VO_ERR_BAD_ARG, "tap size/state");
        int r = arg1 - c->stripe_y0[arg0];
        if (r < 0 || r >= c->stripe_rows[arg0] || arg1 < arg0 * c->stripe_sz)
            return vo_fail(c, VO_ERR_BAD_ARG, "row %d is not an output row of stripe %d (warm-up rows are not kept)", arg1, arg0);
        VO_HIP(c, hipMemcpy2DAsync(out, (size_t)c->D * 2, c->cost + (((size_t)b * c->srows + c->stripe_row_off[arg0] + r) * c->width1) * c->Dp,
                                   (size_t)c->Dp * 2, (size_t)c->D * 2, (size_t)c->width1, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
        return VO_OK;
    }
    case VO_TAP_PYR_LEVEL:
    case VO_TAP_DERIV_LEVEL: {
        if (!c->have_pyr || arg0 < 0 || arg0 > 3 || arg1 < 0 || arg1 > c->nlevels) return vo_fail(c, VO_ERR_BAD_ARG, "tap size/state");
        const PyrLevel& L = c->lv[arg1];
        const size_t esz = tap == VO_TAP_PYR_LEVEL ? 1 : 4;
        if (bytes != (size_t)L.w * L.h * esz) return vo_fail(c, VO_ERR_BAD_ARG, "tap size");
        const char* base = tap == VO_TAP_PYR_LEVEL
            ? (const char*)(c->pyr + ((size_t)b * (c->seq_mode ? 2 : 4) + arg0) * c->pyr_bytes + L.img_off)
            : (const char*)(c->der + ((size_t)b * (c->seq_mode ? 2 : 4) + arg0) * c->der_elems + L.der_off);
        const char* src = base + ((size_t)c->border * L.pitch + c->border) * esz;
        VO_HIP(c, hipMemcpy2DAsync(out, (size_t)L.w * esz, src, (size_t)L.pitch * esz, (size_t)L.w * esz, L.h, hipMemcpyDeviceToHost, c->stream));
        VO_HIP(c, hipStreamSynchronize(c->stream));
        return VO_OK;
    }
    }
    return vo_fail(c, VO_ERR_BAD_ARG, "unknown tap %d", tap);
}
