// Internal context of the gfx950 stereo front-end: device buffers, derived sizes, launch prototypes.
#ifndef VO_CTX_H
#define VO_CTX_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include <chrono>
#include "../../include/vo_abi.h"

#define VO_MAX_LEVELS 8
#define VO_NUM_STAGES 16

struct PyrLevel {
    int w, h;          // level size
    int pitch;         // bordered row pitch in pixels (w + 2*border, rounded up to 16)
    int rows;          // h + 2*border
    size_t img_off;    // byte offset of the bordered u8 level inside one image's pyramid block
    size_t der_off;    // element offset (int16 pairs counted as 2) inside one image's derivative block
};

struct vo_ctx {
    vo_params p;
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // optional: corner / tracking chain of a step, beside the disparity chain on `stream`
    hipEvent_t join2 = nullptr;
    hipEvent_t after_cost = nullptr; // recorded behind the cost kernel: a batch's tracker on stream2 waits for it
    bool tracker_at_once = false, single_stream = false; // VO_DUAL_STREAM: no such wait; VO_SINGLE_STREAM: no second stream
    hipStream_t stream_hi = nullptr; // highest priority: the short scoring round trips of the pose solver
    std::string err;

    // derived
    int W = 0, H = 0, D = 0, width1 = 0;
    int Dp = 0;        // D rounded up to 32: the disparity stride of the cost volumes and what the kernels are instantiated for;
                       // cells d >= D carry a cost no path can prefer and are never winners
    int dead_cost = 0; // that cost
    int stripe_sz = 0, overlap = 0;
    int stripe_y0[4], stripe_rows[4], stripe_row_off[5]; // source row range and offset into the stripe-row volume
    int srows = 0;                                       // rows of the cost volume per frame (sum of stripe rows)
    int ftzero = 15, P1 = 0, P2 = 0, disp12 = 1;
    int nlevels = 0;   // maxLevel actually built (levels 0..nlevels)
    int border = 0;    // = lk_win
    PyrLevel lv[VO_MAX_LEVELS];
    size_t pyr_bytes = 0;   // bytes of one image's bordered pyramid
    size_t der_elems = 0;   // int16 elements of one image's derivative pyramid

    int n = 0; // frames in the current batch
    double P0[12], P1m[12];
    bool have_calib = false;

    // images: [4] = LT1, RT1, LT2, RT2
    const uint8_t* img[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pitch = 0, fstride = 0;
    uint8_t* img_own = nullptr; size_t own_pitch = 0, own_fstride = 0;
    // consecutive frames of one sequence: pair b is (frame b, frame b + 1), so img[2] = img[0] + fstride and
    // img[3] = img[1] + fstride, and the pyramids are built once per frame ([n + 1][2]) instead of [n][4]
    bool seq_mode = false;

    // ---- SGBM ----
    uint8_t* pre = nullptr;      // prefiltered rows: [n][2 images][H][vo_pre_pitch] x 8 bytes (g,g0,g1,0,r,r0,r1,0)
    int16_t* cost = nullptr;     // [n][srows][width1][D]
    int16_t* agg = nullptr;      // [n][srows][width1][D]  L_top, then L_top + L_lr
    int32_t* wta = nullptr;      // [n][srows][W] packed (minS << 16) | (d16 & 0xffff)
    int* ckpt = nullptr;         // [n][H][segments][16 lanes][D / 32 + 1]: left-to-right path state at segment starts
    int16_t* disp_pre = nullptr; // [n][H][W]
    int16_t* disp16 = nullptr;   // [n][H][W]
    float* disp32 = nullptr;     // [n][H][W]
    double* depth = nullptr;     // [n][H][W]
    bool have_disp = false, have_depth = false;

    // ---- FAST ----
    uint8_t* score = nullptr;    // [n][H][W]
    int* row_count = nullptr;    // [n][H]
    unsigned long long* kp_mask = nullptr; // [n][H][ceil((W - 6) / 64)] keypoint flags of a row, column 3 first
    int* n_feat = nullptr;       // [n]  (device)
    vo_point2f* feat = nullptr;  // [n][max_features]
    std::vector<int> h_nfeat;    // host copy
    bool have_features = false;

    // ---- LK ----
    uint8_t* pyr = nullptr;      // [n][4 images][pyr_bytes]
    int16_t* der = nullptr;      // [n][4 images][der_elems]
    vo_point2f* legs = nullptr;  // [n][4][max_features]
    uint8_t* leg_status = nullptr; // [n][4][max_features]
    vo_point2f* m1 = nullptr;    // [n][max_features]
    vo_point2f* m2 = nullptr;
    int* n_match = nullptr;      // [n]
    std::vector<int> h_nmatch;
    bool have_pyr = false, have_matches = false;

    // ---- PnP ----
    vo_point3f* obj = nullptr;   // [n][max_features]
    vo_point2f* obj_img = nullptr;
    int* n_obj = nullptr;
    std::vector<int> h_nobj;
    double* hyp = nullptr;       // [n][chunk][12]
    int* hyp_count = nullptr;    // [n][chunk]
    uint8_t* hyp_mask = nullptr; // [n][chunk][max_features]
    int* hyp_sel = nullptr;      // [max_batch] hypothesis of a round whose inlier mask the host wants (mask pass)
    int* hyp_idx = nullptr;      // pnp_device: [max_batch][chunk][5] subset indices, then [max_batch] hypotheses of the round per frame
    double* hyp_rt = nullptr;    // pnp_device: [n][chunk][6] rvec, tvec of the round's hypotheses
    bool have_obj = false;

    // split step: GPU part enqueued, host part pending
    bool step_pending = false;
    hipEvent_t step_done = nullptr;
    std::chrono::steady_clock::time_point step_t0;

    // pinned host staging
    void* h_stage = nullptr; size_t h_stage_bytes = 0;
    vo_cloud_point* cloud = nullptr; int* cloud_rows = nullptr; // dense point cloud of one frame, built on request
    int* h_counts = nullptr; // pinned: [3][max_batch] feature / match / object-point counts of the pending step

    // optional per-stage timing with HIP events on the context's stream
    bool timing = false;
    bool timing_exclusive = false; // steps stay on one stream, so that a stage's interval holds that stage's kernels alone
    hipEvent_t ev0[VO_NUM_STAGES], ev1[VO_NUM_STAGES];
    bool ev_used[VO_NUM_STAGES];
    double stage_ms[VO_NUM_STAGES];
    long long stage_calls[VO_NUM_STAGES];
};

struct StageTimer {
    vo_ctx* c; int id;
    StageTimer(vo_ctx* c_, int id_) : c(c_), id(id_) { if (c->timing) { (void)hipEventRecord(c->ev0[id], c->stream); } }
    ~StageTimer() { if (c->timing) { (void)hipEventRecord(c->ev1[id], c->stream); c->ev_used[id] = true; } }
};
void vo_collect_timing(vo_ctx* c); // after a stream synchronise: adds the recorded intervals to stage_ms

int vo_fail(vo_ctx* c, int code, const char* fmt, ...);

#define VO_HIP(c, call)                                                                           \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) return vo_fail((c), VO_ERR_HIP, "%s: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// cost kernel tiling: 64 tile columns per block, block_size / 2 of them on each side are halo
inline size_t vo_cost_tiles(const vo_ctx* c)
{
    const int tile_out = 64 - 2 * (c->p.block_size / 2);
    return (size_t)((c->width1 + tile_out - 1) / tile_out);
}

// horizontal paths: the left-to-right state is checkpointed every segment; a segment's columns live in registers,
// so wide disparity ranges (more cells per lane) get shorter segments
// pixels per row of the prefiltered images: the width rounded up to the 4 pixels a prefilter thread writes
inline int vo_pre_pitch(const vo_ctx* c) { return (c->W + 3) & ~3; }
constexpr int vo_hpath_seg(int pairs_per_lane) { return pairs_per_lane <= 4 ? 16 : 8; }
inline int vo_hpath_segments(const vo_ctx* c) { const int seg = vo_hpath_seg(c->Dp / 32); return (c->width1 + seg - 1) / seg; }

// ---- launchers (each enqueues on c->stream, returns a vo_status) ----
int launch_fast(vo_ctx* c);
int launch_sgbm(vo_ctx* c);
int launch_depth_map(vo_ctx* c);
int launch_depth_cloud(vo_ctx* c, int b, int* count);
int vo_noalloc_prepare(int dev); // fast.hip: resolves the current device's scratch array of vo_detect_fast_noalloc
int launch_pyramids(vo_ctx* c);
int launch_match_klt(vo_ctx* c, bool counts_on_host);
int launch_triangulate(vo_ctx* c);
int launch_pnp_score(vo_ctx* c, int b0, int nb, int H, float thresh2, bool want_masks);
int launch_pnp_masks(vo_ctx* c, int nb, float thresh2); // masks of hypothesis c->hyp_sel[b] of every frame
int launch_pnp_solve(vo_ctx* c, int n); // pnp_solve.hip: 5-point solves of a round from c->hyp_idx into c->hyp and c->hyp_rt

#endif
