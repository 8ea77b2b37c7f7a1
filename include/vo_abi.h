/*
 * C ABI of the MI355X (gfx950) stereo visual-odometry front-end.
 *
 * The reference has no FFI: its boundary is the public method set of `class VOClass`
 * (reference Include/VOClass.h:62-123), whose five hot-path methods are thin wrappers over one
 * OpenCV call each.  This header is what a binding of that class binds instead of OpenCV:
 * every entry point below names the reference method (file:line) whose arithmetic it replaces.
 * include/VOClass.h is the header-only C++ class with the reference's method names on top of it.
 *
 * Conventions
 *  - plain C, no exceptions, no C++ types; every call returns 0 (VO_OK) or a negative vo_status;
 *    vo_last_error(ctx) gives the text of the last failure on that context.
 *  - a context owns one GPU stream and all device buffers; it is used by one thread at a time;
 *    different contexts are independent.  Nothing is logged.
 *  - a context processes a BATCH of n independent frame pairs per call (1 <= n <= max_batch);
 *    the reference's one-pair-at-a-time use is n = 1.  Frame b of a batch is addressed by index b.
 *  - images are 8-bit grayscale, row-major, `stride` bytes per row, frames of a batch `frame_stride`
 *    bytes apart.  Points are (x, y) float pairs as cv::Point2f.
 *  - the library fails (VO_ERR_NO_DEVICE / VO_ERR_HIP) when no gfx950 device or code object is
 *    available; there is no CPU fallback.
 */
#ifndef VO_ABI_H
#define VO_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

typedef struct { float x, y; } vo_point2f;
typedef struct { float x, y, z; } vo_point3f;

typedef enum {
    VO_OK = 0,
    VO_ERR_BAD_ARG = -1,      /* null pointer, size out of range, n > max_batch ... */
    VO_ERR_BAD_PARAM = -2,    /* parameter combination outside the supported domain */
    VO_ERR_HIP = -3,          /* a HIP runtime call failed */
    VO_ERR_NO_DEVICE = -4,
    VO_ERR_STATE = -5,        /* stage called before the stage that produces its input */
    VO_ERR_CAPACITY = -6,     /* more features than max_features; results were truncated */
    VO_ERR_NOT_ENOUGH_POINTS = -7, /* a frame has fewer than 4 3D-2D correspondences (where OpenCV throws).  In a batch the
                                    * other frames are complete: see vo_step */
    VO_ERR_UNSUPPORTED = -8
} vo_status;

/* Every literal of the reference's hot path as a run-time field; vo_default_params fills in the
 * reference's values (Source/VOClass.cpp:13-14,238-254,264,400; Source/VOClassFeatures.cpp:27-28,
 * 69-70,101-102,172) and OpenCV 4.5.4's defaults for what the reference leaves unset. */
typedef struct {
    int width, height;        /* frameW, frameH */
    int max_batch;            /* frame pairs per call */
    int max_features;         /* capacity per image; FAST output beyond it is truncated (row-major order) */
    /* StereoSGBM, MODE_SGBM_3WAY */
    int min_disparity;        /* 0 (only 0 is supported) */
    int num_disparities;      /* 96; multiple of 16 in [16, 256] */
    int block_size;           /* 11; odd, <= 13 (aggregated cost must fit 14 bits) */
    int p1, p2;               /* 968, 3872 */
    int disp12_max_diff;      /* 0 -> 1 */
    int pre_filter_cap;       /* 0 -> 15 */
    int uniqueness_ratio;     /* 0 (only 0 is supported) */
    int wta_simd_lanes;       /* 8: winner-take-all tie rule of OpenCV's 128-bit SIMD loop; 0: scalar rule */
    /* FAST */
    int fast_threshold;       /* 20 */
    int fast_nms;             /* 1 */
    /* calcOpticalFlowPyrLK */
    int lk_win;               /* 15 (odd, <= 31) */
    int lk_max_level;         /* 3 */
    int lk_max_count;         /* 50 */
    double lk_epsilon;        /* (double)0.03f */
    double lk_min_eig;        /* 1e-4 */
    int loop_threshold;       /* 2 */
    /* estimateMotion */
    int depth_thresh;         /* 3000 */
    int ransac_iters;         /* 100 */
    float ransac_reproj;      /* 8.0 */
    double ransac_conf;       /* 0.99 */
    int ransac_chunk;         /* hypotheses solved on the host and scored on the GPU per round (8; at most 1024) */
    int host_threads;         /* host worker threads for the minimal solvers of a batch (0 = auto) */
    /* Accumulators of the tracker's structure tensor and mismatch vector.  0: exact integer sums (order free), the only mode
     * this library computes.  OpenCV sums them in float (its integer accumulators belong to 32-bit ARM builds without NEON),
     * partly in 4-lane vector partial sums whose order depends on the instruction set; the CPU oracle restates those orders
     * as modes 1, 4 (NEON) and 8 (x86) of the same field, where 99.9 % of the tracks stay within 0.01 px of mode 0 and no
     * relative pose moves by more than 5e-7 rad / 2e-5 m.  Any other value here: VO_ERR_UNSUPPORTED from vo_create. */
    int lk_accum;             /* 0 */
    /* Where the 5-point minimal solves of the RANSAC rounds run.  0: on the host's worker threads (the reference's split:
     * only hypothesis scoring is batched on the GPU).  1: on the GPU, one thread per hypothesis, from the correspondences that
     * are already in device memory (the host still draws the subsets from OpenCV's generator, replays the sequential accept
     * rule and runs the final iterative solve).  Same arithmetic in the same order; the device's hypot / acos / sin / cos differ
     * from the host's in the last bits, which a five-point solve now and then amplifies into a visibly different hypothesis
     * (3 of 199 steps of the KITTI-shape test sequence end with a few inliers more or less; poses stay within 1e-4 rad / 1e-3 m).  For rounds whose solves outrun the host (1 024 hypotheses
     * per frame: 2.3 ms against 77 ms on 16 threads).  A round costs 1.2 ms of latency however few its solves (one thread's
     * serial work), so at the reference's 8 hypotheses per round the host is the faster place, with 16 threads or with 2. */
    int pnp_device;           /* 0 */
} vo_params;

typedef struct {
    int ok;                   /* solvePnPRansac's return value */
    int iters_run;            /* RANSAC iterations the serial loop would have executed */
    int best_inliers;
    int final_from_refine;    /* 1: pose is the final iterative solve on the inliers */
    int n_features, n_matched, n_object_points;
} vo_frame_info;

/* fills *p with the reference's literals for a width x height frame, max_batch 1 */
int vo_default_params(vo_params* p, int width, int height);

/* Context life cycle.  Replaces `VOClass VO;` (Source/main.cpp:9, Source/VOClass.cpp:8-28). */
int vo_create(const vo_params* p, int device, vo_ctx** out);
void vo_destroy(vo_ctx* ctx);
const char* vo_last_error(const vo_ctx* ctx);
/* last error of a failed vo_create (ctx == NULL there) */
const char* vo_create_error(void);
/* stream of the context as a hipStream_t, for callers that time or order work against it */
void* vo_stream(vo_ctx* ctx);
int vo_synchronize(vo_ctx* ctx);

/* projectionCL / projectionCR, row-major 3x4.  Replaces the state filled by
 * VOClass::getProjectionMatrices (Source/VOClass.cpp:102-138). */
int vo_set_calib(vo_ctx* ctx, const double P0[12], const double P1[12]);

/* imgLT1, imgRT1, imgLT2, imgRT2 of n frame pairs.  Replaces the member images filled by
 * VOClass::readStereoImagesT1T2 (Source/VOClass.cpp:36-83; Include/VOClass.h:65-66).
 * Host pointers are copied to the device; device pointers are used in place and must stay valid
 * until the last stage that reads them has been synchronised. */
int vo_set_images(vo_ctx* ctx, int n, const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2,
                  const uint8_t* rt2, size_t stride, size_t frame_stride);
int vo_set_images_device(vo_ctx* ctx, int n, const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2,
                         const uint8_t* rt2, size_t stride, size_t frame_stride);

/* n_pairs + 1 consecutive frames of one sequence (left[f], right[f] at f * frame_stride): pair b is
 * (frame b, frame b + 1), i.e. what n_pairs iterations of the reference's main loop read one after the other
 * (Source/main.cpp:36-39, Source/VOClass.cpp:36-83, which re-reads frame t+1 as the next step's frame t).
 * Results are identical to vo_set_images with the four shifted arrays; every frame is uploaded once and its
 * pyramid and derivatives are built once instead of twice. */
int vo_set_sequence(vo_ctx* ctx, int n_pairs, const uint8_t* left, const uint8_t* right, size_t stride,
                    size_t frame_stride);
int vo_set_sequence_device(vo_ctx* ctx, int n_pairs, const uint8_t* left, const uint8_t* right, size_t stride,
                           size_t frame_stride);

/* VOClass::computeDisparity(imgLT1, imgRT1) (Source/VOClass.cpp:231-305): StereoSGBM 3-way + median,
 * s16 x16 map and its /16 float form, for all n frames. */
int vo_compute_disparity(vo_ctx* ctx);
int vo_get_disparity16(vo_ctx* ctx, int b, int16_t* out /* H*W */);
int vo_get_disparity32(vo_ctx* ctx, int b, float* out /* H*W */);

/* VOClass::computeDepthMap (Source/VOClass.cpp:312-366): patches the float disparity in place
 * (0 and -1 become 0.1f) and produces the f64 depth map. */
int vo_compute_depth_map(vo_ctx* ctx);
int vo_get_depth(vo_ctx* ctx, int b, double* out /* H*W */);
/* Caller-supplied maps.  The reference's methods work on their ARGUMENTS: computeDepthMap(cv::Mat disparityMap)
 * (Include/VOClass.h:85, Source/VOClass.cpp:312-366) converts whatever float map it is handed, and
 * estimateMotion(..., cv::Mat depthMap) (Include/VOClass.h:98-100, Source/VOClass.cpp:418) looks depths up in whatever f64
 * map it is handed.  vo_set_disparity32 replaces frame b's float disparity map (H*W floats, host memory) with the caller's,
 * vo_set_depth frame b's depth map (H*W doubles); the stages downstream (vo_compute_depth_map, vo_triangulate) then read
 * them.  In a batch of more than one frame the other frames must already hold computed maps. */
int vo_set_disparity32(vo_ctx* ctx, int b, const float* disp /* H*W */);
int vo_set_depth(vo_ctx* ctx, int b, const double* depth /* H*W */);
/* The vertices VOClass::writeToPLY lists for the depth map of frame b (Source/VOClassUtils.cpp:132-148, called from
 * Source/VOClass.cpp:330-360): every pixel whose depth, narrowed to float, is not above depth_thresh, in row-major
 * order, as (row, col, -z, grey of imgLT1).  Compacted on the GPU; out nullable (count only), at most cap records are
 * copied, *n_out is the full count. */
typedef struct { int row, col; float neg_z; uint8_t gray, pad[3]; } vo_cloud_point;
int vo_get_depth_cloud(vo_ctx* ctx, int b, vo_cloud_point* out, int cap, int* n_out);

/* VOClass::getFeaturesFAST(imgLT1) (Source/VOClassFeatures.cpp:7-39). */
int vo_detect_fast(vo_ctx* ctx);
int vo_get_features(vo_ctx* ctx, int b, vo_point2f* out, int cap, int* n_out);
/* caller-supplied feature list for frame b (the vector argument of matchFeatureKLT) */
int vo_set_features(vo_ctx* ctx, int b, const vo_point2f* pts, int n);

/* FAST corners of ONE image, allocating nothing.  No counterpart in the reference: it exists for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same
 * kernels and same results as vo_detect_fast + vo_get_features; the kernels' scratch (score map, row counts)
 * lives in a fixed device array of the library instead of a context, so the image is limited to 2^21 pixels and 4096
 * rows (VO_ERR_CAPACITY beyond; the text is in vo_create_error()).  img_dev, out_dev (cap points) and n_out_dev (one int:
 * min(number of corners, cap)) are device pointers on the CURRENT device; the work is enqueued on `stream` (a hipStream_t,
 * may be NULL) and not waited for.  All calls on one GPU share that GPU's scratch: calls that overlap on one GPU race.
 * The address of a GPU's scratch is looked up once (a code-object load and a lock, neither allowed under stream capture):
 * by vo_create for its device, or by the first call here on a device; a caller that captures makes one of the two first. */
int vo_detect_fast_noalloc(const uint8_t* img_dev, int width, int height, size_t stride, int threshold, int nms,
                           vo_point2f* out_dev, int cap, int* n_out_dev, void* stream);

/* VOClass::matchFeatureKLT (Source/VOClassFeatures.cpp:54-202): four pyramidal LK legs
 * LT1->RT1->RT2->LT2->LT1, bounds filter, 4-way status AND, loop-closure test.
 * t1 receives the re-tracked LT1 points (the reference overwrites its in/out argument with them),
 * t2 the matching LT2 points. */
int vo_match_klt(vo_ctx* ctx);
int vo_get_matches(vo_ctx* ctx, int b, vo_point2f* t1, vo_point2f* t2, int cap, int* m_out);
/* per-leg end points (4 x n) and status bytes after the bounds filter (4 x n); n = feature count */
int vo_get_klt_legs(vo_ctx* ctx, int b, vo_point2f* pts /* 4*n */, uint8_t* status /* 4*n */, int n);
int vo_set_matches(vo_ctx* ctx, int b, const vo_point2f* t1, const vo_point2f* t2, int m);

/* VOClass::estimateMotion part 1 (Source/VOClass.cpp:396-441): depth look-up, depth filter,
 * back-projection.  Uses the depth map if vo_compute_depth_map ran, else derives depth from the
 * disparity on the fly (same values). */
int vo_triangulate(vo_ctx* ctx);
int vo_get_object_points(vo_ctx* ctx, int b, vo_point3f* obj, vo_point2f* img2, int cap, int* k_out);
/* The same for the whole batch with one strided copy per array: frame b's points go to obj[b * cap ..] (img2 likewise; each
 * nullable), at most cap per frame; counts (n ints, nullable) receives every frame's full count.  obj and img2 may point to
 * host memory or to memory of the context's GPU (a result gather over RCCL takes them without a trip to the host); with both
 * NULL the call copies nothing and does not synchronise.  This is the point-cloud
 * output that a sequence-sharded run gathers at its end (objectPoints of Source/VOClass.cpp:409-436, one set per frame pair). */
int vo_get_object_points_batch(vo_ctx* ctx, vo_point3f* obj, vo_point2f* img2, int cap, int* counts);

/* Scoring step of cv::solvePnPRansac (Source/VOClass.cpp:531): reprojection error of every
 * correspondence of frame b under H hypotheses (R row-major 3x3 then t: 12 doubles each), inlier
 * iff squared error <= thresh2.  counts: H ints; masks nullable: H * k bytes. */
int vo_pnp_score(vo_ctx* ctx, int b, const double* Rt12, int H, float thresh2, int* counts, uint8_t* masks);

/* VOClass::estimateMotion parts 2 and 3 (Source/VOClass.cpp:531-599) for all n frames:
 * RANSAC (host minimal solver + GPU scoring), final iterative solve, Rodrigues, Rt.
 * Rt_out: n x 16 doubles (row-major 4x4 [R|t; 0 0 0 1]); the caller integrates
 * pose = pose * inv(Rt) (vo_integrate_pose). */
int vo_estimate_motion(vo_ctx* ctx, double* Rt_out, vo_frame_info* info /* nullable, n */);
/* poseRt = poseRt * Rt.inv() (Source/VOClass.cpp:587), 4x4 row-major f64, LU inverse */
int vo_integrate_pose(double pose_inout[16], const double Rt[16]);

/* One iteration of Source/main.cpp:36-60 for n frame pairs, images already set:
 * disparity, FAST, circular KLT, triangulation, PnP.  Same results as the stage calls in sequence.
 * Frames of a batch are independent.  If some frames end with too few correspondences the call returns
 * VO_ERR_NOT_ENOUGH_POINTS, but Rt_out and info are filled for every frame: the frames that failed have ok = 0 and the
 * identity as [R|t], all others hold their results. */
int vo_step(vo_ctx* ctx, double* Rt_out /* n*16 */, vo_frame_info* info /* nullable, n */);
/* The same step in two halves, so that a caller with two contexts can overlap the host part (minimal
 * solvers, final refine) of one batch with the GPU part of the next: vo_step_submit enqueues every GPU
 * stage and returns without waiting; vo_step_finish waits for them and runs the host part.
 * vo_step(ctx, ...) == vo_step_submit(ctx) followed by vo_step_finish(ctx, ...).
 * Streams: a context enqueues the disparity chain on one stream and the corner / tracking chain on a second one; the pose
 * solver's short round trips use one high-priority stream shared by all contexts of a device.  A step of ONE frame pair
 * leaves most of the GPU idle, so its tracker starts at once (2.2 -> 1.7 ms per step at KITTI shape).  In a batch every
 * large kernel fills the GPU, and the tracker waits for the cost kernel, the other kernel bound by vector issue, so that it
 * runs beside the horizontal path passes, which memory and latency bound (up to 128 disparities and a 15 x 15 tracking window;
 * beyond, where one of the two dwarfs the other, a batch stays on one stream).  A process has 4 hardware queues unless
 * GPU_MAX_HW_QUEUES (read when the runtime starts) says otherwise: three contexts in flight want 8 (5 850 -> 5 995 frames/s;
 * with 4 the second stream neither helps nor hurts).  Environment, read by vo_create: VO_SINGLE_STREAM keeps a context on
 * one stream, VO_DUAL_STREAM starts a batch's tracker at once as well. */
int vo_step_submit(vo_ctx* ctx);
int vo_step_finish(vo_ctx* ctx, double* Rt_out /* n*16 */, vo_frame_info* info /* nullable, n */);

/* Test taps: intermediate buffers of frame b copied to the host. */
typedef enum {
    VO_TAP_FAST_SCORE = 1,     /* u8 H*W */
    VO_TAP_DISP_PREMEDIAN = 2, /* s16 H*W */
    VO_TAP_COST_ROW = 3,       /* s16 width1*D: aggregated cost, arg0 = stripe, arg1 = image row */
    VO_TAP_PYR_LEVEL = 4,      /* u8 h*w of pyramid level arg1 of image arg0 (0 LT1, 1 RT1, 2 LT2, 3 RT2) */
    VO_TAP_DERIV_LEVEL = 5     /* s16 h*w*2 Scharr derivative of the same */
} vo_tap;
int vo_read_tap(vo_ctx* ctx, int tap, int b, int arg0, int arg1, void* out, size_t bytes);
/* size in pixels of pyramid level `level`, and the number of levels built (maxLevel) */
int vo_pyramid_info(vo_ctx* ctx, int level, int* w, int* h, int* max_level);

/* Optional per-stage timing (HIP events on the context's stream around each kernel group; host wall
 * clock for the host-side PnP part and the whole step).  ms: VO_NUM_STAGES accumulated milliseconds,
 * calls: how many times each stage ran since the last reset.  enable = 2 also keeps every step on ONE stream, so that an
 * interval holds its stage's kernels alone (with the tracking chain on its second stream, kernels of the two chains share
 * the GPU and each other's intervals).  No counterpart in the reference. */
#define VO_NUM_STAGES 16
typedef enum {
    VO_ST_PREFILTER = 0, /* 1: not in use */ VO_ST_VPATH = 2, VO_ST_HPATH_FWD = 3, VO_ST_HPATH_BWD = 4, VO_ST_LRCHECK = 5 /* consistency check + median */,
    VO_ST_PNP_SOLVE = 6 /* minimal solves on the GPU (pnp_device = 1) */, VO_ST_FAST = 7, VO_ST_PYRAMID = 8, VO_ST_LK = 9, VO_ST_FILTER = 10, VO_ST_TRIANGULATE = 11,
    VO_ST_PNP_SCORE = 12, VO_ST_PNP_HOST = 13, VO_ST_STEP_WALL = 14,
    VO_ST_PNP_SERIAL = 15 /* host solver sections that found the process's worker pool busy with another thread's section and ran
                           * on the calling thread alone (one pool per process, shared by its contexts) */
} vo_stage;
int vo_set_timing(vo_ctx* ctx, int enable);
int vo_get_timing(vo_ctx* ctx, double* ms, long long* calls, int reset);
const char* vo_stage_name(int stage);

/* Host-side solvers of estimateMotion, callable without a GPU (used by vo_estimate_motion internally).
 * vo_host_epnp: solvePnP(SOLVEPNP_EPNP) on n >= 4 float correspondences; K row-major 3x3.
 * vo_host_pnp_iterative: solvePnP(SOLVEPNP_ITERATIVE) without a guess on f64 correspondences (a planar object starts from
 *   the plane's homography, any other from the 12-parameter DLT, then Levenberg-Marquardt); returns 1,
 *   or -3 fewer than 6 points on a non-planar object, -4 degenerate.
 * vo_host_p3p: solvePnP(SOLVEPNP_P3P) on exactly 4 float correspondences, which is what solvePnPRansac runs when it is
 *   handed 4 points: returns 1, or 0 when there is no real solution.
 * vo_host_rng_subsets: the first nsubsets 5-index subsets cv::RNG((uint64)-1) yields for `count` points.
 * vo_host_rodrigues: rotation vector -> 3x3 matrix. */
int vo_host_epnp(const vo_point3f* obj, const vo_point2f* img, int n, const double K[9], double rvec[3], double tvec[3]);
/* the fixed-size 5-point form of the same solve that the GPU runs with pnp_device = 1, compiled for the host */
int vo_host_epnp5_fixed(const vo_point3f* obj /* 5 */, const vo_point2f* img /* 5 */, const double K[9], double rvec[3], double tvec[3]);
int vo_host_pnp_iterative(const double* obj, const double* img, int n, const double K[9], double rvec[3], double tvec[3]);
int vo_host_p3p(const vo_point3f* obj /* 4 */, const vo_point2f* img /* 4 */, const double K[9], double rvec[3], double tvec[3]);
int vo_host_rng_subsets(int count, int nsubsets, int* idx_out /* nsubsets*5 */);
int vo_host_rodrigues(const double rvec[3], double R[9]);

#ifdef __cplusplus
}
#endif
#endif
