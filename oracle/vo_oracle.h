/*
 * CPU ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, single-pass, obviously-correct restatement of the arithmetic that the
 * reference's per-frame stereo front-end delegates to OpenCV 4.5.4
 * (reference call sites: Source/VOClass.cpp:231-305,312-366,370-623,
 * Source/VOClassFeatures.cpp:7-39,54-202, Source/VOClassUtils.cpp:61-91).
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or runnable
 * binary for this path, and OpenCV is neither vendored in the reference nor
 * installed here, so this restatement follows the published OpenCV 4.5.4
 * algorithms (stereosgbm.cpp, fast.cpp, fast_score.cpp, lkpyramid.cpp,
 * pyramids.cpp, solvepnp.cpp, ptsetreg.cpp, epnp.cpp, calibration.cpp,
 * rand.cpp) from knowledge of those sources.  Hand-checkable known-answer
 * tests live in tests/.
 *
 * Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may link or load this library; the shipped library never does.
 *
 * C ABI so the Python tests can load it with ctypes.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y; } orc_point2f;
typedef struct { float x, y, z; } orc_point3f;

/* Parameters; defaults = the literals in the reference. */
typedef struct {
    int width, height;
    /* StereoSGBM (Source/VOClass.cpp:238-264) */
    int min_disparity;      /* 0 */
    int num_disparities;    /* 96 */
    int block_size;         /* 11 */
    int p1, p2;             /* 968, 3872 */
    int disp12_max_diff;    /* 0 -> treated as 1 by the 3-way code */
    int pre_filter_cap;     /* 0 -> ftzero = max(0,15)|1 = 15 */
    int uniqueness_ratio;   /* 0 */
    int wta_simd_lanes;     /* 8: tie rule of OpenCV's 128-bit SIMD WTA; 0: scalar rule (lowest d) */
    /* FAST (Source/VOClassFeatures.cpp:27-29) */
    int fast_threshold;     /* 20 */
    int fast_nms;           /* 1 */
    /* LK (Source/VOClassFeatures.cpp:69-72,101-102) */
    int lk_win;             /* 15 */
    int lk_max_level;       /* 3 */
    int lk_max_count;       /* 50 */
    double lk_epsilon;      /* (double)0.03f */
    double lk_min_eig;      /* 1e-4 */
    /* circular matching (Source/VOClassFeatures.cpp:172) */
    int loop_threshold;     /* 2 */
    /* triangulation (Source/VOClass.cpp:400) */
    int depth_thresh;       /* 3000 */
    /* solvePnPRansac defaults */
    int ransac_iters;       /* 100 */
    float ransac_reproj;    /* 8.0 */
    double ransac_conf;     /* 0.99 */
    /* accumulators of the LK tracker (oracle/lk.cpp): 0 exact integer sums, 1 float in scan order, 4 float in the order of
     * OpenCV's NEON block (arm64, the reference author's machine), 8 float in the order of its 128-bit x86 block */
    int lk_accum;           /* 0 */
} orc_params;

void orc_default_params(orc_params* p, int width, int height);

/* ---- FAST-9/16 + NMS (cv::FAST(img, kp, threshold, true)) ---- */
/* returns number of keypoints (may exceed cap; only cap are written). score_out nullable (H*W u8). */
int orc_fast(const uint8_t* img, int width, int height, int stride, int threshold, int nms,
             orc_point2f* out, uint8_t* response_out, int cap, uint8_t* score_map_out);

/* ---- StereoSGBM MODE_SGBM_3WAY + medianBlur 3x3; output CV_16S x16 ---- */
/* premedian_out nullable. returns 0 ok. */
int orc_sgbm3way(const uint8_t* left, const uint8_t* right, const orc_params* p,
                 int16_t* disp16_out, int16_t* premedian_out);
/* debug taps for kernel-level tests: aggregated cost C of one stripe-local row. */
int orc_sgbm_cost_row(const uint8_t* left, const uint8_t* right, const orc_params* p,
                      int stripe, int y, int16_t* cost_out /* width1*D */);
/* convertTo(CV_32F, 1/16) (Source/VOClass.cpp:296) */
void orc_disp_to_float(const int16_t* disp16, int n, float* out);
/* computeDepthMap (Source/VOClass.cpp:312-366): patches disp in place, writes f64 depth */
void orc_depth_map(float* disp, int width, int height, double fx, double baseline, double* depth_out);

/* ---- pyramids / derivatives (buildOpticalFlowPyramid, calcSharrDeriv) ---- */
/* number of levels that calcOpticalFlowPyrLK would use: returns maxLevel actually built */
int orc_pyr_levels(int width, int height, int win, int max_level);
void orc_pyr_level_size(int width, int height, int level, int* w, int* h);
/* pyrDown (separable [1 4 6 4 1], (s+128)>>8, REFLECT_101) */
void orc_pyrdown(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh);
/* Scharr derivative, interleaved (Ix,Iy) s16, no border */
void orc_scharr(const uint8_t* src, int w, int h, int16_t* dst);

/* ---- calcOpticalFlowPyrLK(prev,next,prevPts,nextPts,status,err,win,maxLevel,crit) ---- */
int orc_lk(const uint8_t* prev, const uint8_t* next, int width, int height,
           const orc_point2f* prev_pts, int n, orc_point2f* next_pts, uint8_t* status,
           float* err /* nullable: still applies the level-0 final bounds check as if given */,
           const orc_params* p);

/* ---- VOClass::matchFeatureKLT (Source/VOClassFeatures.cpp:54-202) ---- */
/* in/out pts1 (n in, m out), out pts2; returns m. legs_out nullable: 4*n points + 4*n status bytes */
int orc_match_klt(const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                  const orc_params* p, orc_point2f* pts1_inout, int n, orc_point2f* pts2_out,
                  orc_point2f* legs_out, uint8_t* leg_status_out);

/* ---- estimateMotion part 1 (Source/VOClass.cpp:396-441) ---- */
int orc_triangulate(const orc_point2f* t1, const orc_point2f* t2, int m, const double* depth,
                    int width, int height, const double* P0 /*3x4*/, int depth_thresh,
                    orc_point3f* obj_out, orc_point2f* img1_out, orc_point2f* img2_out);

/* ---- OpenCV RNG ---- */
void orc_rng_subsets(int count, int model_points, int nsubsets, int* idx_out);

/* ---- solvePnPRansac (defaults) ---- */
typedef struct {
    int ok;             /* solvePnPRansac return value */
    int iters_run;      /* RANSAC iterations executed */
    int best_inliers;
    int final_from_refine; /* 1: rvec/tvec from the final solvePnP(ITERATIVE) on the inliers */
} orc_pnp_info;
int orc_solve_pnp_ransac(const orc_point3f* obj, const orc_point2f* img, int n, const double* K /*3x3*/,
                         const orc_params* p, double rvec[3], double tvec[3],
                         uint8_t* inlier_mask /* nullable, n */, orc_pnp_info* info);
/* pieces, for kernel-level tests */
/* Sensitivity switches (tests/golden/make_ocvk_sensitivity.py): each bit replaces ONE assumption about OpenCV 4.5.4's
 * internals by its plausible alternative, so that the effect of being wrong about it can be measured.  0 = the restatement. */
enum {
    ORC_VAR_PATH_START_C = 1,      /* first pixel of a path: L = C (instead of C - P2, zero state with zero minimum) */
    ORC_VAR_DISP12_ZERO = 2,       /* disp12MaxDiff <= 0 stays 0 (instead of being taken as 1) */
    ORC_VAR_RAW_BORDER_PIXEL = 4,  /* raw-intensity channel keeps the pixel in columns 0 and W-1 (instead of the table's zero entry) */
    ORC_VAR_LK_FLOAT_ACC = 8,      /* LK sums in float, in scan order (instead of exact integer sums): same as lk_accum = 1 */
    ORC_VAR_LK_NO_ERR_STATUS = 16  /* LK: no re-test of the final window position after the iterations */
};
void orc_set_variant(unsigned flags);
unsigned orc_get_variant(void);

/* exactly four correspondences (solvePnP(SOLVEPNP_P3P)); 1 = solved, 0 = no real solution */
int orc_p3p(const orc_point3f* obj, const orc_point2f* img, const double* K, double rvec[3], double tvec[3]);
/* pose start of the iterative solver for a planar object */
int orc_planar_start(const double* obj, const double* mn, int n, const double* Vt, const double* Mc, double param[6]);
int orc_epnp(const orc_point3f* obj, const orc_point2f* img, int n, const double* K, double rvec[3], double tvec[3]);
int orc_pnp_iterative(const double* obj /*n*3*/, const double* img /*n*2*/, int n, const double* K,
                      double rvec[3], double tvec[3]);
/* PnPRansacCallback::computeError */
void orc_pnp_errors(const orc_point3f* obj, const orc_point2f* img, int n, const double* K,
                    const double rvec[3], const double tvec[3], float* err_out);
void orc_rodrigues_vec2mat(const double r[3], double R[9], double* jac /* nullable 3x9 */);
void orc_rodrigues_mat2vec(const double R[9], double r[3]);
void orc_inv4x4(const double* A, double* Ainv);

/* ---- estimateMotion parts 2,3 (Source/VOClass.cpp:531-599): updates pose (4x4 row-major) in place ---- */
int orc_estimate_motion(const orc_point2f* t1, const orc_point2f* t2, int m, const double* depth,
                        const orc_params* p, const double* P0, double pose_inout[16],
                        double Rt_out[16] /* nullable */, double tpose_out[3], orc_pnp_info* info);

/* ---- one full iteration of Source/main.cpp:36-60 ---- */
typedef struct {
    int n_fast, n_matched, n_obj;
    orc_pnp_info pnp;
    double ms_disparity, ms_fast, ms_klt, ms_motion;
} orc_step_info;
int orc_step(const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
             const orc_params* p, const double* P0, const double* P1, double pose_inout[16],
             double Rt_out[16], orc_step_info* info);

/* the same iteration, also handing out what its stages produced (each pointer nullable): the s16 disparity map (W*H), the FAST
 * list (W*H/4 points at most; info->n_fast of them written) and the circular-matching survivors (info->n_matched pairs) */
int orc_step_ex(const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                const orc_params* p, const double* P0, const double* P1, double pose_inout[16],
                double Rt_out[16], orc_step_info* info, int16_t* disp16_out, orc_point2f* fast_out,
                orc_point2f* match_t1_out, orc_point2f* match_t2_out);

/* RMSE of Source/VOClass.cpp:627-649 (float accumulator as in the reference) */
float orc_trajectory_rmse(const double* est /*n*3*/, const double* gt /*n*3*/, int n);

#ifdef __cplusplus
}
#endif
#endif
