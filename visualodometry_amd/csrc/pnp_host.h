// Host side of VOClass::estimateMotion (reference Source/VOClass.cpp:531-599): the parts of
// cv::solvePnPRansac that stay on the CPU — OpenCV's RNG and subset sampling, the 5-point EPnP minimal
// solver, the sequential accept / iteration-count rule replayed over GPU-scored hypotheses, and the final
// iterative solve (DLT start + Levenberg-Marquardt) on the inliers — plus Rodrigues and the 4x4 inverse.
#ifndef VO_PNP_HOST_H
#define VO_PNP_HOST_H

#include "vo_ctx.h"

namespace pnp_host {

// OpenCV's multiply-with-carry generator, seeded like RNG((uint64)-1)
struct Rng {
    uint64_t state;
    explicit Rng(uint64_t s = (uint64_t)-1) : state(s ? s : 0xffffffffULL) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
void draw_subset(Rng& rng, int count, int model_points, int* idx);
int update_num_iters(double p, double ep, int model_points, int max_iters);

void rodrigues_to_matrix(const double r[3], double R[9], double* jac /* nullable 3x9 */);
void rodrigues_from_matrix(const double R[9], double r[3]);
bool inv4x4(const double* A, double* Ainv);

// solvePnP(SOLVEPNP_EPNP) on float correspondences; K row-major 3x3
void epnp(const vo_point3f* obj, const vo_point2f* img, int n, const double K[9], double rvec[3], double tvec[3]);
// the same solve for exactly five correspondences on fixed-size arrays (epnp5.h, the code the GPU runs with pnp_device = 1)
void epnp5_fixed(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3]);
// solvePnP(SOLVEPNP_ITERATIVE) without an extrinsic guess on f64 correspondences.
// (planar objects start from the plane's homography, others from the 12-parameter DLT);
// returns 1, or -3 fewer than 6 points on a non-planar object, -4 degenerate
int iterative(const double* obj, const double* img, int n, const double K[9], double rvec[3], double tvec[3]);

// solvePnP(SOLVEPNP_P3P) on exactly four float correspondences (what solvePnPRansac runs when handed four points);
// returns 1, or 0 when the three-point problem has no real solution
int p3p4(const vo_point3f* obj, const vo_point2f* img, const double K[9], double rvec[3], double tvec[3]);

// RANSAC + refine for every frame of the current batch; hypotheses are scored by the GPU kernel.
int estimate_motion_batch(vo_ctx* c, double* Rt_out, vo_frame_info* info);

} // namespace pnp_host
#endif
