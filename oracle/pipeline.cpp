/* CPU ORACLE (test infrastructure only) — the reference's own glue around the OpenCV calls:
 * VOClass::estimateMotion (Source/VOClass.cpp:370-623), one iteration of main()'s loop
 * (Source/main.cpp:36-60) and the RMSE of Source/VOClass.cpp:627-649.
 */
#include "vo_oracle.h"
#include <vector>
#include <cstring>
#include <cmath>
#include <chrono>

extern "C" void orc_default_params(orc_params* p, int width, int height)
{
    memset(p, 0, sizeof(*p));
    p->width = width; p->height = height;
    p->min_disparity = 0; p->num_disparities = 96; p->block_size = 11;
    p->p1 = 8 * 1 * 11 * 11; p->p2 = 32 * 1 * 11 * 11;
    p->disp12_max_diff = 0; p->pre_filter_cap = 0; p->uniqueness_ratio = 0;
    p->wta_simd_lanes = 8;
    p->fast_threshold = 20; p->fast_nms = 1;
    p->lk_win = 15; p->lk_max_level = 3; p->lk_max_count = 50;
    p->lk_epsilon = (double)0.03f; p->lk_min_eig = 1e-4;
    p->loop_threshold = 2;
    p->depth_thresh = 3000;
    p->ransac_iters = 100; p->ransac_reproj = 8.0f; p->ransac_conf = 0.99;
}

/* Source/VOClass.cpp:409-436 */
extern "C" int orc_triangulate(const orc_point2f* t1, const orc_point2f* t2, int m, const double* depth,
                               int width, int height, const double* P0, int depth_thresh,
                               orc_point3f* obj_out, orc_point2f* img1_out, orc_point2f* img2_out)
{
    const double cx = P0[2], cy = P0[6], fx = P0[0], fy = P0[5];
    int k = 0;
    for (int i = 0; i < m; i++) {
        float u = t1[i].x, v = t1[i].y;
        int iu = (int)u, iv = (int)v; /* Mat::at<double>(v, u): float -> int truncation */
        if (iu < 0 || iu >= width || iv < 0 || iv >= height) continue; /* the reference would read out of range */
        float z = (float)depth[(size_t)iv * width + iu];
        if (z > depth_thresh) continue;
        float x = (float)(z * (u - cx) / fx);
        float y = (float)(z * (v - cy) / fy);
        if (img1_out) img1_out[k] = t1[i];
        img2_out[k] = t2[i];
        obj_out[k].x = x; obj_out[k].y = y; obj_out[k].z = z;
        k++;
    }
    return k;
}

static void mat4_mul(const double* A, const double* B, double* C)
{
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += A[r * 4 + k] * B[k * 4 + c];
        C[r * 4 + c] = s;
    }
}

extern "C" int orc_estimate_motion(const orc_point2f* t1, const orc_point2f* t2, int m, const double* depth,
                                   const orc_params* p, const double* P0, double pose[16], double Rt_out[16],
                                   double tpose_out[3], orc_pnp_info* info)
{
    std::vector<orc_point3f> obj(m > 0 ? m : 1);
    std::vector<orc_point2f> i2(m > 0 ? m : 1);
    int k = orc_triangulate(t1, t2, m, depth, p->width, p->height, P0, p->depth_thresh, obj.data(), nullptr, i2.data());
    double K[9] = {P0[0], P0[1], P0[2], P0[4], P0[5], P0[6], P0[8], P0[9], P0[10]};
    double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0};
    orc_pnp_info li; memset(&li, 0, sizeof(li));
    int ok = orc_solve_pnp_ransac(obj.data(), i2.data(), k, K, p, rvec, tvec, nullptr, &li);
    if (info) *info = li;
    double R[9], Rt[16] = {0}, Rti[16], np[16];
    orc_rodrigues_vec2mat(rvec, R, nullptr);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rt[r * 4 + c] = R[r * 3 + c]; Rt[r * 4 + 3] = tvec[r]; }
    Rt[15] = 1;
    if (Rt_out) memcpy(Rt_out, Rt, sizeof(Rt));
    orc_inv4x4(Rt, Rti);
    mat4_mul(pose, Rti, np);
    memcpy(pose, np, sizeof(np));
    tpose_out[0] = pose[3]; tpose_out[1] = pose[7]; tpose_out[2] = pose[11];
    return ok ? k : -k - 1;
}

extern "C" int orc_step_ex(const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                           const orc_params* p, const double* P0, const double* P1, double pose[16],
                           double Rt_out[16], orc_step_info* info, int16_t* disp16_out, orc_point2f* fast_out,
                           orc_point2f* match_t1_out, orc_point2f* match_t2_out)
{
    using clk = std::chrono::steady_clock;
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const int W = p->width, H = p->height;
    orc_step_info li; memset(&li, 0, sizeof(li));
    auto t0 = clk::now();
    std::vector<int16_t> d16((size_t)W * H);
    if (orc_sgbm3way(lt1, rt1, p, d16.data(), nullptr) != 0) return -1;
    if (disp16_out) memcpy(disp16_out, d16.data(), sizeof(int16_t) * (size_t)W * H);
    std::vector<float> disp((size_t)W * H);
    orc_disp_to_float(d16.data(), W * H, disp.data());
    std::vector<double> depth((size_t)W * H);
    double fx = P0[0], baseline = P1[3] / (-1 * fx);
    orc_depth_map(disp.data(), W, H, fx, baseline, depth.data());
    auto t1 = clk::now();
    int cap = W * H / 4;
    std::vector<orc_point2f> f1(cap), f2(cap);
    int n = orc_fast(lt1, W, H, W, p->fast_threshold, p->fast_nms, f1.data(), nullptr, cap, nullptr);
    if (n > cap) n = cap;
    if (fast_out) memcpy(fast_out, f1.data(), sizeof(orc_point2f) * (size_t)n);
    auto t2 = clk::now();
    int m = orc_match_klt(lt1, rt1, lt2, rt2, p, f1.data(), n, f2.data(), nullptr, nullptr);
    if (match_t1_out) memcpy(match_t1_out, f1.data(), sizeof(orc_point2f) * (size_t)m);
    if (match_t2_out) memcpy(match_t2_out, f2.data(), sizeof(orc_point2f) * (size_t)m);
    auto t3 = clk::now();
    double tp[3];
    int k = orc_estimate_motion(f1.data(), f2.data(), m, depth.data(), p, P0, pose, Rt_out, tp, &li.pnp);
    auto t4 = clk::now();
    li.n_fast = n; li.n_matched = m; li.n_obj = k >= 0 ? k : -k - 1;
    li.ms_disparity = ms(t0, t1); li.ms_fast = ms(t1, t2); li.ms_klt = ms(t2, t3); li.ms_motion = ms(t3, t4);
    if (info) *info = li;
    return 0;
}

extern "C" int orc_step(const uint8_t* lt1, const uint8_t* rt1, const uint8_t* lt2, const uint8_t* rt2,
                        const orc_params* p, const double* P0, const double* P1, double pose[16],
                        double Rt_out[16], orc_step_info* info)
{
    return orc_step_ex(lt1, rt1, lt2, rt2, p, P0, P1, pose, Rt_out, info, nullptr, nullptr, nullptr, nullptr);
}

extern "C" float orc_trajectory_rmse(const double* est, const double* gt, int n)
{
    float error = 0;
    for (int i = 0; i < n; i++)
        error += pow(gt[3 * i] - est[3 * i], 2) + pow(gt[3 * i + 1] - est[3 * i + 1], 2) + pow(gt[3 * i + 2] - est[3 * i + 2], 2);
    return sqrt(error / n);
}
