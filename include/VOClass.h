/*
 * Header-only C++ class with the reference's `VOClass` method set (reference Include/VOClass.h:62-123) on top of
 * the C ABI in vo_abi.h.  Same method names, argument order and side effects as the reference's hot-path
 * methods; the OpenCV value types are replaced by small POD-style types so that the header needs nothing but
 * the C ABI:
 *
 *     cv::Mat (CV_8UC1 / CV_32FC1 / CV_64FC1)  ->  vo::Image<uint8_t> / vo::Image<float> / vo::Image<double>
 *     std::vector<cv::Point2f>                 ->  std::vector<vo_point2f>
 *     cv::Mat 3x1 / 4x4 CV_64F                 ->  std::array<double, 3> / std::array<double, 16> (row-major)
 *
 * vo::Image shares its pixels between copies (like a cv::Mat header), which is what makes computeDepthMap's
 * in-place patch of the disparity map visible to the caller exactly as in the reference
 * (Source/VOClass.cpp:323-329).  Errors: the reference logs and asserts; this class throws vo::Error carrying
 * the vo_status code and the library's message.
 *
 * Out of scope here (host-side file parsing / GUI of the reference, not part of the GPU path):
 * readStereoImagesT1T2, getGroundTruthPath, the testShow* visualisers.  The caller fills imgLT1..imgRT2 and calls
 * setProjectionMatrices() with what getProjectionMatrices() would have parsed from calib.txt.
 */
#ifndef VO_VOCLASS_H
#define VO_VOCLASS_H

#include "vo_abi.h"

#include <array>
#include <cmath>
#include <memory>
#include <stdexcept>
#include <fstream>
#include <string>
#include <vector>

namespace vo {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

template <class T>
struct Image {
    int rows = 0, cols = 0;
    std::shared_ptr<std::vector<T>> data;
    Image() {}
    Image(int r, int c) : rows(r), cols(c), data(std::make_shared<std::vector<T>>((size_t)r * c)) {}
    bool empty() const { return !data || data->empty(); }
    T* ptr(int r = 0) { return data->data() + (size_t)r * cols; }
    const T* ptr(int r = 0) const { return data->data() + (size_t)r * cols; }
    T& at(int r, int c) { return (*data)[(size_t)r * cols + c]; }
    const T& at(int r, int c) const { return (*data)[(size_t)r * cols + c]; }
    Image clone() const { Image o(rows, cols); *o.data = *data; return o; }
};

} // namespace vo

class VOClass {
    int frameW, frameH;
    int depthThreshold = 3000;
    vo_ctx* ctx = nullptr;
    std::array<double, 12> projectionCL{}, projectionCR{};
    std::vector<std::array<double, 3>> groundTruth;

    void check(int rc) const { if (rc != VO_OK) throw vo::Error(rc, vo_last_error(ctx)); }

    /* uploads the four member images (or the two images passed to computeDisparity / getFeaturesFAST).  Always: the
     * reference re-reads its cv::Mat arguments and members on every call, so pixels rewritten in place must be seen */
    void syncImages(const vo::Image<uint8_t>* lt1 = nullptr, const vo::Image<uint8_t>* rt1 = nullptr)
    {
        const vo::Image<uint8_t>& a = lt1 ? *lt1 : imgLT1;
        const vo::Image<uint8_t>& b = rt1 ? *rt1 : imgRT1;
        const vo::Image<uint8_t>& c = imgLT2.empty() ? a : imgLT2;
        const vo::Image<uint8_t>& d = imgRT2.empty() ? b : imgRT2;
        if (a.rows != frameH || a.cols != frameW) throw vo::Error(VO_ERR_BAD_ARG, "image size differs from frameW x frameH");
        check(vo_set_images(ctx, 1, a.ptr(), b.ptr(), c.ptr(), d.ptr(), (size_t)frameW, (size_t)frameW * frameH));
    }

public:
    /* we need to hold 4 images at a time; 2x at time t and 2x at time (t+1)  (Include/VOClass.h:65-66) */
    vo::Image<uint8_t> imgLT1, imgRT1, imgLT2, imgRT2;
    /* integrated pose matrix [R|t] (4x4, row-major); the first pose is identity  (Include/VOClass.h:95) */
    std::array<double, 16> poseRt{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
    vo_frame_info lastInfo{};

    /* the reference hard-codes the KITTI frame size in its constructor (Source/VOClass.cpp:13-14) */
    explicit VOClass(int width = 1241, int height = 376, int device = 0, const vo_params* params = nullptr)
        : frameW(width), frameH(height)
    {
        vo_params p;
        if (params) p = *params; else vo_default_params(&p, width, height);
        depthThreshold = p.depth_thresh;
        int rc = vo_create(&p, device, &ctx);
        if (rc != VO_OK) throw vo::Error(rc, vo_create_error());
    }
    ~VOClass() { vo_destroy(ctx); }
    VOClass(const VOClass&) = delete;
    VOClass& operator=(const VOClass&) = delete;

    /* what getProjectionMatrices (Source/VOClass.cpp:102-138) leaves in projectionCL / projectionCR */
    bool setProjectionMatrices(const double P0[12], const double P1[12])
    {
        for (int i = 0; i < 12; i++) { projectionCL[i] = P0[i]; projectionCR[i] = P1[i]; }
        check(vo_set_calib(ctx, P0, P1));
        return true;
    }

    /* compute disparity  (Source/VOClass.cpp:231-305) */
    vo::Image<float> computeDisparity(vo::Image<uint8_t> leftImg, vo::Image<uint8_t> rightImg)
    {
        syncImages(&leftImg, &rightImg);
        check(vo_compute_disparity(ctx));
        vo::Image<float> out(frameH, frameW);
        check(vo_get_disparity32(ctx, 0, out.ptr()));
        return out;
    }

    /* disparity to depth map  (Source/VOClass.cpp:312-366); patches disparityMap in place like the reference */
    vo::Image<double> computeDepthMap(vo::Image<float> disparityMap)
    {
        if (disparityMap.rows != frameH || disparityMap.cols != frameW) throw vo::Error(VO_ERR_BAD_ARG, "disparity map size differs from frameW x frameH");
        check(vo_set_disparity32(ctx, 0, disparityMap.ptr())); /* the map the caller hands over is the one converted */
        check(vo_compute_depth_map(ctx));
        vo::Image<double> depth(frameH, frameW);
        check(vo_get_depth(ctx, 0, depth.ptr()));
        for (float& d : *disparityMap.data) if (d == 0.0 || d == -1.0) d = 0.1;
        return depth;
    }

    /* point cloud of a depth map as the reference writes it (VOClass::writeToPLY(depthMap, colors, depthThresh, numVertices),
     * Source/VOClassUtils.cpp:111-157, called from computeDepthMap, Source/VOClass.cpp:330-360): "row col -z g g g" for every
     * pixel whose depth is not above depthThresh, colours from the image handed over.  Works on its arguments like the
     * reference, so it can be called at any point of the loop; the vertices are compacted on the GPU in the writer's order.
     * numVertices is the header's count as the caller computed it (the reference derives it from a histogram), negative: the
     * number of vertices written.  depthThresh must be the context's depth_thresh (3000 in the reference's only call). */
    int writeToPLY(vo::Image<double> depthMap, vo::Image<uint8_t> colors, int depthThresh, int numVertices, const std::string& plyFilePath)
    {
        if (depthMap.rows != frameH || depthMap.cols != frameW) throw vo::Error(VO_ERR_BAD_ARG, "depth map size differs from frameW x frameH");
        if (depthThresh != depthThreshold) throw vo::Error(VO_ERR_BAD_ARG, "depthThresh differs from the context's depth_thresh");
        syncImages(&colors, nullptr);
        check(vo_set_depth(ctx, 0, depthMap.ptr()));
        int n = 0;
        check(vo_get_depth_cloud(ctx, 0, nullptr, 0, &n));
        std::vector<vo_cloud_point> pts((size_t)n);
        if (n) check(vo_get_depth_cloud(ctx, 0, pts.data(), n, &n));
        std::ofstream plyFile(plyFilePath);
        if (!plyFile.is_open()) throw vo::Error(VO_ERR_BAD_ARG, "Unable to open .ply file");
        plyFile << "ply" << std::endl << "format ascii 1.0" << std::endl << "element vertex " << (numVertices < 0 ? n : numVertices) << std::endl;
        plyFile << "property float x" << std::endl << "property float y" << std::endl << "property float z" << std::endl;
        plyFile << "property uchar red" << std::endl << "property uchar green" << std::endl << "property uchar blue" << std::endl;
        plyFile << "end_header" << std::endl;
        for (const vo_cloud_point& v : pts)
            plyFile << v.row << " " << v.col << " " << v.neg_z << " " << (int)v.gray << " " << (int)v.gray << " " << (int)v.gray << "\n";
        return n;
    }

    /* feature detection  (Source/VOClassFeatures.cpp:7-39) */
    std::vector<vo_point2f> getFeaturesFAST(vo::Image<uint8_t> img)
    {
        syncImages(&img, nullptr);
        check(vo_detect_fast(ctx));
        int n = 0;
        check(vo_get_features(ctx, 0, nullptr, 0, &n));
        std::vector<vo_point2f> pts((size_t)n);
        if (n) check(vo_get_features(ctx, 0, pts.data(), n, &n));
        return pts;
    }

    /* feature matching  (Source/VOClassFeatures.cpp:54-202): featurePointsLT1 is replaced by the re-tracked
     * LT1 points of the features that close the circle; returns the matching LT2 points */
    std::vector<vo_point2f> matchFeatureKLT(std::vector<vo_point2f>& featurePointsLT1)
    {
        syncImages();
        check(vo_set_features(ctx, 0, featurePointsLT1.data(), (int)featurePointsLT1.size()));
        check(vo_match_klt(ctx));
        int m = 0;
        check(vo_get_matches(ctx, 0, nullptr, nullptr, 0, &m));
        std::vector<vo_point2f> t1((size_t)m), t2((size_t)m);
        if (m) check(vo_get_matches(ctx, 0, t1.data(), t2.data(), m, &m));
        featurePointsLT1 = t1;
        return t2;
    }

    /* estimate motion  (Source/VOClass.cpp:370-623); depths are looked up in the map the caller hands over */
    std::array<double, 3> estimateMotion(std::vector<vo_point2f> featurePointsT1, std::vector<vo_point2f> featurePointsT2,
                                         vo::Image<double> depthMap)
    {
        if (featurePointsT1.size() != featurePointsT2.size()) throw vo::Error(VO_ERR_BAD_ARG, "matched arrays differ in size");
        if (depthMap.rows != frameH || depthMap.cols != frameW) throw vo::Error(VO_ERR_BAD_ARG, "depth map size differs from frameW x frameH");
        check(vo_set_depth(ctx, 0, depthMap.ptr()));
        check(vo_set_matches(ctx, 0, featurePointsT1.data(), featurePointsT2.data(), (int)featurePointsT1.size()));
        check(vo_triangulate(ctx));
        double Rt[16];
        check(vo_estimate_motion(ctx, Rt, &lastInfo));
        vo_integrate_pose(poseRt.data(), Rt);
        return {poseRt[3], poseRt[7], poseRt[11]};
    }

    /* compute error  (Source/VOClass.cpp:627-649): RMSE of the translations against the ground truth set with
     * setGroundTruth (the reference fills it in getGroundTruthPath) */
    void setGroundTruth(const std::vector<std::array<double, 3>>& gt) { groundTruth = gt; }
    float computeErrorInPoseEstimation(const std::vector<std::array<double, 3>>& estimatedTrajectory) const
    {
        float error = 0;
        for (size_t i = 0; i < estimatedTrajectory.size(); i++)
            error += std::pow(groundTruth[i][0] - estimatedTrajectory[i][0], 2) + std::pow(groundTruth[i][1] - estimatedTrajectory[i][1], 2) +
                     std::pow(groundTruth[i][2] - estimatedTrajectory[i][2], 2);
        return std::sqrt(error / estimatedTrajectory.size());
    }

    vo_ctx* context() { return ctx; }
};

#endif
