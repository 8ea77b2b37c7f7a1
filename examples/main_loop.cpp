// The reference's main loop (Source/main.cpp:30-65) written against include/VOClass.h.
// Input: a raw file "W H N" header line followed by N stereo frames (left then right, W*H bytes each) and a
// calib file with the 12 numbers of P0 then the 12 numbers of P1.  Prints one "x y z" line per pose, like the
// reference's pose dump (Source/VOClass.cpp:613-615).
//
//   g++ -std=c++17 -Iinclude examples/main_loop.cpp -o main_loop -Lvisualodometry_amd -lvo_amd -Wl,-rpath,$PWD/visualodometry_amd
#include "VOClass.h"

#include <cstdio>
#include <fstream>
#include <iostream>

int main(int argc, char** argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s frames.raw calib.txt [num_disparities [first_frame.ply]]\n", argv[0]); return 2; }
    std::ifstream in(argv[1], std::ios::binary);
    int W = 0, H = 0, N = 0;
    in >> W >> H >> N;
    in.get();
    std::ifstream cal(argv[2]);
    double P0[12], P1[12];
    for (double& v : P0) cal >> v;
    for (double& v : P1) cal >> v;
    if (!in || !cal || N < 2) { std::fprintf(stderr, "bad input\n"); return 2; }
    try {
        vo_params p;
        vo_default_params(&p, W, H);
        if (argc > 3) p.num_disparities = std::atoi(argv[3]);
        VOClass VO(W, H, 0, &p);
        VO.setProjectionMatrices(P0, P1);
        auto read_frame = [&](vo::Image<uint8_t>& l, vo::Image<uint8_t>& r) {
            l = vo::Image<uint8_t>(H, W); r = vo::Image<uint8_t>(H, W);
            in.read((char*)l.ptr(), (std::streamsize)W * H);
            in.read((char*)r.ptr(), (std::streamsize)W * H);
        };
        std::vector<std::array<double, 3>> estimatedTrajectory;
        estimatedTrajectory.push_back({0, 0, 0});
        std::cout << 0 << " " << 0 << " " << 0 << std::endl;
        read_frame(VO.imgLT2, VO.imgRT2);
        for (int i = 0; i < N - 1; i++) {
            VO.imgLT1 = VO.imgLT2; VO.imgRT1 = VO.imgRT2;
            read_frame(VO.imgLT2, VO.imgRT2);
            vo::Image<float> disparityMapT1 = VO.computeDisparity(VO.imgLT1, VO.imgRT1);
            vo::Image<double> depthMapT1 = VO.computeDepthMap(disparityMapT1);
            std::vector<vo_point2f> featurePointsT1 = VO.getFeaturesFAST(VO.imgLT1);
            /* the reference writes the cloud of a frame from computeDepthMap when asked to; the writer works on its arguments,
             * so any point of the loop will do (here: after FAST, when the device holds no depth map of its own any more) */
            if (i == 0 && argc > 4) VO.writeToPLY(depthMapT1, VO.imgLT1, 3000, -1, argv[4]);
            std::vector<vo_point2f> featurePointsT2 = VO.matchFeatureKLT(featurePointsT1);
            auto t = VO.estimateMotion(featurePointsT1, featurePointsT2, depthMapT1);
            estimatedTrajectory.push_back(t);
            std::cout << t[0] << " " << t[1] << " " << t[2] << std::endl;
        }
    } catch (const vo::Error& e) {
        std::fprintf(stderr, "vo error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
