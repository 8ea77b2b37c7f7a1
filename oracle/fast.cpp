/* CPU ORACLE (test infrastructure only) — FAST-9/16 + non-max suppression.
 * Restates cv::FAST(img, keypoints, 20, true) as called at
 * reference Source/VOClassFeatures.cpp:27-32 (OpenCV 4.5.4 features2d: fast.cpp FAST_t<16>,
 * fast_score.cpp cornerScore<16>).  Parity unpinned (see vo_oracle.h).
 */
#include "vo_oracle.h"
#include <vector>
#include <algorithm>
#include <cstring>

namespace {

/* Bresenham ring of radius 3, order as in OpenCV's offsets16 table. */
const int kRing[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* cornerScore<16>: the early "continue"s of the original are pure shortcuts. */
int corner_score(const int d[25], int threshold)
{
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

} // namespace

extern "C" int orc_fast(const uint8_t* img, int width, int height, int stride, int threshold, int nms,
                        orc_point2f* out, uint8_t* response_out, int cap, uint8_t* score_map_out)
{
    threshold = std::min(std::max(threshold, 0), 255);
    const int K = 8, N = 25;
    std::vector<uint8_t> score((size_t)width * height, 0);
    std::vector<uint8_t> is_corner((size_t)width * height, 0);

    /* detection domain: rows 3..H-4, cols 3..W-4 */
    for (int y = 3; y < height - 3; y++) {
        for (int x = 3; x < width - 3; x++) {
            const uint8_t* p = img + (size_t)y * stride + x;
            int v = p[0];
            int ring[25];
            for (int k = 0; k < 25; k++) {
                const int* o = kRing[k & 15];
                ring[k] = p[o[0] + o[1] * stride];
            }
            bool corner = false;
            /* dark arc: > K contiguous ring pixels < v - t */
            {
                int vt = v - threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    if (ring[k] < vt) { if (++count > K) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (!corner) {
                int vt = v + threshold, count = 0;
                for (int k = 0; k < N; k++) {
                    if (ring[k] > vt) { if (++count > K) { corner = true; break; } }
                    else count = 0;
                }
            }
            if (corner) {
                is_corner[(size_t)y * width + x] = 1;
                int d[25];
                for (int k = 0; k < 25; k++) d[k] = v - ring[k];
                score[(size_t)y * width + x] = (uint8_t)corner_score(d, threshold);
            }
        }
    }
    if (score_map_out) memcpy(score_map_out, score.data(), score.size());

    /* NMS: strict > over the 8 neighbours; non-corners score 0.  Output row-major. */
    int n = 0;
    for (int y = 3; y < height - 3; y++) {
        for (int x = 3; x < width - 3; x++) {
            size_t i = (size_t)y * width + x;
            if (!is_corner[i]) continue;
            int s = score[i];
            bool keep = true;
            if (nms) {
                keep = s > score[i - 1] && s > score[i + 1] &&
                       s > score[i - width - 1] && s > score[i - width] && s > score[i - width + 1] &&
                       s > score[i + width - 1] && s > score[i + width] && s > score[i + width + 1];
            }
            if (keep) {
                if (n < cap) {
                    if (out) { out[n].x = (float)x; out[n].y = (float)y; }
                    if (response_out) response_out[n] = (uint8_t)s;
                }
                n++;
            }
        }
    }
    return n;
}
