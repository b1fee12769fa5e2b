// Host-side cost of the pose step (sst_tracker::step, csrc/ss_track.cpp) on a synthetic track: n points in front of a camera that
// moves sideways; every frame sees all points (identity matches, Hamming distance 10), octaves 0..7.  Prints ms per frame in
// tracking state OK.  Build: g++ -O2 -std=c++17 -pthread [-DSST_PHASE_TIMING] -I include -o /tmp/track_timing profiles/tools/track_timing.cpp send-slam_amd/csrc/ss_track.cpp
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#include "../../send-slam_amd/csrc/ss_track.h"

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2000, frames = argc > 2 ? atoi(argv[2]) : 60;
    std::mt19937 rng(7);
    std::uniform_real_distribution<double> ux(-4, 4), uy(-2.5, 2.5), uz(4, 12);
    std::vector<double> X(3 * n);
    std::vector<int32_t> oct(n), idx(n);
    std::vector<uint16_t> d1(n, 10);
    for (int i = 0; i < n; i++) {
        X[3 * i] = ux(rng); X[3 * i + 1] = uy(rng); X[3 * i + 2] = uz(rng);
        oct[i] = i % 8;
        idx[i] = i;
    }
    sst_tracker tr;
    tr.cam = sst_camera{1000, 1000, 640, 360, 0.01, -0.002, 0.0005, -0.0003};
    tr.scale_factor = 1.2;
    std::vector<float> xy(2 * n);
    std::normal_distribution<double> noise(0, 0.3);
    double total = 0;
    int ok = 0;
    for (int f = 0; f < frames; f++) {
        const double tx = 0.02 * f;
        for (int i = 0; i < n; i++) {
            const double x = X[3 * i] - tx, y = X[3 * i + 1], z = X[3 * i + 2];
            xy[2 * i] = (float)(1000 * x / z + 640 + noise(rng));
            xy[2 * i + 1] = (float)(1000 * y / z + 360 + noise(rng));
        }
        sst_pose_out o;
        const auto t0 = std::chrono::steady_clock::now();
        tr.step(n, xy.data(), oct.data(), idx.data(), d1.data(), o);
        const double ms = std::chrono::duration_cast<std::chrono::duration<double, std::milli>>(std::chrono::steady_clock::now() - t0).count();
        if (o.state == 2 && f > 2) { total += ms; ok++; }
        if (f < 4 || f % 20 == 0 || f == frames - 1) printf("frame %d state %d matches %d inliers %d map %d  %.3f ms\n", f, o.state, o.n_matches, o.n_inliers, o.n_map_points, ms);
    }
    printf("n %d: %.3f ms per frame in state OK (%d frames)\n", n, ok ? total / ok : 0.0, ok);
#ifdef SST_PHASE_TIMING
    extern double sst_phase_ms[8];
    static const char *names[6] = {"undistort + arrays", "unique matches", "projection gate", "pose-only", "new points", "history + hand-over"};
    for (int k = 0; k < 6; k++) printf("  %-20s %.3f ms per frame (all %d frames)\n", names[k], sst_phase_ms[k] / frames, frames);
#endif
    return 0;
}
