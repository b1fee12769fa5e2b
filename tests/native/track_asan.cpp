/* ASan/UBSan driver for the host geometry of ss_track (send-slam_amd/csrc/ss_track.cpp): a synthetic
 * rigid scene, a sliding camera, noisy projections, some wrong matches, keypoints that come and go. */
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../send-slam_amd/csrc/ss_track.h"

static uint64_t rng_state = 12345;
static double urand()
{
    rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

int main()
{
    const int N = 500;
    std::vector<double> X((size_t)3 * N);
    for (int i = 0; i < N; i++) {
        X[3 * i] = -4 + 8 * urand();
        X[3 * i + 1] = -3 + 6 * urand();
        X[3 * i + 2] = 3 + 6 * urand();
    }
    sst_tracker tr;
    tr.cam = sst_camera{500, 500, 320, 240, -0.1, 0.02, 1e-3, -1e-3};
    tr.scale_factor = 1.2f;
    std::vector<int> prev_ids, ref_ids;
    int ok_frames = 0, lost = 0;
    for (int f = 0; f < 40; f++) {
        const double cx = -0.08 * f, ang = 0.002 * f;
        std::vector<float> xy;
        std::vector<int32_t> oct;
        std::vector<int> ids;
        for (int i = 0; i < N; i++) {
            if (urand() < 0.15) continue; /* not detected in this frame */
            const double x = X[3 * i] - cx, y = X[3 * i + 1], z = X[3 * i + 2];
            const double xr = std::cos(ang) * x + std::sin(ang) * z, zr = -std::sin(ang) * x + std::cos(ang) * z;
            if (zr <= 0.5) continue;
            const double u = 500 * xr / zr + 320 + (urand() - 0.5), v = 500 * y / zr + 240 + (urand() - 0.5);
            if (u < 0 || u >= 640 || v < 0 || v >= 480) continue;
            xy.push_back((float)u); xy.push_back((float)v);
            oct.push_back((int)(urand() * 8));
            ids.push_back(i);
        }
        if (f == 20) { xy.resize(2 * 40); oct.resize(40); ids.resize(40); } /* a nearly empty frame: tracking is lost */
        const int n = (int)ids.size(), want = tr.want_match();
        const std::vector<int> &train = want == SST_MATCH_REF ? ref_ids : prev_ids;
        std::vector<int32_t> idx((size_t)(n > 0 ? n : 1), -1);
        std::vector<uint16_t> d1((size_t)(n > 0 ? n : 1), 0xFFFF);
        if (want != SST_MATCH_NONE)
            for (int i = 0; i < n; i++) {
                for (size_t j = 0; j < train.size(); j++)
                    if (train[j] == ids[i]) { idx[i] = (int)j; d1[i] = (uint16_t)(urand() * 40); break; }
                if (urand() < 0.05 && !train.empty()) { idx[i] = (int)(urand() * train.size()); d1[i] = 45; } /* wrong match */
            }
        sst_pose_out o;
        const int keep = tr.step(n, xy.data(), oct.data(), idx.data(), d1.data(), o);
        if (keep == SST_KEEP_AS_REF) ref_ids = ids;
        if (keep == SST_KEEP_AS_PREV) prev_ids = ids;
        ok_frames += o.state == 2;
        lost += o.state == 4;
    }
    tr.reset();
    printf("ok_frames=%d lost=%d\n", ok_frames, lost);
    return ok_frames >= 25 && lost == 1 ? 0 : 1;
}
