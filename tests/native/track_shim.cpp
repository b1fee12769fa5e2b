/* Test-only C wrappers around send-slam_amd/csrc/ss_track.{h,cpp} (host geometry of ss_track), so
 * the CPU suite can compare every block with oracle/vo_oracle.py without a GPU.  Built by
 * tests/test_track.py with g++ against the product source; not part of libsendslam_orb.so's ABI. */
#include <cstring>

#include "../../send-slam_amd/csrc/ss_track.h"

static sst_camera cam_of(const double *k) { return sst_camera{k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7]}; }

extern "C" {

void shim_undistort(const double *cam, int n, const float *xy, double *out) { sst_undistort(cam_of(cam), n, xy, out); }

int shim_last_model = 0;
int shim_two_view_model(void) { return shim_last_model; }

int shim_two_view(const double *cam, int n, const double *x1, const double *x2, double *R, double *t, uint8_t *tri, double *p3d)
{
    std::vector<uint8_t> tr;
    std::vector<double> p;
    const int r = sst_two_view(cam_of(cam), n, x1, x2, R, t, tr, p, &shim_last_model);
    if (n > 0) {
        memcpy(tri, tr.data(), (size_t)n);
        memcpy(p3d, p.data(), sizeof(double) * 3 * (size_t)n);
    }
    return r;
}

int shim_two_view_ba(const double *cam, int n, const double *o1, const double *o2, const double *w1, const double *w2, double *R,
                     double *t, double *X, int iterations)
{
    return sst_two_view_ba(cam_of(cam), n, o1, o2, w1, w2, R, t, X, iterations);
}

int shim_pose_only(int n, const double *P, const double *obs, const double *w, const double *cam, double *R, double *t, uint8_t *inl)
{
    std::vector<uint8_t> in;
    const int r = sst_pose_only(n, P, obs, w, cam_of(cam), R, t, in);
    if (n > 0) memcpy(inl, in.data(), (size_t)n);
    return r;
}

int shim_triangulate(const double *cam, const double *x1, const double *x2, const double *R1, const double *t1, const double *R2,
                     const double *t2, double s1, double s2, double *X)
{
    return sst_triangulate(cam_of(cam), x1, x2, R1, t1, R2, t2, s1, s2, X) ? 1 : 0;
}

void shim_pose_to_twc(const double *R, const double *t, double *pos, double *q) { sst_pose_to_twc(R, t, pos, q); }

void *shim_tracker_new(const double *cam, double scale_factor)
{
    sst_tracker *tr = new sst_tracker();
    tr->cam = cam_of(cam);
    tr->scale_factor = scale_factor;
    return tr;
}
void shim_tracker_free(void *p) { delete (sst_tracker *)p; }
void shim_tracker_set_hist_cap(void *p, int cap) { ((sst_tracker *)p)->pose_hist_cap = cap; }
int shim_tracker_hist_len(void *p) { return (int)(((sst_tracker *)p)->pose_hist.size() / 12); }
int shim_tracker_want(void *p) { return ((sst_tracker *)p)->want_match(); }
int shim_tracker_n_train(void *p) { return ((sst_tracker *)p)->n_train(); }
int shim_tracker_step(void *p, int n, const float *xy, const int32_t *oct, const int32_t *idx, const uint16_t *d1, double *pose7, int32_t *counts4)
{
    sst_pose_out o;
    const int keep = ((sst_tracker *)p)->step(n, xy, oct, idx, d1, o);
    for (int k = 0; k < 3; k++) pose7[k] = o.pos[k];
    for (int k = 0; k < 4; k++) pose7[3 + k] = o.quat[k];
    counts4[0] = o.state; counts4[1] = o.n_matches; counts4[2] = o.n_inliers; counts4[3] = o.n_map_points;
    return keep;
}
}
