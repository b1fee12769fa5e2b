/*
 * ss_track.h -- pose from matches: the small dense geometry that
 * follows the GPU hot path.  Host code, double precision: <= 2000 points per frame.
 *
 * The reference obtains poses from ORB_SLAM3::System::TrackMonocular
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:594) and ships them with
 * SendPosePacket (:225-282) only in tracking state OK (:596).  ORB-SLAM3's full tracking / mapping
 * is out of scope; this is a monocular visual-odometry front-end made of the same building blocks
 * (restated from the published algorithms, parity unpinned like the rest):
 *   sst_undistort        Frame::UndistortKeyPoints = cv::undistortPoints (k1 k2 p1 p2, 5 iterations)
 *   sst_two_view         TwoViewReconstruction: homography (DLT) and fundamental matrix (normalised
 *                        8-point) on the same 200 RANSAC sets, chi2 scores, model choice by
 *                        SH / (SH + SF) > 0.45; ReconstructF (E = K^T F K, four hypotheses) or ReconstructH
 *                        (Faugeras-Lustman, eight hypotheses); linear triangulation, cheirality /
 *                        reprojection / parallax checks
 *   sst_pose_only        Optimizer::PoseOptimization (g2o EdgeSE3ProjectXYZOnlyPose): 4 x 10 damped
 *                        Gauss-Newton steps, Huber sqrt(5.991), outliers at chi2 > 5.991
 *   sst_triangulate      linear two-view triangulation (DLT)
 *   sst_two_view_ba      two-view bundle adjustment of the initial map (GlobalBundleAdjustemnt, 20 its)
 * oracle/vo_oracle.py is the numpy re-derivation the tests compare against (tolerance 1e-4 rel).
 */
#ifndef SS_TRACK_H
#define SS_TRACK_H

#include <cstdint>
#include <vector>

#define SST_RH_THRESHOLD 0.45 /* homography if SH / (SH + SF) exceeds it: see sst_two_view */

struct sst_camera {
    double fx, fy, cx, cy, k1, k2, p1, p2;
};

/* pixels (x, y) -> undistorted pixels; no-op when every distortion coefficient is 0 */
void sst_undistort(const sst_camera &c, int n, const float *xy_in /* stride 2 */, double *xy_out);

/* n matches: x1 (reference), x2 (current), undistorted pixels.  On success fills R, t (current from
 * reference, |t| = 1), triangulated[n] flags and pts3d (reference frame) and returns the number of
 * triangulated points; 0 if the pair does not reconstruct (too little parallax, ambiguous, < 50). */
int sst_two_view(const sst_camera &c, int n, const double *x1, const double *x2, double R[9], double t[3],
                 std::vector<uint8_t> &triangulated, std::vector<double> &pts3d, int *model = nullptr /* 1 F, 2 H */);

/* Two-view bundle adjustment (what Optimizer::GlobalBundleAdjustemnt does to the initial map of two
 * keyframes, CreateInitialMapMonocular): camera 1 fixed at the origin, camera 2 (R, t) and the n
 * points X free, Huber sqrt(5.991), Levenberg-Marquardt on the Schur-reduced system, `iterations`
 * trial steps.  Returns the number of accepted steps. */
int sst_two_view_ba(const sst_camera &c, int n, const double *obs1, const double *obs2, const double *w1, const double *w2,
                    double R[9], double t[3], double *X, int iterations);

/* pose-only optimisation from the initial Tcw in (R, t); returns inliers (or < 0) */
int sst_pose_only(int n, const double *pts3d, const double *obs, const double *inv_sigma2, const sst_camera &c,
                  double R[9], double t[3], std::vector<uint8_t> &inlier);

/* triangulate one match seen from Tcw1 = (R1,t1) and Tcw2 = (R2,t2); false if it fails the
 * depth / reprojection (chi2 5.991 * sigma2) / parallax (cos < 0.9998) checks */
bool sst_triangulate(const sst_camera &c, const double x1[2], const double x2[2], const double R1[9], const double t1[3],
                     const double R2[9], const double t2[3], double sigma2_1, double sigma2_2, double X[3]);

/* Twc = Tcw^-1 as position + unit quaternion (x, y, z, w), Eigen's matrix->quaternion branch rule */
void sst_pose_to_twc(const double R[9], const double t[3], double pos[3], double quat_xyzw[4]);

/* ---- the frame-to-frame state machine that strings the blocks above together ----
 * States are ORB_SLAM3::Tracking::eTrackingState values (the shim ships a pose only in state 2,
 * orbslam3_mono_networked.cc:596): 0 NO_IMAGES_YET, 1 NOT_INITIALIZED, 2 OK, 4 LOST.
 * Monocular initialisation follows Tracking::MonocularInitialization: a frame with > 100 keypoints
 * becomes the reference; a later frame with > 100 keypoints and >= 100 matches to it (inside the
 * 100-px window of SearchForInitialization; all octaves, where ORB-SLAM3 uses octave 0 of a 5x
 * extractor) is handed to sst_two_view, then sst_two_view_ba; on success the reference frame is the world origin, the map is scaled so the median
 * depth is 1 (CreateInitialMapMonocular), and every later frame is tracked against its predecessor
 * (constant-velocity prediction, SearchByProjection's th * scale^octave gate around the projected
 * point, pose-only optimisation on the carried-over points, >= 30 inliers as in TrackLocalMap), new
 * matches being triangulated between the first observation of their track and the current one.  No keyframes, local mapping, loop closing or
 * relocalisation: a lost tracker re-initialises from the next frame.  */
struct sst_frame {
    int n = 0;
    std::vector<double> und;     /* 2n undistorted pixels */
    std::vector<int32_t> octave; /* n */
    std::vector<uint8_t> has3d;  /* n */
    std::vector<double> p3d;     /* 3n world points */
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0}; /* Tcw */
    /* keypoints without a 3-D point that continue a track: its first observation */
    std::vector<int32_t> anchor;       /* n: index into sst_tracker::pose_hist, -1 = none */
    std::vector<double> anchor_xy;     /* 2n */
    std::vector<double> anchor_sigma2; /* n */
};

struct sst_pose_out {
    int state = 0, n_matches = 0, n_inliers = 0, n_map_points = 0;
    double pos[3] = {0, 0, 0}, quat[4] = {0, 0, 0, 1};
};

enum { SST_MATCH_NONE = 0, SST_MATCH_REF = 1, SST_MATCH_PREV = 2 };
/* sst_pose_only: a Gauss-Newton round ends when no component of the se(3) step exceeds this */
#define SST_POSE_STEP_EPS 1e-10
/* sst_two_view: threads per model for the 200 RANSAC hypotheses (results do not depend on it) */
#ifndef SST_RANSAC_THREADS
#define SST_RANSAC_THREADS 4
#endif
enum { SST_KEEP_NONE = 0, SST_KEEP_AS_REF = 1, SST_KEEP_AS_PREV = 2 };

struct sst_tracker {
    sst_camera cam{};
    double scale_factor = 1.2;
    int state = 0;
    bool have_ref = false, have_vel = false;
    double vel_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, vel_t[3] = {0, 0, 0}; /* Tcw(k) * Twc(k-1) */
    sst_frame ref, prev;
    std::vector<double> pose_hist; /* 12 doubles (R, t) per tracked frame an anchored track still refers to */
    int pose_hist_cap = 256;       /* poses kept before the unreferenced ones are dropped (results do not depend on it) */

    void reset();
    /* which stored descriptor set the next frame's descriptors are to be matched against */
    int want_match() const;
    int n_train() const;
    /* one frame: xy = 2n distorted pixels, match_idx[i] = row of the wanted set or -1, d1 = its
     * distance.  Returns SST_KEEP_*: whether the caller stores this frame's descriptors. */
    int step(int n, const float *xy, const int32_t *octave, const int32_t *match_idx, const uint16_t *d1, sst_pose_out &out);
};

#endif
