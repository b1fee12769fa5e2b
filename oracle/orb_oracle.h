/*
 * orb_oracle.h -- CPU restatement of the ORB extract + match hot path (plain C11).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call this.  PARITY UNPINNED: see orb_constants.h for why and for
 * every constant that is recalled rather than read from a source present here.
 *
 * What it restates (reference entry point
 * slam_backends/orb_slam_3/orbslam3_mono_networked.cc:594 TrackMonocular):
 *   K0 orc_gray            cv::cvtColor RGB2GRAY/BGR2GRAY 8U fixed point
 *   K1 orc_pyramid         ORBextractor::ComputePyramid (cv::resize INTER_LINEAR 8U)
 *   K2 orc_fast_cell       cv::FAST(TYPE_9_16, nonmax) on one cell sub-image
 *   K3 orc_candidates      ComputeKeyPointsOctTree cell grid, 20 -> 7 fallback
 *   K4 orc_distribute      ORBextractor::DistributeOctTree (+ libstdc++ std::sort order)
 *   K5 orc_ic_angle        IC_Angle + cv::fastAtan2
 *   K6 orc_blur, orc_descriptor   GaussianBlur 7x7 s2 fixed point; computeOrbDescriptor
 *   -- orc_extract         ORBextractor::operator() incl. lapping-area output order
 *   K7 orc_match           all-pairs Hamming, best / second best, ratio + threshold
 *                          (this rule is the build's own)
 */
#ifndef ORC_ORACLE_H
#define ORC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_LEVELS 16

typedef struct {
    int n_features;
    float scale_factor;
    int n_levels;
    int ini_th_fast;
    int min_th_fast;
    int lapping_x0; /* ORB-SLAM3 vLappingArea; mono default {0, 1000} */
    int lapping_x1;
    /* rBRIEF tap coordinates: 0 = x*b + y*a with every operation rounded (the expression as written); 1 = the
     * first product fused, fmaf(x, b, y*a) / fmaf(x, a, -(y*b)): GCC's contraction of the same expression under
     * upstream's -O3 -march=native (CMakeLists.txt:10-13).  Which one the reference binary runs is unpinned. */
    int steer_fma;
} orc_params;

typedef struct {
    float x, y;     /* level-0 pixel coordinates (level coords * scale, float) */
    float size;     /* (int)(31 * scale[level]) */
    float angle;    /* degrees, fastAtan2 */
    float response; /* FAST score */
    int octave;
} orc_keypoint;

/* candidate / per-level keypoint in LEVEL coordinates (integers) */
typedef struct {
    int x, y;
    int response;
} orc_point;

typedef struct {
    int n_levels;
    int w[ORC_MAX_LEVELS], h[ORC_MAX_LEVELS];
    float scale[ORC_MAX_LEVELS];     /* mvScaleFactor */
    float inv_scale[ORC_MAX_LEVELS]; /* mvInvScaleFactor */
    int quota[ORC_MAX_LEVELS];       /* mnFeaturesPerLevel */
    int umax[16];
} orc_geometry;

void orc_default_params(orc_params *p);

/* level sizes, scale factors, per-level quotas, umax; returns 0 or <0 if a level is too
 * small for the cell grid */
int orc_geometry_init(orc_geometry *g, const orc_params *p, int width, int height);

/* K0: 3/4-channel interleaved -> gray.  rgb != 0: byte 0 is treated as R (Camera.RGB: 1),
 * else byte 0 is B. */
void orc_gray(const uint8_t *src, int w, int h, int channels, int stride, int rgb,
              uint8_t *dst /* w*h tight */);

/* K1: one bilinear down-scale step, tight pitches */
void orc_resize_linear(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);

/* K1: full pyramid; levels[l] must hold w[l]*h[l] bytes; level 0 is a copy of src */
void orc_pyramid(const uint8_t *src, int stride, const orc_geometry *g, uint8_t **levels);

/* K2: FAST-9-16 score at 'threshold' for every pixel of a tight w*h image: out = score
 * (>= threshold) if the pixel is a corner, else 0; a 3-pixel frame is 0.  Whole-image
 * form used to check the device response map. */
void orc_fast_score_map(const uint8_t *img, int w, int h, int threshold, uint8_t *out);

/* K2: cv::FAST(cell, kps, threshold, true) on the sub-image [x0,x1) x [y0,y1) of a tight
 * w-pitch image; appends keypoints in cell-local coordinates, row-major; returns count */
int orc_fast_cell(const uint8_t *img, int pitch, int x0, int y0, int x1, int y1,
                  int threshold, orc_point *out, int max_out);

/* K3: candidates of one level in upstream order, coordinates relative to
 * (minBorderX, minBorderY) = (16, 16); returns count or <0 on overflow */
int orc_candidates(const uint8_t *img, int w, int h, int ini_th, int min_th,
                   orc_point *out, int max_out);

/* K4: quadtree distribution; in/out coordinates as orc_candidates; returns count kept */
int orc_distribute(const orc_point *cand, int n_cand, int min_x, int max_x, int min_y,
                   int max_y, int n_wanted, orc_point *out, int max_out);

/* K5 */
float orc_fast_atan2(float y, float x);
float orc_ic_angle(const uint8_t *img, int pitch, int x, int y, const int *umax);

/* K6 */
void orc_blur(const uint8_t *src, int w, int h, uint8_t *dst);
void orc_descriptor(const uint8_t *blurred, int pitch, int x, int y, float angle_deg,
                    uint8_t desc[32]);
/* steer_fma: see orc_params */
void orc_descriptor_ex(const uint8_t *blurred, int pitch, int x, int y, float angle_deg, int steer_fma,
                       uint8_t desc[32]);

/* full extractor; returns number of keypoints (<= max_kp) or <0 on error.
 * level_counts (optional, n_levels ints) receives per-level keypoint counts. */
int orc_extract(const uint8_t *gray, int w, int h, int stride, const orc_params *p,
                orc_keypoint *kps, uint8_t *desc, int max_kp, int *level_counts);

/* K7.  exclude_self: skip j == i (self-match).  idx[i] = accepted best index or -1;
 * d1/d2 = best / second-best distance (0xFFFF if none). */
void orc_match(const uint8_t *q, int nq, const uint8_t *t, int nt, int th, int ratio_num,
               int ratio_den, int exclude_self, int32_t *idx, uint16_t *d1, uint16_t *d2);

/* libstdc++ std::sort restated for the (size, UL.x) node comparator; exposed so
 * tests can pin it against the real std::sort of this container's g++. */
typedef struct {
    int size;
    int ulx;
    int id;
} orc_sort_item;
void orc_std_sort(orc_sort_item *a, int n);
extern int orc_std_sort_heap_calls;

#ifdef __cplusplus
}
#endif
#endif
