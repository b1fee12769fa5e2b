/*
 * sendslam_orb.h -- C ABI of libsendslam_orb.so: the MI355X-native ORB tracking front-end
 * that replaces SEND-SLAM's dockerised ORB-SLAM3 CPU backend for ONE path: per-frame ORB
 * extraction (pyramid, FAST-9 + NMS, quadtree distribution, orientation, rBRIEF) and
 * brute-force Hamming matching with a ratio test.
 *
 * Plain C linkage, plain pointers and sizes, int status (0 = ok, < 0 = ss_status), no
 * exception crosses.  What each entry point replaces in the reference
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc unless noted):
 *
 *   ss_create / ss_destroy     make_unique<ORB_SLAM3::System>(voc, yaml, MONOCULAR, false) :511,
 *                              Shutdown :654; ORB parameters = the YAML literals :193-206
 *   ss_set_calibration         the "calibration" message branch :477-519 with the 16 scalars of
 *                              CameraCalibration :59-77 / ParseCameraCalibration :109-137
 *   ss_extract                 the "frame" branch :521-627 up to and inside TrackMonocular :594
 *                              (ORBextractor::operator()), pixels as cv::imdecode leaves them :546
 *   ss_extract_batch_device    same, for a batch of frames already resident in HBM (cameras /
 *                              frame batches shard one per GPU; no counterpart in the reference,
 *                              which handles one camera on one thread :594)
 *   ss_pipe_*                  the same frame branch :521-627 for a STREAM of host frames: pinned ring,
 *                              H2D copy / kernels / D2H copy of different batches overlapped, results in
 *                              host memory -- what the copy at :325 + imdecode :546 + TrackMonocular :594
 *                              do one frame at a time; host half slam_handler.ex:59-88
 *   ss_track_features[_matched] the pose half of TrackMonocular :594 for a frame whose features a pipe
 *                              extracted (front door read-ahead of queued frames)
 *   ss_match_partial_device /  the local and the cross-shard half of a query against a database
 *   ss_match_fold_device       partitioned over GPUs
 *   ss_xchg_*                  the exchange step of configs 4 / 5 (all-gather, broadcast) as direct peer writes; no
 *                              counterpart in the reference (one TCP link :387-388)
 *   ss_match*                  ORBmatcher::DescriptorDistance + best/second-best search inside
 *                              TrackMonocular :594 (all-pairs rule: SURVEY.md Appendix A.6)
 *   ss_track                   TrackMonocular :594 -> Twc, tracking state :596 (bounded monocular
 *                              front-end; the pose SendPosePacket :225-282 ships)
 *   ss_stats                   vTimesTrack median/mean summary :615-616, :656-664
 *   ss_last_error              the cerr diagnostics of the shim (:457-469, :523-551)
 *
 * Threading: a context is single-threaded (one HIP stream, one camera or one batch in
 * flight); distinct contexts are independent and may live on different devices.  A call
 * that fails leaves the context usable (the shim's log-and-skip policy, :523-551).
 * Every entry point takes longer than 1 ms on first use: NIF callers flag them dirty
 * (INTEGRATION.md).
 *
 * The library has NO CPU fallback: without a usable HIP device ss_create fails with
 * SS_ERR_NO_DEVICE.
 */
#ifndef SENDSLAM_ORB_H
#define SENDSLAM_ORB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SS_ABI_VERSION 5
#define SS_MAX_LEVELS 16
#define SS_DESC_BYTES 32

typedef enum {
    SS_OK = 0,
    SS_ERR_INVALID_ARG = -1,
    SS_ERR_NO_DEVICE = -2,
    SS_ERR_HIP = -3,
    SS_ERR_TOO_SMALL = -4,      /* image too small for the cell grid at some level */
    SS_ERR_OVERFLOW = -5,       /* an internal capacity was exceeded; nothing was truncated */
    SS_ERR_NOT_CALIBRATED = -6, /* frame before calibration (shim :523-527) */
    SS_ERR_BAD_FRAME = -7,
    SS_ERR_NO_MEMORY = -8,
    SS_ERR_STATE = -9,
    SS_ERR_BUSY = -10 /* ss_pipe_acquire: every slot of the ring is in flight or not yet released */
} ss_status;

typedef struct ss_ctx ss_ctx;

/* ORB parameters.  Defaults are the reference's YAML literals (:193-206): 1250 features,
 * scale 1.2, 8 levels, FAST 20 / 7; lapping area {0, 1000} is ORB-SLAM3's monocular
 * Frame constructor (ExtractORB(0, im, 0, 1000)). */
typedef struct {
    int32_t n_features;
    float scale_factor;
    int32_t n_levels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t lapping_x0;
    int32_t lapping_x1;
    int32_t max_batch; /* frames per ss_extract_batch_device call; >= 1 */
    /* How the rotated rBRIEF tap coordinates cvRound(x*b + y*a), cvRound(x*a - y*b) of computeOrbDescriptor are
     * evaluated.  0 (default): as written, every product and the sum rounded.  1: the first product fused into the
     * sum, fma(x, b, y*a) / fma(x, a, -(y*b)) -- what GCC's FMA contraction makes of the expression when upstream is
     * built -O3 -march=native (slam_backends/orb_slam_3/CMakeLists.txt:10-13) on an FMA-capable host.  Which one the
     * reference binary runs is unpinned (tests/golden/ref_dump/README.md); a few descriptor bits per frame depend on it. */
    int32_t steer_fma;
} ss_orb_params;

/* The 16 calibration scalars of the wire protocol (shim :59-77; produced by
 * send_slam/lib/send_slam/slam_handler.ex:209-226). */
typedef struct {
    char type[16]; /* "PinHole" */
    double fx, fy, cx, cy;
    double k1, k2, p1, p2;
    int32_t width, height;
    double fps;
    int32_t rgb; /* Camera.RGB: 1 = byte 0 of a 3-channel pixel is treated as R */
    double th_depth, baseline, depth_map_factor;
} ss_camera;

/* cv::KeyPoint fields the extractor fills */
typedef struct {
    float x, y;     /* level-0 pixel coordinates */
    float size;     /* (int)(31 * scale^octave) */
    float angle;    /* degrees [0, 360) */
    float response; /* FAST score */
    int32_t octave;
} ss_keypoint;

/* Result of ss_extract: host arrays owned by the context, valid until the next call on it */
typedef struct {
    int32_t n_keypoints;
    int32_t camera_id;
    double timestamp;
    const ss_keypoint *keypoints; /* n_keypoints */
    const uint8_t *descriptors;   /* n_keypoints x 32, row-major (CV_8U N x 32) */
    int32_t level_counts[SS_MAX_LEVELS];
} ss_frame_result;

/* Device-resident results of the last ss_extract_batch_device call */
typedef struct {
    int32_t n_frames;
    int32_t kp_capacity;           /* rows per frame in the two arrays below */
    const ss_keypoint *keypoints;  /* device: [n_frames][kp_capacity] */
    const uint8_t *descriptors;    /* device: [n_frames][kp_capacity][32] */
    const int32_t *n_keypoints;    /* device: [n_frames] */
    const int32_t *level_counts;   /* device: [n_frames][SS_MAX_LEVELS] */
    const int32_t *frame_error;    /* device: [n_frames]; 0 or the ss_status of a frame whose capacity was exceeded */
} ss_batch_view;

typedef struct {
    char name[32];
    int64_t launches;
    double total_ms;  /* HIP-event time on the context's stream */
    double mean_ms;   /* per launch */
    double median_ms; /* per launch */
    int64_t algorithmic_bytes; /* per launch of the last shape run */
} ss_stage_stats;

int ss_abi_version(void);
int ss_orb_params_default(ss_orb_params *p);

int ss_create(int device_ordinal, const ss_orb_params *params, ss_ctx **out);
int ss_destroy(ss_ctx *ctx);
/* ctx may be NULL: returns the message of the last failed ss_create on this thread */
const char *ss_last_error(const ss_ctx *ctx);

int ss_set_calibration(ss_ctx *ctx, int camera_id, const ss_camera *cam);

/* Host pixels in, host keypoints/descriptors out; synchronous.  channels 1 (gray), 3 or 4
 * (converted with the calibration's rgb flag, exactly as GrabImageMonocular does);
 * camera_id must be non-zero (shim :528); caller keeps ownership of pix. */
int ss_extract(ss_ctx *ctx, int camera_id, const uint8_t *pix, int width, int height,
               int channels, int row_stride, double timestamp, ss_frame_result *out);

/* n_frames <= max_batch frames already in device memory (row_stride / frame_stride in
 * bytes).  Asynchronous on the context's stream; results stay on the device. */
int ss_extract_batch_device(ss_ctx *ctx, const void *d_pix, int n_frames, int width,
                            int height, int channels, int64_t row_stride,
                            int64_t frame_stride);
int ss_get_batch_view(ss_ctx *ctx, ss_batch_view *out);
/* Copies frame `frame` of the last batch to the context's host arrays (same ownership
 * rule as ss_extract); synchronises the context's stream. */
int ss_fetch_frame(ss_ctx *ctx, int frame, ss_frame_result *out);

/* K7.  idx[i] = index of the accepted best train descriptor or -1; d1/d2 = best and
 * second-best distance (0xFFFF when absent).  Accept iff d1 <= th and d1*ratio_den <
 * d2*ratio_num.  exclude_self skips j == i.  Ties: lowest index.  th < 0 = raw mode: no
 * acceptance test, idx = best index (or -1 when there is no train row): what a shard of a
 * partitioned database reports before the cross-shard merge. */
int ss_match(ss_ctx *ctx, const uint8_t *query, int n_query, const uint8_t *train,
             int n_train, int th, int ratio_num, int ratio_den, int exclude_self,
             int32_t *idx, uint16_t *d1, uint16_t *d2);
/* same with device pointers, asynchronous on the context's stream */
int ss_match_device(ss_ctx *ctx, const void *d_query, int n_query, const void *d_train,
                    int n_train, int th, int ratio_num, int ratio_den, int exclude_self,
                    void *d_idx, void *d_d1, void *d_d2);
/* Matches every frame of the last batch: mode 0 = self-match (exclude j == i), mode 1 =
 * frame b against frame b-1 (frame 0 against itself, excluding j == i).  Outputs are
 * device arrays [n_frames][kp_capacity]; rows >= n_keypoints[b] are idx -1. */
int ss_match_batch_device(ss_ctx *ctx, int mode, int th, int ratio_num, int ratio_den,
                          void *d_idx, void *d_d1, void *d_d2);

/* n_frames independent (query frame, train frame) pairs in one launch, e.g. the frames of this GPU's eye against the
 * all-gathered frames of the peer eye.  Both sides are device arrays
 * [n_frames][rows_per_frame][32] with per-frame row counts d_n_query / d_n_train (device int32 [n_frames]); outputs
 * are [n_frames][rows_per_frame], rows >= n_query[b] get idx -1 / 0xFFFF.  No self-exclusion. */
int ss_match_pairs_device(ss_ctx *ctx, const void *d_query, const void *d_n_query, const void *d_train,
                          const void *d_n_train, int n_frames, int rows_per_frame, int th, int ratio_num,
                          int ratio_den, void *d_idx, void *d_d1, void *d_d2);

/* Pose of one frame (shim :225-282 SendPosePacket: position + quaternion x y z w of Twc, shipped
 * only in tracking state OK :596).  tracking_state uses ORB_SLAM3::Tracking::eTrackingState values:
 * 0 NO_IMAGES_YET, 1 NOT_INITIALIZED, 2 OK, 4 LOST. */
typedef struct {
    int32_t tracking_state;
    int32_t camera_id;
    double timestamp;
    double position[3];
    double quaternion[4]; /* x y z w */
    int32_t n_keypoints;
    int32_t n_matches;    /* one-to-one matches to the reference / previous frame */
    int32_t n_inliers;    /* triangulated (initialisation) or pose-optimisation inliers */
    int32_t n_map_points; /* keypoints of this frame that carry a 3-D point */
} ss_pose;

/* The whole "frame" branch :521-627 = TrackMonocular :594 for a bounded monocular front-end:
 * ss_extract, device match against the initial / previous frame's descriptors (th 50, ratio
 * 0.9), then host double-precision geometry (csrc/ss_track.h: two-view initialisation, pose-only
 * optimisation, triangulation).  Requires ss_set_calibration (fx fy cx cy k1 k2 p1 p2 are used).
 * No keyframes, local mapping, loop closing or relocalisation. */
int ss_track(ss_ctx *ctx, int camera_id, const uint8_t *pix, int width, int height, int channels,
             int row_stride, double timestamp, ss_pose *out);
/* back to NO_IMAGES_YET (System::Reset / the "terminate" message :462-469) */
int ss_track_reset(ss_ctx *ctx);

/* The matrix-core matcher reads descriptors as rows of 256 FP4 values (one per bit, +1 / -1: 128 bytes).  The frames of a
 * batch get that form from the extraction itself; a database that is matched against again and again (loop closure,
 * relocalisation: SURVEY.md section 8(e) config 5) is expanded ONCE: n descriptors of 32 B at d_packed -> rows of 128 B
 * at d_expanded, which must hold n rounded up to a multiple of 32 rows (SS_EXPANDED_BYTES(n)).  ss_match_expanded_device
 * is ss_match_device on two expanded operands (same rule, same outputs, any sizes); ss_match_partial_expanded_device is
 * ss_match_partial_device on them. */
#define SS_EXPANDED_ROW_BYTES 128
#define SS_EXPANDED_BYTES(n) ((((int64_t)(n) + 31) & ~(int64_t)31) * SS_EXPANDED_ROW_BYTES)
int ss_expand_descriptors_device(ss_ctx *ctx, const void *d_packed, int n, void *d_expanded);
int ss_match_expanded_device(ss_ctx *ctx, const void *d_query_x, int n_query, const void *d_train_x, int n_train, int th,
                             int ratio_num, int ratio_den, int exclude_self, void *d_idx, void *d_d1, void *d_d2);
int ss_match_partial_expanded_device(ss_ctx *ctx, const void *d_query_x, int n_query, const void *d_train_x, int n_train,
                                     int64_t row_offset, void *d_part);

/* Pose step alone (ss_track without its ss_extract): the frame's descriptors are n rows of 32 bytes in DEVICE memory
 * (e.g. ss_pipe_result.d_descriptors of a completed slot), its keypoints are host memory.  Same state machine, same
 * match (th 50, ratio 0.9) and geometry as ss_track; frames must arrive in camera order. */
int ss_track_features(ss_ctx *ctx, int camera_id, double timestamp, const void *d_descriptors,
                      const ss_keypoint *keypoints, int n_keypoints, ss_pose *out);

/* ss_track_features for a caller that has matched the frames of a batch against each other already (ss_pipe match_mode 1,
 * ss_match_batch_device mode 1, with th 50 and ratio 9 / 10 -- the pose step's own rule): match_idx / match_d1 are host
 * arrays of n_keypoints entries, this frame's matches against the frame of the PREVIOUS pose-step call on this context
 * that returned SS_OK (NULL, NULL: none -- also the thing to pass after a call that failed).  They are used when that
 * frame is the one the tracker is about to match against (tracking, or the frame right after a new reference); in every
 * other case the tracker runs its own device match, as ss_track_features does, so the poses are the same either way.  flags: SS_TRACK_DESC_STAYS_VALID = d_descriptors stays valid and unchanged
 * until the next pose-step call on this context has returned (the rows of a pipe slot that is released after its last
 * frame): the tracker then refers to them instead of copying them.  With both, a tracked frame costs no device work. */
#define SS_TRACK_DESC_STAYS_VALID 1
int ss_track_features_matched(ss_ctx *ctx, int camera_id, double timestamp, const void *d_descriptors,
                              const ss_keypoint *keypoints, int n_keypoints, const int32_t *match_idx,
                              const uint16_t *match_d1, int flags, ss_pose *out);

/* ---- a database partitioned over GPUs ------------------------------------
 * A shard reports, per query descriptor, ss_match_part = (best distance, second-best distance, GLOBAL row of the
 * best or -1): 8 bytes, the unit every rank all-gathers.  ss_match_partial_device runs the raw local match of
 * n_query device descriptors against this shard's n_train rows (global row = row_offset + local row) and writes
 * d_part[n_query].  ss_match_fold_device folds n_parts such arrays laid out [part][n_query], parts in ASCENDING
 * row order, with the rule the match kernels use across their train chunks (ties keep the lower row; second best =
 * min over the losers' best and everyone's second best), then applies the acceptance test of ss_match.  One
 * launch; asynchronous on the context's stream. */
typedef struct {
    uint16_t d1, d2; /* 0xFFFF = none */
    int32_t row;     /* global row of the best, -1 = none */
} ss_match_part;
int ss_match_partial_device(ss_ctx *ctx, const void *d_query, int n_query, const void *d_train, int n_train,
                            int64_t row_offset, void *d_part);
int ss_match_fold_device(ss_ctx *ctx, const void *d_parts, int n_parts, int n_query, int th, int ratio_num,
                         int ratio_den, void *d_idx, void *d_d1, void *d_d2);
/* same, part p at d_parts + p * part_stride_bytes (a multiple of 8, >= n_query * 8): the layout ss_xchg_allgather leaves */
int ss_match_fold_strided_device(ss_ctx *ctx, const void *d_parts, int n_parts, int64_t part_stride_bytes, int n_query, int th,
                                 int ratio_num, int ratio_den, void *d_idx, void *d_d1, void *d_d2);

/* ---- the exchange step of configs 4 and 5 without PyTorch / RCCL ---------------------------------
 * One process per GPU.  ss_xchg_create is collective: every rank calls it with the same world, max_bytes and rendezvous
 * (the path of a Unix-domain socket rank 0 listens on while the hipIpc handles of the ranks' slabs are swapped); every rank
 * then has every peer's slab mapped -- different GPUs of one node (xGMI) or the same GPU.  A message is ONE hop: each rank
 * stores its block straight into every peer's slab and raises a flag there (csrc/ss_xchg.hip).
 *
 * ss_xchg_allgather: every rank contributes the same number of bytes, as up to 4 device segments laid back to back (each
 * padded to 16 bytes); on return *d_gathered points at [world][*rank_stride] bytes of LOCAL device memory, rank r's
 * block at r * *rank_stride, valid until the second-next message of this exchange.  ss_xchg_broadcast: d_buf of `root` ->
 * d_buf of everybody (in place, like ncclBroadcast).  Both are asynchronous on ctx's stream: enqueue the consumers on the same
 * stream.  Every rank must issue the same sequence of messages.  timeout_ms bounds the rendezvous of ss_xchg_create (0 = 10 s);
 * a message waits at most min(timeout_ms, 10 s).  A peer that does not show up within that limit ends
 * the waiting kernel (it never hangs the GPU) and poisons the exchange: ss_xchg_status / the next call return SS_ERR_STATE.
 * ss_xchg_destroy is collective too (a last flag-only message, so that nobody unmaps memory a peer still writes).
 * The reference has nothing here: one camera, one TCP link (orbslam3_mono_networked.cc:387-388, application.ex:80). */
typedef struct ss_xchg ss_xchg;
int ss_xchg_create(int device_ordinal, int rank, int world, int64_t max_bytes, const char *rendezvous, int timeout_ms,
                   ss_xchg **out);
int ss_xchg_destroy(ss_xchg *x);
/* x may be NULL: message of the last failed ss_xchg_create on this thread */
const char *ss_xchg_last_error(const ss_xchg *x);
int ss_xchg_status(ss_xchg *x);
int ss_xchg_allgather(ss_xchg *x, ss_ctx *ctx, const void *const *d_segments, const int64_t *segment_bytes, int n_segments,
                      const void **d_gathered, int64_t *rank_stride);
int ss_xchg_broadcast(ss_xchg *x, ss_ctx *ctx, int root, void *d_buf, int64_t bytes);
/* Config 4 in one call, for hosts that hold no device memory of their own (the TCP front door with SENDSLAM_SHARD=r/2, the NIF):
 * all-gathers the descriptor block and keypoint count of the frame ctx extracted last (ss_extract / ss_track; frame 0 of a
 * batch) with the peer rank's, matches this eye's descriptors against the peer eye's (th, ratio as in ss_match) and copies
 * idx / d1 / d2 (kp_capacity entries each; any of them may be NULL) to the host.  Both ranks call it once per stereo pair, in
 * lockstep.  *n_own / *n_peer: the two keypoint counts.  Synchronous. */
int ss_stereo_exchange_match(ss_ctx *ctx, ss_xchg *x, int peer_rank, int th, int ratio_num, int ratio_den, int32_t *idx,
                             uint16_t *d1, uint16_t *d2, int32_t *n_own, int32_t *n_peer);

int ss_synchronize(ss_ctx *ctx);
/* Orders the context's stream after everything enqueued so far on another stream of the same device
 * (hipStream_t; NULL = the legacy default stream): for callers that produce the inputs of a *_device call on their own
 * stream (a collective's output, a decoder) and must not launch the match before they are written. */
int ss_wait_stream(ss_ctx *ctx, void *hip_stream);
/* the hipStream_t every kernel of this context is launched on */
int ss_get_stream(ss_ctx *ctx, void **hip_stream);

/* Per-kernel HIP-event timing on the context's stream (off by default). */
int ss_profile_enable(ss_ctx *ctx, int on);
int ss_profile_reset(ss_ctx *ctx);
/* fills up to max_stages entries, returns the number of stages (or < 0) */
int ss_stats(ss_ctx *ctx, ss_stage_stats *out, int max_stages);

/* Intermediate buffers of the last batch, for stage-by-stage parity tests.  what:
 * 0 pyramid level, 1 blurred level, 2 FAST score map (tight w*h u8 each; no kernel reads the map, so it
 * is not kept in normal operation: the first request allocates it, re-runs the FAST kernel on the last
 * batch's pyramid, and the context keeps it from then on); 3 candidates,
 * 4 quadtree-selected keypoints of a level (int32 triples x, y, response; candidates are
 * relative to the (16,16) border origin, selected are level coordinates).  Returns the
 * number of bytes written to dst (<= dst_bytes) or < 0. */
int ss_debug_fetch(ss_ctx *ctx, int what, int frame, int level, void *dst, int64_t dst_bytes);
/* Test hook: sorts n <= 2048 items in place with the device's restatement of libstdc++
 * std::sort for ORB-SLAM3's compareNodes.  Item = size << 32 | UL.x << 20 | id (20 bits); the
 * comparator looks at (size, UL.x) only, so the placement of equal keys is what is tested. */
int ss_debug_sort(ss_ctx *ctx, uint64_t *items, int n);

/* ---- pipelined host-memory path ----------------------------------------------------------------------------
 * A pipe owns a ring of `depth` slots.  A slot = pinned host memory for `batch` frames + its own extraction
 * context (HBM buffers, HIP stream) + pinned host memory for the results.  Producer side: ss_pipe_acquire hands out
 * a free slot, the caller decodes / receives / copies frames straight into slot.pixels (frame i at
 * pixels + i * frame_stride, rows of row_stride bytes) and calls ss_pipe_submit; or ss_pipe_submit_frames gathers
 * caller-owned frames into a slot with a few host threads and submits it.  Submission enqueues, on the slot's stream,
 * H2D copy -> extraction (-> match) -> D2H copy of the results and returns at once, so the copies of one batch overlap
 * the kernels of the others.  Consumer side: ss_pipe_wait / ss_pipe_poll return completed batches in submission
 * order; the result arrays are the slot's pinned host memory and stay valid until ss_pipe_release(slot), which puts
 * the slot back into the ring.  A frame that is bad (NULL pointer, camera id 0) or exceeds an internal capacity gets
 * its own status; the other frames of the batch are unaffected (the shim's log-and-skip policy :523-551).
 * One producer thread and one consumer thread (or one thread doing both) per pipe. */
typedef struct ss_pipe ss_pipe;

typedef struct {
    int32_t width, height, channels; /* every frame of the pipe has this shape */
    int32_t batch;                   /* frames per slot, 1..256 */
    int32_t depth;                   /* slots, 2..16 */
    int32_t match_mode;              /* -1 none; 0 self-match; 1 frame b against frame b-1 of the batch (ss_match_batch_device) */
    int32_t match_th, ratio_num, ratio_den; /* 0 0 0 = the defaults 50, 9, 10 */
    int32_t copy_threads;            /* host threads of ss_pipe_submit_frames; 0 = 4 */
} ss_pipe_config;

typedef struct {
    int32_t slot;
    uint8_t *pixels; /* pinned host memory, batch * frame_stride bytes */
    int64_t row_stride, frame_stride;
} ss_pipe_slot;

typedef struct {
    int32_t slot;
    int32_t n_frames;
    int32_t kp_capacity;          /* rows per frame in the per-keypoint arrays */
    uint64_t sequence;            /* 0, 1, 2 ... in submission order */
    const int32_t *status;        /* [n_frames] SS_OK or the frame's ss_status */
    const int32_t *camera_id;     /* [n_frames] as submitted */
    const double *timestamp;      /* [n_frames] as submitted */
    const int32_t *n_keypoints;   /* [n_frames]; 0 for a frame whose status is not SS_OK */
    const int32_t *level_counts;  /* [n_frames][SS_MAX_LEVELS] */
    const ss_keypoint *keypoints; /* [n_frames][kp_capacity] */
    const uint8_t *descriptors;   /* [n_frames][kp_capacity][32] */
    const int32_t *match_idx;     /* [n_frames][kp_capacity], NULL when match_mode < 0; match_mode 1: all -1 for a frame whose train frame (the one before) is bad */
    const uint16_t *match_d1, *match_d2;
    const void *d_descriptors;    /* DEVICE copy of `descriptors` (same layout), valid until ss_pipe_release */
} ss_pipe_result;

/* cam may be NULL for 1-channel frames (its rgb flag decides the gray weights of 3/4-channel frames) */
int ss_pipe_create(int device_ordinal, const ss_orb_params *params, const ss_camera *cam,
                   const ss_pipe_config *cfg, ss_pipe **out);
int ss_pipe_destroy(ss_pipe *pipe);
/* pipe may be NULL: message of the last failed ss_pipe_create on this thread */
const char *ss_pipe_last_error(const ss_pipe *pipe);
int ss_pipe_acquire(ss_pipe *pipe, ss_pipe_slot *out);
/* camera_ids / timestamps: n_frames entries each, or NULL (camera 1, timestamp 0) */
int ss_pipe_submit(ss_pipe *pipe, int slot, int n_frames, const int32_t *camera_ids, const double *timestamps);
/* frames[i]: caller-owned host image of the pipe's shape with rows of row_stride bytes (NULL = a bad frame);
 * consumed before the call returns.  SS_ERR_BUSY when no slot is free (ss_pipe_last_error is not updated for SS_ERR_BUSY:
 * producers poll it). */
int ss_pipe_submit_frames(ss_pipe *pipe, const uint8_t *const *frames, int n_frames, int64_t row_stride,
                          const int32_t *camera_ids, const double *timestamps);
/* oldest submitted batch: wait blocks until it has completed; poll returns 1 (completed, *out filled), 0 (still
 * running, or nothing submitted) or < 0 */
int ss_pipe_wait(ss_pipe *pipe, ss_pipe_result *out);
int ss_pipe_poll(ss_pipe *pipe, ss_pipe_result *out);
int ss_pipe_release(ss_pipe *pipe, int slot);
/* batches submitted and not yet returned by wait / poll */
int ss_pipe_in_flight(const ss_pipe *pipe);
/* Test hook: the next submission fails (SS_ERR_HIP, "injected failure ...") after `after_operations` of its enqueues have
 * been issued.  A submission that fails half-way drains its streams before it returns, leaves the slot ACQUIRED (release or
 * resubmit it) and the pipe usable; ss_pipe_submit_frames frees its slot itself. */
int ss_pipe_debug_inject_failure(ss_pipe *pipe, int after_operations);

#ifdef __cplusplus
}
#endif
#endif
