/* ss_kernels.h -- host-callable launch wrappers of ss_kernels.hip (all asynchronous on `s`). */
#ifndef SS_KERNELS_H
#define SS_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sendslam_orb.h"
#include "ss_layout.h"

/* level 0 read in place from the caller's buffer (ptr != NULL: 1-channel image, 16-byte aligned base, row and frame
 * strides; no k_ingest pass), or from the pyramid block (ptr == NULL) */
struct ss_lvl0 {
    const uint8_t *ptr = nullptr;
    int pitch = 0;
    int64_t frame_stride = 0;
};

void ssk_ingest(hipStream_t s, const void *src, int channels, int64_t row_stride, int64_t frame_stride,
                int c0, int c1, int c2, uint8_t *pyr, const ss_geom *dg, const ss_geom &hg, int n_frames);
void ssk_resize(hipStream_t s, uint8_t *pyr, const ss_geom *dg, const ss_geom &hg, const ss_rtab *rtab,
                int level, int n_frames, const ss_lvl0 &l0);
/* levels `level` and `level + 1` in one launch (k_resize_pair), where ssk_resize_pair_fits(host geometry, HOST tap tables)
 * said so */
bool ssk_resize_pair_fits(const ss_geom &hg, const ss_rtab *host_rtab, int level);
void ssk_resize_pair(hipStream_t s, uint8_t *pyr, const ss_geom *dg, const ss_geom &hg, const ss_rtab *rtab, int level,
                     int n_frames, const ss_lvl0 &l0);
/* K2 + K3a + K6a fused: FAST response map, in-window NMS into per-tile survivor sub-lists, blurred pyramid -- one
 * staged tile, no global atomics */
void ssk_fast_blur_nms(hipStream_t s, const uint8_t *pyr, uint8_t *score, uint8_t *blur, const ss_geom *dg, const ss_geom &hg,
                       const uint32_t *tile_recs, const uint16_t *cinfo, uint32_t *tsurv, uint32_t *thdr, ss_level_state *state,
                       int n_frames, const ss_lvl0 &l0);
/* tile sub-lists -> per-cell buckets + count words (one thread per cell) */
void ssk_bucket_gather(hipStream_t s, const ss_geom *dg, const ss_geom &hg, const uint32_t *cell_units, const uint32_t *tsurv,
                       const uint32_t *thdr, uint32_t *bucket, uint32_t *cell_cnt, ss_level_state *state, int n_frames);
/* K3b: ranks the bucket entries of every cell into upstream's candidate order */
void ssk_cells_emit(hipStream_t s, const uint32_t *bucket, const ss_geom *dg, const ss_geom &hg, const uint32_t *cell_cnt,
                    uint32_t *cand, ss_level_state *state, int n_frames);
void ssk_quadtree(hipStream_t s, const ss_geom *dg, const ss_geom &hg, const uint32_t *cand, uint32_t *qbuf0,
                  uint32_t *qbuf1, ss_qnode *nodes, int32_t *lists, uint32_t *sel, ss_level_state *state,
                  int n_frames);
void ssk_slots(hipStream_t s, const ss_geom *dg, const uint32_t *sel, const ss_level_state *state, uint32_t *kp_ref,
               int32_t *n_kp, int32_t *level_counts, int32_t *frame_error, int n_frames);
void ssk_orient_describe(hipStream_t s, const ss_geom *dg, const ss_geom &hg, const uint8_t *pyr, const uint8_t *blur,
                         const uint32_t *sel, const uint32_t *kp_ref, const int32_t *n_kp, ss_keypoint *kps,
                         uint8_t *desc, int n_frames, const ss_lvl0 &l0, bool steer_fma, uint8_t *desc_x);

/* train split so that a launch has >> 256 workgroups and local indices fit 16 bits */
int ssk_match_chunks(int n_query_max, int n_train_max, int n_frames, int *chunk_len);
/* strides in 32-bit words; exclude_self_mode: 0 never, 1 always, 2 when train frame == query frame */
void ssk_match(hipStream_t s, const void *query, const void *train, const int32_t *nq_arr, const int32_t *nt_arr,
               int nq_fixed, int nt_fixed, int64_t q_frame_stride_words, int64_t t_frame_stride_words,
               int train_frame_shift, int chunk_len, int n_chunks, int exclude_self_mode, int th, int rnum, int rden,
               int out_stride, void *partial, int32_t *idx, uint16_t *d1, uint16_t *d2, int n_frames);
/* (idx, d1, d2) of a raw match -> ss_match_part records with global rows (row_offset + idx) */
void ssk_pack_partial(hipStream_t s, const int32_t *idx, const uint16_t *d1, const uint16_t *d2, int n, int32_t row_offset,
                      void *part);
/* cross-shard fold: parts [n_parts][nq] in ascending row order -> final outputs (k_match_merge's rule) */
void ssk_match_fold(hipStream_t s, const void *parts, int n_parts, int nq, int th, int rnum, int rden, int32_t *idx,
                    uint16_t *d1, uint16_t *d2);
void ssk_match_fold_strided(hipStream_t s, const void *parts, int64_t part_stride_bytes, int n_parts, int nq, int th, int rnum, int rden,
                            int32_t *idx, uint16_t *d1, uint16_t *d2);
/* the matrix-core matcher on descriptors already expanded to one FP4 value (+1 / -1) per bit (desc_x, SSK_X_ROW bytes per
 * row, written by ssk_orient_describe): batches of frames; frame strides in BYTES; multiples of 32 rows allocated */
#define SSK_X_ROW 128
void ssk_match_x(hipStream_t s, const uint8_t *query_x, const uint8_t *train_x, const int32_t *nq_arr, const int32_t *nt_arr,
                 int nq_fixed, int nt_fixed, int64_t q_frame_stride, int64_t t_frame_stride, int train_frame_shift, int chunk_len,
                 int n_chunks, int exclude_self_mode, int th, int rnum, int rden, int out_stride, void *partial, int32_t *idx,
                 uint16_t *d1, uint16_t *d2, int n_frames, const uint8_t *query_p = nullptr, const uint8_t *train_p = nullptr,
                 int64_t qp_frame_stride = 0, int64_t tp_frame_stride = 0);
/* (query_p / train_p: the same rows as packed 32-byte descriptors where the caller has them, frame strides in bytes: the second
 * launch, which recomputes 15 distances per query, then reads a quarter of the bytes) */
/* packed 32-B descriptors -> expanded SSK_X_ROW-byte rows; `out` holds n rounded up to 32 rows */
void ssk_expand_desc(hipStream_t s, const void *packed, int n, void *out);
/* [n_frames][rows][32] packed -> [n_frames][rows rounded up to 32][SSK_X_ROW] */
void ssk_expand_desc_frames(hipStream_t s, const void *packed, int rows, int n_frames, void *out);
/* one expanded query set against one expanded train set (any size): chunk plan + launch (+ merge of the chunk partials) */
int ssk_match_x_chunks(int n_query, int n_train, int *chunk_len);
/* chunk plan of a batch of frames (rows_q query rows against rows_t train rows per frame) */
int ssk_match_x_batch_chunks(int rows_q, int rows_t, int n_frames, int *chunk_len);
void ssk_match_x_single(hipStream_t s, const uint8_t *query_x, int nq, const uint8_t *train_x, int nt, int chunk_len, int n_chunks,
                        int exclude_self, int th, int rnum, int rden, void *partial, int32_t *idx, uint16_t *d1, uint16_t *d2,
                        const uint8_t *query_p = nullptr, const uint8_t *train_p = nullptr);
/* test hook: run the device std::sort restatement on n <= 2048 items (size << 32 | UL.x << 20 | id) */
int ssk_debug_sort(hipStream_t s, uint64_t *d_items, int n);
#define SSK_MATCH_PARTIAL_BYTES 8
#define SSK_MATCH_MFMA_MIN_QUERIES 128 /* from this many query rows on, ssk_match runs the matrix-core kernel */
/* database-streaming form for n_query <= 8 and n_train >= 65536: plan (false = not applicable), the HBM-bound kernel,
 * the merge of its per-chunk partials */
bool ssk_match_stream_plan(int nq, int nt, size_t partial_bytes, int *chunk_len, int *n_chunks);
void ssk_match_stream_kernel(hipStream_t s, const void *query, const void *train, int nq, int nt, int chunk_len, int n_chunks,
                             void *partial);
void ssk_match_stream_merge(hipStream_t s, const void *partial, int nq, int n_chunks, int th, int rnum, int rden, int32_t *idx,
                            uint16_t *d1, uint16_t *d2);
#define SSK_STREAM_PARTIAL_MAX (2048 * 8 * SSK_MATCH_PARTIAL_BYTES)

#endif
