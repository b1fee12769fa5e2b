/*
 * ss_api.cpp -- the C ABI of libsendslam_orb.so (include/sendslam_orb.h): context, HBM
 * buffers, the per-batch kernel sequence on ONE HIP stream, HIP-event stage timing.
 *
 * Host-side counterpart of the reference shim's frame branch
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:521-627): same
 * guards and the same log-and-skip error policy, with the ORB-SLAM3 call at :594 replaced
 * by the kernel sequence below.  No CPU fallback exists: every stage runs on the device.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sendslam_orb.h"
#include "ss_constants.h"
#include "ss_geometry.h"
#include "ss_kernels.h"
#include "ss_layout.h"
#include "ss_track.h"

namespace {

thread_local std::string g_create_error;

struct stage_rec {
    std::string name;
    int64_t bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<float> ms;
};

template <typename T> void dev_free(T *&p)
{
    if (p) (void)hipFree((void *)p);
    p = nullptr;
}

} // namespace

struct ss_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    ss_orb_params params{};
    std::string err;
    bool calibrated = false;
    bool force_ingest = false; /* SENDSLAM_FORCE_INGEST=1: always copy level 0 into the pyramid block (tests) */
    bool resize_pair[SS_MAX_LEVELS] = {}; /* levels l, l + 1 built by one k_resize_pair launch (checked on the tap tables) */
    int skip_after = 0, n_extracts = 0; /* SENDSLAM_SKIP_AFTER=k: the mask applies from the k-th batch of the context on, and the
                                * per-level state of the batch before is kept (so that what follows a skipped stage still has work) */
    int skip_stages = 0;       /* SENDSLAM_SKIP_STAGES bit mask, timing experiments only (results invalid): 1 quadtree, 2 orient_describe,
                                * 4 resize, 8 fast_blur_nms, 16 gather + emit, 32 match */
    bool no_desc_x = false;    /* SENDSLAM_MATCH_PACKED=1: batch matches run k_match_mfma on the packed descriptors (A/B tests) */
    ss_camera cam{};
    int cam_id = 0;

    bool have_geom = false;
    ss_geom hg{};
    ss_host_tables tabs;
    ss_geom *dg = nullptr;
    ss_rtab *d_rtab = nullptr;
    uint32_t *d_tiles2 = nullptr; /* per-tile records of the FAST kernel (SS_TILE_REC_WORDS each) */

    uint8_t *pyr = nullptr, *blur = nullptr, *score = nullptr;
    uint32_t *cell_cnt = nullptr;
    uint16_t *d_cinfo = nullptr;
    uint32_t *cand = nullptr, *qbuf0 = nullptr, *qbuf1 = nullptr;
    uint32_t *bucket = nullptr; /* per cell: NMS survivors, unordered */
    uint32_t *tsurv = nullptr, *thdr = nullptr; /* per 64x32 tile: survivor sub-lists and their count words */
    uint32_t *d_cell_units = nullptr;
    ss_qnode *nodes = nullptr;
    int32_t *lists = nullptr;
    uint32_t *sel = nullptr;
    ss_level_state *state = nullptr;
    uint32_t *kp_ref = nullptr;
    int32_t *n_kp = nullptr, *level_counts = nullptr, *frame_error = nullptr;
    ss_keypoint *kps = nullptr;
    uint8_t *desc = nullptr;
    uint8_t *desc_x = nullptr; /* the descriptors as 256 FP4 values (+1 / -1) per row, 128 B: operand of the batch matcher */

    uint8_t *d_in = nullptr;
    size_t d_in_bytes = 0;
    void *match_partial = nullptr;
    size_t match_partial_bytes = 0;
    uint8_t *d_mq = nullptr, *d_mt = nullptr, *d_mout = nullptr, *d_part_tmp = nullptr;
    size_t d_mq_bytes = 0, d_mt_bytes = 0, d_mout_bytes = 0, d_part_tmp_bytes = 0;
    uint8_t *d_qx = nullptr, *d_tx = nullptr; /* caller descriptors expanded to the matrix-core matcher's operand rows */
    size_t d_qx_bytes = 0, d_tx_bytes = 0;

    /* host results of ss_extract */
    std::vector<ss_keypoint> h_kps;
    std::vector<uint8_t> h_desc;
    std::vector<int32_t> h_err;

    int last_n_frames = 0;
    ss_lvl0 last_lvl0; /* where level 0 of the last batch lives (ptr == NULL: in the pyramid block) */

    /* ss_track: descriptors of the initialisation reference / the previous frame, host geometry */
    uint8_t *d_ref_desc = nullptr, *d_prev_desc = nullptr;
    size_t d_ref_desc_bytes = 0, d_prev_desc_bytes = 0;
    uint8_t *d_ref_desc_x = nullptr, *d_prev_desc_x = nullptr; /* the same rows as matrix-core operands (128 B each) */
    size_t d_ref_desc_x_bytes = 0, d_prev_desc_x_bytes = 0;
    sst_tracker tracker;
    std::vector<float> h_xy;
    std::vector<int32_t> h_oct, h_midx;
    std::vector<uint16_t> h_md1;

    /* SENDSLAM_TRACK_TIMING=1: host seconds of the pose step, printed at ss_destroy (match = enqueue + wait for the device match,
     * geometry = sst_tracker::step, keep = copies of the descriptors the next frame matches against) */
    bool track_timing = false;
    double t_match = 0, t_geom = 0, t_keep = 0;
    /* calls of the pose step are numbered: which call's frame the tracker holds as its previous / reference frame, and
     * whether the previous frame's descriptors are still the caller's rows (ss_track_features_matched) */
    int64_t track_serial = 0, prev_serial = -1, ref_serial = -1;
    const uint8_t *d_prev_ext = nullptr;
    int64_t n_tracked = 0;
    bool profile = false;
    std::vector<stage_rec> stages;
    std::vector<hipEvent_t> event_pool;
};

namespace {

int fail(ss_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(c, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((c), e_ == hipErrorOutOfMemory ? SS_ERR_NO_MEMORY : SS_ERR_HIP,       \
                        std::string(#call) + ": " + hipGetErrorString(e_));                   \
    } while (0)

stage_rec &stage(ss_ctx *c, const char *name)
{
    for (auto &s : c->stages)
        if (s.name == name) return s;
    c->stages.emplace_back();
    c->stages.back().name = name;
    return c->stages.back();
}

hipEvent_t get_event(ss_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct stage_timer {
    ss_ctx *c;
    stage_rec *s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t on;
    stage_timer(ss_ctx *ctx, const char *name, int64_t bytes, hipStream_t stream = nullptr)
        : c(ctx), on(stream ? stream : ctx->stream)
    {
        if (!c->profile) return;
        s = &stage(c, name);
        s->bytes = bytes;
        a = get_event(c);
        b = get_event(c);
        (void)hipEventRecord(a, on); /* on the stream the kernel is launched on */
    }
    ~stage_timer()
    {
        if (!s) return;
        (void)hipEventRecord(b, on);
        s->pending.emplace_back(a, b);
    }
};

void collect_events(ss_ctx *c)
{
    for (auto &s : c->stages) {
        for (auto &p : s.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) s.ms.push_back(ms);
            c->event_pool.push_back(p.first);
            c->event_pool.push_back(p.second);
        }
        s.pending.clear();
    }
}

void free_geometry_buffers(ss_ctx *c)
{
    dev_free(c->dg);
    dev_free(c->d_rtab);
    dev_free(c->d_tiles2);
    dev_free(c->pyr);
    dev_free(c->blur);
    dev_free(c->score);
    dev_free(c->d_cinfo);
    dev_free(c->cell_cnt);
    dev_free(c->cand);
    dev_free(c->qbuf0);
    dev_free(c->qbuf1);
    dev_free(c->bucket);
    dev_free(c->tsurv);
    dev_free(c->thdr);
    dev_free(c->d_cell_units);
    dev_free(c->nodes);
    dev_free(c->lists);
    dev_free(c->sel);
    dev_free(c->state);
    dev_free(c->kp_ref);
    dev_free(c->n_kp);
    dev_free(c->level_counts);
    dev_free(c->frame_error);
    dev_free(c->kps);
    dev_free(c->desc);
    dev_free(c->desc_x);
    c->have_geom = false;
}

int ensure_geometry(ss_ctx *c, int w, int h)
{
    if (c->have_geom && c->hg.w == w && c->hg.h == h) return SS_OK;
    (void)hipStreamSynchronize(c->stream);
    free_geometry_buffers(c);
    std::string msg;
    ss_geom g;
    int rc = ss_build_geometry(c->params, w, h, &g, &c->tabs, &msg);
    if (rc != SS_OK) return fail(c, rc, msg);
    for (int l = 0; l < g.n_levels; l++)
        if (g.lv[l].item_cap > 2048)
            return fail(c, SS_ERR_INVALID_ARG, "n_features too large: per-level quota exceeds 2032");
    c->hg = g;
    {
        /* SENDSLAM_RESIZE_PAIRS=1: two pyramid levels per launch (k_resize_pair: four launches instead of seven, bit-exact).
         * Off by default: alone it takes the same 0.126 ms per 64 frames, with four batches in flight it costs 7.6 % frames/s
         * (31 KB of LDS per block and 10 % more instructions for the overlapping windows) */
        const char *e = getenv("SENDSLAM_RESIZE_PAIRS");
        std::fill(std::begin(c->resize_pair), std::end(c->resize_pair), false); /* flags of the previous geometry do not survive */
        for (int l = 1; l + 1 < g.n_levels && e && atoi(e);)
            if (ssk_resize_pair_fits(g, c->tabs.rtab.data(), l)) { c->resize_pair[l] = true; l += 2; } else l += 1;
    }
    const size_t B = (size_t)c->params.max_batch;
    HIP_TRY(c, hipMalloc((void **)&c->dg, sizeof(ss_geom)));
    HIP_TRY(c, hipMemcpy(c->dg, &g, sizeof(ss_geom), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc((void **)&c->d_rtab, std::max<size_t>(c->tabs.rtab.size(), 1) * sizeof(ss_rtab)));
    if (!c->tabs.rtab.empty())
        HIP_TRY(c, hipMemcpy(c->d_rtab, c->tabs.rtab.data(), c->tabs.rtab.size() * sizeof(ss_rtab), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc((void **)&c->d_tiles2, c->tabs.tile_recs.size() * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpy(c->d_tiles2, c->tabs.tile_recs.data(), c->tabs.tile_recs.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc((void **)&c->pyr, B * g.block_bytes));
    HIP_TRY(c, hipMalloc((void **)&c->blur, B * g.block_bytes));
    /* c->score (the FAST response map) is allocated by the first ss_debug_fetch(2): no kernel reads it */
    HIP_TRY(c, hipMalloc((void **)&c->d_cinfo, c->tabs.cinfo.size() * sizeof(uint16_t)));
    HIP_TRY(c, hipMemcpy(c->d_cinfo, c->tabs.cinfo.data(), c->tabs.cinfo.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc((void **)&c->cell_cnt, B * g.n_cells * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->cand, B * g.cand_total * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->qbuf0, B * g.cand_total * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->qbuf1, B * g.cand_total * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->bucket, B * g.bucket_total * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->tsurv, B * g.tiles2_total * (size_t)SS_TS_CAP * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->thdr, B * g.tiles2_total * (size_t)SS_TS_HDR * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->d_cell_units, c->tabs.cell_units.size() * sizeof(uint32_t)));
    HIP_TRY(c, hipMemcpy(c->d_cell_units, c->tabs.cell_units.data(), c->tabs.cell_units.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc((void **)&c->nodes, B * g.node_total * sizeof(ss_qnode)));
    HIP_TRY(c, hipMalloc((void **)&c->lists, B * g.item_total * 2 * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->sel, B * g.sel_total * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->state, B * SS_MAX_LEVELS * sizeof(ss_level_state)));
    HIP_TRY(c, hipMalloc((void **)&c->kp_ref, B * g.kcap * 2 * sizeof(uint32_t))); /* (reference, record) per output slot */
    HIP_TRY(c, hipMalloc((void **)&c->n_kp, B * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->level_counts, B * SS_MAX_LEVELS * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->frame_error, B * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void **)&c->kps, B * g.kcap * sizeof(ss_keypoint)));
    HIP_TRY(c, hipMalloc((void **)&c->desc, B * g.kcap * SS_DESC_BYTES));
    HIP_TRY(c, hipMemset(c->n_kp, 0, B * sizeof(int32_t)));
    HIP_TRY(c, hipMemset(c->kps, 0, B * g.kcap * sizeof(ss_keypoint)));
    HIP_TRY(c, hipMemset(c->desc, 0, B * g.kcap * SS_DESC_BYTES));
    if (!c->no_desc_x && g.kcap >= SSK_MATCH_MFMA_MIN_QUERIES) {
        /* zeroed once: a row that was never written contributes 0 to every dot product (and is masked out anyway) */
        HIP_TRY(c, hipMalloc((void **)&c->desc_x, B * g.kcap * SSK_X_ROW));
        HIP_TRY(c, hipMemset(c->desc_x, 0, B * g.kcap * SSK_X_ROW));
    }
    c->have_geom = true;
    c->last_n_frames = 0;
    return SS_OK;
}

int64_t level_px(const ss_geom &g, int l) { return (int64_t)g.lv[l].w * g.lv[l].h; }

/* the per-batch kernel sequence; d_pix is device memory */
int run_extract(ss_ctx *c, const void *d_pix, int n, int channels, int64_t row_stride, int64_t frame_stride)
{
    const ss_geom &g = c->hg;
    hipStream_t s = c->stream;
    int64_t all_px = 0;
    for (int l = 0; l < g.n_levels; l++) all_px += level_px(g, l);

#ifdef SS_TIMING_KNOBS /* profiles/tools/build_variant.sh only: the shipped library has no way to skip a stage */
    const int skip_mask = c->n_extracts++ >= c->skip_after ? c->skip_stages : 0;
#else
    const int skip_mask = 0;
#endif
    if (!(skip_mask && c->skip_after > 0))
        HIP_TRY(c, hipMemsetAsync(c->state, 0, (size_t)n * SS_MAX_LEVELS * sizeof(ss_level_state), s));
    /* A 1-channel image whose base, rows and frames are 16-byte aligned IS pyramid level 0: the kernels read it in
     * place (aligned dword loads work on it as they do on the pyramid block) and the ingest copy is skipped.  The
     * caller's buffer must stay untouched until the batch has finished (it is asynchronous, as before). */
    ss_lvl0 l0;
    if (channels == 1 && ((uintptr_t)d_pix % 16) == 0 && row_stride % 16 == 0 && frame_stride % 16 == 0 &&
        row_stride < (1 << 24) && !c->force_ingest) {
        l0.ptr = (const uint8_t *)d_pix;
        l0.pitch = (int)row_stride;
        l0.frame_stride = frame_stride;
    }
    c->last_lvl0 = l0;
    if (!l0.ptr) {
        int c0 = 0, c1 = 0, c2 = 0;
        if (channels != 1) { /* Camera.RGB: 1 -> byte 0 weighs as R */
            const bool rgb = c->calibrated ? c->cam.rgb != 0 : false;
            c0 = rgb ? SS_GRAY_RY : SS_GRAY_BY;
            c1 = SS_GRAY_GY;
            c2 = rgb ? SS_GRAY_BY : SS_GRAY_RY;
        }
        stage_timer t(c, "ingest", n * level_px(g, 0) * (channels + 1));
        ssk_ingest(s, d_pix, channels, row_stride, frame_stride, c0, c1, c2, c->pyr, c->dg, g, n);
    }
    for (int l = 1; l < g.n_levels && !(skip_mask & 4);) {
        if (c->resize_pair[l]) { /* two pyramid steps, the middle level never read back */
            stage_timer t(c, "resize", n * (level_px(g, l - 1) + level_px(g, l) + level_px(g, l + 1)));
            ssk_resize_pair(s, c->pyr, c->dg, g, c->d_rtab, l, n, l0);
            l += 2;
        } else {
            stage_timer t(c, "resize", n * (level_px(g, l - 1) + level_px(g, l)));
            ssk_resize(s, c->pyr, c->dg, g, c->d_rtab, l, n, l0);
            l += 1;
        }
    }
    if (!(skip_mask & 8)) {
        /* algorithmic bytes: read the pyramid once, write the blurred pyramid (the score map and the
         * survivor lists are this design's own intermediates) */
        stage_timer t(c, "fast_blur_nms", n * 2 * all_px);
        ssk_fast_blur_nms(s, c->pyr, c->score, c->blur, c->dg, g, c->d_tiles2, c->d_cinfo, c->tsurv, c->thdr, c->state, n, l0);
    }
    if (!(skip_mask & 16)) {
        stage_timer t(c, "bucket_gather", 0);
        ssk_bucket_gather(s, c->dg, g, c->d_cell_units, c->tsurv, c->thdr, c->bucket, c->cell_cnt, c->state, n);
    }
    if (!(skip_mask & 16)) {
        stage_timer t(c, "cells_emit", 0);
        ssk_cells_emit(s, c->bucket, c->dg, g, c->cell_cnt, c->cand, c->state, n);
    }
    if (!(skip_mask & 1)) {
        stage_timer t(c, "quadtree", 0);
        ssk_quadtree(s, c->dg, g, c->cand, c->qbuf0, c->qbuf1, c->nodes, c->lists, c->sel, c->state, n);
    }
    {
        stage_timer t(c, "slots", 0);
        ssk_slots(s, c->dg, c->sel, c->state, c->kp_ref, c->n_kp, c->level_counts, c->frame_error, n);
    }
    if (!(skip_mask & 2)) {
        stage_timer t(c, "orient_describe", (int64_t)n * g.n_features * (709 + 512 + 32 + 24));
        ssk_orient_describe(s, c->dg, g, c->pyr, c->blur, c->sel, c->kp_ref, c->n_kp, c->kps, c->desc, n, l0, c->params.steer_fma != 0, c->desc_x);
    }
    HIP_TRY(c, hipGetLastError());
    c->last_n_frames = n;
    return SS_OK;
}

int check_frame_errors(ss_ctx *c)
{
    const int n = c->last_n_frames;
    if (n <= 0) return SS_OK;
    c->h_err.resize((size_t)n);
    HIP_TRY(c, hipMemcpyAsync(c->h_err.data(), c->frame_error, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++)
        if (c->h_err[i] != 0)
            return fail(c, SS_ERR_OVERFLOW, "frame " + std::to_string(i) + ": an internal capacity was exceeded (code " +
                                                std::to_string(c->h_err[i]) + "); no result was truncated");
    return SS_OK;
}

template <typename T> int grow(ss_ctx *c, T *&p, size_t &have, size_t want)
{
    if (have >= want) return SS_OK;
    (void)hipStreamSynchronize(c->stream);
    dev_free(p);
    have = 0;
    HIP_TRY(c, hipMalloc((void **)&p, want));
    have = want;
    return SS_OK;
}

} // namespace

static int match_expanded(ss_ctx *c, const void *d_query_x, int n_query, const void *d_train_x, int n_train, int th, int ratio_num,
                          int ratio_den, int exclude_self, void *d_idx, void *d_d1, void *d_d2, const uint8_t *q_packed, const uint8_t *t_packed);

extern "C" {

int ss_abi_version(void) { return SS_ABI_VERSION; }

int ss_orb_params_default(ss_orb_params *p)
{
    if (!p) return SS_ERR_INVALID_ARG;
    p->n_features = SS_DEFAULT_NFEATURES;
    p->scale_factor = SS_DEFAULT_SCALE;
    p->n_levels = SS_DEFAULT_NLEVELS;
    p->ini_th_fast = SS_DEFAULT_INI_TH;
    p->min_th_fast = SS_DEFAULT_MIN_TH;
    p->lapping_x0 = SS_DEFAULT_LAPPING_X0;
    p->lapping_x1 = SS_DEFAULT_LAPPING_X1;
    p->max_batch = 1;
    p->steer_fma = 0;
    return SS_OK;
}

int ss_create(int device_ordinal, const ss_orb_params *params, ss_ctx **out)
{
    if (!out) return fail(nullptr, SS_ERR_INVALID_ARG, "ss_create: out is NULL");
    *out = nullptr;
    ss_orb_params p;
    ss_orb_params_default(&p);
    if (params) p = *params;
    if (p.max_batch < 1 || p.max_batch > 4096) return fail(nullptr, SS_ERR_INVALID_ARG, "max_batch out of range (1..4096)");
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, SS_ERR_NO_DEVICE,
                    std::string("no HIP device: this library has no CPU path (") + hipGetErrorString(e) + ")");
    if (device_ordinal < 0 || device_ordinal >= n_dev)
        return fail(nullptr, SS_ERR_NO_DEVICE, "device ordinal " + std::to_string(device_ordinal) + " out of range");
    HIP_TRY((ss_ctx *)nullptr, hipSetDevice(device_ordinal));
    {
        /* validate the parameters on a nominal size now, so a bad parameter fails here */
        ss_geom g;
        ss_host_tables tabs;
        std::string msg;
        int rc = ss_build_geometry(p, 640, 480, &g, &tabs, &msg);
        if (rc == SS_ERR_INVALID_ARG) return fail(nullptr, rc, msg);
    }
    ss_ctx *c = new ss_ctx();
    c->device = device_ordinal;
    c->params = p;
    if (const char *e = getenv("SENDSLAM_FORCE_INGEST")) c->force_ingest = atoi(e) != 0;
    if (const char *e = getenv("SENDSLAM_TRACK_TIMING")) c->track_timing = atoi(e) != 0;
    if (const char *e = getenv("SENDSLAM_MATCH_PACKED")) c->no_desc_x = atoi(e) != 0;
#ifdef SS_TIMING_KNOBS
    if (const char *e = getenv("SENDSLAM_SKIP_STAGES")) c->skip_stages = atoi(e);
    if (const char *e = getenv("SENDSLAM_SKIP_AFTER")) c->skip_after = atoi(e);
#endif
    hipError_t se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) {
        delete c;
        return fail(nullptr, SS_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(se));
    }
    *out = c;
    return SS_OK;
}

int ss_destroy(ss_ctx *c)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->track_timing && c->n_tracked)
        fprintf(stderr, "ss_track timing over %lld frames: match %.3f ms, geometry %.3f ms, keep %.3f ms per frame\n", (long long)c->n_tracked,
                1e3 * c->t_match / c->n_tracked, 1e3 * c->t_geom / c->n_tracked, 1e3 * c->t_keep / c->n_tracked);
    collect_events(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    free_geometry_buffers(c);
    dev_free(c->d_in);
    dev_free(c->match_partial);
    dev_free(c->d_mq);
    dev_free(c->d_mt);
    dev_free(c->d_mout);
    dev_free(c->d_part_tmp);
    dev_free(c->d_qx);
    dev_free(c->d_tx);
    dev_free(c->d_ref_desc);
    dev_free(c->d_prev_desc);
    dev_free(c->d_ref_desc_x);
    dev_free(c->d_prev_desc_x);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return SS_OK;
}

const char *ss_last_error(const ss_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ss_set_calibration(ss_ctx *c, int camera_id, const ss_camera *cam)
{
    if (!c) return SS_ERR_INVALID_ARG;
    if (camera_id == 0) return fail(c, SS_ERR_INVALID_ARG, "Calibration message missing camera identifier.");
    if (!cam) return fail(c, SS_ERR_INVALID_ARG, "Calibration message missing structured parameter payload.");
    c->cam = *cam;
    c->cam.type[sizeof(c->cam.type) - 1] = 0;
    c->cam_id = camera_id;
    c->calibrated = true;
    /* the reference rebuilds the whole ORB_SLAM3::System on every calibration message (:491-518): the map, the
     * reference frame and the motion model do not survive it */
    c->tracker.reset();
    c->prev_serial = c->ref_serial = -1;
    c->d_prev_ext = nullptr;
    return SS_OK;
}

int ss_extract(ss_ctx *c, int camera_id, const uint8_t *pix, int width, int height, int channels, int row_stride,
               double timestamp, ss_frame_result *out)
{
    if (!c || !out) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (camera_id == 0) return fail(c, SS_ERR_BAD_FRAME, "Frame message missing camera identifier.");
    if (!pix || width <= 0 || height <= 0) return fail(c, SS_ERR_BAD_FRAME, "Frame message missing binary image data.");
    if (channels != 1 && channels != 3 && channels != 4) return fail(c, SS_ERR_BAD_FRAME, "unsupported channel count");
    if (channels != 1 && !c->calibrated)
        return fail(c, SS_ERR_NOT_CALIBRATED, "Received frame before calibration. Ignoring.");
    if (row_stride < width * channels) return fail(c, SS_ERR_BAD_FRAME, "row_stride smaller than a row");
    int rc = ensure_geometry(c, width, height);
    if (rc != SS_OK) return rc;
    /* the last row of a tight caller buffer ends after width * channels bytes, not after row_stride */
    const size_t bytes = (size_t)row_stride * (height - 1) + (size_t)width * channels;
    const size_t alloc = ((size_t)row_stride * height + 15) & ~(size_t)15;
    rc = grow(c, c->d_in, c->d_in_bytes, alloc);
    if (rc != SS_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_in, pix, bytes, hipMemcpyHostToDevice, c->stream));
    /* the caller keeps ownership of pix: it is consumed before we return */
    rc = run_extract(c, c->d_in, 1, channels, row_stride, (int64_t)alloc);
    if (rc != SS_OK) return rc;
    int32_t nk = 0;
    HIP_TRY(c, hipMemcpyAsync(&nk, c->n_kp, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(out->level_counts, c->level_counts, SS_MAX_LEVELS * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    rc = check_frame_errors(c); /* synchronises */
    if (rc != SS_OK) return rc;
    c->h_kps.resize((size_t)std::max(nk, 1));
    c->h_desc.resize((size_t)std::max(nk, 1) * SS_DESC_BYTES);
    if (nk > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->h_kps.data(), c->kps, (size_t)nk * sizeof(ss_keypoint), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_desc.data(), c->desc, (size_t)nk * SS_DESC_BYTES, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    out->n_keypoints = nk;
    out->camera_id = camera_id;
    out->timestamp = timestamp;
    out->keypoints = c->h_kps.data();
    out->descriptors = c->h_desc.data();
    return SS_OK;
}

int ss_extract_batch_device(ss_ctx *c, const void *d_pix, int n_frames, int width, int height, int channels,
                            int64_t row_stride, int64_t frame_stride)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!d_pix || n_frames < 1 || width <= 0 || height <= 0) return fail(c, SS_ERR_BAD_FRAME, "empty batch");
    if (n_frames > c->params.max_batch) return fail(c, SS_ERR_INVALID_ARG, "n_frames exceeds max_batch of this context");
    if (channels != 1 && channels != 3 && channels != 4) return fail(c, SS_ERR_BAD_FRAME, "unsupported channel count");
    if (channels != 1 && !c->calibrated)
        return fail(c, SS_ERR_NOT_CALIBRATED, "Received frame before calibration. Ignoring.");
    if (row_stride < (int64_t)width * channels || frame_stride < row_stride * height)
        return fail(c, SS_ERR_BAD_FRAME, "strides smaller than the frame");
    int rc = ensure_geometry(c, width, height);
    if (rc != SS_OK) return rc;
    return run_extract(c, d_pix, n_frames, channels, row_stride, frame_stride);
}

int ss_get_batch_view(ss_ctx *c, ss_batch_view *out)
{
    if (!c || !out) return SS_ERR_INVALID_ARG;
    if (!c->have_geom || c->last_n_frames <= 0) return fail(c, SS_ERR_STATE, "no batch has been extracted");
    out->n_frames = c->last_n_frames;
    out->kp_capacity = c->hg.kcap;
    out->keypoints = c->kps;
    out->descriptors = c->desc;
    out->n_keypoints = c->n_kp;
    out->level_counts = c->level_counts;
    out->frame_error = c->frame_error;
    return SS_OK;
}

int ss_fetch_frame(ss_ctx *c, int frame, ss_frame_result *out)
{
    if (!c || !out) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!c->have_geom || frame < 0 || frame >= c->last_n_frames) return fail(c, SS_ERR_STATE, "ss_fetch_frame: no such frame");
    int rc = check_frame_errors(c);
    if (rc != SS_OK) return rc;
    const int kcap = c->hg.kcap;
    int32_t nk = 0;
    HIP_TRY(c, hipMemcpy(&nk, c->n_kp + frame, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(out->level_counts, c->level_counts + (size_t)frame * SS_MAX_LEVELS, SS_MAX_LEVELS * sizeof(int32_t), hipMemcpyDeviceToHost));
    c->h_kps.resize((size_t)std::max(nk, 1));
    c->h_desc.resize((size_t)std::max(nk, 1) * SS_DESC_BYTES);
    if (nk > 0) {
        HIP_TRY(c, hipMemcpy(c->h_kps.data(), c->kps + (size_t)frame * kcap, (size_t)nk * sizeof(ss_keypoint), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_desc.data(), c->desc + (size_t)frame * kcap * SS_DESC_BYTES, (size_t)nk * SS_DESC_BYTES, hipMemcpyDeviceToHost));
    }
    out->n_keypoints = nk;
    out->camera_id = c->cam_id;
    out->timestamp = 0.0;
    out->keypoints = c->h_kps.data();
    out->descriptors = c->h_desc.data();
    return SS_OK;
}

int ss_match_device(ss_ctx *c, const void *d_query, int n_query, const void *d_train, int n_train, int th,
                    int ratio_num, int ratio_den, int exclude_self, void *d_idx, void *d_d1, void *d_d2)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_query < 0 || n_train < 0 || ratio_den <= 0 || ratio_num < 0) return fail(c, SS_ERR_INVALID_ARG, "bad match arguments");
    if (n_query == 0) return SS_OK;
    if (!d_query || (!d_train && n_train > 0) || !d_idx || !d_d1 || !d_d2) return fail(c, SS_ERR_INVALID_ARG, "NULL match buffer");
    if (n_query <= 8 && n_train >= 65536 && !exclude_self) {
        /* a handful of queries against a large database: stream the database once (HBM-bound) */
        int rc = grow(c, c->match_partial, c->match_partial_bytes, (size_t)SSK_STREAM_PARTIAL_MAX);
        if (rc != SS_OK) return rc;
        int s_len = 0, s_chunks = 0;
        if (ssk_match_stream_plan(n_query, n_train, c->match_partial_bytes, &s_len, &s_chunks)) {
            {
                /* the kernel that reads the database exactly once: timed on its own (bench.py match_stream_roofline) */
                stage_timer t(c, "match_stream_kernel", (int64_t)n_query * 32 + (int64_t)n_train * 32 + (int64_t)s_chunks * n_query * 8);
                ssk_match_stream_kernel(c->stream, d_query, d_train, n_query, n_train, s_len, s_chunks, c->match_partial);
            }
            {
                stage_timer t(c, "match_stream_merge", (int64_t)s_chunks * n_query * 8 + (int64_t)n_query * 8);
                ssk_match_stream_merge(c->stream, c->match_partial, n_query, s_chunks, th, ratio_num, ratio_den, (int32_t *)d_idx,
                                       (uint16_t *)d_d1, (uint16_t *)d_d2);
            }
            HIP_TRY(c, hipGetLastError());
            return SS_OK;
        }
    }
    if (n_query >= SSK_MATCH_MFMA_MIN_QUERIES && n_train > 0 && !c->no_desc_x) {
        /* ONE matrix-core matcher for every entry point: the caller's packed rows are expanded to its operand format
         * (k_expand_desc: 32 -> 128 bytes per row) and k_match_mfma_x runs on them.  SENDSLAM_MATCH_PACKED=1 keeps round 1's
         * k_match_mfma, which expands every tile in every query block through an LDS table. */
        const bool same = d_train == d_query && n_train == n_query;
        int rc = grow(c, c->d_qx, c->d_qx_bytes, (size_t)SS_EXPANDED_BYTES(n_query));
        if (rc == SS_OK && !same) rc = grow(c, c->d_tx, c->d_tx_bytes, (size_t)SS_EXPANDED_BYTES(n_train));
        if (rc != SS_OK) return rc;
        {
            stage_timer t(c, "expand", ((int64_t)n_query + (same ? 0 : n_train)) * (32 + SSK_X_ROW));
            ssk_expand_desc(c->stream, d_query, n_query, c->d_qx);
            if (!same) ssk_expand_desc(c->stream, d_train, n_train, c->d_tx);
        }
        return match_expanded(c, c->d_qx, n_query, same ? c->d_qx : c->d_tx, n_train, th, ratio_num, ratio_den, exclude_self, d_idx, d_d1, d_d2,
                              (const uint8_t *)d_query, (const uint8_t *)d_train);
    }
    int chunk_len = 4;
    const int n_chunks = ssk_match_chunks(n_query, std::max(n_train, 1), 1, &chunk_len);
    if (n_chunks > 1) {
        int rc = grow(c, c->match_partial, c->match_partial_bytes, (size_t)n_chunks * n_query * SSK_MATCH_PARTIAL_BYTES);
        if (rc != SS_OK) return rc;
    }
    {
        stage_timer t(c, "match", (int64_t)n_query * 32 + (int64_t)n_train * 32 + (int64_t)n_query * 8);
        ssk_match(c->stream, d_query, d_train ? d_train : d_query, nullptr, nullptr, n_query, n_train, 0, 0, 0, chunk_len,
                  n_chunks, exclude_self ? 1 : 0, th, ratio_num, ratio_den, n_query, c->match_partial, (int32_t *)d_idx,
                  (uint16_t *)d_d1, (uint16_t *)d_d2, 1);
    }
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

int ss_match(ss_ctx *c, const uint8_t *query, int n_query, const uint8_t *train, int n_train, int th, int ratio_num,
             int ratio_den, int exclude_self, int32_t *idx, uint16_t *d1, uint16_t *d2)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_query < 0 || n_train < 0) return fail(c, SS_ERR_INVALID_ARG, "bad match arguments");
    if (n_query == 0) return SS_OK;
    if (!query || (!train && n_train > 0) || !idx || !d1 || !d2) return fail(c, SS_ERR_INVALID_ARG, "NULL match buffer");
    int rc = grow(c, c->d_mq, c->d_mq_bytes, (size_t)n_query * 32);
    if (rc == SS_OK) rc = grow(c, c->d_mt, c->d_mt_bytes, (size_t)std::max(n_train, 1) * 32);
    if (rc == SS_OK) rc = grow(c, c->d_mout, c->d_mout_bytes, (size_t)n_query * 8);
    if (rc != SS_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_mq, query, (size_t)n_query * 32, hipMemcpyHostToDevice, c->stream));
    if (n_train > 0) HIP_TRY(c, hipMemcpyAsync(c->d_mt, train, (size_t)n_train * 32, hipMemcpyHostToDevice, c->stream));
    int32_t *di = (int32_t *)c->d_mout;
    uint16_t *dd1 = (uint16_t *)(c->d_mout + (size_t)n_query * 4), *dd2 = (uint16_t *)(c->d_mout + (size_t)n_query * 6);
    rc = ss_match_device(c, c->d_mq, n_query, c->d_mt, n_train, th, ratio_num, ratio_den, exclude_self, di, dd1, dd2);
    if (rc != SS_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(idx, di, (size_t)n_query * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d1, dd1, (size_t)n_query * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(d2, dd2, (size_t)n_query * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

int ss_match_batch_device(ss_ctx *c, int mode, int th, int ratio_num, int ratio_den, void *d_idx, void *d_d1, void *d_d2)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!c->have_geom || c->last_n_frames <= 0) return fail(c, SS_ERR_STATE, "no batch has been extracted");
    if ((mode != 0 && mode != 1) || !d_idx || !d_d1 || !d_d2 || ratio_den <= 0) return fail(c, SS_ERR_INVALID_ARG, "bad match arguments");
    const int n = c->last_n_frames, kcap = c->hg.kcap;
    int chunk_len = 4;
    int n_chunks = ssk_match_chunks(kcap, kcap, n, &chunk_len);
    if (c->desc_x) n_chunks = ssk_match_x_batch_chunks(kcap, kcap, n, &chunk_len);
    if (n_chunks > 1 || c->desc_x) { /* the matrix-core matcher always writes partials: its second launch finishes them */
        int rc = grow(c, c->match_partial, c->match_partial_bytes, (size_t)n * n_chunks * kcap * SSK_MATCH_PARTIAL_BYTES);
        if (rc != SS_OK) return rc;
    }
#ifdef SS_TIMING_KNOBS
    const bool skip_match = ((c->n_extracts > c->skip_after ? c->skip_stages : 0) & 32) != 0;
#else
    const bool skip_match = false;
#endif
    if (!skip_match) {
        const int64_t nf = c->hg.n_features;
        stage_timer t(c, "match", (int64_t)n * (nf * 32 * 2 + nf * 8));
        if (c->desc_x)
            ssk_match_x(c->stream, c->desc_x, c->desc_x, c->n_kp, c->n_kp, 0, 0, (int64_t)kcap * SSK_X_ROW, (int64_t)kcap * SSK_X_ROW,
                        mode == 0 ? 0 : -1, chunk_len, n_chunks, mode == 0 ? 1 : 2, th, ratio_num, ratio_den, kcap,
                        c->match_partial, (int32_t *)d_idx, (uint16_t *)d_d1, (uint16_t *)d_d2, n, c->desc, c->desc, (int64_t)kcap * SS_DESC_BYTES,
                        (int64_t)kcap * SS_DESC_BYTES);
        else
            ssk_match(c->stream, c->desc, c->desc, c->n_kp, c->n_kp, 0, 0, (int64_t)kcap * 8, (int64_t)kcap * 8,
                      mode == 0 ? 0 : -1, chunk_len, n_chunks, mode == 0 ? 1 : 2, th, ratio_num, ratio_den, kcap,
                      c->match_partial, (int32_t *)d_idx, (uint16_t *)d_d1, (uint16_t *)d_d2, n);
    }
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

int ss_match_pairs_device(ss_ctx *c, const void *d_query, const void *d_n_query, const void *d_train, const void *d_n_train,
                          int n_frames, int rows_per_frame, int th, int ratio_num, int ratio_den, void *d_idx, void *d_d1,
                          void *d_d2)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_frames < 0 || rows_per_frame < 1 || ratio_den <= 0 || ratio_num < 0) return fail(c, SS_ERR_INVALID_ARG, "bad match arguments");
    if (n_frames == 0) return SS_OK;
    if (!d_query || !d_n_query || !d_train || !d_n_train || !d_idx || !d_d1 || !d_d2) return fail(c, SS_ERR_INVALID_ARG, "NULL match buffer");
    if (rows_per_frame >= SSK_MATCH_MFMA_MIN_QUERIES && !c->no_desc_x) {
        /* both sides expanded frame by frame ([n_frames][rows rounded up to 32][128 B]), then the batch matcher of the metric
         * path with per-frame counts on both sides */
        const int rows_alloc = (rows_per_frame + 31) & ~31;
        const size_t xb = (size_t)n_frames * rows_alloc * SSK_X_ROW;
        int rc = grow(c, c->d_qx, c->d_qx_bytes, xb);
        if (rc == SS_OK) rc = grow(c, c->d_tx, c->d_tx_bytes, xb);
        if (rc != SS_OK) return rc;
        int x_chunk = 32;
        const int x_chunks = ssk_match_x_batch_chunks(rows_per_frame, rows_per_frame, n_frames, &x_chunk);
        rc = grow(c, c->match_partial, c->match_partial_bytes, (size_t)n_frames * x_chunks * rows_per_frame * SSK_MATCH_PARTIAL_BYTES);
        if (rc != SS_OK) return rc;
        {
            stage_timer t(c, "expand", (int64_t)2 * n_frames * rows_per_frame * (32 + SSK_X_ROW));
            ssk_expand_desc_frames(c->stream, d_query, rows_per_frame, n_frames, c->d_qx);
            ssk_expand_desc_frames(c->stream, d_train, rows_per_frame, n_frames, c->d_tx);
        }
        {
            stage_timer t(c, "match", (int64_t)n_frames * rows_per_frame * (32 * 2 + 8));
            ssk_match_x(c->stream, c->d_qx, c->d_tx, (const int32_t *)d_n_query, (const int32_t *)d_n_train, 0, 0, (int64_t)rows_alloc * SSK_X_ROW,
                        (int64_t)rows_alloc * SSK_X_ROW, 0, x_chunk, x_chunks, 0, th, ratio_num, ratio_den, rows_per_frame, c->match_partial,
                        (int32_t *)d_idx, (uint16_t *)d_d1, (uint16_t *)d_d2, n_frames, (const uint8_t *)d_query, (const uint8_t *)d_train,
                        (int64_t)rows_per_frame * SS_DESC_BYTES, (int64_t)rows_per_frame * SS_DESC_BYTES);
        }
        HIP_TRY(c, hipGetLastError());
        return SS_OK;
    }
    int chunk_len = 4;
    const int n_chunks = ssk_match_chunks(rows_per_frame, rows_per_frame, n_frames, &chunk_len);
    if (n_chunks > 1) {
        int rc = grow(c, c->match_partial, c->match_partial_bytes, (size_t)n_frames * n_chunks * rows_per_frame * SSK_MATCH_PARTIAL_BYTES);
        if (rc != SS_OK) return rc;
    }
    {
        stage_timer t(c, "match", (int64_t)n_frames * rows_per_frame * (32 * 2 + 8));
        ssk_match(c->stream, d_query, d_train, (const int32_t *)d_n_query, (const int32_t *)d_n_train, 0, rows_per_frame,
                  (int64_t)rows_per_frame * 8, (int64_t)rows_per_frame * 8, 0, chunk_len, n_chunks, 0, th, ratio_num, ratio_den,
                  rows_per_frame, c->match_partial, (int32_t *)d_idx, (uint16_t *)d_d1, (uint16_t *)d_d2, n_frames);
    }
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

/* the pose half of the frame branch: device match against the initial / previous frame's descriptors, then the host
 * geometry (csrc/ss_track.cpp).  d_desc: n rows of 32 bytes in device memory, written on c->stream or complete. */
static int track_step(ss_ctx *c, int camera_id, double timestamp, const uint8_t *d_desc, const uint8_t *d_desc_x, const ss_keypoint *kps,
                      int n, ss_pose *out, const int32_t *given_idx = nullptr, const uint16_t *given_d1 = nullptr, int flags = 0)
{
    /* d_desc_x: the same n rows already expanded (the extraction's desc_x), or NULL: expanded here when the matrix-core matcher
     * is going to read them (as the query now, or as the next frames' train set).
     * given_idx / given_d1 (ss_track_features_matched): this frame's matches against the frame of the PREVIOUS call, made by
     * the caller's batch matcher; used when that frame is the one the tracker is about to match against, and then nothing
     * is enqueued on the device for this frame. */
    const bool use_x = !c->no_desc_x && n > 0;
    const int64_t serial = ++c->track_serial;
    auto expanded = [&]() -> int {
        if (!use_x || d_desc_x) return SS_OK;
        int rcx = grow(c, c->d_qx, c->d_qx_bytes, (size_t)SS_EXPANDED_BYTES(n));
        if (rcx != SS_OK) return rcx;
        stage_timer t(c, "expand", (int64_t)n * (32 + SSK_X_ROW));
        ssk_expand_desc(c->stream, d_desc, n, c->d_qx);
        d_desc_x = c->d_qx;
        return SS_OK;
    };
    sst_tracker &tr = c->tracker;
    tr.cam = sst_camera{c->cam.fx, c->cam.fy, c->cam.cx, c->cam.cy, c->cam.k1, c->cam.k2, c->cam.p1, c->cam.p2};
    tr.scale_factor = c->params.scale_factor;
    c->h_xy.resize((size_t)2 * std::max(n, 1));
    c->h_oct.resize((size_t)std::max(n, 1));
    c->h_midx.assign((size_t)std::max(n, 1), -1);
    c->h_md1.assign((size_t)std::max(n, 1), 0xFFFF);
    for (int i = 0; i < n; i++) {
        c->h_xy[2 * i] = kps[i].x;
        c->h_xy[2 * i + 1] = kps[i].y;
        c->h_oct[i] = kps[i].octave;
    }
    int rc;
    const int want = tr.want_match();
    const auto tm0 = std::chrono::steady_clock::now();
    const int32_t *m_idx = c->h_midx.data();
    const uint16_t *m_d1 = c->h_md1.data();
    if (want != SST_MATCH_NONE && n > 0) {
        const int64_t train_serial = want == SST_MATCH_REF ? c->ref_serial : c->prev_serial;
        if (given_idx && given_d1 && train_serial == serial - 1 && tr.n_train() > 0) {
            m_idx = given_idx;
            m_d1 = given_d1;
        } else {
            /* the previous frame's descriptors may still be the caller's (SS_TRACK_DESC_STAYS_VALID): packed rows only */
            const bool prev_ext = want == SST_MATCH_PREV && c->d_prev_ext != nullptr;
            const uint8_t *train = want == SST_MATCH_REF ? c->d_ref_desc : prev_ext ? c->d_prev_ext : c->d_prev_desc;
            const uint8_t *train_x = want == SST_MATCH_REF ? c->d_ref_desc_x : prev_ext ? nullptr : c->d_prev_desc_x;
            rc = grow(c, c->d_mout, c->d_mout_bytes, (size_t)n * 8);
            if (rc != SS_OK) return rc;
            int32_t *di = (int32_t *)c->d_mout;
            uint16_t *dd1 = (uint16_t *)(c->d_mout + (size_t)n * 4), *dd2 = (uint16_t *)(c->d_mout + (size_t)n * 6);
            if (use_x && train_x && n >= SSK_MATCH_MFMA_MIN_QUERIES && tr.n_train() > 0) { /* both operands are expanded already */
                rc = expanded();
                if (rc == SS_OK) rc = match_expanded(c, d_desc_x, n, train_x, tr.n_train(), SS_TH_LOW, 9, 10, 0, di, dd1, dd2, d_desc, train);
            } else {
                rc = ss_match_device(c, d_desc, n, train, tr.n_train(), SS_TH_LOW, 9, 10, 0, di, dd1, dd2);
            }
            if (rc != SS_OK) return rc;
            HIP_TRY(c, hipMemcpyAsync(c->h_midx.data(), di, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->h_md1.data(), dd1, (size_t)n * 2, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    sst_pose_out po;
    const auto tg0 = std::chrono::steady_clock::now();
    const int keep = tr.step(n, c->h_xy.data(), c->h_oct.data(), m_idx, m_d1, po);
    const auto tk0 = std::chrono::steady_clock::now();
    struct timing_guard {
        ss_ctx *c;
        std::chrono::steady_clock::time_point a, b, d;
        ~timing_guard()
        {
            if (!c->track_timing) return;
            const auto e = std::chrono::steady_clock::now();
            c->t_match += std::chrono::duration<double>(b - a).count();
            c->t_geom += std::chrono::duration<double>(d - b).count();
            c->t_keep += std::chrono::duration<double>(e - d).count();
            c->n_tracked++;
        }
    } tguard{c, tm0, tg0, tk0};
    if (keep == SST_KEEP_AS_PREV) {
        c->prev_serial = serial;
        c->d_prev_ext = nullptr;
    } else if (keep == SST_KEEP_AS_REF) {
        c->ref_serial = serial;
    }
    if (keep == SST_KEEP_AS_PREV && n > 0 && (flags & SS_TRACK_DESC_STAYS_VALID)) {
        c->d_prev_ext = d_desc; /* the caller keeps the rows alive until the next call has returned: nothing to copy */
    } else if (keep != SST_KEEP_NONE && n > 0) {
        uint8_t *&dst = keep == SST_KEEP_AS_REF ? c->d_ref_desc : c->d_prev_desc;
        size_t &dst_bytes = keep == SST_KEEP_AS_REF ? c->d_ref_desc_bytes : c->d_prev_desc_bytes;
        rc = grow(c, dst, dst_bytes, (size_t)n * SS_DESC_BYTES);
        if (rc != SS_OK) return rc;
        HIP_TRY(c, hipMemcpyAsync(dst, d_desc, (size_t)n * SS_DESC_BYTES, hipMemcpyDeviceToDevice, c->stream));
        if (use_x) {
            rc = expanded();
            if (rc != SS_OK) return rc;
            uint8_t *&dst_x = keep == SST_KEEP_AS_REF ? c->d_ref_desc_x : c->d_prev_desc_x;
            size_t &dst_x_bytes = keep == SST_KEEP_AS_REF ? c->d_ref_desc_x_bytes : c->d_prev_desc_x_bytes;
            rc = grow(c, dst_x, dst_x_bytes, (size_t)SS_EXPANDED_BYTES(n));
            if (rc != SS_OK) return rc;
            HIP_TRY(c, hipMemcpyAsync(dst_x, d_desc_x, (size_t)SS_EXPANDED_BYTES(n), hipMemcpyDeviceToDevice, c->stream));
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    out->tracking_state = po.state;
    out->camera_id = camera_id;
    out->timestamp = timestamp;
    for (int k = 0; k < 3; k++) out->position[k] = po.pos[k];
    for (int k = 0; k < 4; k++) out->quaternion[k] = po.quat[k];
    out->n_keypoints = n;
    out->n_matches = po.n_matches;
    out->n_inliers = po.n_inliers;
    out->n_map_points = po.n_map_points;
    return SS_OK;
}

int ss_track(ss_ctx *c, int camera_id, const uint8_t *pix, int width, int height, int channels, int row_stride,
             double timestamp, ss_pose *out)
{
    if (!c || !out) return SS_ERR_INVALID_ARG;
    if (!c->calibrated) return fail(c, SS_ERR_NOT_CALIBRATED, "Received frame before calibration. Ignoring.");
    ss_frame_result res;
    int rc = ss_extract(c, camera_id, pix, width, height, channels, row_stride, timestamp, &res);
    if (rc != SS_OK) return rc;
    /* this frame's descriptors are still in HBM (frame 0 of the batch arrays) */
    return track_step(c, camera_id, timestamp, c->desc, c->desc_x, res.keypoints, res.n_keypoints, out);
}

int ss_track_features(ss_ctx *c, int camera_id, double timestamp, const void *d_descriptors, const ss_keypoint *keypoints,
                      int n_keypoints, ss_pose *out)
{
    if (!c || !out || n_keypoints < 0) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!c->calibrated) return fail(c, SS_ERR_NOT_CALIBRATED, "Received frame before calibration. Ignoring.");
    if (camera_id == 0) return fail(c, SS_ERR_BAD_FRAME, "Frame message missing camera identifier.");
    if (n_keypoints > 0 && (!d_descriptors || !keypoints)) return fail(c, SS_ERR_INVALID_ARG, "ss_track_features: NULL feature arrays");
    return track_step(c, camera_id, timestamp, (const uint8_t *)d_descriptors, nullptr, keypoints, n_keypoints, out);
}

int ss_track_features_matched(ss_ctx *c, int camera_id, double timestamp, const void *d_descriptors, const ss_keypoint *keypoints,
                              int n_keypoints, const int32_t *match_idx, const uint16_t *match_d1, int flags, ss_pose *out)
{
    if (!c || !out || n_keypoints < 0) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!c->calibrated) return fail(c, SS_ERR_NOT_CALIBRATED, "Received frame before calibration. Ignoring.");
    if (camera_id == 0) return fail(c, SS_ERR_BAD_FRAME, "Frame message missing camera identifier.");
    if (n_keypoints > 0 && (!d_descriptors || !keypoints)) return fail(c, SS_ERR_INVALID_ARG, "ss_track_features_matched: NULL feature arrays");
    if ((match_idx == nullptr) != (match_d1 == nullptr)) return fail(c, SS_ERR_INVALID_ARG, "ss_track_features_matched: match_idx and match_d1 go together");
    if (flags & ~SS_TRACK_DESC_STAYS_VALID) return fail(c, SS_ERR_INVALID_ARG, "ss_track_features_matched: unknown flag");
    return track_step(c, camera_id, timestamp, (const uint8_t *)d_descriptors, nullptr, keypoints, n_keypoints, out, match_idx, match_d1, flags);
}

int ss_expand_descriptors_device(ss_ctx *c, const void *d_packed, int n, void *d_expanded)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n < 0) return fail(c, SS_ERR_INVALID_ARG, "bad descriptor count");
    if (n == 0) return SS_OK;
    if (!d_packed || !d_expanded) return fail(c, SS_ERR_INVALID_ARG, "NULL descriptor buffer");
    stage_timer t(c, "expand", (int64_t)n * (32 + SSK_X_ROW));
    ssk_expand_desc(c->stream, d_packed, n, d_expanded);
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

int ss_match_expanded_device(ss_ctx *c, const void *d_query_x, int n_query, const void *d_train_x, int n_train, int th, int ratio_num,
                             int ratio_den, int exclude_self, void *d_idx, void *d_d1, void *d_d2)
{
    return match_expanded(c, d_query_x, n_query, d_train_x, n_train, th, ratio_num, ratio_den, exclude_self, d_idx, d_d1, d_d2, nullptr, nullptr);
}

/* q_packed / t_packed: the same rows as packed descriptors when the caller has them (the finishing launch reads those) */
static int match_expanded(ss_ctx *c, const void *d_query_x, int n_query, const void *d_train_x, int n_train, int th, int ratio_num,
                          int ratio_den, int exclude_self, void *d_idx, void *d_d1, void *d_d2, const uint8_t *q_packed, const uint8_t *t_packed)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_query < 0 || n_train < 0 || ratio_den <= 0 || ratio_num < 0) return fail(c, SS_ERR_INVALID_ARG, "bad match arguments");
    if (n_query == 0) return SS_OK;
    if (!d_query_x || (!d_train_x && n_train > 0) || !d_idx || !d_d1 || !d_d2) return fail(c, SS_ERR_INVALID_ARG, "NULL match buffer");
    int chunk_len = 32;
    const int n_chunks = ssk_match_x_chunks(n_query, std::max(n_train, 1), &chunk_len);
    {
        int rc = grow(c, c->match_partial, c->match_partial_bytes, (size_t)n_chunks * n_query * SSK_MATCH_PARTIAL_BYTES);
        if (rc != SS_OK) return rc;
    }
    {
        stage_timer t(c, "match", (int64_t)n_query * SSK_X_ROW + (int64_t)n_train * SSK_X_ROW + (int64_t)n_query * 8);
        ssk_match_x_single(c->stream, (const uint8_t *)d_query_x, n_query, (const uint8_t *)(d_train_x ? d_train_x : d_query_x), n_train,
                           chunk_len, n_chunks, exclude_self, th, ratio_num, ratio_den, c->match_partial, (int32_t *)d_idx, (uint16_t *)d_d1,
                           (uint16_t *)d_d2, q_packed, t_packed);
    }
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

static int partial_common(ss_ctx *c, bool expanded, const void *d_query, int n_query, const void *d_train, int n_train,
                          int64_t row_offset, void *d_part)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_query < 0 || n_train < 0 || row_offset < 0 || row_offset + n_train > 0x7FFFFFFFll)
        return fail(c, SS_ERR_INVALID_ARG, "ss_match_partial_device: rows must fit 31 bits");
    if (n_query == 0) return SS_OK;
    if (!d_query || !d_part || (!d_train && n_train > 0)) return fail(c, SS_ERR_INVALID_ARG, "NULL match buffer");
    int rc = grow(c, c->d_part_tmp, c->d_part_tmp_bytes, (size_t)n_query * 8);
    if (rc != SS_OK) return rc;
    int32_t *di = (int32_t *)c->d_part_tmp;
    uint16_t *dd1 = (uint16_t *)(c->d_part_tmp + (size_t)n_query * 4), *dd2 = (uint16_t *)(c->d_part_tmp + (size_t)n_query * 6);
    rc = expanded ? ss_match_expanded_device(c, d_query, n_query, d_train, n_train, -1, 1, 1, 0, di, dd1, dd2)
                  : ss_match_device(c, d_query, n_query, d_train, n_train, -1, 1, 1, 0, di, dd1, dd2);
    if (rc != SS_OK) return rc;
    ssk_pack_partial(c->stream, di, dd1, dd2, n_query, (int32_t)row_offset, d_part);
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

int ss_match_partial_expanded_device(ss_ctx *c, const void *d_query_x, int n_query, const void *d_train_x, int n_train,
                                     int64_t row_offset, void *d_part)
{
    return partial_common(c, true, d_query_x, n_query, d_train_x, n_train, row_offset, d_part);
}

int ss_match_partial_device(ss_ctx *c, const void *d_query, int n_query, const void *d_train, int n_train, int64_t row_offset,
                            void *d_part)
{
    return partial_common(c, false, d_query, n_query, d_train, n_train, row_offset, d_part);
}

int ss_match_fold_device(ss_ctx *c, const void *d_parts, int n_parts, int n_query, int th, int ratio_num, int ratio_den,
                         void *d_idx, void *d_d1, void *d_d2)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_parts < 1 || n_query < 0 || ratio_den <= 0 || ratio_num < 0) return fail(c, SS_ERR_INVALID_ARG, "bad fold arguments");
    if (n_query == 0) return SS_OK;
    if (!d_parts || !d_idx || !d_d1 || !d_d2) return fail(c, SS_ERR_INVALID_ARG, "NULL fold buffer");
    stage_timer t(c, "match_fold", (int64_t)n_parts * n_query * 8 + (int64_t)n_query * 8);
    ssk_match_fold(c->stream, d_parts, n_parts, n_query, th, ratio_num, ratio_den, (int32_t *)d_idx, (uint16_t *)d_d1, (uint16_t *)d_d2);
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

int ss_match_fold_strided_device(ss_ctx *c, const void *d_parts, int n_parts, int64_t part_stride_bytes, int n_query, int th,
                                 int ratio_num, int ratio_den, void *d_idx, void *d_d1, void *d_d2)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n_parts < 1 || n_query < 0 || ratio_den <= 0 || ratio_num < 0 || part_stride_bytes % 8 != 0 || part_stride_bytes < (int64_t)n_query * 8)
        return fail(c, SS_ERR_INVALID_ARG, "bad fold arguments");
    if (n_query == 0) return SS_OK;
    if (!d_parts || !d_idx || !d_d1 || !d_d2) return fail(c, SS_ERR_INVALID_ARG, "NULL fold buffer");
    stage_timer t(c, "match_fold", (int64_t)n_parts * n_query * 8 + (int64_t)n_query * 8);
    ssk_match_fold_strided(c->stream, d_parts, part_stride_bytes, n_parts, n_query, th, ratio_num, ratio_den, (int32_t *)d_idx, (uint16_t *)d_d1,
                           (uint16_t *)d_d2);
    HIP_TRY(c, hipGetLastError());
    return SS_OK;
}

int ss_stereo_exchange_match(ss_ctx *c, ss_xchg *x, int peer_rank, int th, int ratio_num, int ratio_den, int32_t *idx, uint16_t *d1,
                             uint16_t *d2, int32_t *n_own, int32_t *n_peer)
{
    if (!c || !x) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!c->have_geom || c->last_n_frames <= 0) return fail(c, SS_ERR_STATE, "ss_stereo_exchange_match: no frame has been extracted");
    if (peer_rank < 0) return fail(c, SS_ERR_INVALID_ARG, "ss_stereo_exchange_match: bad peer rank");
    const int kcap = c->hg.kcap;
    const int64_t blk = (int64_t)kcap * SS_DESC_BYTES;
    const void *segs[2] = {c->desc, c->n_kp};
    const int64_t sizes[2] = {blk, (int64_t)sizeof(int32_t)};
    const void *gathered = nullptr;
    int64_t stride = 0;
    int rc = ss_xchg_allgather(x, c, segs, sizes, 2, &gathered, &stride);
    if (rc != SS_OK) return fail(c, rc, std::string("ss_stereo_exchange_match: ") + ss_xchg_last_error(x));
    rc = grow(c, c->d_mout, c->d_mout_bytes, (size_t)kcap * 8);
    if (rc != SS_OK) return rc;
    const uint8_t *peer = (const uint8_t *)gathered + (int64_t)peer_rank * stride;
    int32_t *di = (int32_t *)c->d_mout;
    uint16_t *dd1 = (uint16_t *)(c->d_mout + (size_t)kcap * 4), *dd2 = (uint16_t *)(c->d_mout + (size_t)kcap * 6);
    rc = ss_match_pairs_device(c, c->desc, c->n_kp, peer, peer + blk, 1, kcap, th, ratio_num, ratio_den, di, dd1, dd2);
    if (rc != SS_OK) return rc;
    int32_t counts[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&counts[0], c->n_kp, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&counts[1], peer + blk, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (idx) HIP_TRY(c, hipMemcpyAsync(idx, di, (size_t)kcap * 4, hipMemcpyDeviceToHost, c->stream));
    if (d1) HIP_TRY(c, hipMemcpyAsync(d1, dd1, (size_t)kcap * 2, hipMemcpyDeviceToHost, c->stream));
    if (d2) HIP_TRY(c, hipMemcpyAsync(d2, dd2, (size_t)kcap * 2, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = ss_xchg_status(x);
    if (rc != SS_OK) return fail(c, rc, std::string("ss_stereo_exchange_match: ") + ss_xchg_last_error(x));
    if (n_own) *n_own = counts[0];
    if (n_peer) *n_peer = counts[1];
    return SS_OK;
}

int ss_wait_stream(ss_ctx *c, void *hip_stream)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    hipEvent_t e = nullptr;
    HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipError_t r = hipEventRecord(e, (hipStream_t)hip_stream);
    if (r == hipSuccess) r = hipStreamWaitEvent(c->stream, e, 0);
    (void)hipEventDestroy(e); /* destruction is deferred until the event has completed */
    if (r != hipSuccess) return fail(c, SS_ERR_HIP, std::string("ss_wait_stream: ") + hipGetErrorString(r));
    return SS_OK;
}

int ss_track_reset(ss_ctx *c)
{
    if (!c) return SS_ERR_INVALID_ARG;
    c->tracker.reset();
    c->prev_serial = c->ref_serial = -1;
    c->d_prev_ext = nullptr;
    return SS_OK;
}

int ss_synchronize(ss_ctx *c)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (c->have_geom && c->last_n_frames > 0) return check_frame_errors(c);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}

int ss_get_stream(ss_ctx *c, void **hip_stream)
{
    if (!c || !hip_stream) return SS_ERR_INVALID_ARG;
    *hip_stream = (void *)c->stream;
    return SS_OK;
}

int ss_profile_enable(ss_ctx *c, int on)
{
    if (!c) return SS_ERR_INVALID_ARG;
    c->profile = on != 0;
    return SS_OK;
}

int ss_profile_reset(ss_ctx *c)
{
    if (!c) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    collect_events(c);
    for (auto &s : c->stages) s.ms.clear();
    return SS_OK;
}

int ss_stats(ss_ctx *c, ss_stage_stats *out, int max_stages)
{
    if (!c || (!out && max_stages > 0)) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    collect_events(c);
    int n = 0;
    for (auto &s : c->stages) {
        if (n < max_stages) {
            ss_stage_stats &o = out[n];
            memset(&o, 0, sizeof(o));
            snprintf(o.name, sizeof(o.name), "%s", s.name.c_str());
            o.launches = (int64_t)s.ms.size();
            o.algorithmic_bytes = s.bytes;
            if (!s.ms.empty()) {
                std::vector<float> v = s.ms;
                std::sort(v.begin(), v.end());
                double tot = 0;
                for (float x : v) tot += x;
                o.total_ms = tot;
                o.mean_ms = tot / v.size();
                o.median_ms = v[v.size() / 2]; /* the shim's median rule, :661 */
            }
        }
        n++;
    }
    return n;
}

int ss_debug_fetch(ss_ctx *c, int what, int frame, int level, void *dst, int64_t dst_bytes)
{
    if (!c || !dst) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (!c->have_geom || frame < 0 || frame >= c->last_n_frames || level < 0 || level >= c->hg.n_levels)
        return fail(c, SS_ERR_INVALID_ARG, "ss_debug_fetch: no such frame / level");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const ss_geom &g = c->hg;
    const ss_level &L = g.lv[level];
    if (what >= 0 && what <= 2) {
        if (what == 2 && !c->score) {
            /* the response map is not kept in normal operation: allocate it and run the FAST kernel
             * again on the pyramid of the last batch (same kernel, same outputs, plus the map);
             * from now on this context keeps it */
            HIP_TRY(c, hipMalloc((void **)&c->score, (size_t)c->params.max_batch * g.block_bytes));
            ssk_fast_blur_nms(c->stream, c->pyr, c->score, c->blur, c->dg, g, c->d_tiles2, c->d_cinfo, c->tsurv,
                              c->thdr, c->state, c->last_n_frames, c->last_lvl0);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        const uint8_t *base = what == 0 ? c->pyr : what == 1 ? c->blur : c->score;
        const int64_t need = (int64_t)L.w * L.h;
        if (dst_bytes < need) return fail(c, SS_ERR_INVALID_ARG, "ss_debug_fetch: dst too small");
        if (what == 0 && level == 0 && c->last_lvl0.ptr) { /* level 0 was read in place from the caller's buffer */
            HIP_TRY(c, hipMemcpy2D(dst, (size_t)L.w, c->last_lvl0.ptr + (int64_t)frame * c->last_lvl0.frame_stride,
                                   (size_t)c->last_lvl0.pitch, (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
            return (int)need;
        }
        HIP_TRY(c, hipMemcpy2D(dst, (size_t)L.w, base + (size_t)frame * g.block_bytes + L.off, (size_t)L.pitch,
                               (size_t)L.w, (size_t)L.h, hipMemcpyDeviceToHost));
        return (int)need;
    }
    if (what == 3 || what == 4) {
        ss_level_state st;
        HIP_TRY(c, hipMemcpy(&st, c->state + (size_t)frame * SS_MAX_LEVELS + level, sizeof(st), hipMemcpyDeviceToHost));
        const int n = what == 3 ? st.n_cand : st.n_sel;
        if (dst_bytes < (int64_t)n * 12) return fail(c, SS_ERR_INVALID_ARG, "ss_debug_fetch: dst too small");
        std::vector<uint32_t> packed((size_t)std::max(n, 1));
        const uint32_t *src = what == 3 ? c->cand + (size_t)frame * g.cand_total + L.cand_base
                                        : c->sel + (size_t)frame * g.sel_total + L.sel_base;
        if (n > 0) HIP_TRY(c, hipMemcpy(packed.data(), src, (size_t)n * 4, hipMemcpyDeviceToHost));
        int32_t *o = (int32_t *)dst;
        for (int i = 0; i < n; i++) {
            o[3 * i] = SS_PX(packed[i]);
            o[3 * i + 1] = SS_PY(packed[i]);
            o[3 * i + 2] = SS_PR(packed[i]);
        }
        return n * 12;
    }
    return fail(c, SS_ERR_INVALID_ARG, "ss_debug_fetch: unknown selector");
}

} /* extern "C" */

extern "C" int ss_debug_sort(ss_ctx *c, uint64_t *items, int n)
{
    if (!c || (!items && n > 0)) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(c->device);
    if (n == 0) return SS_OK;
    int rc = grow(c, c->d_mq, c->d_mq_bytes, (size_t)n * 8);
    if (rc != SS_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->d_mq, items, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if (ssk_debug_sort(c->stream, (uint64_t *)c->d_mq, n) != 0) return fail(c, SS_ERR_INVALID_ARG, "ss_debug_sort: n > 2048");
    HIP_TRY(c, hipMemcpyAsync(items, c->d_mq, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SS_OK;
}
