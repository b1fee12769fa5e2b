/*
 * ss_pipe.cpp -- the pipelined host-memory path of the C ABI (include/sendslam_orb.h, ss_pipe_*).
 *
 * What it replaces in the reference: the per-frame loop of the backend shim
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:521-627) -- payload copy :325, imdecode :546,
 * TrackMonocular :594 -- and, on the host side, slam_handler.ex:59-88, which hands the shim ONE frame at a time over
 * loopback TCP.  Here a STREAM of host frames goes through a ring of pinned slots: while the kernels of one batch
 * run, the next batch crosses PCIe (H2D) and the previous one's keypoints / descriptors / matches come back (D2H).
 *
 * Built on the public entry points only (ss_create, ss_extract_batch_device, ss_match_batch_device,
 * ss_get_batch_view, ss_get_stream) plus the HIP runtime for pinned memory, copies and events: every slot owns an
 * extraction context, so slots share nothing and need no ordering between them.  Per batch:
 *     H2D copy (pinned -> HBM) on the pipe's ONE upload stream, in submission order
 *     -> event -> the slot's context stream: extraction kernels (level 0 read in place) | match | D2H copies | event
 * One upload stream on purpose: uploads issued on the slots' own streams run CONCURRENTLY, share the link and all
 * finish together (measured: three 59 MB copies took 2.7 / 1.9 / 1.4 ms side by side against 1.04 ms alone), so the
 * kernels of all batches in flight start together and the chip idles while the next convoy of copies crosses PCIe.
 * In order, copy i + 1 crosses the link while the kernels of batch i run.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sendslam_orb.h"

namespace {

thread_local std::string g_pipe_create_error;

/* a few host threads that gather caller-owned frames into a pinned slot (one memcpy thread moves ~10 GB/s; the PCIe
 * link wants 25-50) */
class copy_pool {
public:
    explicit copy_pool(int n)
    {
        for (int i = 0; i < n; i++) workers.emplace_back([this] { run(); });
    }
    ~copy_pool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
    /* runs job(i) for i in [0, n) on the pool and the calling thread; returns when all are done */
    void parallel_for(int n, const std::function<void(int)> &job)
    {
        if (n <= 0) return;
        {
            std::lock_guard<std::mutex> g(m);
            fn = &job;
            next = 0;
            total = n;
            pending = n;
        }
        cv.notify_all();
        work();
        std::unique_lock<std::mutex> l(m);
        done_cv.wait(l, [this] { return pending == 0; });
        fn = nullptr;
    }

private:
    void work()
    {
        for (;;) {
            int i;
            const std::function<void(int)> *f;
            {
                std::lock_guard<std::mutex> g(m);
                if (!fn || next >= total) return;
                i = next++;
                f = fn;
            }
            (*f)(i);
            {
                std::lock_guard<std::mutex> g(m);
                if (--pending == 0) done_cv.notify_all();
            }
        }
    }
    void run()
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [this] { return stop || (fn && next < total); });
                if (stop) return;
            }
            work();
        }
    }
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv, done_cv;
    const std::function<void(int)> *fn = nullptr;
    int next = 0, total = 0, pending = 0;
    bool stop = false;
};

enum slot_state { SLOT_FREE = 0, SLOT_ACQUIRED, SLOT_IN_FLIGHT, SLOT_RETURNED };

struct pipe_slot {
    int state = SLOT_FREE;
    ss_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr, uploaded = nullptr;
    uint8_t *h_pix = nullptr; /* pinned */
    uint8_t *d_pix = nullptr;
    uint8_t *h_res = nullptr; /* pinned: one block, carved below */
    uint8_t *d_match = nullptr;
    int n_frames = 0;
    uint64_t sequence = 0;
    std::vector<int32_t> status, camera_id;
    std::vector<double> timestamp;
    /* device result arrays of the slot's context */
    ss_batch_view view{};
    /* carved host result arrays */
    int32_t *h_nkp = nullptr, *h_levels = nullptr, *h_err = nullptr, *h_midx = nullptr;
    ss_keypoint *h_kps = nullptr;
    uint8_t *h_desc = nullptr;
    uint16_t *h_md1 = nullptr, *h_md2 = nullptr;
};

} // namespace

struct ss_pipe {
    int device = 0;
    ss_pipe_config cfg{};
    int64_t row_stride = 0, frame_stride = 0;
    int kcap = 0;
    std::vector<pipe_slot> slots;
    hipStream_t upload = nullptr; /* every H2D copy, in submission order */
    std::deque<int> in_flight; /* slot ids in submission order */
    uint64_t next_sequence = 0;
    mutable std::mutex m;
    /* last error of the producer calls (acquire, submit*) and of the consumer calls (wait, poll, release): one thread each may
     * use the pipe at the same time, so each side owns its string; ss_pipe_last_error returns the younger one */
    std::string err_producer, err_consumer;
    std::atomic<int> err_side{0}; /* 0 none, 1 producer, 2 consumer */
    bool broken = false;        /* a submission failed half-way and its streams could not be drained */
    int inject_fail_after = -1; /* test hook (ss_pipe_debug_inject_failure): the next submission fails after this many enqueued operations */
    copy_pool *pool = nullptr;
};

namespace {

enum { SIDE_PRODUCER = 1, SIDE_CONSUMER = 2 };

int pfail(ss_pipe *p, int code, const std::string &msg, int side = SIDE_PRODUCER)
{
    if (!p) {
        g_pipe_create_error = msg;
        return code;
    }
    (side == SIDE_CONSUMER ? p->err_consumer : p->err_producer) = msg;
    p->err_side.store(side, std::memory_order_release);
    return code;
}

#define PIPE_HIP_SIDE(p, call, side)                                                                            \
    do {                                                                                                        \
        hipError_t e_ = (call);                                                                                 \
        if (e_ != hipSuccess)                                                                                   \
            return pfail((p), e_ == hipErrorOutOfMemory ? SS_ERR_NO_MEMORY : SS_ERR_HIP,                        \
                         std::string(#call) + ": " + hipGetErrorString(e_), (side));                            \
    } while (0)
#define PIPE_HIP(p, call) PIPE_HIP_SIDE(p, call, SIDE_PRODUCER)
/* an enqueue of a submission: counts towards the injected failure of the test hook */
#define PIPE_ENQ(p, call)                                                                                       \
    do {                                                                                                        \
        if ((p)->inject_fail_after == 0) {                                                                      \
            (p)->inject_fail_after = -1;                                                                        \
            return pfail((p), SS_ERR_HIP, "injected failure before " #call);                                    \
        }                                                                                                       \
        if ((p)->inject_fail_after > 0) (p)->inject_fail_after--;                                               \
        PIPE_HIP(p, call);                                                                                      \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

void free_slot(pipe_slot &s)
{
    if (s.ctx) {
        (void)ss_synchronize(s.ctx);
        (void)ss_destroy(s.ctx);
    }
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.uploaded) (void)hipEventDestroy(s.uploaded);
    if (s.h_pix) (void)hipHostFree(s.h_pix);
    if (s.h_res) (void)hipHostFree(s.h_res);
    if (s.d_pix) (void)hipFree(s.d_pix);
    if (s.d_match) (void)hipFree(s.d_match);
    s = pipe_slot();
}

void fill_result(ss_pipe *p, pipe_slot &s, int id, ss_pipe_result *out)
{
    /* a frame's own status: what the producer flagged, else what the kernels reported */
    for (int i = 0; i < s.n_frames; i++) {
        if (s.status[i] == SS_OK && s.h_err[i] != 0) s.status[i] = SS_ERR_OVERFLOW;
        if (s.status[i] != SS_OK) {
            s.h_nkp[i] = 0;
            if (s.h_midx)
                for (int k = 0; k < p->kcap; k++) s.h_midx[(size_t)i * p->kcap + k] = -1;
        }
    }
    /* match_mode 1: frame i was matched against frame i - 1's rows; when that frame is bad its rows are not reported, so
     * frame i's indices would point nowhere */
    if (p->cfg.match_mode == 1 && s.h_midx)
        for (int i = 1; i < s.n_frames; i++)
            if (s.status[i - 1] != SS_OK)
                for (int k = 0; k < p->kcap; k++) s.h_midx[(size_t)i * p->kcap + k] = -1;
    out->slot = id;
    out->n_frames = s.n_frames;
    out->kp_capacity = p->kcap;
    out->sequence = s.sequence;
    out->status = s.status.data();
    out->camera_id = s.camera_id.data();
    out->timestamp = s.timestamp.data();
    out->n_keypoints = s.h_nkp;
    out->level_counts = s.h_levels;
    out->keypoints = s.h_kps;
    out->descriptors = s.h_desc;
    out->match_idx = s.h_midx;
    out->match_d1 = s.h_md1;
    out->match_d2 = s.h_md2;
    out->d_descriptors = s.view.descriptors;
}

} // namespace

extern "C" {

const char *ss_pipe_last_error(const ss_pipe *p)
{
    if (!p) return g_pipe_create_error.c_str();
    return p->err_side.load(std::memory_order_acquire) == SIDE_CONSUMER ? p->err_consumer.c_str() : p->err_producer.c_str();
}

int ss_pipe_debug_inject_failure(ss_pipe *p, int after_operations)
{
    if (!p) return SS_ERR_INVALID_ARG;
    p->inject_fail_after = after_operations;
    return SS_OK;
}

int ss_pipe_destroy(ss_pipe *p)
{
    if (!p) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(p->device);
    delete p->pool;
    for (auto &s : p->slots) free_slot(s);
    if (p->upload) (void)hipStreamDestroy(p->upload);
    delete p;
    return SS_OK;
}

int ss_pipe_create(int device_ordinal, const ss_orb_params *params, const ss_camera *cam, const ss_pipe_config *cfg,
                   ss_pipe **out)
{
    if (!out) return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: out is NULL");
    *out = nullptr;
    if (!cfg) return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: cfg is NULL");
    ss_pipe_config c = *cfg;
    if (c.width <= 0 || c.height <= 0 || (c.channels != 1 && c.channels != 3 && c.channels != 4))
        return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: bad frame shape");
    if (c.batch < 1 || c.batch > 256) return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: batch out of range (1..256)");
    if (c.depth < 2 || c.depth > 16) return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: depth out of range (2..16)");
    if (c.match_mode < -1 || c.match_mode > 1) return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: bad match_mode");
    if (c.channels != 1 && !cam) return pfail(nullptr, SS_ERR_NOT_CALIBRATED, "ss_pipe_create: colour frames need the calibration (rgb flag)");
    if (c.match_th == 0 && c.ratio_num == 0 && c.ratio_den == 0) {
        c.match_th = 50;
        c.ratio_num = 9;
        c.ratio_den = 10;
    }
    if (c.ratio_den <= 0 || c.ratio_num < 0) return pfail(nullptr, SS_ERR_INVALID_ARG, "ss_pipe_create: bad ratio");
    if (c.copy_threads <= 0) c.copy_threads = 4;
    if (c.copy_threads > 64) c.copy_threads = 64;

    ss_orb_params prm;
    ss_orb_params_default(&prm);
    if (params) prm = *params;
    prm.max_batch = c.batch;

    ss_pipe *p = new ss_pipe();
    p->device = device_ordinal;
    p->cfg = c;
    /* rows padded to 16 bytes: a 1-channel slot IS pyramid level 0 for the kernels (no ingest copy) */
    p->row_stride = (int64_t)align_up((size_t)c.width * c.channels, 16);
    p->frame_stride = p->row_stride * c.height;
    p->slots.resize((size_t)c.depth);
    const size_t pix_bytes = (size_t)p->frame_stride * c.batch;

    auto bail = [&](int code) {
        const std::string msg = p->err_producer;
        ss_pipe_destroy(p);
        g_pipe_create_error = msg;
        return code;
    };
    (void)hipSetDevice(device_ordinal);
    for (int i = 0; i < c.depth; i++) {
        pipe_slot &s = p->slots[(size_t)i];
        int rc = ss_create(device_ordinal, &prm, &s.ctx);
        if (rc != SS_OK) {
            p->err_producer = ss_last_error(nullptr);
            return bail(rc);
        }
        if (cam) {
            rc = ss_set_calibration(s.ctx, 1, cam);
            if (rc != SS_OK) {
                p->err_producer = ss_last_error(s.ctx);
                return bail(rc);
            }
        }
        void *st = nullptr;
        ss_get_stream(s.ctx, &st);
        s.stream = (hipStream_t)st;
        hipError_t e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming);
        if (e == hipSuccess && !p->upload) e = hipStreamCreateWithFlags(&p->upload, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipHostMalloc((void **)&s.h_pix, pix_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&s.d_pix, pix_bytes);
        if (e == hipSuccess) e = hipMemsetAsync(s.d_pix, 0, pix_bytes, s.stream);
        if (e != hipSuccess) {
            p->err_producer = std::string("ss_pipe_create: ") + hipGetErrorString(e);
            return bail(e == hipErrorOutOfMemory ? SS_ERR_NO_MEMORY : SS_ERR_HIP);
        }
        memset(s.h_pix, 0, pix_bytes);
        /* one blank batch through the slot's context: allocates its HBM buffers, loads the kernels, and tells us
         * kp_capacity (fixed by the geometry) before the first real frame arrives */
        rc = ss_extract_batch_device(s.ctx, s.d_pix, c.batch, c.width, c.height, c.channels, p->row_stride, p->frame_stride);
        if (rc == SS_OK) rc = ss_synchronize(s.ctx);
        if (rc == SS_OK) rc = ss_get_batch_view(s.ctx, &s.view);
        if (rc != SS_OK) {
            p->err_producer = ss_last_error(s.ctx);
            return bail(rc);
        }
        p->kcap = s.view.kp_capacity;
        const size_t B = (size_t)c.batch, K = (size_t)p->kcap;
        /* pinned result block */
        size_t off = 0;
        auto carve = [&](size_t bytes) {
            const size_t o = off;
            off = align_up(off + bytes, 64);
            return o;
        };
        const size_t o_nkp = carve(B * 4), o_lv = carve(B * SS_MAX_LEVELS * 4), o_err = carve(B * 4);
        const size_t o_kps = carve(B * K * sizeof(ss_keypoint)), o_desc = carve(B * K * SS_DESC_BYTES);
        const size_t o_mi = carve(B * K * 4), o_m1 = carve(B * K * 2), o_m2 = carve(B * K * 2);
        e = hipHostMalloc((void **)&s.h_res, off, hipHostMallocDefault);
        if (e == hipSuccess && c.match_mode >= 0) e = hipMalloc((void **)&s.d_match, B * K * 8);
        if (e != hipSuccess) {
            p->err_producer = std::string("ss_pipe_create: ") + hipGetErrorString(e);
            return bail(e == hipErrorOutOfMemory ? SS_ERR_NO_MEMORY : SS_ERR_HIP);
        }
        memset(s.h_res, 0, off);
        s.h_nkp = (int32_t *)(s.h_res + o_nkp);
        s.h_levels = (int32_t *)(s.h_res + o_lv);
        s.h_err = (int32_t *)(s.h_res + o_err);
        s.h_kps = (ss_keypoint *)(s.h_res + o_kps);
        s.h_desc = s.h_res + o_desc;
        if (c.match_mode >= 0) {
            s.h_midx = (int32_t *)(s.h_res + o_mi);
            s.h_md1 = (uint16_t *)(s.h_res + o_m1);
            s.h_md2 = (uint16_t *)(s.h_res + o_m2);
            /* one match of the blank batch: the context allocates its chunk-partial buffer now, not inside the first
             * real submission */
            rc = ss_match_batch_device(s.ctx, c.match_mode, c.match_th, c.ratio_num, c.ratio_den, s.d_match, s.d_match + B * K * 4,
                                       s.d_match + B * K * 6);
            if (rc == SS_OK) rc = ss_synchronize(s.ctx);
            if (rc != SS_OK) {
                p->err_producer = ss_last_error(s.ctx);
                return bail(rc);
            }
        }
        s.status.assign(B, SS_OK);
        s.camera_id.assign(B, 1);
        s.timestamp.assign(B, 0.0);
    }
    p->pool = new copy_pool(c.copy_threads - 1); /* the submitting thread copies too */
    *out = p;
    return SS_OK;
}

int ss_pipe_acquire(ss_pipe *p, ss_pipe_slot *out)
{
    if (!p || !out) return SS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(p->m);
    for (size_t i = 0; i < p->slots.size(); i++) {
        pipe_slot &s = p->slots[i];
        if (s.state != SLOT_FREE) continue;
        s.state = SLOT_ACQUIRED;
        out->slot = (int)i;
        out->pixels = s.h_pix;
        out->row_stride = p->row_stride;
        out->frame_stride = p->frame_stride;
        return SS_OK;
    }
    /* polled in a loop by producers: no message is formatted (ss_pipe_last_error is not updated for SS_ERR_BUSY) */
    return p->broken ? pfail(p, SS_ERR_STATE, "ss_pipe: a failed submission could not be drained; destroy the pipe") : SS_ERR_BUSY;
}

/* the enqueues of one submission, on the upload stream and on the slot's stream */
static int enqueue_batch(ss_pipe *p, pipe_slot &s, int n)
{
    const ss_pipe_config &c = p->cfg;
    const size_t K = (size_t)p->kcap, N = (size_t)n;
    PIPE_ENQ(p, hipMemcpyAsync(s.d_pix, s.h_pix, (size_t)p->frame_stride * N, hipMemcpyHostToDevice, p->upload));
    PIPE_ENQ(p, hipEventRecord(s.uploaded, p->upload));
    PIPE_ENQ(p, hipStreamWaitEvent(s.stream, s.uploaded, 0));
    if (p->inject_fail_after == 0) {
        p->inject_fail_after = -1;
        return pfail(p, SS_ERR_HIP, "injected failure before ss_extract_batch_device");
    }
    if (p->inject_fail_after > 0) p->inject_fail_after--;
    int rc = ss_extract_batch_device(s.ctx, s.d_pix, n, c.width, c.height, c.channels, p->row_stride, p->frame_stride);
    if (rc != SS_OK) return pfail(p, rc, ss_last_error(s.ctx));
    if (c.match_mode >= 0) {
        uint8_t *dm = s.d_match;
        const size_t B = (size_t)c.batch;
        rc = ss_match_batch_device(s.ctx, c.match_mode, c.match_th, c.ratio_num, c.ratio_den, dm, dm + B * K * 4, dm + B * K * 6);
        if (rc != SS_OK) return pfail(p, rc, ss_last_error(s.ctx));
        PIPE_ENQ(p, hipMemcpyAsync(s.h_midx, dm, N * K * 4, hipMemcpyDeviceToHost, s.stream));
        PIPE_ENQ(p, hipMemcpyAsync(s.h_md1, dm + B * K * 4, N * K * 2, hipMemcpyDeviceToHost, s.stream));
        PIPE_ENQ(p, hipMemcpyAsync(s.h_md2, dm + B * K * 6, N * K * 2, hipMemcpyDeviceToHost, s.stream));
    }
    PIPE_ENQ(p, hipMemcpyAsync(s.h_nkp, s.view.n_keypoints, N * 4, hipMemcpyDeviceToHost, s.stream));
    PIPE_ENQ(p, hipMemcpyAsync(s.h_levels, s.view.level_counts, N * SS_MAX_LEVELS * 4, hipMemcpyDeviceToHost, s.stream));
    PIPE_ENQ(p, hipMemcpyAsync(s.h_err, s.view.frame_error, N * 4, hipMemcpyDeviceToHost, s.stream));
    PIPE_ENQ(p, hipMemcpyAsync(s.h_kps, s.view.keypoints, N * K * sizeof(ss_keypoint), hipMemcpyDeviceToHost, s.stream));
    PIPE_ENQ(p, hipMemcpyAsync(s.h_desc, s.view.descriptors, N * K * SS_DESC_BYTES, hipMemcpyDeviceToHost, s.stream));
    PIPE_ENQ(p, hipEventRecord(s.done, s.stream));
    return SS_OK;
}

/* caller holds no lock; slot must be ACQUIRED.  status[] may already carry producer-side errors. */
static int submit_locked(ss_pipe *p, int slot, int n, const int32_t *camera_ids, const double *timestamps, bool keep_status)
{
    if (slot < 0 || slot >= (int)p->slots.size()) return pfail(p, SS_ERR_INVALID_ARG, "ss_pipe_submit: no such slot");
    pipe_slot &s = p->slots[(size_t)slot];
    {
        std::lock_guard<std::mutex> g(p->m);
        if (p->broken) return pfail(p, SS_ERR_STATE, "ss_pipe: a failed submission could not be drained; destroy the pipe");
        if (s.state != SLOT_ACQUIRED) return pfail(p, SS_ERR_STATE, "ss_pipe_submit: slot was not acquired");
    }
    if (n < 1 || n > p->cfg.batch) return pfail(p, SS_ERR_INVALID_ARG, "ss_pipe_submit: n_frames out of range (1..batch)");
    (void)hipSetDevice(p->device);
    for (int i = 0; i < n; i++) {
        if (!keep_status) s.status[(size_t)i] = SS_OK;
        s.camera_id[(size_t)i] = camera_ids ? camera_ids[i] : 1;
        s.timestamp[(size_t)i] = timestamps ? timestamps[i] : 0.0;
        /* the shim skips a frame without a camera identifier (:528); here it costs the frame, not the batch */
        if (s.camera_id[(size_t)i] == 0) s.status[(size_t)i] = SS_ERR_BAD_FRAME;
    }
    s.n_frames = n;
    const int rc = enqueue_batch(p, s, n);
    if (rc != SS_OK) {
        /* part of the batch may be enqueued (the upload, kernels, some result copies) and still use h_pix, d_pix and h_res:
         * both streams are drained before the slot can be handed out again; if even that fails the pipe refuses further
         * work.  The slot stays ACQUIRED: the caller releases or resubmits it. */
        const std::string first = p->err_producer;
        const hipError_t e1 = hipStreamSynchronize(p->upload), e2 = hipStreamSynchronize(s.stream);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            std::lock_guard<std::mutex> g(p->m);
            p->broken = true;
        }
        (void)hipGetLastError();
        return pfail(p, rc, first);
    }
    {
        std::lock_guard<std::mutex> g(p->m);
        s.state = SLOT_IN_FLIGHT;
        s.sequence = p->next_sequence++;
        p->in_flight.push_back(slot);
    }
    return SS_OK;
}

int ss_pipe_submit(ss_pipe *p, int slot, int n_frames, const int32_t *camera_ids, const double *timestamps)
{
    if (!p) return SS_ERR_INVALID_ARG;
    return submit_locked(p, slot, n_frames, camera_ids, timestamps, false);
}

int ss_pipe_submit_frames(ss_pipe *p, const uint8_t *const *frames, int n_frames, int64_t row_stride,
                          const int32_t *camera_ids, const double *timestamps)
{
    if (!p || !frames) return SS_ERR_INVALID_ARG;
    const ss_pipe_config &c = p->cfg;
    if (n_frames < 1 || n_frames > c.batch) return pfail(p, SS_ERR_INVALID_ARG, "ss_pipe_submit_frames: n_frames out of range (1..batch)");
    const int64_t tight = (int64_t)c.width * c.channels;
    if (row_stride < tight) return pfail(p, SS_ERR_BAD_FRAME, "row_stride smaller than a row");
    ss_pipe_slot sl;
    int rc = ss_pipe_acquire(p, &sl);
    if (rc != SS_OK) return rc;
    pipe_slot &s = p->slots[(size_t)sl.slot];
    for (int i = 0; i < n_frames; i++) s.status[(size_t)i] = frames[i] ? SS_OK : SS_ERR_BAD_FRAME;
    /* gather: frame i in `parts` pieces of rows so that a handful of frames still spreads over the threads */
    const int parts = n_frames >= 2 * c.copy_threads ? 1 : (2 * c.copy_threads + n_frames - 1) / n_frames;
    const int rows_per = (c.height + parts - 1) / parts;
    const std::function<void(int)> job = [&](int k) {
        const int i = k / parts, y0 = (k % parts) * rows_per, y1 = std::min(c.height, y0 + rows_per);
        uint8_t *dst = sl.pixels + (size_t)i * (size_t)p->frame_stride;
        if (!frames[i]) {
            if (y0 < y1) memset(dst + (size_t)y0 * (size_t)p->row_stride, 0, (size_t)(y1 - y0) * (size_t)p->row_stride);
            return;
        }
        if (row_stride == p->row_stride) {
            /* the caller's last row may end after width * channels bytes */
            const size_t bytes = (size_t)(y1 - y0 - 1) * (size_t)row_stride + (size_t)tight;
            if (y0 < y1) memcpy(dst + (size_t)y0 * (size_t)row_stride, frames[i] + (size_t)y0 * (size_t)row_stride, bytes);
        } else {
            for (int y = y0; y < y1; y++) memcpy(dst + (size_t)y * (size_t)p->row_stride, frames[i] + (size_t)y * (size_t)row_stride, (size_t)tight);
        }
    };
    p->pool->parallel_for(n_frames * parts, job);
    rc = submit_locked(p, sl.slot, n_frames, camera_ids, timestamps, true);
    if (rc != SS_OK) {
        std::lock_guard<std::mutex> g(p->m);
        s.state = SLOT_FREE;
    }
    return rc;
}

static int take_oldest(ss_pipe *p, bool block, ss_pipe_result *out)
{
    if (!p || !out) return SS_ERR_INVALID_ARG;
    int id;
    {
        std::lock_guard<std::mutex> g(p->m);
        if (p->in_flight.empty()) return block ? pfail(p, SS_ERR_STATE, "ss_pipe_wait: nothing has been submitted", SIDE_CONSUMER) : 0;
        id = p->in_flight.front();
    }
    pipe_slot &s = p->slots[(size_t)id];
    (void)hipSetDevice(p->device);
    if (block) {
        PIPE_HIP_SIDE(p, hipEventSynchronize(s.done), SIDE_CONSUMER);
    } else {
        const hipError_t e = hipEventQuery(s.done);
        if (e == hipErrorNotReady) return 0;
        if (e != hipSuccess) return pfail(p, SS_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e), SIDE_CONSUMER);
    }
    {
        std::lock_guard<std::mutex> g(p->m);
        p->in_flight.pop_front();
        s.state = SLOT_RETURNED;
    }
    fill_result(p, s, id, out);
    return block ? SS_OK : 1;
}

int ss_pipe_wait(ss_pipe *p, ss_pipe_result *out) { return take_oldest(p, true, out); }
int ss_pipe_poll(ss_pipe *p, ss_pipe_result *out) { return take_oldest(p, false, out); }

int ss_pipe_release(ss_pipe *p, int slot)
{
    if (!p) return SS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(p->m);
    if (slot < 0 || slot >= (int)p->slots.size()) return pfail(p, SS_ERR_INVALID_ARG, "ss_pipe_release: no such slot", SIDE_CONSUMER);
    pipe_slot &s = p->slots[(size_t)slot];
    if (s.state != SLOT_RETURNED && s.state != SLOT_ACQUIRED)
        return pfail(p, SS_ERR_STATE, "ss_pipe_release: slot is free or still in flight", SIDE_CONSUMER);
    s.state = SLOT_FREE;
    return SS_OK;
}

int ss_pipe_in_flight(const ss_pipe *p)
{
    if (!p) return SS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(p->m);
    return (int)p->in_flight.size();
}

} /* extern "C" */
