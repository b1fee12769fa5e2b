/*
 * ss_xchg.hip -- the C ABI's own exchange step for the two multi-GPU shapes of the ORB path that have one
 * (SURVEY.md section 8(e)): stereo (config 4: all-gather of the descriptor blocks of two eyes) and loop closure
 * (config 5: query broadcast + all-gather of world x nq x 8-byte match records).  Nothing in the reference
 * corresponds to it: the reference has one camera, one backend process and one TCP link
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc:387-388, send_slam/lib/send_slam/application.ex:80).
 *
 * Shape of the thing.  Every message of these shapes is <= 64 KB per rank (stereo batches: B x 68 KB): latency-bound.
 * xGMI is a full mesh of point-to-point links, so the natural all-gather is ONE hop: every rank stores its block straight
 * into every peer's memory and raises a flag there; nobody forwards anything (a ring would pay world - 1 hops of latency
 * for bandwidth these messages do not need).
 *
 *   slab (one per rank, device memory, uncached / fine-grained so that a consumer on any XCD reads what arrived and
 *   not what an L2 still holds):   [world flags, 128 B apart] [parity 0: world x slot] [parity 1: world x slot]
 *
 *   create     every rank allocates and zeroes its slab, exports it (hipIpcGetMemHandle), the handles are swapped over a
 *              Unix-domain socket (rank 0 listens at `rendezvous`), every rank maps every peer's slab
 *              (hipIpcOpenMemHandle; works between processes on different GPUs and on the same GPU)
 *   send       k_xchg_send on the context's stream: grid (blocks, peers); 16-byte stores into peer p's slab at
 *              [parity][rank]; every block ends with a system-scope fence, the last block of a peer (a counter in local
 *              memory) stores the message number into flag[rank] of that peer with system-scope release
 *   wait       k_xchg_wait on the same stream: lane r polls the local flag of rank r (relaxed, with s_sleep) until it
 *              carries this message's number, then ONE system-scope acquire.  The spin is bounded by the wall clock
 *              (s_memrealtime): a peer that never arrives makes the kernel END with the error word set, never hang.
 *   consume    kernels enqueued on that stream afterwards read the local slab.
 *
 * Reuse of a parity buffer.  Message m + 2 of a rank lands where its message m did.  The writer enqueues send(m + 2) after
 * its own wait(m + 1), which saw the reader's flag m + 1; the reader enqueued send(m + 1) after its consumers of message m
 * (same stream).  So the reader is done with m before m + 2 can arrive -- provided EVERY rank takes part in EVERY message
 * (a broadcast is a message in which the non-root ranks send only their flag) and consumers run on the context's stream.
 *
 * No PyTorch, no RCCL: a non-Python host (the front door, the NIF) reaches configs 4 and 5 with this.  RCCL through
 * torch.distributed stays the other route (send_slam_amd/multi.py); bench.py --exchange native|rccl reports both.
 */
#include <hip/hip_runtime.h>

#include <errno.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/sendslam_orb.h"

#define XC_FLAG_WORDS 16 /* 128 bytes between two flags */
#define XC_MAX_SEGMENTS 4
#define XC_MAX_WORLD 64

namespace {

thread_local std::string g_xchg_create_error;

struct xc_segs {
    const uint8_t *ptr[XC_MAX_SEGMENTS];
    int64_t bytes[XC_MAX_SEGMENTS];
    int64_t off[XC_MAX_SEGMENTS]; /* offset inside the rank's block, multiples of 16 */
    int n;
};

/* grid (blocks_per_peer, n_peers).  Plain loads of the source (written by earlier kernels of this stream), 16-byte
 * stores into the peer's slab. */
__global__ __launch_bounds__(256) void k_xchg_send(xc_segs segs, uint8_t *const *__restrict__ peer_slab, uint64_t *const *__restrict__ peer_flags,
                                                   uint32_t *__restrict__ done, int rank, int64_t dst_off, uint64_t seq)
{
    const int peer = blockIdx.y, nb = gridDim.x, tid = threadIdx.x;
    uint8_t *dst = peer_slab[peer] + dst_off;
    for (int s = 0; s < segs.n; s++) {
        const uint8_t *src = segs.ptr[s];
        uint8_t *d = dst + segs.off[s];
        const int64_t bytes = segs.bytes[s];
        if ((((uintptr_t)src) & 15) == 0) {
            const int64_t units = bytes >> 4;
            for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < units; i += (int64_t)nb * 256)
                ((uint4 *)d)[i] = ((const uint4 *)src)[i];
            if (blockIdx.x == 0 && tid < (int)(bytes & 15)) d[(units << 4) + tid] = src[(units << 4) + tid];
        } else {
            for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < bytes; i += (int64_t)nb * 256) d[i] = src[i];
        }
    }
    __threadfence_system(); /* this thread's stores are out of every cache of this GPU */
    __syncthreads();
    if (tid == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(&done[peer], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_SYSTEM);
        if (prev + 1 == (uint32_t)nb) { /* every block of this peer has fenced its stores */
            __hip_atomic_store(&done[peer], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* next message: later in this stream */
            __threadfence_system();
            __hip_atomic_store(&peer_flags[peer][rank * XC_FLAG_WORDS], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

/* one wave; lane r waits for rank r's message `seq`.  Every lane reaches the end: the spin is bounded. */
__global__ __launch_bounds__(64) void k_xchg_wait(const uint64_t *__restrict__ flags, int world, uint64_t seq, uint64_t timeout_ticks,
                                                  int32_t *__restrict__ err)
{
    const int lane = threadIdx.x;
    bool ok = true;
    if (lane < world) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime(); /* 100 MHz, independent of the shader clock */
        while (__hip_atomic_load(&flags[lane * XC_FLAG_WORDS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); /* system scope: nothing read after this kernel is older than the flags */
    if (!ok) __hip_atomic_store(err, 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct xc_entry {
    hipIpcMemHandle_t handle;
    int32_t rank;
    int32_t pid;
};

double now_s()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

bool io_all(int fd, void *buf, size_t n, bool wr, double deadline)
{
    uint8_t *p = (uint8_t *)buf;
    while (n > 0) {
        const double left = deadline - now_s();
        if (left <= 0) return false;
        pollfd pf{fd, (short)(wr ? POLLOUT : POLLIN), 0};
        const int pr = poll(&pf, 1, (int)std::min(left * 1e3 + 1, 1e6));
        if (pr < 0 && errno == EINTR) continue;
        if (pr <= 0) return false;
        const ssize_t k = wr ? send(fd, p, n, MSG_NOSIGNAL) : recv(fd, p, n, 0);
        if (k < 0 && (errno == EINTR || errno == EAGAIN)) continue;
        if (k <= 0) return false;
        p += k;
        n -= (size_t)k;
    }
    return true;
}

bool fill_addr(sockaddr_un *a, const char *path)
{
    memset(a, 0, sizeof(*a));
    a->sun_family = AF_UNIX;
    if (strlen(path) >= sizeof(a->sun_path)) return false;
    strcpy(a->sun_path, path);
    return true;
}

/* all ranks end up with every rank's entry; returns an error message or "" */
std::string swap_handles(int rank, int world, const char *path, double deadline, const xc_entry &mine, std::vector<xc_entry> &table)
{
    table.assign((size_t)world, xc_entry{});
    table[(size_t)rank] = mine;
    sockaddr_un addr;
    if (!fill_addr(&addr, path)) return "rendezvous path too long for a Unix-domain socket";
    if (rank == 0) {
        const int ls = socket(AF_UNIX, SOCK_STREAM, 0);
        if (ls < 0) return std::string("socket: ") + strerror(errno);
        unlink(path);
        if (bind(ls, (sockaddr *)&addr, sizeof(addr)) != 0 || listen(ls, world) != 0) {
            const std::string m = std::string("bind/listen ") + path + ": " + strerror(errno);
            close(ls);
            return m;
        }
        std::vector<int> conns;
        std::string msg;
        while ((int)conns.size() < world - 1 && msg.empty()) {
            const double left = deadline - now_s();
            pollfd pf{ls, POLLIN, 0};
            const int pr = left > 0 ? poll(&pf, 1, (int)(left * 1e3) + 1) : 0;
            if (pr < 0 && errno == EINTR) continue;
            if (pr <= 0) {
                msg = "rendezvous: " + std::to_string(world - 1 - (int)conns.size()) + " rank(s) did not connect in time";
                break;
            }
            const int fd = accept(ls, nullptr, nullptr);
            if (fd < 0) continue;
            xc_entry e;
            if (!io_all(fd, &e, sizeof(e), false, deadline) || e.rank <= 0 || e.rank >= world) {
                close(fd);
                msg = "rendezvous: bad hello from a peer";
                break;
            }
            table[(size_t)e.rank] = e;
            conns.push_back(fd);
        }
        for (int fd : conns) {
            if (msg.empty() && !io_all(fd, table.data(), sizeof(xc_entry) * (size_t)world, true, deadline)) msg = "rendezvous: a peer went away";
            close(fd);
        }
        close(ls);
        unlink(path);
        return msg;
    }
    /* rank 0 may not be listening yet */
    int fd = -1;
    while (true) {
        fd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) return std::string("socket: ") + strerror(errno);
        if (connect(fd, (sockaddr *)&addr, sizeof(addr)) == 0) break;
        close(fd);
        if (now_s() > deadline) return std::string("rendezvous: cannot reach rank 0 at ") + path + ": " + strerror(errno);
        usleep(5000);
    }
    std::string msg;
    xc_entry e = mine;
    if (!io_all(fd, &e, sizeof(e), true, deadline) || !io_all(fd, table.data(), sizeof(xc_entry) * (size_t)world, false, deadline))
        msg = "rendezvous: rank 0 went away";
    close(fd);
    return msg;
}

} // namespace

struct ss_xchg {
    int device = 0, rank = 0, world = 1;
    int64_t max_bytes = 0, slot = 0, data_off = 0, slab_bytes = 0;
    uint8_t *slab = nullptr;
    std::vector<uint8_t *> peer_slab;
    uint8_t **d_peer_slab = nullptr;
    uint64_t **d_peer_flags = nullptr;
    uint32_t *d_done = nullptr;
    int32_t *h_err = nullptr; /* pinned, device-visible */
    uint64_t seq = 0;
    uint64_t timeout_ticks = 0;
    std::string err;
};

namespace {

int xfail(ss_xchg *x, int code, const std::string &msg)
{
    if (x) x->err = msg;
    else g_xchg_create_error = msg;
    return code;
}

#define XC_TRY(x, call)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return xfail((x), e_ == hipErrorOutOfMemory ? SS_ERR_NO_MEMORY : SS_ERR_HIP,                     \
                         std::string(#call) + ": " + hipGetErrorString(e_));                                 \
    } while (0)

void xc_free(ss_xchg *x)
{
    (void)hipSetDevice(x->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < (int)x->peer_slab.size(); r++)
        if (r != x->rank && x->peer_slab[(size_t)r]) (void)hipIpcCloseMemHandle(x->peer_slab[(size_t)r]);
    if (x->d_peer_slab) (void)hipFree(x->d_peer_slab);
    if (x->d_peer_flags) (void)hipFree(x->d_peer_flags);
    if (x->d_done) (void)hipFree(x->d_done);
    if (x->slab) (void)hipFree(x->slab);
    if (x->h_err) (void)hipHostFree(x->h_err);
    delete x;
}

int xc_check(ss_xchg *x)
{
    const int32_t e = __atomic_load_n(x->h_err, __ATOMIC_RELAXED);
    if (e != 0)
        return xfail(x, SS_ERR_STATE, "ss_xchg: rank " + std::to_string(e - 1) + "'s message did not arrive within the time limit (peer gone?); the exchange is unusable");
    return SS_OK;
}

/* one message: this rank's segments (possibly none) to every peer at dst_off, then the wait */
int xc_message(ss_xchg *x, hipStream_t s, const xc_segs &segs, int64_t total, int64_t dst_off, uint64_t seq)
{
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(32, (total + 16383) / 16384));
    hipLaunchKernelGGL(k_xchg_send, dim3((unsigned)nb, (unsigned)x->world), dim3(256), 0, s, segs, (uint8_t *const *)x->d_peer_slab,
                       (uint64_t *const *)x->d_peer_flags, x->d_done, x->rank, dst_off, seq);
    hipLaunchKernelGGL(k_xchg_wait, dim3(1), dim3(64), 0, s, (const uint64_t *)x->slab, x->world, seq, x->timeout_ticks, x->h_err);
    XC_TRY(x, hipGetLastError());
    return SS_OK;
}

} // namespace

extern "C" {

int ss_xchg_create(int device_ordinal, int rank, int world, int64_t max_bytes, const char *rendezvous, int timeout_ms, ss_xchg **out)
{
    if (!out) return xfail(nullptr, SS_ERR_INVALID_ARG, "ss_xchg_create: out is NULL");
    *out = nullptr;
    if (world < 1 || world > XC_MAX_WORLD || rank < 0 || rank >= world || max_bytes < 16 || max_bytes > ((int64_t)1 << 32))
        return xfail(nullptr, SS_ERR_INVALID_ARG, "ss_xchg_create: bad rank / world / max_bytes");
    if (world > 1 && (!rendezvous || !*rendezvous)) return xfail(nullptr, SS_ERR_INVALID_ARG, "ss_xchg_create: a rendezvous path is needed for world > 1");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return xfail(nullptr, SS_ERR_NO_DEVICE, "no HIP device: this library has no CPU path");
    if (device_ordinal < 0 || device_ordinal >= n_dev) return xfail(nullptr, SS_ERR_NO_DEVICE, "device ordinal out of range");
    XC_TRY((ss_xchg *)nullptr, hipSetDevice(device_ordinal));
    if (timeout_ms <= 0) timeout_ms = 10000;
    ss_xchg *x = new ss_xchg();
    x->device = device_ordinal;
    x->rank = rank;
    x->world = world;
    x->max_bytes = max_bytes;
    x->slot = (max_bytes + 255) & ~(int64_t)255;
    x->data_off = ((int64_t)world * XC_FLAG_WORDS * 8 + 255) & ~(int64_t)255;
    x->slab_bytes = x->data_off + 2 * (int64_t)world * x->slot;
    /* timeout_ms bounds the rendezvous (ranks may reach it far apart: first imports, database setup); a message's wait is
     * bounded by min(timeout_ms, 10 s) -- a spinning kernel should not outlive a watchdog -- or by SENDSLAM_XCHG_WAIT_MS */
    int wait_ms = timeout_ms < 10000 ? timeout_ms : 10000;
    if (const char *e = getenv("SENDSLAM_XCHG_WAIT_MS")) wait_ms = atoi(e) > 0 ? atoi(e) : wait_ms;
    x->timeout_ticks = (uint64_t)wait_ms * 100000ull;
    x->peer_slab.assign((size_t)world, nullptr);
#define XC_CREATE_TRY(call)                                                                                   \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            const int rc_ = xfail(nullptr, e_ == hipErrorOutOfMemory ? SS_ERR_NO_MEMORY : SS_ERR_HIP,         \
                                  std::string(#call) + ": " + hipGetErrorString(e_));                         \
            xc_free(x);                                                                                       \
            return rc_;                                                                                       \
        }                                                                                                     \
    } while (0)
    /* uncached device memory: what a peer stored is what a local load returns, whatever XCD the reader runs on */
    if (hipExtMallocWithFlags((void **)&x->slab, (size_t)x->slab_bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        x->slab = nullptr;
        XC_CREATE_TRY(hipExtMallocWithFlags((void **)&x->slab, (size_t)x->slab_bytes, hipDeviceMallocFinegrained));
    }
    XC_CREATE_TRY(hipMemset(x->slab, 0, (size_t)x->slab_bytes));
    XC_CREATE_TRY(hipHostMalloc((void **)&x->h_err, sizeof(int32_t), hipHostMallocMapped));
    *x->h_err = 0;
    XC_CREATE_TRY(hipMalloc((void **)&x->d_done, (size_t)world * sizeof(uint32_t)));
    XC_CREATE_TRY(hipMemset(x->d_done, 0, (size_t)world * sizeof(uint32_t)));
    XC_CREATE_TRY(hipMalloc((void **)&x->d_peer_slab, (size_t)world * sizeof(uint8_t *)));
    XC_CREATE_TRY(hipMalloc((void **)&x->d_peer_flags, (size_t)world * sizeof(uint64_t *)));
    XC_CREATE_TRY(hipDeviceSynchronize()); /* the slab is zero before anybody can learn its handle */
    x->peer_slab[(size_t)rank] = x->slab;
    if (world > 1) {
        xc_entry mine{};
        XC_CREATE_TRY(hipIpcGetMemHandle(&mine.handle, x->slab));
        mine.rank = rank;
        mine.pid = (int32_t)getpid();
        std::vector<xc_entry> table;
        const std::string msg = swap_handles(rank, world, rendezvous, now_s() + timeout_ms * 1e-3, mine, table);
        if (!msg.empty()) {
            xc_free(x);
            return xfail(nullptr, SS_ERR_STATE, "ss_xchg_create: " + msg);
        }
        for (int r = 0; r < world; r++) {
            if (r == rank) continue;
            void *p = nullptr;
            XC_CREATE_TRY(hipIpcOpenMemHandle(&p, table[(size_t)r].handle, hipIpcMemLazyEnablePeerAccess));
            x->peer_slab[(size_t)r] = (uint8_t *)p;
        }
    }
    std::vector<uint8_t *> ps((size_t)world);
    std::vector<uint64_t *> pf((size_t)world);
    for (int r = 0; r < world; r++) {
        ps[(size_t)r] = x->peer_slab[(size_t)r] + x->data_off;
        pf[(size_t)r] = (uint64_t *)x->peer_slab[(size_t)r];
    }
    XC_CREATE_TRY(hipMemcpy(x->d_peer_slab, ps.data(), (size_t)world * sizeof(uint8_t *), hipMemcpyHostToDevice));
    XC_CREATE_TRY(hipMemcpy(x->d_peer_flags, pf.data(), (size_t)world * sizeof(uint64_t *), hipMemcpyHostToDevice));
#undef XC_CREATE_TRY
    *out = x;
    return SS_OK;
}

const char *ss_xchg_last_error(const ss_xchg *x) { return x ? x->err.c_str() : g_xchg_create_error.c_str(); }

int ss_xchg_status(ss_xchg *x)
{
    if (!x) return SS_ERR_INVALID_ARG;
    return xc_check(x);
}

int ss_xchg_allgather(ss_xchg *x, ss_ctx *ctx, const void *const *d_segments, const int64_t *segment_bytes, int n_segments,
                      const void **d_gathered, int64_t *rank_stride)
{
    if (!x || !ctx || !d_gathered || !rank_stride) return SS_ERR_INVALID_ARG;
    if (n_segments < 1 || n_segments > XC_MAX_SEGMENTS || !d_segments || !segment_bytes)
        return xfail(x, SS_ERR_INVALID_ARG, "ss_xchg_allgather: 1..4 segments");
    (void)hipSetDevice(x->device);
    int rc = xc_check(x);
    if (rc != SS_OK) return rc;
    xc_segs segs{};
    int64_t total = 0;
    for (int i = 0; i < n_segments; i++) {
        if (segment_bytes[i] < 0 || (segment_bytes[i] > 0 && !d_segments[i])) return xfail(x, SS_ERR_INVALID_ARG, "ss_xchg_allgather: bad segment");
        segs.ptr[i] = (const uint8_t *)d_segments[i];
        segs.bytes[i] = segment_bytes[i];
        segs.off[i] = total;
        total += (segment_bytes[i] + 15) & ~(int64_t)15;
    }
    segs.n = n_segments;
    if (total > x->max_bytes) return xfail(x, SS_ERR_INVALID_ARG, "ss_xchg_allgather: message larger than max_bytes of this exchange");
    void *sv = nullptr;
    rc = ss_get_stream(ctx, &sv);
    if (rc != SS_OK) return xfail(x, rc, "ss_xchg_allgather: bad context");
    const uint64_t seq = ++x->seq;
    const int64_t base = (int64_t)(seq & 1) * x->world * x->slot;
    rc = xc_message(x, (hipStream_t)sv, segs, total, base + (int64_t)x->rank * total, seq);
    if (rc != SS_OK) return rc;
    *d_gathered = x->slab + x->data_off + base;
    *rank_stride = total;
    return SS_OK;
}

int ss_xchg_broadcast(ss_xchg *x, ss_ctx *ctx, int root, void *d_buf, int64_t bytes)
{
    if (!x || !ctx) return SS_ERR_INVALID_ARG;
    if (root < 0 || root >= x->world || bytes < 0 || bytes > x->max_bytes || (bytes > 0 && !d_buf))
        return xfail(x, SS_ERR_INVALID_ARG, "ss_xchg_broadcast: bad root / size");
    (void)hipSetDevice(x->device);
    int rc = xc_check(x);
    if (rc != SS_OK) return rc;
    void *sv = nullptr;
    rc = ss_get_stream(ctx, &sv);
    if (rc != SS_OK) return xfail(x, rc, "ss_xchg_broadcast: bad context");
    xc_segs segs{};
    if (x->rank == root) { /* the others send their flag only: every rank takes part in every message */
        segs.ptr[0] = (const uint8_t *)d_buf;
        segs.bytes[0] = bytes;
        segs.n = 1;
    }
    const uint64_t seq = ++x->seq;
    const int64_t base = (int64_t)(seq & 1) * x->world * x->slot;
    rc = xc_message(x, (hipStream_t)sv, segs, x->rank == root ? bytes : 0, base, seq);
    if (rc != SS_OK) return rc;
    if (x->rank != root && bytes > 0)
        XC_TRY(x, hipMemcpyAsync(d_buf, x->slab + x->data_off + base, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)sv));
    return SS_OK;
}

int ss_xchg_destroy(ss_xchg *x)
{
    if (!x) return SS_ERR_INVALID_ARG;
    (void)hipSetDevice(x->device);
    if (x->world > 1 && *x->h_err == 0) {
        /* a last flag-only message on a stream of our own: once every peer's has arrived, nobody writes into this slab
         * any more, and our stores into theirs have completed with the synchronize below (bounded like every wait) */
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess) {
            (void)hipDeviceSynchronize();
            xc_segs segs{};
            const uint64_t seq = ++x->seq;
            (void)xc_message(x, s, segs, 0, (int64_t)(seq & 1) * x->world * x->slot, seq);
            (void)hipStreamSynchronize(s);
            (void)hipStreamDestroy(s);
        }
    }
    xc_free(x);
    return SS_OK;
}

} /* extern "C" */
