/*
 * sendslam_nif.c -- dirty-NIF glue between the BEAM and libsendslam_orb.so (C ABI:
 * include/sendslam_orb.h).  Pure marshalling: no arithmetic lives here.
 *
 * NOT COMPILED IN THIS REPOSITORY'S CONTAINER: erl_nif.h / OTP are absent (SURVEY.md section
 * 8(c)).  Build on a host with Erlang/OTP:
 *     cc -O2 -fPIC -shared -I$ERL_ROOT/usr/include -I../../include sendslam_nif.c \
 *        -L../lib -lsendslam_orb -Wl,-rpath,'$ORIGIN/../lib' -o priv/sendslam_nif.so
 *
 * It replaces, for the per-frame path, what SendSlam.SlamHandler does over TCP
 * (/root/reference/send_slam/lib/send_slam/slam_handler.ex:59-88): instead of PPM-encoding the
 * Evision.Mat and writing 2.7 MB to a socket, HipBackend passes the Mat's binary straight to
 * ss_extract.  Every entry point is flagged ERL_NIF_DIRTY_JOB_CPU_BOUND (calls exceed 1 ms) and
 * none raises: errors come back as {:error, {code, message}}.
 */
#include <erl_nif.h>
#include <string.h>

#include "sendslam_orb.h"

static ErlNifResourceType *CTX_TYPE;
typedef struct { ss_ctx *ctx; } ctx_res;

static void ctx_dtor(ErlNifEnv *env, void *obj)
{
    (void)env;
    ctx_res *r = (ctx_res *)obj;
    if (r->ctx) ss_destroy(r->ctx);
    r->ctx = NULL;
}

static ERL_NIF_TERM mk_error(ErlNifEnv *env, int code, const char *msg)
{
    ERL_NIF_TERM m = enif_make_string(env, msg ? msg : "", ERL_NIF_LATIN1);
    return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_tuple2(env, enif_make_int(env, code), m));
}

/* create(device, n_features) -> {:ok, ref} | {:error, {code, msg}} */
static ERL_NIF_TERM nif_create(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    int device, n_features;
    (void)argc;
    if (!enif_get_int(env, argv[0], &device) || !enif_get_int(env, argv[1], &n_features)) return enif_make_badarg(env);
    ss_orb_params p;
    ss_orb_params_default(&p);
    if (n_features > 0) p.n_features = n_features;
    ss_ctx *ctx = NULL;
    int rc = ss_create(device, &p, &ctx);
    if (rc != SS_OK) return mk_error(env, rc, ss_last_error(NULL));
    ctx_res *r = enif_alloc_resource(CTX_TYPE, sizeof(ctx_res));
    r->ctx = ctx;
    ERL_NIF_TERM t = enif_make_resource(env, r);
    enif_release_resource(r);
    return enif_make_tuple2(env, enif_make_atom(env, "ok"), t);
}

/* set_calibration(ref, camera_id, {fx,fy,cx,cy,k1,k2,p1,p2}, width, height, fps, rgb) -> :ok | error */
static ERL_NIF_TERM nif_set_calibration(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    ctx_res *r;
    int cam_id, arity, w, h, rgb;
    double fps, v[8];
    const ERL_NIF_TERM *tup;
    (void)argc;
    if (!enif_get_resource(env, argv[0], CTX_TYPE, (void **)&r) || !enif_get_int(env, argv[1], &cam_id) ||
        !enif_get_tuple(env, argv[2], &arity, &tup) || arity != 8 || !enif_get_int(env, argv[3], &w) ||
        !enif_get_int(env, argv[4], &h) || !enif_get_double(env, argv[5], &fps) || !enif_get_int(env, argv[6], &rgb))
        return enif_make_badarg(env);
    for (int i = 0; i < 8; i++)
        if (!enif_get_double(env, tup[i], &v[i])) return enif_make_badarg(env);
    ss_camera c;
    memset(&c, 0, sizeof(c));
    strcpy(c.type, "PinHole");
    c.fx = v[0]; c.fy = v[1]; c.cx = v[2]; c.cy = v[3]; c.k1 = v[4]; c.k2 = v[5]; c.p1 = v[6]; c.p2 = v[7];
    c.width = w; c.height = h; c.fps = fps; c.rgb = rgb;
    c.th_depth = 40.0; c.baseline = 0.0; c.depth_map_factor = 1000.0; /* slam_handler.ex:223-225 */
    int rc = ss_set_calibration(r->ctx, cam_id, &c);
    return rc == SS_OK ? enif_make_atom(env, "ok") : mk_error(env, rc, ss_last_error(r->ctx));
}

/* extract(ref, camera_id, pixels :: binary, width, height, channels, timestamp)
 *   -> {:ok, n, keypoints :: binary (n x 24 B), descriptors :: binary (n x 32 B)} | error */
static ERL_NIF_TERM nif_extract(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    ctx_res *r;
    ErlNifBinary pix;
    int cam_id, w, h, ch;
    double ts;
    (void)argc;
    if (!enif_get_resource(env, argv[0], CTX_TYPE, (void **)&r) || !enif_get_int(env, argv[1], &cam_id) ||
        !enif_inspect_binary(env, argv[2], &pix) || !enif_get_int(env, argv[3], &w) || !enif_get_int(env, argv[4], &h) ||
        !enif_get_int(env, argv[5], &ch) || !enif_get_double(env, argv[6], &ts))
        return enif_make_badarg(env);
    if (w <= 0 || h <= 0 || ch <= 0 || pix.size < (size_t)w * h * ch) return mk_error(env, SS_ERR_BAD_FRAME, "binary smaller than the frame");
    ss_frame_result res;
    int rc = ss_extract(r->ctx, cam_id, pix.data, w, h, ch, w * ch, ts, &res);
    if (rc != SS_OK) return mk_error(env, rc, ss_last_error(r->ctx));
    ERL_NIF_TERM kb, db;
    unsigned char *kp = enif_make_new_binary(env, (size_t)res.n_keypoints * sizeof(ss_keypoint), &kb);
    unsigned char *dp = enif_make_new_binary(env, (size_t)res.n_keypoints * SS_DESC_BYTES, &db);
    memcpy(kp, res.keypoints, (size_t)res.n_keypoints * sizeof(ss_keypoint));
    memcpy(dp, res.descriptors, (size_t)res.n_keypoints * SS_DESC_BYTES);
    return enif_make_tuple4(env, enif_make_atom(env, "ok"), enif_make_int(env, res.n_keypoints), kb, db);
}

/* match(ref, query :: binary, train :: binary, th, ratio_num, ratio_den) -> {:ok, idx :: binary (int32 x nq)} | error */
static ERL_NIF_TERM nif_match(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    ctx_res *r;
    ErlNifBinary q, t;
    int th, num, den;
    (void)argc;
    if (!enif_get_resource(env, argv[0], CTX_TYPE, (void **)&r) || !enif_inspect_binary(env, argv[1], &q) ||
        !enif_inspect_binary(env, argv[2], &t) || !enif_get_int(env, argv[3], &th) || !enif_get_int(env, argv[4], &num) ||
        !enif_get_int(env, argv[5], &den))
        return enif_make_badarg(env);
    const int nq = (int)(q.size / SS_DESC_BYTES), nt = (int)(t.size / SS_DESC_BYTES);
    ERL_NIF_TERM ib;
    int32_t *idx = (int32_t *)enif_make_new_binary(env, (size_t)nq * 4, &ib);
    uint16_t *d = (uint16_t *)enif_alloc((size_t)(nq > 0 ? nq : 1) * 4);
    int rc = ss_match(r->ctx, q.data, nq, t.data, nt, th, num, den, 0, idx, d, d + nq);
    enif_free(d);
    if (rc != SS_OK) return mk_error(env, rc, ss_last_error(r->ctx));
    return enif_make_tuple2(env, enif_make_atom(env, "ok"), ib);
}

/* track(ref, camera_id, pixels :: binary, width, height, channels, timestamp)
 *   -> {:ok, tracking_state, {px, py, pz}, {qx, qy, qz, qw}, {n_keypoints, n_matches, n_inliers, n_map_points}} | error */
static ERL_NIF_TERM nif_track(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    ctx_res *r;
    ErlNifBinary pix;
    int cam_id, w, h, ch;
    double ts;
    (void)argc;
    if (!enif_get_resource(env, argv[0], CTX_TYPE, (void **)&r) || !enif_get_int(env, argv[1], &cam_id) ||
        !enif_inspect_binary(env, argv[2], &pix) || !enif_get_int(env, argv[3], &w) || !enif_get_int(env, argv[4], &h) ||
        !enif_get_int(env, argv[5], &ch) || !enif_get_double(env, argv[6], &ts))
        return enif_make_badarg(env);
    if (w <= 0 || h <= 0 || ch <= 0 || pix.size < (size_t)w * h * ch) return mk_error(env, SS_ERR_BAD_FRAME, "binary smaller than the frame");
    ss_pose po;
    int rc = ss_track(r->ctx, cam_id, pix.data, w, h, ch, w * ch, ts, &po);
    if (rc != SS_OK) return mk_error(env, rc, ss_last_error(r->ctx));
    ERL_NIF_TERM pos = enif_make_tuple3(env, enif_make_double(env, po.position[0]), enif_make_double(env, po.position[1]),
                                        enif_make_double(env, po.position[2]));
    ERL_NIF_TERM quat = enif_make_tuple4(env, enif_make_double(env, po.quaternion[0]), enif_make_double(env, po.quaternion[1]),
                                         enif_make_double(env, po.quaternion[2]), enif_make_double(env, po.quaternion[3]));
    ERL_NIF_TERM cnt = enif_make_tuple4(env, enif_make_int(env, po.n_keypoints), enif_make_int(env, po.n_matches),
                                        enif_make_int(env, po.n_inliers), enif_make_int(env, po.n_map_points));
    return enif_make_tuple5(env, enif_make_atom(env, "ok"), enif_make_int(env, po.tracking_state), pos, quat, cnt);
}

/* track_reset(ref) -> :ok   (System::Reset; ss_set_calibration resets by itself, like the shim's rebuild :491-518) */
static ERL_NIF_TERM nif_track_reset(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    ctx_res *r;
    (void)argc;
    if (!enif_get_resource(env, argv[0], CTX_TYPE, (void **)&r)) return enif_make_badarg(env);
    ss_track_reset(r->ctx);
    return enif_make_atom(env, "ok");
}

/* ---- pipelined host-memory path (ss_pipe_*): a GenServer that receives frames faster than one ss_track call per
 * frame returns (several cameras, a replay producer) hands lists of Mat binaries to pipe_submit and collects the
 * batches with pipe_wait; copies and kernels of different batches overlap inside the library. ---- */
static ErlNifResourceType *PIPE_TYPE;
typedef struct { ss_pipe *pipe; int w, h, ch; } pipe_res;

static void pipe_dtor(ErlNifEnv *env, void *obj)
{
    (void)env;
    pipe_res *r = (pipe_res *)obj;
    if (r->pipe) ss_pipe_destroy(r->pipe);
    r->pipe = NULL;
}

/* pipe_open(device, n_features, width, height, channels, batch, depth, match_mode, rgb) -> {:ok, pipe} | error */
static ERL_NIF_TERM nif_pipe_open(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    int v[9];
    (void)argc;
    for (int i = 0; i < 9; i++)
        if (!enif_get_int(env, argv[i], &v[i])) return enif_make_badarg(env);
    ss_orb_params p;
    ss_orb_params_default(&p);
    if (v[1] > 0) p.n_features = v[1];
    ss_pipe_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.width = v[2]; cfg.height = v[3]; cfg.channels = v[4]; cfg.batch = v[5]; cfg.depth = v[6]; cfg.match_mode = v[7];
    ss_camera cam;
    memset(&cam, 0, sizeof(cam));
    strcpy(cam.type, "PinHole");
    cam.width = v[2]; cam.height = v[3]; cam.rgb = v[8];
    ss_pipe *pipe = NULL;
    int rc = ss_pipe_create(v[0], &p, &cam, &cfg, &pipe);
    if (rc != SS_OK) return mk_error(env, rc, ss_pipe_last_error(NULL));
    pipe_res *r = enif_alloc_resource(PIPE_TYPE, sizeof(pipe_res));
    r->pipe = pipe; r->w = v[2]; r->h = v[3]; r->ch = v[4];
    ERL_NIF_TERM t = enif_make_resource(env, r);
    enif_release_resource(r);
    return enif_make_tuple2(env, enif_make_atom(env, "ok"), t);
}

/* pipe_submit(pipe, frames :: [binary], camera_id, first_timestamp, dt) -> :ok | {:error, {-10, _}} when the ring is full */
static ERL_NIF_TERM nif_pipe_submit(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    pipe_res *r;
    int cam_id;
    double ts0, dt;
    unsigned n = 0;
    (void)argc;
    if (!enif_get_resource(env, argv[0], PIPE_TYPE, (void **)&r) || !enif_get_list_length(env, argv[1], &n) ||
        !enif_get_int(env, argv[2], &cam_id) || !enif_get_double(env, argv[3], &ts0) || !enif_get_double(env, argv[4], &dt))
        return enif_make_badarg(env);
    if (n == 0 || n > 256) return mk_error(env, SS_ERR_INVALID_ARG, "1..256 frames per batch");
    const uint8_t *ptrs[256];
    int32_t cams[256];
    double stamps[256];
    ERL_NIF_TERM head, tail = argv[1];
    const size_t need = (size_t)r->w * r->h * r->ch;
    for (unsigned i = 0; i < n; i++) {
        ErlNifBinary b;
        if (!enif_get_list_cell(env, tail, &head, &tail)) return enif_make_badarg(env);
        ptrs[i] = (enif_inspect_binary(env, head, &b) && b.size >= need) ? b.data : NULL; /* NULL = this frame is skipped */
        cams[i] = cam_id;
        stamps[i] = ts0 + dt * i;
    }
    int rc = ss_pipe_submit_frames(r->pipe, ptrs, (int)n, (int64_t)r->w * r->ch, cams, stamps);
    return rc == SS_OK ? enif_make_atom(env, "ok") : mk_error(env, rc, ss_pipe_last_error(r->pipe));
}

/* pipe_wait(pipe) -> {:ok, [{status, camera_id, timestamp, n, keypoints :: binary, descriptors :: binary, match_idx :: binary}]} */
static ERL_NIF_TERM nif_pipe_wait(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    pipe_res *r;
    (void)argc;
    if (!enif_get_resource(env, argv[0], PIPE_TYPE, (void **)&r)) return enif_make_badarg(env);
    ss_pipe_result res;
    int rc = ss_pipe_wait(r->pipe, &res);
    if (rc != SS_OK) return mk_error(env, rc, ss_pipe_last_error(r->pipe));
    ERL_NIF_TERM items[256];
    for (int i = 0; i < res.n_frames; i++) {
        const size_t n = (size_t)res.n_keypoints[i], row = (size_t)i * res.kp_capacity;
        ERL_NIF_TERM kb, db, mb;
        memcpy(enif_make_new_binary(env, n * sizeof(ss_keypoint), &kb), res.keypoints + row, n * sizeof(ss_keypoint));
        memcpy(enif_make_new_binary(env, n * SS_DESC_BYTES, &db), res.descriptors + row * SS_DESC_BYTES, n * SS_DESC_BYTES);
        unsigned char *mp = enif_make_new_binary(env, res.match_idx ? n * 4 : 0, &mb);
        if (res.match_idx) memcpy(mp, res.match_idx + row, n * 4);
        items[i] = enif_make_tuple(env, 7, enif_make_int(env, res.status[i]), enif_make_int(env, res.camera_id[i]),
                                   enif_make_double(env, res.timestamp[i]), enif_make_int(env, (int)n), kb, db, mb);
    }
    ERL_NIF_TERM list = enif_make_list_from_array(env, items, (unsigned)res.n_frames);
    ss_pipe_release(r->pipe, res.slot); /* everything was copied into BEAM binaries */
    return enif_make_tuple2(env, enif_make_atom(env, "ok"), list);
}

/* ---- config 4 from Elixir: one HipBackend GenServer per eye / GPU, the two exchange descriptor blocks through ss_xchg_*
 * (peer-mapped device memory, no BEAM message carries a descriptor).  xchg_open is collective: both GenServers call it
 * with the same rendezvous path; stereo_match once per stereo pair, after track / extract of the pair's frame. ---- */
static ErlNifResourceType *XCHG_TYPE;
typedef struct { ss_xchg *x; int rank; } xchg_res;

static void xchg_dtor(ErlNifEnv *env, void *obj)
{
    (void)env;
    xchg_res *r = (xchg_res *)obj;
    if (r->x) ss_xchg_destroy(r->x);
    r->x = NULL;
}

/* xchg_open(device, rank, world, max_bytes, rendezvous :: binary, timeout_ms) -> {:ok, xchg} | error */
static ERL_NIF_TERM nif_xchg_open(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    int device, rank, world, max_bytes, timeout_ms;
    ErlNifBinary path;
    char buf[108];
    (void)argc;
    if (!enif_get_int(env, argv[0], &device) || !enif_get_int(env, argv[1], &rank) || !enif_get_int(env, argv[2], &world) ||
        !enif_get_int(env, argv[3], &max_bytes) || !enif_inspect_binary(env, argv[4], &path) || !enif_get_int(env, argv[5], &timeout_ms) ||
        path.size >= sizeof(buf))
        return enif_make_badarg(env);
    memcpy(buf, path.data, path.size);
    buf[path.size] = 0;
    ss_xchg *x = NULL;
    int rc = ss_xchg_create(device, rank, world, max_bytes, buf, timeout_ms, &x);
    if (rc != SS_OK) return mk_error(env, rc, ss_xchg_last_error(NULL));
    xchg_res *r = enif_alloc_resource(XCHG_TYPE, sizeof(xchg_res));
    r->x = x; r->rank = rank;
    ERL_NIF_TERM t = enif_make_resource(env, r);
    enif_release_resource(r);
    return enif_make_tuple2(env, enif_make_atom(env, "ok"), t);
}

/* stereo_match(ref, xchg, peer_rank) -> {:ok, n_own, n_peer, match_idx :: binary (n_own x int32, -1 = no match)} | error */
static ERL_NIF_TERM nif_stereo_match(ErlNifEnv *env, int argc, const ERL_NIF_TERM argv[])
{
    ctx_res *c;
    xchg_res *x;
    int peer;
    (void)argc;
    if (!enif_get_resource(env, argv[0], CTX_TYPE, (void **)&c) || !enif_get_resource(env, argv[1], XCHG_TYPE, (void **)&x) ||
        !enif_get_int(env, argv[2], &peer))
        return enif_make_badarg(env);
    static int32_t idx[65536]; /* >= kp_capacity of any geometry; dirty NIFs of one GenServer do not overlap */
    int32_t n_own = 0, n_peer = 0;
    int rc = ss_stereo_exchange_match(c->ctx, x->x, peer, 50, 9, 10, idx, NULL, NULL, &n_own, &n_peer);
    if (rc != SS_OK) return mk_error(env, rc, ss_last_error(c->ctx));
    ERL_NIF_TERM mb;
    memcpy(enif_make_new_binary(env, (size_t)n_own * 4, &mb), idx, (size_t)n_own * 4);
    return enif_make_tuple4(env, enif_make_atom(env, "ok"), enif_make_int(env, n_own), enif_make_int(env, n_peer), mb);
}

static int on_load(ErlNifEnv *env, void **priv, ERL_NIF_TERM info)
{
    (void)priv; (void)info;
    CTX_TYPE = enif_open_resource_type(env, NULL, "sendslam_ctx", ctx_dtor, ERL_NIF_RT_CREATE, NULL);
    PIPE_TYPE = enif_open_resource_type(env, NULL, "sendslam_pipe", pipe_dtor, ERL_NIF_RT_CREATE, NULL);
    XCHG_TYPE = enif_open_resource_type(env, NULL, "sendslam_xchg", xchg_dtor, ERL_NIF_RT_CREATE, NULL);
    return CTX_TYPE && PIPE_TYPE && XCHG_TYPE ? 0 : 1;
}

static ErlNifFunc funcs[] = {
    {"create", 2, nif_create, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"set_calibration", 7, nif_set_calibration, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"extract", 7, nif_extract, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"match", 6, nif_match, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"track", 7, nif_track, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"track_reset", 1, nif_track_reset, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"pipe_open", 9, nif_pipe_open, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"pipe_submit", 5, nif_pipe_submit, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"pipe_wait", 1, nif_pipe_wait, ERL_NIF_DIRTY_JOB_IO_BOUND}, /* blocks on the GPU, burns no CPU */
    {"xchg_open", 6, nif_xchg_open, ERL_NIF_DIRTY_JOB_IO_BOUND},
    {"stereo_match", 3, nif_stereo_match, ERL_NIF_DIRTY_JOB_IO_BOUND},
};

ERL_NIF_INIT(Elixir.SendSlam.HipNif, funcs, on_load, NULL, NULL, NULL)
