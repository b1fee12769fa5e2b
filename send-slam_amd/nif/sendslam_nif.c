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

static int on_load(ErlNifEnv *env, void **priv, ERL_NIF_TERM info)
{
    (void)priv; (void)info;
    CTX_TYPE = enif_open_resource_type(env, NULL, "sendslam_ctx", ctx_dtor, ERL_NIF_RT_CREATE, NULL);
    return CTX_TYPE ? 0 : 1;
}

static ErlNifFunc funcs[] = {
    {"create", 2, nif_create, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"set_calibration", 7, nif_set_calibration, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"extract", 7, nif_extract, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"match", 6, nif_match, ERL_NIF_DIRTY_JOB_CPU_BOUND},
    {"track", 7, nif_track, ERL_NIF_DIRTY_JOB_CPU_BOUND},
};

ERL_NIF_INIT(Elixir.SendSlam.HipNif, funcs, on_load, NULL, NULL, NULL)
