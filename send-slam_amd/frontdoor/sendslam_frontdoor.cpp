/*
 * sendslam_frontdoor.cpp -- stand-in for the reference's containerised backend shim
 * (/root/reference/slam_backends/orb_slam_3/orbslam3_mono_networked.cc, 684 lines): same
 * environment variable, same TCP client role, same length-prefixed MessagePack protocol, same
 * guards and log-and-skip policy, same shutdown summary -- with ORB_SLAM3::System replaced by
 * libsendslam_orb.so (HIP kernels on an MI355X).  The Elixir side (SlamHandler,
 * send_slam/lib/send_slam/slam_handler.ex) needs no change to talk to it.
 *
 * Mapping to the reference:
 *   ParseMessage / MessagePacket            :78-88, :284-339   -> parse_message
 *   ParseCameraCalibration / RequireScalar  :90-156            -> parse_calibration
 *   BuildCalibrationYaml (ORB literals)     :158-223           -> settings_text (for logs) +
 *                                                                 ss_orb_params / ss_camera
 *   main: env, connect, framing, guards     :341-627           -> main
 *   cv::imdecode(IMREAD_UNCHANGED) on PNM   :546               -> pnm_header + pnm_copy (P5 -> 1
 *                                                                 channel, P6 -> 3 channels in BGR order)
 *   TrackMonocular                          :594               -> ss_track (HIP extraction + HIP
 *                                                                 match + host geometry; a bounded
 *                                                                 monocular front-end)
 *   SendPosePacket                          :225-282           -> send_pose_packet (emitted only
 *                                                                 when tracking_state == OK, :596)
 *   pacing sleep, timing summary            :618-624, :656-664 -> same
 *
 * Extras, all off by default so the binary stays a strict drop-in:
 *   SENDSLAM_EMIT_FEATURES=1   after every frame send {"type":"features", ...} (the reference
 *                              host logs unknown types at debug level and ignores them,
 *                              slam_handler.ex:131-132)
 *   SENDSLAM_ORB_NFEATURES=n   override the 1250 literal (BASELINE.json benches use 2000)
 *   SENDSLAM_DEVICE=k          HIP device ordinal (one backend process per GPU / camera)
 *   SENDSLAM_SHARD=r/2         stereo (BASELINE.json config 4): this process is eye r of a pair of front doors, one per camera / GPU
 *                              (SENDSLAM_DEVICE); after every tracked frame the two exchange their descriptor blocks through
 *                              ss_xchg_* (peer-mapped device memory, rendezvous at SENDSLAM_XCHG_PATH) and each matches its eye
 *                              against the other's: the count goes into the "features" message (stereo_matches) and the log.
 *                              Both eyes must receive frame k before either can answer it (lockstep cameras); frame by frame only
 *   SENDSLAM_TIMING=1          print where the connection's wall time went with the shutdown summary (bench.py "frontdoor")
 *   SENDSLAM_NO_PACING=1       no sleep between frames (:618-624 switched off) AND read-ahead: frames already
 *                              queued on the socket are decoded straight into a pinned slot of an ss_pipe (up to
 *                              SENDSLAM_READAHEAD per batch, default 16), extracted as one batch while the next
 *                              ones are received, then tracked in order (ss_track_features): same poses, same
 *                              messages, in the same order as frame-by-frame ss_track (tests/test_wire.py).
 *                              The pose step and the answers run on a second thread (SENDSLAM_TRACK_THREAD=0: on the
 *                              receiving one): batch k is tracked while batch k + 1 is received, decoded and submitted
 *   --selftest-pose            print the pose packet for fixed values as hex and exit (golden
 *                              wire bytes, tests/test_wire.py; needs no GPU)
 */
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iostream>
#include <mutex>
#include <numeric>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sendslam_orb.h"
#include "ss_msgpack.h"

using namespace std;

namespace {

/* ORB_SLAM3::Tracking::eTrackingState */
enum tracking_state { SYSTEM_NOT_READY = -1, NO_IMAGES_YET = 0, NOT_INITIALIZED = 1, TRACKING_OK = 2, RECENTLY_LOST = 3, LOST = 4 };

struct CameraCalibration {
    string type;
    double fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0;
    int width = 0, height = 0;
    double fps = 0;
    int rgb = 0;
    double stereoThDepth = 0, stereoBaseline = 0, depthMapFactor = 0;
};

struct MessagePacket {
    string type;
    const uint8_t *imageData = nullptr; /* view into the payload (the reference copies, :325) */
    size_t imageSize = 0;
    double timestamp = 0.0;
    bool hasImage = false, hasTimestamp = false;
    int camera_id = 0;
    bool hasCalibrationParameters = false;
    CameraCalibration calibrationParameters;
};

template <typename F> auto require_scalar(const ssmp::value &node, const string &section, const char *key, F conv)
{
    const ssmp::value *v = node.find(key);
    if (!v) throw runtime_error("Calibration section '" + section + "' is missing key '" + key + "'");
    try {
        return conv(*v);
    } catch (const exception &ex) {
        throw runtime_error("Failed to parse key '" + section + "." + key + "': " + string(ex.what()));
    }
}

CameraCalibration parse_camera_calibration(const ssmp::value &obj)
{
    if (obj.t != ssmp::type::MAP) throw runtime_error("Calibration 'camera' field must be a map");
    auto D = [](const ssmp::value &v) { return v.as_double(); };
    auto I = [](const ssmp::value &v) { return v.as_int(); };
    auto S = [](const ssmp::value &v) { return v.as_string(); };
    CameraCalibration c;
    c.type = require_scalar(obj, "camera", "type", S);
    c.fx = require_scalar(obj, "camera", "fx", D);
    c.fy = require_scalar(obj, "camera", "fy", D);
    c.cx = require_scalar(obj, "camera", "cx", D);
    c.cy = require_scalar(obj, "camera", "cy", D);
    c.k1 = require_scalar(obj, "camera", "k1", D);
    c.k2 = require_scalar(obj, "camera", "k2", D);
    c.p1 = require_scalar(obj, "camera", "p1", D);
    c.p2 = require_scalar(obj, "camera", "p2", D);
    c.width = require_scalar(obj, "camera", "width", I);
    c.height = require_scalar(obj, "camera", "height", I);
    c.fps = require_scalar(obj, "camera", "fps", D);
    c.rgb = require_scalar(obj, "camera", "rgb", I);
    c.stereoThDepth = require_scalar(obj, "camera", "th_depth", D);
    c.stereoBaseline = require_scalar(obj, "camera", "baseline", D);
    c.depthMapFactor = require_scalar(obj, "camera", "depth_map_factor", D);
    return c;
}

CameraCalibration parse_calibration(const ssmp::value &obj)
{
    if (obj.t != ssmp::type::MAP) throw runtime_error("Calibration payload must be a map");
    if (const ssmp::value *cam = obj.find("camera")) return parse_camera_calibration(*cam);
    return parse_camera_calibration(obj);
}

bool parse_message(const ssmp::value &root, MessagePacket &packet)
{
    if (root.t != ssmp::type::MAP) throw runtime_error("MessagePack payload must be a map at the top level");
    for (const auto &kv : root.map) {
        const string key = kv.first.as_string();
        const ssmp::value &value = kv.second;
        if (key == "type") packet.type = value.as_string();
        else if (key == "calibration" || key == "calibration_params") {
            packet.calibrationParameters = parse_calibration(value);
            packet.hasCalibrationParameters = true;
        } else if (key == "timestamp") {
            packet.timestamp = value.as_double();
            packet.hasTimestamp = true;
        } else if (key == "image" || key == "frame") {
            if (value.t != ssmp::type::BIN) throw runtime_error("Image data must be encoded as MessagePack bin");
            packet.imageData = value.data;
            packet.imageSize = value.size;
            packet.hasImage = true;
        } else if (key == "camera_id") packet.camera_id = value.as_int();
        /* other fields are ignored (:332-335) */
    }
    return !packet.type.empty();
}

/* What BuildCalibrationYaml would have written; logged, not parsed */
string settings_text(const CameraCalibration &c, const ss_orb_params &p)
{
    ostringstream o;
    o << "Camera.type: \"" << c.type << "\"\n"
      << "Camera1.fx: " << c.fx << "\nCamera1.fy: " << c.fy << "\nCamera1.cx: " << c.cx << "\nCamera1.cy: " << c.cy << "\n"
      << "Camera1.k1: " << c.k1 << "\nCamera1.k2: " << c.k2 << "\nCamera1.p1: " << c.p1 << "\nCamera1.p2: " << c.p2 << "\n"
      << "Camera.width: " << c.width << "\nCamera.height: " << c.height << "\nCamera.fps: " << c.fps << "\n"
      << "Camera.RGB: " << c.rgb << "\nStereo.ThDepth: " << c.stereoThDepth << "\nStereo.b: " << c.stereoBaseline << "\n"
      << "RGBD.DepthMapFactor: " << c.depthMapFactor << "\n"
      << "ORBextractor.nFeatures: " << p.n_features << "\nORBextractor.scaleFactor: " << p.scale_factor << "\n"
      << "ORBextractor.nLevels: " << p.n_levels << "\nORBextractor.iniThFAST: " << p.ini_th_fast << "\n"
      << "ORBextractor.minThFAST: " << p.min_th_fast << "\n";
    return o.str();
}

/* cv::imdecode(IMREAD_UNCHANGED) for the two encodings the host sends (PPM from
 * Evision.imencode(".ppm"), slam_handler.ex:275-281; PGM for gray frames).  Returns false
 * if the buffer is not a binary 8-bit PNM ("Failed to decode frame image data.", :547-551). */
bool pnm_header(const uint8_t *p, size_t n, int &w, int &h, int &channels, size_t &data_off)
{
    size_t i = 0;
    auto token = [&](long &out) -> bool {
        for (;;) {
            while (i < n && isspace(p[i])) i++;
            if (i < n && p[i] == '#') {
                while (i < n && p[i] != '\n') i++;
                continue;
            }
            break;
        }
        if (i >= n || !isdigit(p[i])) return false;
        long v = 0;
        while (i < n && isdigit(p[i])) {
            v = v * 10 + (p[i] - '0');
            if (v > 100000000) return false;
            i++;
        }
        out = v;
        return true;
    };
    if (n < 7 || p[0] != 'P' || (p[1] != '5' && p[1] != '6')) return false;
    channels = p[1] == '6' ? 3 : 1;
    i = 2;
    long lw, lh, maxv;
    if (!token(lw) || !token(lh) || !token(maxv)) return false;
    if (lw <= 0 || lh <= 0 || lw > 16384 || lh > 16384 || maxv <= 0 || maxv > 255) return false;
    if (i >= n || !isspace(p[i])) return false;
    i++; /* single whitespace after maxval */
    const size_t need = (size_t)lw * lh * channels;
    if (n - i < need) return false;
    w = (int)lw;
    h = (int)lh;
    data_off = i;
    return true;
}

/* PNM raster -> the layout a cv::Mat holds (P6 stores R,G,B; a cv::Mat is B,G,R), rows of dst_stride bytes */
void pnm_copy(const uint8_t *src, int w, int h, int channels, uint8_t *dst, size_t dst_stride)
{
    const size_t row = (size_t)w * channels;
    for (int y = 0; y < h; y++) {
        const uint8_t *s = src + (size_t)y * row;
        uint8_t *d = dst + (size_t)y * dst_stride;
        if (channels == 1) memcpy(d, s, row);
        else
            for (size_t k = 0; k < row; k += 3) {
                d[k] = s[k + 2];
                d[k + 1] = s[k + 1];
                d[k + 2] = s[k];
            }
    }
}

struct pose {
    double px, py, pz, qx, qy, qz, qw; /* Twc translation + unit quaternion, world-from-camera */
};

vector<uint8_t> build_pose_packet(const pose &T, double timestamp, int cameraId, int trackingState)
{
    ssmp::packer pk;
    pk.pack_map(6);
    pk.pack("type");           pk.pack("pose");
    pk.pack("timestamp");      pk.pack(timestamp);
    pk.pack("camera_id");      pk.pack(cameraId);
    pk.pack("tracking_state"); pk.pack(trackingState);
    pk.pack("position");
    pk.pack_map(3);
    pk.pack("x"); pk.pack(T.px);
    pk.pack("y"); pk.pack(T.py);
    pk.pack("z"); pk.pack(T.pz);
    pk.pack("orientation");
    pk.pack_map(4);
    pk.pack("x"); pk.pack(T.qx);
    pk.pack("y"); pk.pack(T.qy);
    pk.pack("z"); pk.pack(T.qz);
    pk.pack("w"); pk.pack(T.qw);
    return pk.buf;
}

bool write_all(int fd, const uint8_t *p, size_t n)
{
    while (n) {
        const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k <= 0) return false;
        p += k;
        n -= (size_t)k;
    }
    return true;
}

bool send_framed(int fd, const vector<uint8_t> &payload)
{
    const uint32_t len = (uint32_t)payload.size();
    const uint8_t header[4] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len};
    return write_all(fd, header, 4) && write_all(fd, payload.data(), payload.size());
}

bool send_pose_packet(int fd, const pose &T, double timestamp, int cameraId, int trackingState)
{
    if (!send_framed(fd, build_pose_packet(T, timestamp, cameraId, trackingState))) {
        cerr << "Failed to send pose packet: socket write failed" << endl;
        return false;
    }
    return true;
}

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return atoi(v);
}

} // namespace

int main(int argc, char **argv)
{
    if (argc > 1 && string(argv[1]) == "--selftest-pose") {
        const pose T{0.25, -1.5, 3.0, 0.0, 0.7071067811865476, 0.0, 0.7071067811865476};
        for (uint8_t b : build_pose_packet(T, 12.5, 1, TRACKING_OK)) printf("%02x", b);
        printf("\n");
        return 0;
    }

    const char *portEnv = getenv("ORB_SLAM3_WS_PORT");
    if (portEnv == nullptr) {
        cerr << "ORB_SLAM3_WS_PORT environment variable is not set." << endl;
        return 1;
    }
    int port = 0;
    try {
        port = stoi(portEnv);
    } catch (const exception &e) {
        cerr << "Failed to parse ORB_SLAM3_WS_PORT: " << e.what() << endl;
        return 1;
    }
    if (port <= 0 || port > 65535) {
        cerr << "ORB_SLAM3_WS_PORT must be a valid TCP port (1-65535)." << endl;
        return 1;
    }

    ss_orb_params params;
    ss_orb_params_default(&params);
    params.n_features = env_int("SENDSLAM_ORB_NFEATURES", params.n_features);
    const int device = env_int("SENDSLAM_DEVICE", 0);
    const bool emitFeatures = env_int("SENDSLAM_EMIT_FEATURES", 0) != 0;
    int shardRank = -1, shardWorld = 0;
    if (const char *sh = getenv("SENDSLAM_SHARD")) {
        if (sscanf(sh, "%d/%d", &shardRank, &shardWorld) != 2 || shardWorld != 2 || shardRank < 0 || shardRank >= shardWorld) {
            cerr << "SENDSLAM_SHARD must be r/2 (stereo: eye r of two); ignoring '" << sh << "'" << endl;
            shardRank = -1;
        }
    }
    ss_xchg *xchg = nullptr;
    int stereoMatches = -1;

    ss_ctx *ctx = nullptr;
    vector<float> trackSeconds;
    double lastFrameStamp = -1.0;

    cout << endl << "-------" << endl;
    cout << "Connecting to tcp://127.0.0.1:" << port << " ..." << endl;

    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_port = htons((uint16_t)port);
    inet_pton(AF_INET, "127.0.0.1", &addr.sin_addr);
    if (fd < 0 || ::connect(fd, (sockaddr *)&addr, sizeof(addr)) != 0) {
        cerr << "TCP message processing failed: connect: " << strerror(errno) << endl;
        return 1;
    }
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));

    /* returns 1 ok, 0 clean EOF before any byte, -1 error / EOF mid-message */
    auto readExact = [fd](uint8_t *dst, size_t length) -> int {
        size_t total = 0;
        while (total < length) {
            const ssize_t k = ::recv(fd, dst + total, length - total, 0);
            if (k == 0) return total == 0 ? 0 : -1;
            if (k < 0) {
                if (errno == EINTR) continue;
                return -1;
            }
            total += (size_t)k;
        }
        return 1;
    };

    constexpr size_t kMaxMessageSize = 50 * 1024 * 1024; /* 50 MB safety guard (:412) */
    bool haveCalibration = false;
    atomic<int> exitCode{0};
    vector<uint8_t> payload, pix;

    /* what TrackMonocular's outputs turn into on the wire (:596-616) */
    auto emit_tracked = [&](const ss_pose &tracked, int camera_id, double timestamp) {
        const int trackingState = tracked.tracking_state;
        if (trackingState == TRACKING_OK) { /* :596: a pose is shipped only while tracking is OK */
            const pose T{tracked.position[0], tracked.position[1], tracked.position[2], tracked.quaternion[0],
                         tracked.quaternion[1], tracked.quaternion[2], tracked.quaternion[3]};
            send_pose_packet(fd, T, timestamp, camera_id, trackingState);
        }
        if (emitFeatures) {
            ssmp::packer pk;
            pk.pack_map(stereoMatches >= 0 ? 9 : 8);
            if (stereoMatches >= 0) { pk.pack("stereo_matches"); pk.pack(stereoMatches); }
            pk.pack("type");           pk.pack("features");
            pk.pack("timestamp");      pk.pack(timestamp);
            pk.pack("camera_id");      pk.pack(camera_id);
            pk.pack("tracking_state"); pk.pack(trackingState);
            pk.pack("n_keypoints");    pk.pack((int)tracked.n_keypoints);
            pk.pack("n_matches");      pk.pack((int)tracked.n_matches);
            pk.pack("n_inliers");      pk.pack((int)tracked.n_inliers);
            pk.pack("n_map_points");   pk.pack((int)tracked.n_map_points);
            send_framed(fd, pk.buf);
        }
    };

    /* Read-ahead (SENDSLAM_NO_PACING=1): queued frames go through an ss_pipe in batches; poses come out in frame order. */
    const bool noPacing = env_int("SENDSLAM_NO_PACING", 0) != 0;
    const int readAhead = noPacing && shardRank < 0 ? max(1, min(64, env_int("SENDSLAM_READAHEAD", 16))) : 1;
    ss_pipe *pipe = nullptr;
    ss_camera pipeCam{};
    int pipeW = 0, pipeH = 0, pipeCh = 0;
    ss_pipe_slot openSlot{};
    int openN = 0, batchesInFlight = 0;
    vector<int32_t> openCams;
    vector<double> openStamps;
    vector<chrono::steady_clock::time_point> submitTimes; /* per batch in flight, oldest first */

    /* SENDSLAM_TIMING=1: where the wall time of the connection goes, printed with the shutdown summary (seconds):
     * recv = blocked in recv() for message bytes, parse = MessagePack decode, decode = PNM payload -> pinned slot / frame
     * buffer, submit = ss_pipe_submit, wait = blocked in ss_pipe_wait (GPU not done yet), track = ss_track_features /
     * ss_track, send = pose / features packets */
    const bool timing = env_int("SENDSLAM_TIMING", 0) != 0;
    double tRecv = 0, tParse = 0, tDecode = 0, tSubmit = 0, tWait = 0, tTrack = 0, tSend = 0;
    const auto tStart = chrono::steady_clock::now();
    auto secs_since = [](chrono::steady_clock::time_point t0) {
        return chrono::duration_cast<chrono::duration<double>>(chrono::steady_clock::now() - t0).count();
    };
    /* A pipe call that fails for a reason other than a bad frame (a HIP error: the GPU or its driver is in trouble) ends
     * the process with a non-zero code, so that the supervisor starts a fresh backend (docker_handler.ex:117-145 stops itself
     * on a dead container for exactly that).  Carrying on would mean polling slots that can never complete. */
    atomic<bool> pipeFailed{false};
    auto pipe_fatal = [&](const char *what) {
        cerr << "GPU pipeline failed (" << what << "): " << ss_pipe_last_error(pipe) << " -- exiting for a supervised restart" << endl;
        pipeFailed = true;
        exitCode = 3;
    };
    /* Read-ahead runs on two threads: this one receives, decodes and submits; the tracker takes completed batches in
     * order, runs the pose step and sends the answers.  They share the count of submitted-but-unanswered batches and
     * their submit times (under qm); everything else has one writer: the pipe's producer calls are made here, its
     * consumer calls and ss_track_features (ctx) there, and this thread touches ctx only with nothing in flight. */
    const bool trackThread = readAhead > 1 && env_int("SENDSLAM_TRACK_THREAD", 1) != 0;
    mutex qm;
    condition_variable qcv;
    bool trackerStop = false, trackerBusy = false;
    long batchesAnswered = 0;
    auto in_flight = [&]() {
        lock_guard<mutex> g(qm);
        return batchesInFlight;
    };
    /* takes the oldest completed batch (blocking), tracks its frames in order, ships poses; false = the pipe has failed */
    auto finish_batch_now = [&]() -> bool {
        ss_pipe_result r{};
        chrono::steady_clock::time_point submitted;
        {
            lock_guard<mutex> g(qm);
            if (pipeFailed || batchesInFlight <= 0 || submitTimes.empty()) return false;
            submitted = submitTimes.front();
        }
        const auto tw = chrono::steady_clock::now();
        const int wrc = ss_pipe_wait(pipe, &r);
        tWait += secs_since(tw);
        if (wrc != SS_OK) {
            pipe_fatal("ss_pipe_wait");
            lock_guard<mutex> g(qm);
            qcv.notify_all();
            return false;
        }
        const double extractShare = chrono::duration_cast<chrono::duration<double>>(chrono::steady_clock::now() - submitted).count() / max(1, r.n_frames);
        bool prevTracked = false; /* frame i - 1 of this batch went through the pose step */
        for (int i = 0; i < r.n_frames; i++) {
            if (r.status[i] != SS_OK) {
                cerr << "Frame skipped: extraction failed (status " << r.status[i] << ")" << endl; /* bad frame => log + skip */
                prevTracked = false;
                continue;
            }
            const auto t1 = chrono::steady_clock::now();
            ss_pose tracked{};
            /* the slot's batch matcher has matched frame i against frame i - 1 (match_mode 1): the pose step takes those when
             * frame i - 1 is the frame it saw last, and the slot's rows stay put until the slot is released below */
            const bool haveMatches = r.match_idx && prevTracked;
            prevTracked = false;
            const int rc = ss_track_features_matched(ctx, r.camera_id[i], r.timestamp[i],
                                                     (const uint8_t *)r.d_descriptors + (size_t)i * r.kp_capacity * SS_DESC_BYTES,
                                                     r.keypoints + (size_t)i * r.kp_capacity, r.n_keypoints[i],
                                                     haveMatches ? r.match_idx + (size_t)i * r.kp_capacity : nullptr,
                                                     haveMatches ? r.match_d1 + (size_t)i * r.kp_capacity : nullptr,
                                                     i + 1 < r.n_frames ? SS_TRACK_DESC_STAYS_VALID : 0, &tracked);
            if (rc != SS_OK) {
                cerr << "Frame skipped: " << ss_last_error(ctx) << endl;
                continue;
            }
            prevTracked = true;
            tTrack += secs_since(t1);
            const auto ts1 = chrono::steady_clock::now();
            emit_tracked(tracked, r.camera_id[i], r.timestamp[i]);
            tSend += secs_since(ts1);
            const double ttrack = chrono::duration_cast<chrono::duration<double>>(chrono::steady_clock::now() - t1).count();
            trackSeconds.push_back((float)(ttrack + extractShare));
        }
        ss_pipe_release(pipe, r.slot);
        {
            lock_guard<mutex> g(qm);
            submitTimes.erase(submitTimes.begin());
            batchesInFlight--;
            batchesAnswered++;
        }
        qcv.notify_all();
        return true;
    };
    thread tracker;
    if (trackThread)
        tracker = thread([&]() {
            while (true) {
                {
                    unique_lock<mutex> l(qm);
                    qcv.wait(l, [&] { return trackerStop || (batchesInFlight > 0 && !pipeFailed); });
                    if (trackerStop) return;
                    trackerBusy = true;
                }
                finish_batch_now();
                {
                    lock_guard<mutex> g(qm);
                    trackerBusy = false;
                }
                qcv.notify_all();
            }
        });
    /* joins the tracker on every way out of main (after the loop, or an early return) */
    struct at_exit {
        function<void()> f;
        ~at_exit() { f(); }
    } stopTracker{[&]() {
        if (!tracker.joinable()) return;
        {
            lock_guard<mutex> g(qm);
            trackerStop = true;
        }
        qcv.notify_all();
        tracker.join();
    }};
    /* one more batch answered (or none left to wait for); false = the pipe has failed or nothing is in flight */
    auto finish_batch = [&]() -> bool {
        if (!trackThread) return finish_batch_now();
        unique_lock<mutex> l(qm);
        if (pipeFailed || batchesInFlight <= 0) return false;
        const long seen = batchesAnswered;
        qcv.wait(l, [&] { return pipeFailed || batchesAnswered != seen; });
        return !pipeFailed;
    };
    auto submit_open = [&]() {
        if (!pipe || openN == 0) return;
        /* test hook: SENDSLAM_TEST_PIPE_FAIL_BATCH=n makes the n-th submission of this process fail half-way */
        static int submissions = 0;
        if (env_int("SENDSLAM_TEST_PIPE_FAIL_BATCH", -1) == submissions++) ss_pipe_debug_inject_failure(pipe, 4);
        if (pipeFailed) {
        } else if ([&] { const auto t0 = chrono::steady_clock::now(); const int rc = ss_pipe_submit(pipe, openSlot.slot, openN, openCams.data(), openStamps.data()); tSubmit += secs_since(t0); return rc; }() != SS_OK) {
            pipe_fatal("ss_pipe_submit");
        } else {
            {
                lock_guard<mutex> g(qm);
                batchesInFlight++;
                submitTimes.push_back(chrono::steady_clock::now());
            }
            qcv.notify_all();
        }
        openN = 0;
        openCams.clear();
        openStamps.clear();
    };
    /* everything received so far is tracked and answered before the caller goes on (other message types, EOF, idle socket) */
    auto drain_pipe = [&]() {
        submit_open();
        while (pipe && in_flight() > 0 && finish_batch()) {}
    };
    auto destroy_pipe = [&]() {
        if (!pipeFailed) drain_pipe();
        if (trackThread) { /* a failed pipe is not drained: the tracker may still be inside the batch it took before the failure */
            unique_lock<mutex> l(qm);
            qcv.wait(l, [&] { return !trackerBusy; });
        }
        if (pipe) ss_pipe_destroy(pipe);
        pipe = nullptr;
    };
    auto input_queued = [&]() {
        pollfd pf{fd, POLLIN, 0};
        return ::poll(&pf, 1, 0) > 0 && (pf.revents & POLLIN);
    };

    cout << "Connection established. Awaiting calibration parameters..." << endl;

    while (true) {
        if (pipe && (openN > 0 || in_flight() > 0) && !input_queued()) drain_pipe(); /* idle socket: answer now */
        if (pipeFailed) break;
        uint8_t lengthBuffer[4];
        const auto tr0 = chrono::steady_clock::now();
        const int r = readExact(lengthBuffer, 4);
        tRecv += secs_since(tr0);
        if (r <= 0) drain_pipe();
        if (r == 0) {
            cout << "Connection closed by server." << endl;
            break;
        }
        if (r < 0) {
            cerr << "TCP message processing failed: Unexpected EOF while reading from TCP socket" << endl;
            exitCode = 1;
            break;
        }
        const uint32_t messageLength = ((uint32_t)lengthBuffer[0] << 24) | ((uint32_t)lengthBuffer[1] << 16) |
                                       ((uint32_t)lengthBuffer[2] << 8) | (uint32_t)lengthBuffer[3];
        if (messageLength == 0) {
            cerr << "Received empty MessagePack payload. Skipping." << endl;
            continue;
        }
        if (messageLength > kMaxMessageSize) {
            cerr << "Message exceeds safety limit (" << messageLength << " bytes)." << endl;
            destroy_pipe();
            if (ctx) ss_destroy(ctx);
            return 1;
        }
        payload.resize(messageLength);
        const auto tr1 = chrono::steady_clock::now();
        const int gotPayload = readExact(payload.data(), payload.size());
        tRecv += secs_since(tr1);
        if (gotPayload != 1) {
            cerr << "Connection closed before full message was received." << endl;
            break;
        }

        MessagePacket packet;
        ssmp::value root;
        const auto tp0 = chrono::steady_clock::now();
        try {
            root = ssmp::decoder(payload.data(), payload.size()).parse();
            if (!parse_message(root, packet)) {
                cerr << "Ignoring MessagePack payload without 'type' field." << endl;
                continue;
            }
        } catch (const exception &ex) {
            cerr << "Failed to parse MessagePack payload: " << ex.what() << endl;
            continue;
        }

        tParse += secs_since(tp0);
        if (packet.type != "frame") drain_pipe(); /* messages are answered in order */
        if (pipeFailed) break;
        if (packet.type == "terminate" || packet.type == "shutdown") {
            cout << "Received termination request from server." << endl;
            break;
        }

        if (packet.type == "calibration") {
            if (!packet.camera_id) {
                cerr << "Calibration message missing camera identifier." << endl;
                continue;
            }
            if (!packet.hasCalibrationParameters) {
                cerr << "Calibration message missing structured parameter payload." << endl;
                continue;
            }
            /* a second calibration rebuilds the whole system (:491-518) */
            destroy_pipe();
            if (ctx) {
                ss_destroy(ctx);
                ctx = nullptr;
            }
            const int rc = ss_create(device, &params, &ctx);
            if (rc != SS_OK) {
                /* the reference's System ctor would throw into the outer try (:636-650): exit 1.
                 * No GPU means no service: there is no CPU path to fall back to. */
                cerr << "TCP message processing failed: " << ss_last_error(nullptr) << endl;
                ::close(fd);
                return 1;
            }
            const CameraCalibration &c = packet.calibrationParameters;
            ss_camera cam{};
            snprintf(cam.type, sizeof(cam.type), "%s", c.type.c_str());
            cam.fx = c.fx; cam.fy = c.fy; cam.cx = c.cx; cam.cy = c.cy;
            cam.k1 = c.k1; cam.k2 = c.k2; cam.p1 = c.p1; cam.p2 = c.p2;
            cam.width = c.width; cam.height = c.height; cam.fps = c.fps; cam.rgb = c.rgb;
            cam.th_depth = c.stereoThDepth; cam.baseline = c.stereoBaseline; cam.depth_map_factor = c.depthMapFactor;
            if (ss_set_calibration(ctx, packet.camera_id, &cam) != SS_OK) {
                cerr << ss_last_error(ctx) << endl;
                continue;
            }
            pipeCam = cam;
            if (env_int("SENDSLAM_PRINT_SETTINGS", 0)) cout << settings_text(c, params);
            haveCalibration = true;
            trackSeconds.clear();
            lastFrameStamp = -1.0;
            cout << "Calibration parameters received. SLAM system ready to process frames." << endl;
            continue;
        }

        if (packet.type == "frame") {
            if (!haveCalibration) {
                cerr << "Received frame before calibration. Ignoring." << endl;
                continue;
            }
            if (!packet.camera_id) {
                cerr << "Frame message missing camera identifier." << endl;
                continue;
            }
            if (!packet.hasImage || packet.imageSize == 0) {
                cerr << "Frame message missing binary image data." << endl;
                continue;
            }
            if (!packet.hasTimestamp) {
                cerr << "Frame message missing timestamp." << endl;
                continue;
            }
            int w = 0, h = 0, ch = 0;
            size_t pnmOff = 0;
            if (!pnm_header(packet.imageData, packet.imageSize, w, h, ch, pnmOff)) {
                cerr << "Failed to decode frame image data." << endl;
                continue;
            }
            if (!ctx) {
                cerr << "SLAM system is not initialized. Skipping frame." << endl;
                continue;
            }

            if (readAhead > 1) {
                /* decode straight into a pinned slot; the batch goes out when it is full or the socket has run dry */
                if (pipe && (w != pipeW || h != pipeH || ch != pipeCh)) destroy_pipe();
                if (!pipe) {
                    ss_pipe_config cfg{};
                    cfg.width = w; cfg.height = h; cfg.channels = ch;
                    cfg.batch = readAhead; cfg.depth = 3;
                    /* frame b against frame b - 1 inside a batch, by the pose step's own rule: one launch per batch instead of
                     * one match + two copies + two waits per frame (SENDSLAM_BATCH_MATCH=0: the pose step matches) */
                    cfg.match_mode = env_int("SENDSLAM_BATCH_MATCH", 1) ? 1 : -1;
                    cfg.match_th = 50; cfg.ratio_num = 9; cfg.ratio_den = 10;
                    /* a P6 raster is R,G,B on the wire and a B,G,R cv::Mat after imdecode (:546), which Camera.RGB then
                     * labels; the slots keep the wire order (rows copied as they are, no per-pixel swap on this thread) and
                     * the pipe's gray conversion is told the opposite label: the same weights on the same bytes */
                    ss_camera slotCam = pipeCam;
                    if (ch == 3) slotCam.rgb = pipeCam.rgb ? 0 : 1;
                    if (ss_pipe_create(device, &params, &slotCam, &cfg, &pipe) != SS_OK) {
                        cerr << "Frame skipped: " << ss_pipe_last_error(nullptr) << endl;
                        pipe = nullptr;
                        continue;
                    }
                    pipeW = w; pipeH = h; pipeCh = ch;
                }
                if (openN == 0) {
                    int arc;
                    while ((arc = ss_pipe_acquire(pipe, &openSlot)) == SS_ERR_BUSY)
                        if (!finish_batch()) break; /* every slot busy and nothing can complete: the pipe has failed */
                    if (arc != SS_OK && !pipeFailed) pipe_fatal("ss_pipe_acquire");
                    if (pipeFailed) break;
                }
                const auto td0 = chrono::steady_clock::now();
                {
                    const size_t row = (size_t)w * ch;
                    const uint8_t *src = packet.imageData + pnmOff;
                    uint8_t *dst = openSlot.pixels + (size_t)openN * openSlot.frame_stride;
                    if ((size_t)openSlot.row_stride == row) memcpy(dst, src, row * h);
                    else
                        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * openSlot.row_stride, src + (size_t)y * row, row);
                }
                tDecode += secs_since(td0);
                openCams.push_back(packet.camera_id);
                openStamps.push_back(packet.timestamp);
                openN++;
                if (openN == readAhead || !input_queued()) submit_open();
                if (pipeFailed) break;
                lastFrameStamp = packet.timestamp;
                continue;
            }

            pix.resize((size_t)w * h * ch);
            const auto td0 = chrono::steady_clock::now();
            pnm_copy(packet.imageData + pnmOff, w, h, ch, pix.data(), (size_t)w * ch);
            tDecode += secs_since(td0);
            const auto t1 = chrono::steady_clock::now();
            /* TrackMonocular :594 -> Twc + tracking state :596 */
            ss_pose tracked{};
            const int rc = ss_track(ctx, packet.camera_id, pix.data(), w, h, ch, w * ch, packet.timestamp, &tracked);
            if (rc != SS_OK) {
                cerr << "Frame skipped: " << ss_last_error(ctx) << endl; /* bad frame => log + skip */
                continue;
            }
            if (shardRank >= 0) {
                /* config 4: exchange this frame's descriptors with the other eye's front door, match across the eyes */
                if (!xchg) {
                    const char *xp = getenv("SENDSLAM_XCHG_PATH");
                    if (ss_xchg_create(device, shardRank, shardWorld, 4 << 20, xp ? xp : "/tmp/sendslam_stereo.sock", env_int("SENDSLAM_XCHG_TIMEOUT_MS", 20000), &xchg) != SS_OK) {
                        cerr << "Stereo exchange unavailable: " << ss_xchg_last_error(nullptr) << endl;
                        shardRank = -1;
                    }
                }
                stereoMatches = -1;
                if (xchg) {
                    static vector<int32_t> sidx(65536); /* >= kp_capacity of any geometry (<= 16 levels x 2048 per level) */
                    int32_t nOwn = 0, nPeer = 0;
                    const int src = ss_stereo_exchange_match(ctx, xchg, 1 - shardRank, 50, 9, 10, sidx.data(), nullptr, nullptr, &nOwn, &nPeer);
                    if (src != SS_OK) {
                        cerr << "Stereo exchange failed: " << ss_last_error(ctx) << endl;
                        ss_xchg_destroy(xchg);
                        xchg = nullptr;
                        shardRank = -1;
                    } else {
                        stereoMatches = 0;
                        for (int i = 0; i < nOwn && i < (int)sidx.size(); i++) stereoMatches += sidx[(size_t)i] >= 0;
                        cout << "stereo: " << stereoMatches << " of " << nOwn << " keypoints matched in the other eye (" << nPeer << " there)" << endl;
                    }
                }
            }
            tTrack += secs_since(t1);
            const auto ts1 = chrono::steady_clock::now();
            emit_tracked(tracked, packet.camera_id, packet.timestamp);
            tSend += secs_since(ts1);
            const auto t2 = chrono::steady_clock::now();
            const double ttrack = chrono::duration_cast<chrono::duration<double>>(t2 - t1).count();
            trackSeconds.push_back((float)ttrack);

            if (lastFrameStamp > 0.0) { /* never outrun the timestamps (:618-624) */
                const double interval = packet.timestamp - lastFrameStamp;
                if (ttrack < interval && !noPacing) usleep((useconds_t)((interval - ttrack) * 1e6));
            }
            lastFrameStamp = packet.timestamp;
            continue;
        }

        cerr << "Received MessagePack with unsupported type: '" << packet.type << "'." << endl;
    }

    destroy_pipe();
    if (xchg) ss_xchg_destroy(xchg);
    ::shutdown(fd, SHUT_RDWR);
    ::close(fd);

    if (ctx) {
        if (!trackSeconds.empty()) {
            sort(trackSeconds.begin(), trackSeconds.end());
            const float totaltime = accumulate(trackSeconds.begin(), trackSeconds.end(), 0.0f);
            cout << "-------" << endl;
            cout << "Frames processed: " << trackSeconds.size() << endl;
            cout << "median tracking time: " << trackSeconds[trackSeconds.size() / 2] << endl;
            cout << "mean tracking time: " << totaltime / trackSeconds.size() << endl;
            if (timing)
                cout << "timing: wall " << secs_since(tStart) << " recv " << tRecv << " parse " << tParse << " decode " << tDecode << " submit " << tSubmit
                     << " wait " << tWait << " track " << tTrack << " send " << tSend << endl;
        } else {
            cout << "No frames processed." << endl;
        }
        ss_destroy(ctx);
    }
    return exitCode;
}
