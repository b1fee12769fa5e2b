"""Drop-in boundary tests: wire-format golden bytes and the front-door binary against a fake
host that plays SlamHandler's role (the counterpart of the reference's
missing "fake backend").  CPU tests need no GPU; the end-to-end frame test is marked gpu.

Reference: protocol produced at send_slam/lib/send_slam/slam_handler.ex:140-156,189-230,283-291,
consumed at slam_backends/orb_slam_3/orbslam3_mono_networked.cc:284-339,423-630; pose packet
:225-282 (157-byte payload, header 0000009d).
"""
import os
import socket
import struct
import subprocess
import threading

import msgpack
import numpy as np
import pytest

from send_slam_amd import backend, synth, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRONTDOOR = backend.FRONTDOOR


@pytest.fixture(scope="module")
def frontdoor():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "send-slam_amd"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "send-slam_amd", "frontdoor"), "-s"])
    return FRONTDOOR


def test_pose_packet_golden_bytes(frontdoor):
    hexed = subprocess.check_output([frontdoor, "--selftest-pose"], text=True).strip()
    got = bytes.fromhex(hexed)
    want = wire.pose_packet(12.5, 1, 2, (0.25, -1.5, 3.0), (0.0, 0.7071067811865476, 0.0, 0.7071067811865476))
    assert len(got) == 157 and want[:4] == bytes.fromhex("0000009d")  # SURVEY.md section 8(c) pin
    assert got == want[4:]
    pose = wire.handle_incoming_packet(got)
    assert pose["type"] == "pose" and pose["camera_id"] == 1 and pose["tracking_state"] == 2
    assert set(pose) == {"type", "timestamp", "camera_id", "tracking_state", "position", "orientation"}
    assert pose["position"] == {"x": 0.25, "y": -1.5, "z": 3.0} and pose["orientation"]["w"] == 0.7071067811865476


def test_host_packets_shape_and_framing():
    dims = {"width": 640, "height": 480, "channels": 3}
    k = [[500.0, 0, 320.0], [0, 510.0, 240.0], [0, 0, 1]]
    pkt = wire.build_calibration_packet(k, [0.1, -0.2, 0.001, 0.002, 0.5], dims, camera_id=4, fps=30)
    (n,) = struct.unpack(">I", pkt[:4])
    assert n == len(pkt) - 4
    m = msgpack.unpackb(pkt[4:], raw=False)
    cam = m["calibration"]["camera"]
    assert m["type"] == "calibration" and m["camera_id"] == 4
    assert (cam["fx"], cam["fy"], cam["cx"], cam["cy"]) == (500.0, 510.0, 320.0, 240.0)
    assert (cam["k1"], cam["k2"], cam["p1"], cam["p2"]) == (0.1, -0.2, 0.001, 0.002)
    assert cam["rgb"] == 1 and cam["th_depth"] == 40.0 and cam["type"] == "PinHole" and len(cam) == 16
    ppm = wire.encode_to_ppm(np.zeros((720, 1280, 3), np.uint8))
    assert len(ppm) == 2764816  # 1280x720x3 PPM on the wire
    fp = wire.build_frame_packet(ppm[:100], dims, camera_id=4, timestamp=1.5)
    fm = msgpack.unpackb(fp[4:], raw=False)
    assert fm["type"] == "frame" and isinstance(fm["frame"], bytes) and fm["encoding"] == "ppm"
    pk, rest = wire.extract_packets(pkt + fp + fp[:7])
    assert pk == [pkt[4:], fp[4:]] and rest == fp[:7]
    assert wire.extract_packets(b"\x00\x00")[0] == []


class FakeHost:
    """Plays ThousandIsland + SlamHandler: listens, sends packets, collects what comes back."""

    def __init__(self):
        self.srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.srv.bind(("127.0.0.1", 0))
        self.srv.listen(1)
        self.port = self.srv.getsockname()[1]
        self.conn = None
        self.inbound = []
        self._buf = b""

    def accept(self, timeout=60):
        self.srv.settimeout(timeout)
        self.conn, _ = self.srv.accept()
        self.conn.settimeout(timeout)

    def send(self, data: bytes):
        self.conn.sendall(data)

    def recv_packets(self, n, timeout=60):
        self.conn.settimeout(timeout)
        while len(self.inbound) < n:
            chunk = self.conn.recv(65536)
            if not chunk:
                break
            pk, self._buf = wire.extract_packets(self._buf + chunk)
            self.inbound.extend(msgpack.unpackb(p, raw=False) for p in pk)
        return self.inbound

    def close(self):
        for s in (self.conn, self.srv):
            try:
                if s:
                    s.close()
            except OSError:
                pass


def run_backend(host, env=None):
    e = {"SENDSLAM_NO_PACING": "1"}
    e.update(env or {})
    b = backend.HipBackend(port=host.port, env=e)
    assert b.status() == {"state": "initial", "container_id": None, "last_seen": None}
    tag, cid = b.start_container()
    assert tag == "ok" and b.start_container() == ("ok", cid)  # idempotent like docker_handler.ex:82-84
    return b


def test_frontdoor_requires_port_env(frontdoor):
    env = {k: v for k, v in os.environ.items() if k != "ORB_SLAM3_WS_PORT"}
    r = subprocess.run([frontdoor], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "ORB_SLAM3_WS_PORT environment variable is not set." in r.stderr
    r = subprocess.run([frontdoor], env=dict(env, ORB_SLAM3_WS_PORT="70000"), capture_output=True, text=True)
    assert r.returncode == 1 and "must be a valid TCP port" in r.stderr


def test_frontdoor_log_and_skip_policy_without_gpu_work(frontdoor):
    """Everything the shim does before it needs the GPU: skip empty / malformed / typeless /
    unknown messages, refuse frames before calibration, stop on "terminate"."""
    host = FakeHost()
    b = run_backend(host)
    try:
        host.accept()
        host.send(struct.pack(">I", 0))                                   # empty payload -> skipped
        host.send(struct.pack(">I", 3) + b"\xc1\xc1\xc1")                 # not MessagePack -> skipped
        host.send(wire.encode_payload({"camera_id": 1}))                  # no 'type' -> ignored
        host.send(wire.encode_payload({"type": "bogus"}))                 # unsupported type
        host.send(wire.encode_payload({"type": "frame", "camera_id": 1, "timestamp": 0.0, "frame": b"P5"}))
        host.send(wire.encode_payload({"type": "frame", "camera_id": 1, "timestamp": 0.0, "frame": "text"}))
        host.send(wire.build_calibration_packet(np.eye(3), [0, 0, 0, 0], {"width": 8, "height": 8, "channels": 1}, camera_id=0))
        host.send(wire.encode_payload({"type": "calibration", "camera_id": 1}))
        host.send(wire.encode_payload({"type": "calibration", "camera_id": 1, "calibration": {"camera": {"fx": 1.0}}}))
        host.send(wire.build_terminate_packet())
        rc = b.wait(timeout=60)
    finally:
        host.close()
    tag, text = b.logs(200)
    assert tag == "ok" and rc == 0, text
    for line in ["Received empty MessagePack payload. Skipping.", "Failed to parse MessagePack payload",
                 "Ignoring MessagePack payload without 'type' field.",
                 "Received MessagePack with unsupported type: 'bogus'.",
                 "Received frame before calibration. Ignoring.",
                 "Image data must be encoded as MessagePack bin",
                 "Calibration message missing camera identifier.",
                 "Calibration message missing structured parameter payload.",
                 "Calibration section 'camera' is missing key 'type'",
                 "Received termination request from server."]:
        assert line in text, line
    assert b.status()["state"] == "exited"


def test_frontdoor_oversize_message_is_fatal(frontdoor):
    host = FakeHost()
    b = run_backend(host)
    try:
        host.accept()
        host.send(struct.pack(">I", 50 * 1024 * 1024 + 1))
        rc = b.wait(timeout=60)
    finally:
        host.close()
    assert rc == 1 and "Message exceeds safety limit" in b.logs()[1]


def test_frontdoor_without_gpu_fails_loudly_at_calibration(frontdoor):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    host = FakeHost()
    b = run_backend(host)
    try:
        host.accept()
        dims = {"width": 320, "height": 240, "channels": 1}
        host.send(wire.build_calibration_packet(np.eye(3), [0, 0, 0, 0], dims))
        rc = b.wait(timeout=60)
    finally:
        host.close()
    assert rc == 1 and "no CPU path" in b.logs()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("extra_env", [{"SENDSLAM_READAHEAD": "1"}, {"SENDSLAM_READAHEAD": "4"},
                                       {"SENDSLAM_READAHEAD": "4", "SENDSLAM_TRACK_THREAD": "0"},
                                       {"SENDSLAM_READAHEAD": "4", "SENDSLAM_BATCH_MATCH": "0"}],
                         ids=["frame_by_frame", "read_ahead", "read_ahead_one_thread", "read_ahead_own_match"])
def test_frontdoor_end_to_end_frames(frontdoor, oracle, extra_env):
    """Config 1 of BASELINE.json: fake host <-> front door, calibration then frames of a parallax
    sequence.  Every frame goes through ss_track on the GPU; a pose packet is sent exactly for the
    frames whose tracking state is OK (shim :596) and equals the all-CPU pipeline's pose; the optional
    'features' message carries the counts of that pipeline.  read_ahead: the same stream (run_backend sets
    SENDSLAM_NO_PACING=1) with SENDSLAM_READAHEAD=4 instead of 1 -- queued frames are decoded into pinned slots of an ss_pipe, extracted in batches of up
    to 4 and tracked in order, on a second thread, by ss_track_features_matched with the matches the slot's batch matcher
    made (the colour frame changes the geometry mid-stream: the pipe is drained and rebuilt): same messages in the same
    order.  read_ahead_one_thread: the pose step on the receiving thread; read_ahead_own_match: the pose step matches
    frame by frame on the device (no batch matches)."""
    import track_ref
    from oracle import vo_oracle as vo
    w, h, seed = 640, 480, 77
    sc = synth.scene(seed, w, h)
    frames = [synth.parallax_frame(seed, w, h, t, sc=sc) for t in range(6)]
    col = synth.color_frame(41, w, h)
    host = FakeHost()
    b = run_backend(host, dict({"SENDSLAM_EMIT_FEATURES": "1"}, **extra_env))
    gray = lambda c: oracle.gray(np.ascontiguousarray(c), 1)  # PPM is R,G,B on the wire, a BGR Mat after decode, rgb: 1
    want = track_ref.run(oracle, frames + [col], vo.Camera(500, 500, 320, 240), 1250, gray=gray)
    n_expected = len(want) + sum(o["state"] == 2 for o in want)
    try:
        host.accept()
        dims = {"width": w, "height": h, "channels": 1}
        host.send(wire.build_calibration_packet([[500, 0, 320], [0, 500, 240], [0, 0, 1]], [0, 0, 0, 0], dims))
        for t, f in enumerate(frames):
            host.send(wire.build_frame_packet(wire.encode_to_ppm(f), dims, camera_id=1, timestamp=1.0 + t / 30))
        host.send(wire.build_frame_packet(b"P6\n2 2\n255\n", dims, camera_id=1, timestamp=2.0))  # truncated -> skipped
        host.send(wire.build_frame_packet(wire.encode_to_ppm(col), dict(dims, channels=3), camera_id=1, timestamp=2.1))
        msgs = host.recv_packets(n_expected)
        host.send(wire.build_terminate_packet())
        rc = b.wait(timeout=60)
    finally:
        host.close()
    text = b.logs(200)[1]
    assert rc == 0, text
    assert "Failed to decode frame image data." in text and "Frames processed: 7" in text
    assert "median tracking time:" in text and "mean tracking time:" in text
    stamps = [1.0 + t / 30 for t in range(6)] + [2.1]
    k = 0
    n_pose = 0
    for o, ts in zip(want, stamps):
        if o["state"] == 2:
            m = msgs[k]
            k += 1
            n_pose += 1
            pose = wire.handle_incoming_packet(msgpack.packb(m))  # what SlamHandler broadcasts on PoseRegistry
            assert pose["type"] == "pose" and pose["camera_id"] == 1 and pose["tracking_state"] == 2
            assert pose["timestamp"] == ts
            got_p = np.array([pose["position"][a] for a in "xyz"])
            got_q = np.array([pose["orientation"][a] for a in "xyzw"])
            assert np.allclose(got_p, o["position"], rtol=0, atol=1e-6) and np.allclose(got_q, o["quaternion"], rtol=0, atol=1e-6)
        m = msgs[k]
        k += 1
        assert m["type"] == "features" and wire.handle_incoming_packet(msgpack.packb(m)) is None  # ignored by the host
        assert (m["tracking_state"], m["n_keypoints"], m["n_matches"], m["n_inliers"], m["n_map_points"]) == \
               (o["state"], o["n_keypoints"], o["n_matches"], o["n_inliers"], o["n_map_points"]), (ts, m, o)
    assert k == len(msgs) and n_pose >= 3
    assert want[-1]["state"] == 4  # the unrelated colour frame loses tracking: no pose for it


@pytest.mark.gpu
def test_frontdoor_exits_nonzero_when_the_gpu_pipeline_fails(frontdoor):
    """read-ahead: a pipe call that fails for a reason other than a bad frame (here: an injected half-way failure of the second
    batch's submission) must end the process with a non-zero code for a supervised restart -- never spin on busy slots"""
    w, h = 640, 480
    frames = [synth.parallax_frame(77, w, h, t) for t in range(6)]
    host = FakeHost()
    b = run_backend(host, {"SENDSLAM_READAHEAD": "2", "SENDSLAM_TEST_PIPE_FAIL_BATCH": "1"})
    try:
        host.accept()
        dims = {"width": w, "height": h, "channels": 1}
        host.send(wire.build_calibration_packet([[500, 0, 320], [0, 500, 240], [0, 0, 1]], [0, 0, 0, 0], dims))
        blob = b"".join(wire.build_frame_packet(wire.encode_to_ppm(f), dims, camera_id=1, timestamp=1.0 + t / 30) for t, f in enumerate(frames))
        try:
            host.send(blob)
        except OSError:
            pass  # the backend may be gone before everything is written
        rc = b.wait(timeout=60)
    finally:
        host.close()
    text = b.logs(200)[1]
    assert rc == 3, text
    assert "GPU pipeline failed (ss_pipe_submit): injected failure" in text and "supervised restart" in text


def test_docker_cli_shim_drives_the_frontdoor_like_dockerhandler(frontdoor, tmp_path):
    """The four docker invocations of SendSlam.DockerHandler (docker_handler.ex:117-182), answered by
    frontdoor/run_frontdoor.sh: run -d --rm --name ... -e K=V IMAGE / inspect -f ... / logs --tail / rm -f."""
    sh = os.path.join(ROOT, "send-slam_amd", "frontdoor", "run_frontdoor.sh")
    env = dict(os.environ, SENDSLAM_RUN_DIR=str(tmp_path))
    host = FakeHost()
    try:
        run = subprocess.run([sh, "run", "-d", "--rm", "--name", "net-orbslam", "--network=host",
                              "-e", f"ORB_SLAM3_WS_PORT={host.port}", "-e", "ORBSLAM3_MAP_PATH=/tmp/x", "net-orbslam"],
                             capture_output=True, text=True, env=env)
        assert run.returncode == 0 and len(run.stdout.strip()) == 64  # an id, like `docker run -d`
        host.accept()
        ins = subprocess.run([sh, "inspect", "-f", "{{.State.Running}}", "net-orbslam"], capture_output=True, text=True, env=env)
        assert ins.stdout.startswith("true")
        dup = subprocess.run([sh, "run", "-d", "--rm", "--name", "net-orbslam", "--network=host", "net-orbslam"],
                             capture_output=True, text=True, env=env)
        assert dup.returncode != 0 and "already in use" in dup.stderr
        host.send(wire.encode_payload({"type": "bogus"}))
        import time
        time.sleep(0.5)
        logs = subprocess.run([sh, "logs", "--tail", "50", "net-orbslam"], capture_output=True, text=True, env=env)
        assert "Connection established. Awaiting calibration parameters..." in logs.stdout
        assert "unsupported type: 'bogus'" in logs.stdout
        rm = subprocess.run([sh, "rm", "-f", "net-orbslam"], capture_output=True, text=True, env=env)
        assert rm.returncode == 0
        ins = subprocess.run([sh, "inspect", "-f", "{{.State.Running}}", "net-orbslam"], capture_output=True, text=True, env=env)
        assert ins.stdout.startswith("false")
    finally:
        subprocess.run([sh, "rm", "-f", "net-orbslam"], capture_output=True, env=env)
        host.close()


@pytest.mark.gpu
def test_two_front_doors_as_a_stereo_pair_exchange_and_match(frontdoor, tmp_path):
    """BASELINE.json config 4 from the hosts the reference has: two front door processes (SENDSLAM_SHARD=0/2 and 1/2, here both on
    GPU 0), one fake SlamHandler each.  The right eye sees the left eye's frames 24 px further left; after every frame the two
    processes exchange descriptor blocks through ss_xchg_* and match across the eyes: the features message carries the count."""
    w, h = 640, 480
    sc = synth.scene(4242, w + 64, h)
    m = synth._MARGIN
    eyes = [[np.clip(sc[m + 2 * t:m + 2 * t + h, m + 24 * e + 3 * t:m + 24 * e + 3 * t + w], 0, 255).astype(np.uint8) for t in range(4)] for e in range(2)]
    hosts = [FakeHost(), FakeHost()]
    xp = str(tmp_path / "stereo.sock")
    backs = [run_backend(hosts[e], {"SENDSLAM_SHARD": f"{e}/2", "SENDSLAM_XCHG_PATH": xp, "SENDSLAM_EMIT_FEATURES": "1"}) for e in range(2)]
    try:
        dims = {"width": w, "height": h, "channels": 1}
        for e in range(2):
            hosts[e].accept()
            hosts[e].send(wire.build_calibration_packet([[500, 0, 320], [0, 500, 240], [0, 0, 1]], [0, 0, 0, 0], dims, camera_id=e + 1))
        for t in range(4):  # lockstep cameras
            for e in range(2):
                hosts[e].send(wire.build_frame_packet(wire.encode_to_ppm(eyes[e][t]), dims, camera_id=e + 1, timestamp=1.0 + t / 30))
        msgs = []
        for e in range(2):  # a pose packet comes only while tracking is OK: read until four "features" messages are in
            while sum(x.get("type") == "features" for x in hosts[e].inbound) < 4:
                hosts[e].recv_packets(len(hosts[e].inbound) + 1)
            msgs.append([x for x in hosts[e].inbound if x.get("type") == "features"][:4])
        for e in range(2):
            hosts[e].send(wire.build_terminate_packet())
        rcs = [b.wait(timeout=60) for b in backs]
    finally:
        for hst in hosts:
            hst.close()
    for e in range(2):
        text = backs[e].logs(200)[1]
        assert rcs[e] == 0, text
        assert len(msgs[e]) == 4, text
        for x in msgs[e]:
            assert x["stereo_matches"] > 0.5 * x["n_keypoints"] > 100, (e, x, text)
        assert "keypoints matched in the other eye" in text
