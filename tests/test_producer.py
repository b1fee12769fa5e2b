"""Replay producer: Python mirror of nif/replay_producer.ex, modelled on
VideoProducer's image-sequence mode (video_producer.ex:212-245, 303-357)."""
import os
import threading

import numpy as np
import pytest

from send_slam_amd import producer, synth


class FakeClock:
    def __init__(self):
        self.t = 100.0
        self.sleeps = []

    def now(self):
        return self.t

    def sleep(self, s):
        self.sleeps.append(s)
        self.t += s


def test_sequence_filenames_like_cap_images():
    f = producer.format_sequence_filename
    assert f("/d/frame_%06d.png", 7) == "/d/frame_000007.png"
    assert f("img%d.pgm", 12) == "img12.pgm"
    assert f("img%3d.pgm", 5) == "img005.pgm"  # the reference pads %Nd with zeros too (video_producer.ex:228-245)
    assert f("a%02d_%d.pgm", 3) == "a03_%d.pgm"  # only the first field is a counter
    assert producer.looks_like_image_sequence("x_%04d.pgm") and not producer.looks_like_image_sequence("movie.mp4")
    with pytest.raises(ValueError):
        f("plain.pgm", 0)


def test_pnm_round_trip(tmp_path):
    g = synth.frame(1, 97, 61)
    c = synth.color_frame(2, 33, 20)
    producer.write_pnm(str(tmp_path / "g.pgm"), g)
    producer.write_pnm(str(tmp_path / "c.ppm"), c)
    assert np.array_equal(producer.read_pnm(str(tmp_path / "g.pgm")), g)
    assert np.array_equal(producer.read_pnm(str(tmp_path / "c.ppm")), c)
    open(tmp_path / "bad.pgm", "wb").write(b"P2\n1 1\n255\n0\n")
    with pytest.raises(ValueError):
        producer.read_pnm(str(tmp_path / "bad.pgm"))


def test_pacing_eof_loop_and_start_index(tmp_path):
    frames = [synth.frame(3, 64, 48, t) for t in range(4)]
    pat = producer.write_sequence(str(tmp_path / "a"), frames)
    clk = FakeClock()
    p = producer.ReplayProducer(pat, fps=25, camera_id=3, calibration={"k": 1}, clock=clk.now, sleep=clk.sleep)
    evs = list(p.events())
    assert len(evs) == 4 and p.stop_reason == "eof"
    assert all(np.array_equal(e["frame"], f) for e, f in zip(evs, frames))
    assert [e["camera_id"] for e in evs] == [3] * 4 and evs[0]["fps"] == 25 and evs[0]["calibration"] == {"k": 1}
    assert np.allclose(np.diff([e["timestamp"] for e in evs]), 0.04) and clk.sleeps == [0.04] * 4
    # loop: rewinds to the first frame, max_frames bounds the replay
    p = producer.ReplayProducer(pat, fps=0, loop=True, clock=clk.now, sleep=clk.sleep)
    evs = list(p.events(max_frames=10))
    assert len(evs) == 10 and p.stop_reason is None
    assert all(np.array_equal(evs[i]["frame"], frames[i % 4]) for i in range(10))
    # a sequence that starts at 1 (cv::CAP_IMAGES accepts 0 or 1)
    pat1 = producer.write_sequence(str(tmp_path / "b"), frames[:2], "f%03d.pgm", start=1)
    evs = list(producer.ReplayProducer(pat1, fps=0, clock=clk.now, sleep=clk.sleep).events())
    assert len(evs) == 2 and np.array_equal(evs[0]["frame"], frames[0])
    with pytest.raises(FileNotFoundError):
        producer.ReplayProducer(str(tmp_path / "none_%d.pgm"))
    with pytest.raises(ValueError):
        producer.ReplayProducer(str(tmp_path / "a" / "frame_000000.pgm"))


def test_warmup_rebroadcasts_the_first_frame(tmp_path):
    frames = [synth.frame(4, 64, 48, t) for t in range(3)]
    pat = producer.write_sequence(str(tmp_path), frames)
    clk = FakeClock()
    p = producer.ReplayProducer(pat, fps=10, warmup_ms=450, clock=clk.now, sleep=clk.sleep)
    evs = list(p.events())
    # 0.1 s ticks for 450 ms -> 5 warm-up copies of frame 0, then the 3 frames
    assert len(evs) == 5 + 3
    assert all(np.array_equal(e["frame"], frames[0]) for e in evs[:6])
    assert np.array_equal(evs[6]["frame"], frames[1]) and np.array_equal(evs[7]["frame"], frames[2])
    # after a loop rewind the warm-up runs again (video_producer.ex:318-321 restarts with is_first_frame? = true)
    p = producer.ReplayProducer(pat, fps=10, warmup_ms=150, loop=True, clock=clk.now, sleep=clk.sleep)
    evs = list(p.events(max_frames=9))
    firsts = [i for i, e in enumerate(evs) if np.array_equal(e["frame"], frames[0])]
    assert firsts == [0, 1, 2, 5, 6, 7]  # two warm-up copies + the frame itself, again after the rewind
    p.set_calibration({"fx": 1})
    assert p.calibration == {"fx": 1}


def test_backend_auto_restart_relaunches(tmp_path):
    """The lifecycle gap of SURVEY 8(f) rank 4: with auto_restart the poll relaunches an exited backend."""
    from send_slam_amd import backend
    script = tmp_path / "fake_backend.sh"
    script.write_text("#!/bin/sh\necho started $ORB_SLAM3_WS_PORT\nexit 0\n")
    os.chmod(script, 0o755)
    b = backend.HipBackend(port=1234, binary=str(script))
    assert b.start_container()[0] == "ok"
    b.wait(timeout=10)
    assert b.poll() == ("error", "container_not_running") and b.status()["state"] == "exited"
    b = backend.HipBackend(port=1234, binary=str(script), auto_restart=True)
    tag, first = b.start_container()
    b.wait(timeout=10)
    tag2, second = b.poll()
    assert (tag, tag2) == ("ok", "ok") and second != first and b.restarts == 1 and b.status()["state"] == "running"
    b.wait(timeout=10)
    assert "started 1234" in b.logs()[1]


@pytest.mark.gpu
def test_two_cameras_replayed_into_two_backends(tmp_path):
    """Multi-camera drive (BASELINE.json config 3 in miniature): two replay producers, two front doors on one
    GPU, each camera gets its own poses."""
    import msgpack
    from test_wire import FakeHost, run_backend
    from send_slam_amd import wire
    w, h = 640, 480
    pats = []
    for cam, seed in ((1, 77), (2, 78)):
        sc = synth.scene(seed, w, h)
        frames = [synth.parallax_frame(seed, w, h, t, sc=sc) for t in range(6)]
        pats.append(producer.write_sequence(str(tmp_path / f"cam{cam}"), frames))
    results = {}

    def one(cam, pat):
        host = FakeHost()
        b = run_backend(host)
        try:
            host.accept()
            dims = {"width": w, "height": h, "channels": 1}
            host.send(wire.build_calibration_packet([[500, 0, 320], [0, 500, 240], [0, 0, 1]], [0, 0, 0, 0], dims, camera_id=cam))
            prod = producer.ReplayProducer(pat, fps=0, camera_id=cam)
            stamps = producer.drive_frontdoor(host.send, prod)
            host.send(wire.build_terminate_packet())
            rc = b.wait(timeout=120)
            msgs = []
            buf = b""
            while True:
                chunk = host.conn.recv(65536)
                if not chunk:
                    break
                pk, buf = wire.extract_packets(buf + chunk)
                msgs.extend(msgpack.unpackb(p, raw=False) for p in pk)
            results[cam] = (rc, stamps, msgs, b.logs(50)[1])
        finally:
            host.close()

    threads = [threading.Thread(target=one, args=(cam, pat)) for cam, pat in zip((1, 2), pats)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    for cam in (1, 2):
        rc, stamps, msgs, log = results[cam]
        assert rc == 0 and "Frames processed: 6" in log, log
        poses = [m for m in msgs if m["type"] == "pose"]
        assert len(poses) >= 3 and all(m["camera_id"] == cam and m["tracking_state"] == 2 for m in poses)
        assert {m["timestamp"] for m in poses} <= set(stamps)
    p1 = [m["position"]["x"] for m in results[1][2] if m["type"] == "pose"]
    assert p1 == sorted(p1, reverse=True) and p1[-1] < 0  # camera 1 slides along -x
