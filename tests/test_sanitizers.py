"""CPU-only sanitizer runs (GPU AddressSanitizer is not available on the pool): the oracle, the
front door's MessagePack decoder and the host geometry of ss_track under ASan + UBSan."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_msgpack_decoder_fuzz_under_asan(tmp_path):
    exe = str(tmp_path / "msgpack_fuzz")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-o", exe,
                           os.path.join(ROOT, "tests/native/msgpack_fuzz.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "decoded=" in out.stdout, out.stdout + out.stderr


def test_track_geometry_under_asan_ubsan(tmp_path):
    """40 frames of a synthetic sequence through sst_tracker: initialisation, two-view BA, tracking with
    wrong matches, anchored triangulation, a lost frame and re-initialisation."""
    exe = str(tmp_path / "track_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-o", exe, os.path.join(ROOT, "tests/native/track_asan.cpp"),
                           os.path.join(ROOT, "send-slam_amd/csrc/ss_track.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "lost=1" in out.stdout, out.stdout + out.stderr[-3000:]


def test_oracle_under_asan_ubsan(tmp_path):
    so = str(tmp_path / "liborb_oracle_asan.so")
    subprocess.check_call(["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-ffp-contract=off", "-std=c11", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "oracle/orb_oracle.c"), "-lm"])
    script = textwrap.dedent(f"""
        import sys, ctypes as C
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'send-slam_amd')!r})
        from oracle import orb_oracle as O
        O._lib = C.CDLL({so!r})
        O._lib.orc_fast_atan2.restype = C.c_float
        O._lib.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        O._lib.orc_ic_angle.restype = C.c_float
        from send_slam_amd import synth
        for (w, h, nf, nl) in [(320, 240, 500, 8), (323, 243, 300, 8), (131, 99, 200, 3), (640, 480, 40, 8)]:
            k, d, c = O.extract(synth.frame(3, w, h), O.default_params(n_features=nf, n_levels=nl))
            O.match(d, d, exclude_self=True)
            O.match(d[:0], d)
        print("clean")
    """)
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env)
    assert out.returncode == 0 and "clean" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
