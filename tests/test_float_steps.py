"""Pins the float steps the device executes (send-slam_amd/csrc/ss_float_steps.h, host build)
against real implementations in this container: glibc 2.35 sinf/cosf (the libm of the
reference's ubuntu:22.04 image, /root/reference/dockerfile:1) over EVERY float in
[2^-15, 120), and the oracle's fastAtan2 on 4e6 integer moment pairs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sincosf_and_atan2_restatements(tmp_path, oracle):
    exe = str(tmp_path / "float_steps_pin")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests/native/float_steps_pin.c"),
                           "-L" + os.path.join(ROOT, "oracle"), "-lorb_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lm"])
    out = subprocess.run([exe, "1"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sin_mismatch=0 cos_mismatch=0 atan2_mismatch=0" in out.stdout
