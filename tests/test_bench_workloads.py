"""bench.py's multi-rank workloads rehearsed on ONE card (the driver runs the real multi-GPU benches on an 8-GPU node):
two ranks on device 0, gloo for the collectives (SENDSLAM_BENCH_ONE_DEVICE=1, SENDSLAM_BENCH_BACKEND=gloo).  Checks that
every workload runs end to end through the C ABI, prints the contract's JSON line with a roofline object, and that the
exchanged descriptors are really matched (stereo: the right eye sees the left eye's scene 24 px away)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(n_ranks, *args):
    env = dict(os.environ, SENDSLAM_BENCH_ONE_DEVICE="1", SENDSLAM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks)] + list(args)
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def check_contract(j, n):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config"):
        assert k in j, k
    assert j["n_gpus"] == n and j["value"] > 0 and j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"]
    assert j["roofline"] and j["roofline"]["achieved"] > 0 and 0 < j["roofline"]["frac"] < 1


def test_stereo_workload_two_ranks_one_card():
    j = run_bench(2, "--workload", "stereo", "--steps", "3", "--warmup", "1", "--batch", "4")
    check_contract(j, 2)
    assert j["unit"] == "pairs/s" and j["config"]["backend"] == "gloo"
    assert j["fraction_of_keypoints_matched_across_eyes"] > 0.5


def test_loop_closure_workload_two_ranks_one_card():
    j = run_bench(2, "--workload", "loop_closure", "--steps", "2", "--warmup", "1")
    check_contract(j, 2)
    assert j["unit"] == "queries/s" and j["scaling"] == "strong" and j["pairs_per_s"] > 1e11
    assert any(k["name"] == "match_fold" for k in j["kernels"])


def test_metric_workload_two_ranks_one_card_carries_rooflines():
    j = run_bench(2, "--steps", "8", "--warmup", "4", "--batch", "8", "--contexts", "2", "--no-cpu-baseline")
    check_contract(j, 2)
    assert j["scaling"] == "weak" and j["config"]["frames_per_step_per_gpu"] == 8 and j["rounds"] >= 3 and j["value_spread"]
    assert j["match_roofline"] and j["match_stream_roofline"]["frac"] > 0.3


def run_plain(n, *args, env_extra=None):
    """`python bench.py --gpus N ...` started PLAINLY (no torch.distributed.run): bench.py launches its own ranks"""
    env = dict(os.environ, SENDSLAM_BENCH_ONE_DEVICE="1", SENDSLAM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + list(args), capture_output=True, text=True,
                         env=env, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_invocation_gpus_2_launches_its_own_ranks():
    """the driver's command shape with N = 2 and --steps 20: no launcher, rounds of exactly 20 steps that together time
    >= 0.5 s, value_spread filled, rooflines on the N > 1 line, the rank count the backend itself reports"""
    j = run_plain(2, "--steps", "20", "--warmup", "5", "--no-cpu-baseline")
    check_contract(j, 2)
    assert j["steps"] == 20 and j["rounds"] >= 3 and j["timed_region_s"] >= 0.5
    sp = j["value_spread"]
    assert sp and sp["rounds"] == j["rounds"] and sp["min"] <= sp["median"] <= sp["max"] and abs(sp["median"] - j["value"]) < 1e-6 * j["value"] + 0.02
    assert j["ranks_reported_by_backend"] == {"backend": "gloo", "world_size": 2, "allreduce_of_ones": 2}
    assert j["sustained"] and j["sustained"]["steps"] >= 1000 and j["sustained"]["chunks"]["n"] >= 20
    assert j["valu_roofline"]["frac_of_architectural_peak"] < j["valu_roofline"]["frac"]
    assert j["cpu_baseline"] is None  # rank 0 at N = 1 only


def test_native_exchange_stereo_and_loop_closure_two_ranks_one_card():
    j = run_plain(2, "--workload", "stereo", "--exchange", "native", "--steps", "3", "--warmup", "1", "--batch", "4")
    check_contract(j, 2)
    assert j["config"]["exchange"].startswith("native") and j["fraction_of_keypoints_matched_across_eyes"] > 0.5 and j["value_spread"]["rounds"] >= 3
    j = run_plain(2, "--workload", "loop_closure", "--exchange", "native", "--steps", "2", "--warmup", "1")
    check_contract(j, 2)
    assert j["config"]["exchange"].startswith("native") and any(k["name"] == "match_fold" for k in j["kernels"])


def test_forced_rccl_world1_executes_the_collectives():
    """SENDSLAM_BENCH_FORCE_DIST=1: a real "nccl" (RCCL) process group at world size 1; the loop-closure step's broadcast and
    all_gather_into_tensor run through RCCL on the context's stream"""
    env = dict(os.environ, SENDSLAM_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    env.pop("SENDSLAM_BENCH_BACKEND", None)
    for wl, extra in (("loop_closure", []), ("stereo", ["--batch", "4"])):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", wl, "--steps", "3", "--warmup", "1"] + extra,
                             capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        j = json.loads(next(ln for ln in out.stdout.splitlines() if ln.startswith("{")))
        assert j["config"]["backend"] == "nccl" and j["ranks_reported_by_backend"] == {"backend": "nccl", "world_size": 1, "allreduce_of_ones": 1}
        assert j["value"] > 0 and j["value_spread"]["rounds"] >= 3
