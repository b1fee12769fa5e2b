"""PCIe H2D / D2H rate of this box with pinned host memory: one stream, then several streams at once, and both
directions together.  Denominator of bench.py's host_pipeline numbers.  usage: python profiles/tools/pcie_probe.py"""
import json
import time

import torch

dev = torch.device("cuda:0")
N = 59 << 20  # one batch of 64 1280x720 frames
out = {}
host = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(4)]
devb = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]


def run(n_streams, reps, h2d=True, d2h=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(reps):
        for s in range(n_streams):
            with torch.cuda.stream(streams[s]):
                if h2d:
                    devb[s].copy_(host[s], non_blocking=True)
                if d2h:
                    host[(s + 2) % 4].copy_(devb[(s + 2) % 4], non_blocking=True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return reps * n_streams * N / el / 1e9


run(1, 2)
for ns in (1, 2, 4):
    out[f"h2d_{ns}_streams_GBps"] = round(run(ns, 20), 2)
out["d2h_1_stream_GBps"] = round(run(1, 20, h2d=False, d2h=True), 2)
out["both_directions_2_streams_each_GBps"] = round(run(2, 20, h2d=True, d2h=True), 2)
pageable = torch.empty(N, dtype=torch.uint8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    devb[0].copy_(pageable)
torch.cuda.synchronize()
out["h2d_pageable_GBps"] = round(5 * N / (time.perf_counter() - t0) / 1e9, 2)
# host memcpy rate into pinned memory (what ss_pipe_submit_frames' gather threads do), one thread
src = torch.randint(0, 255, (N,), dtype=torch.uint8)
t0 = time.perf_counter()
for _ in range(5):
    host[0].copy_(src)
out["host_memcpy_to_pinned_1_thread_GBps"] = round(5 * N / (time.perf_counter() - t0) / 1e9, 2)
print(json.dumps(out))
