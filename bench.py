#!/usr/bin/env python3
"""bench.py -- frames/s of ORB extract + self-match on synthetic 1280x720 frames, 2000 kp/frame
(BASELINE.json metric, config "1xMI355X: synthetic 1280x720 frames, ORB extract + self-match").

A "step" is one pass of the hot path over ONE BATCH of frames already resident in HBM:
ss_extract_batch_device (pyramid, FAST + NMS, quadtree, orientation, blur, rBRIEF) followed by
ss_match_batch_device (self-match, j == i excluded).  With N GPUs every rank runs the same step
on its own camera batch (cameras shard one per GPU, no data-path collective): weak scaling,
value = frames all ranks processed / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel: ALGORITHMIC bytes per launch / HIP-event mean duration,
                measured on the context's own stream inside the timed region
  kernels       the same for every stage
  cpu_baseline  the CPU oracle (a port: the reference's ORB-SLAM3 cannot be built here,
                DESIGN.md) timed on this host on a bounded sample of the same frames, 1 thread
                like the reference shim (orbslam3_mono_networked.cc:594), plus all cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "send-slam_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

INT8_MFMA_PEAK_OPS = 5.0e15  # dense int8 MFMA, 2 x the 2.5e15 dense bf16 peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def make_frames(rank, batch, w, h):
    """batch frames = ceil(batch/8) scenes x 8 time steps (consecutive frames move by (3,-2) px)."""
    from send_slam_amd import synth
    out = np.empty((batch, h, w), np.uint8)
    n_scenes = (batch + 7) // 8
    i = 0
    for s in range(n_scenes):
        seed = 1000 * rank + s
        sc = synth.scene(seed, w, h)
        for t in range(8):
            if i >= batch:
                break
            out[i] = synth.frame_from_scene(sc, seed, w, h, t)
            i += 1
    return out


def _cpu_one(args):
    from oracle import orb_oracle as O
    frame, nf = args
    p = O.default_params(n_features=nf)
    t0 = time.perf_counter()
    kps, desc, _ = O.extract(frame, p)
    idx, d1, d2 = O.match(desc, desc, 50, 9, 10, exclude_self=True)
    return time.perf_counter() - t0, kps, desc, idx, d1, d2


def cpu_baseline(frames, nf, budget_s=12.0):
    """Oracle timed on host cores BEFORE the GPU is touched (a process pool forks).
    Returns the JSON object and the per-frame oracle outputs for the parity spot-check."""
    import multiprocessing as mp
    from oracle import orb_oracle as O
    O.build()
    times, outs = [], []
    _cpu_one((frames[0], nf))  # warm-up (page in, first malloc)
    t_start = time.perf_counter()
    for i in range(len(frames)):
        r = _cpu_one((frames[i], nf))
        times.append(r[0])
        outs.append(r[1:])
        if time.perf_counter() - t_start > budget_s / 2 and len(times) >= 5:
            break
    times.sort()
    median = times[len(times) // 2]  # the shim's median rule (orbslam3_mono_networked.cc:661)
    cores = len(os.sched_getaffinity(0))
    all_cores = None
    if cores > 1:
        n_jobs = min(len(frames), max(cores, int(cores * (budget_s / 2) / max(median, 1e-3))))
        n_jobs = min(n_jobs, 4 * cores)
        jobs = [(frames[i % len(frames)], nf) for i in range(n_jobs)]
        with mp.get_context("fork").Pool(cores) as pool:
            t0 = time.perf_counter()
            pool.map(_cpu_one, jobs, chunksize=1)
            all_cores = n_jobs / (time.perf_counter() - t0)
    obj = {"value": round(1.0 / median, 3), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{len(times)} of the bench's own 1280x720 frames, extract + self-match, median per frame "
                     f"{median * 1e3:.1f} ms, 1 thread (oracle/orb_oracle.c, -O3)",
           "all_cores_value": None if all_cores is None else round(all_cores, 2), "all_cores": cores}
    return obj, outs


def bench_loop_closure(a):
    """Config 5 of BASELINE.json: query-vs-all Hamming match against a keyframe database partitioned
    in contiguous slabs over the ranks; one broadcast + one all_gather per query (multi.py)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from send_slam_amd import binding, multi
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    backend = os.environ.get("SENDSLAM_BENCH_BACKEND", "nccl")
    if os.environ.get("SENDSLAM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    n_db, nq = 10000 * 2000, a.features
    b, e = multi.slab(n_db, world, rank)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    db = torch.randint(0, 256, (e - b, 32), dtype=torch.uint8, device=dev, generator=gen)
    query = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=gen)
    ctx = binding.OrbContext(local_rank)
    local = multi.hip_local_match(ctx)
    for _ in range(a.warmup):
        multi.loop_closure_query(query, db, b, local)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        idx, d1, d2 = multi.loop_closure_query(query, db, b, local)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "loop-closure queries/sec (2000 descriptors vs 10k-keyframe database)", "value": round(a.steps / elapsed, 3),
            "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"loop closure: {nq}-descriptor query vs {n_db} descriptors (640 MB) sharded over {world} GPU(s), "
                                   "raw local match + all_gather of (d1, j1, d2) + fold", "parallelism": f"db slabs x {world}"},
            "pairs_per_s": float(f"{nq * n_db * a.steps / elapsed:.4g}")}))
    ctx.close()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--contexts", type=int, default=4, help="camera batches in flight per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="only warm-up + timed steps: no isolated pass, latency or tracking extras (what the rocprofv3 "
                         "--stats runs use, so every launch they average is a full-batch launch of the timed loop)")
    ap.add_argument("--workload", default="extract_match", choices=["extract_match", "loop_closure"],
                    help="extract_match = the BASELINE.json metric (default); loop_closure = config 5: one 2000-descriptor "
                         "query against a 10 000-keyframe descriptor database sharded over the ranks (strong scaling)")
    a = ap.parse_args()
    if a.workload == "loop_closure":
        return bench_loop_closure(a)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    w, h, nf, B = a.width, a.height, a.features, a.batch

    frames = make_frames(rank, B, w, h)

    cpu_obj, cpu_outs = None, []
    if rank == 0 and a.gpus == 1 and not a.no_cpu_baseline:
        cpu_obj, cpu_outs = cpu_baseline(frames, nf)  # before any HIP call in this process

    import torch
    import torch.distributed as dist
    from send_slam_amd import binding

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal hooks for a ONE-GPU box (tests / gpurun): SENDSLAM_BENCH_BACKEND=gloo moves the two
    # timing collectives to the CPU, SENDSLAM_BENCH_ONE_DEVICE=1 puts every rank on device 0.
    backend = os.environ.get("SENDSLAM_BENCH_BACKEND", "nccl")
    if os.environ.get("SENDSLAM_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    # A few camera batches in flight per GPU: step i runs on context i % n (own stream, own HBM
    # buffers), so the latency-bound quadtree of one batch overlaps the dense kernels of the others.
    n_ctx = max(1, a.contexts)
    ctxs = [binding.OrbContext(local_rank, n_features=nf, max_batch=B) for _ in range(n_ctx)]
    ctx = ctxs[0]
    d_frames = torch.from_numpy(frames).to(dev)
    torch.cuda.synchronize()

    # shape the outputs once (kp_capacity is known after the first extraction)
    ctx.extract_batch_device(d_frames.data_ptr(), B, w, h)
    ctx.synchronize()
    view = ctx.batch_view()
    kcap = view.kp_capacity
    outs = [(torch.empty((B, kcap), dtype=torch.int32, device=dev), torch.empty((B, kcap), dtype=torch.int16, device=dev),
             torch.empty((B, kcap), dtype=torch.int16, device=dev)) for _ in range(n_ctx)]
    d_idx = outs[0][0]

    def step(i):
        c, (o_idx, o_d1, o_d2) = ctxs[i % n_ctx], outs[i % n_ctx]
        c.extract_batch_device(d_frames.data_ptr(), B, w, h)
        c.match_batch_device(0, o_idx.data_ptr(), o_d1.data_ptr(), o_d2.data_ptr())

    for i in range(max(a.warmup, n_ctx)):
        step(i)
    for c in ctxs:
        c.synchronize()

    for c in ctxs:
        c.profile(True)
        c.profile_reset()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    for c in ctxs:
        c.synchronize()  # drains the context's stream and checks the per-frame error words
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    for c in ctxs:
        c.profile(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel HIP-event statistics, summed over the contexts
    stats = []
    for c in ctxs:
        for s_ in c.stats():
            m = next((x for x in stats if x["name"] == s_["name"]), None)
            if m is None:
                stats.append(dict(s_))
            else:
                tot = m["total_ms"] + s_["total_ms"]
                n_l = m["launches"] + s_["launches"]
                m.update(total_ms=tot, launches=n_l, mean_ms=tot / n_l if n_l else 0.0)

    # The timed region keeps two batches in flight, so a kernel's live duration includes the other
    # stream's kernels sharing the chip.  A short extra pass on ONE context gives each kernel's
    # duration when it has the GPU to itself (reported next to the live number, never instead).
    iso = {}
    ctx.profile(True)
    ctx.profile_reset()
    for _ in range(0 if a.timed_only else 5):
        ctx.extract_batch_device(d_frames.data_ptr(), B, w, h)
        ctx.match_batch_device(0, outs[0][0].data_ptr(), outs[0][1].data_ptr(), outs[0][2].data_ptr())
        ctx.synchronize()
    for s_ in ctx.stats():
        iso[s_["name"]] = s_["mean_ms"]
    ctx.profile(False)

    # parity spot-check of the measured configuration against the oracle outputs of the baseline leg
    parity = None
    if cpu_outs:
        idx = d_idx.cpu().numpy()
        parity = True
        for b, (okps, odesc, oidx, od1, od2) in enumerate(cpu_outs):
            kps_b, desc_b, _ = ctx.fetch_frame(b)
            n = len(kps_b)
            parity &= n == len(okps) and kps_b.tobytes() == okps.tobytes()
            parity &= bool(np.array_equal(desc_b, odesc)) and bool(np.array_equal(idx[b, :n], oidx))
        if not parity:
            sys.exit("bench.py: GPU results differ from the oracle on the benchmark frames")

    # Latency path of the drop-in boundary (what the NIF / front door call per camera frame):
    # host pixels in, host keypoints + descriptors out, PCIe copies included.  Never `value`.
    lat = []
    for i in range(0 if a.timed_only else 12):
        t1 = time.perf_counter()
        ctx.extract(frames[i % B])
        lat.append(time.perf_counter() - t1)
    lat = sorted(lat[2:])
    single_frame_ms = lat[len(lat) // 2] * 1e3 if lat else None

    # Same boundary one step further (ss_track: extraction + match on the GPU, pose geometry on the
    # host) on a parallax sequence, rank 0 only: per-frame time in tracking state OK.  Never `value`.
    track_ms = track_ok = None
    if rank == 0 and not a.timed_only:
        from send_slam_amd import synth
        cam = binding.Camera(type=b"PinHole", fx=800.0, fy=800.0, cx=w / 2.0, cy=h / 2.0, width=w, height=h, fps=30.0,
                             rgb=1, th_depth=40.0, baseline=0.0, depth_map_factor=1000.0)
        ctx.set_calibration(1, cam)
        sc = synth.scene(1000, w, h)
        tl, states = [], []
        for t in range(14):
            img = synth.parallax_frame(1000, w, h, t, sc=sc)
            t1 = time.perf_counter()
            po = ctx.track(img, 1, t / 30.0)
            dt = time.perf_counter() - t1
            states.append(po["state"])
            if po["state"] == 2 and t > 0 and states[-2] == 2:
                tl.append(dt)
        if tl:
            tl.sort()
            track_ms, track_ok = tl[len(tl) // 2] * 1e3, states.count(2)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    kernels = []
    for s in stats:
        if s["launches"] == 0:
            continue
        gbs = s["algorithmic_bytes"] / (s["mean_ms"] * 1e-3) / 1e9 if s["mean_ms"] > 0 and s["algorithmic_bytes"] else None
        im = iso.get(s["name"])
        kernels.append({"name": s["name"], "launches_per_step": s["launches"] / a.steps, "mean_ms": round(s["mean_ms"], 5),
                        "total_ms": round(s["total_ms"], 3), "algorithmic_bytes_per_launch": s["algorithmic_bytes"],
                        "achieved_GBps": None if gbs is None else round(gbs, 1),
                        "isolated_mean_ms": None if not im else round(im, 5),
                        "isolated_GBps": None if not im or not s["algorithmic_bytes"] else round(s["algorithmic_bytes"] / (im * 1e-3) / 1e9, 1)})
    with_bytes = [k for k in kernels if k["achieved_GBps"] is not None]
    dom = max(with_bytes, key=lambda k: k["total_ms"]) if with_bytes else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if dom and os.path.exists(tpath):
        tj = json.load(open(tpath))
        key = f"{dom['name']}@batch{B}_{w}x{h}_n{nf}"
        traffic = tj.get(key)
    roofline = None
    if dom:
        # `achieved` uses the kernel's duration with one batch in flight (HIP events on the launch stream, the
        # pass above): that is the number rocprofv3's kernel trace reports for the same kernel (profiles/
        # r01_bench_kernel_stats*.csv: 419 us alone, 442 us while the other batches' kernels share the chip).  HIP
        # events recorded while several streams are in flight also count the time a kernel waits for its turn, so the
        # timed-region event mean is reported separately and is not a kernel duration.
        ach = dom["isolated_GBps"] if dom["isolated_GBps"] else dom["achieved_GBps"]
        roofline = {"kernel": dom["name"], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "kernel_ms": dom["isolated_mean_ms"] if dom["isolated_GBps"] else dom["mean_ms"],
                    "timed_region_event_ms": dom["mean_ms"], "timed_region_event_GBps": dom["achieved_GBps"],
                    "note": "integer-VALU-bound kernel (DESIGN.md section 5: 86 % of the measured VALU issue rate): the HBM "
                            "fraction is reported because the contract asks for it, not because HBM limits it"}

    # the Hamming-match kernel (k_match_mfma) against the two pipes it uses: the distance of a
    # (query, train) pair is a 256-term i8 contraction on the matrix cores (512 int8 ops), the best / second-best
    # selection stays on the VALU: per pair one v_lshl_add_u32 (key), one v_med3_u32, one v_min_u32, priced with the
    # measured instruction costs (profiles/r01_valu_rates.json, profiles/tools/valu_rates.hip).  Peak int8 MFMA: 2 x
    # the dense bf16 rate (MI355X_MICROARCH.md "Matrix cores": i8 = bf16 cycles at 2 x K) = 5.0e15 op/s.
    match_roofline = None
    mk = next((k for k in kernels if k["name"] == "match"), None)
    rpath = os.path.join(ROOT, "profiles", "r01_valu_rates.json")
    if mk and mk.get("isolated_mean_ms") and os.path.exists(rpath):
        pairs = B * float(nf) ** 2  # the quotas are saturated on these frames (2000 <= n <= 2024)
        t = mk["isolated_mean_ms"] * 1e-3
        rates = json.load(open(rpath))
        c = rates["cycles_per_wave64_instruction_per_simd"]
        sel = c["v_lshl_add_u32"] + c["v_med3_u32"] + c["v_min_u32"]
        simd_cycles_per_s = rates["cus"] * 4 * rates["clock_mhz"] * 1e6
        valu_floor_ms = pairs / 64 * sel / simd_cycles_per_s * 1e3
        mfma_floor_ms = pairs * 512 / INT8_MFMA_PEAK_OPS * 1e3
        match_roofline = {"kernel": "match", "bound": "mfma", "achieved": float(f"{pairs * 512 / t / 1e12:.4g}"),
                          "peak": INT8_MFMA_PEAK_OPS / 1e12, "unit": "Top/s (int8)", "frac": round(pairs * 512 / t / INT8_MFMA_PEAK_OPS, 4),
                          "kernel_ms": mk["isolated_mean_ms"], "mfma_floor_ms": round(mfma_floor_ms, 4),
                          "valu_select_cycles_per_64_pairs": round(sel, 1), "valu_select_floor_ms": round(valu_floor_ms, 4),
                          "frac_of_the_larger_floor": round(max(valu_floor_ms, mfma_floor_ms) / mk["isolated_mean_ms"], 4),
                          "pairs_per_s": float(f"{pairs / t:.4g}")}

    # the dominant kernel against the resource that bounds it: VALU issue.  Instruction count per launch from the
    # committed PMC pass of this workload (profiles/r01_pmc_sq_mix.json, SQ_INSTS_VALU of a full-batch launch),
    # duration live; peak = the measured issue rate (profiles/r01_peaks.json).
    valu_roofline = None
    ppath = os.path.join(ROOT, "profiles", "r01_peaks.json")
    mix_path = os.path.join(ROOT, "profiles", "r01_pmc_sq_mix.json")
    if dom and dom.get("isolated_mean_ms") and os.path.exists(mix_path) and os.path.exists(ppath) and (B, w, h, nf) == (64, 1280, 720, 2000):
        stage_kernel = {"fast_blur_nms": "k_fast_score", "match": "k_match", "orient_describe": "k_orient_describe"}.get(dom["name"])
        insts = json.load(open(mix_path)).get(stage_kernel, {}).get("SQ_INSTS_VALU")
        if insts:
            peak = json.load(open(ppath))["xor_popc_lane_ops_per_s"]
            ach = insts * 64 / (dom["isolated_mean_ms"] * 1e-3)
            valu_roofline = {"kernel": dom["name"], "bound": "int_valu", "achieved": float(f"{ach:.4g}"), "peak": peak,
                             "unit": "lane-ops/s", "frac": round(ach / peak, 4), "valu_wave_instructions_per_launch": insts}

    total_frames = B * a.steps * world
    out = {
        "metric": "frames/sec ORB extract+match @1280x720, 2000 kp/frame",
        "value": round(total_frames / elapsed, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{world}xMI355X: synthetic {w}x{h} frames, ORB extract + self-match, {nf} kp/frame",
                   "frames_per_step_per_gpu": B, "batches_in_flight_per_gpu": n_ctx, "n_features": nf, "n_levels": 8, "scale_factor": 1.2,
                   "match": "self-match all-pairs, j==i excluded, TH 50, ratio 9/10",
                   "parallelism": f"one camera batch per GPU x {world}, no collective"},
        "roofline": roofline, "valu_roofline": valu_roofline, "match_roofline": match_roofline, "kernels": kernels, "cpu_baseline": cpu_obj, "parity_checked_vs_oracle": parity,
        "single_frame_host_to_host_ms": None if single_frame_ms is None else round(single_frame_ms, 3),
        "track_frame_host_to_host_ms": None if track_ms is None else round(track_ms, 3), "track_ok_frames_of_14": track_ok,
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
