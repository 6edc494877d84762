#!/usr/bin/env python3
"""depth-maps/sec of the MVSTER cost-volume path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one ``MVS4net.forward`` (B=1, eval, fp32) = one depth map of the workload
"DTU 512x640, 5 views, 4-stage cascade (8/8/4/4 hypotheses)" (BASELINE.json configs[1]) on seeded
synthetic inputs already resident in HBM.  Multi-GPU = independent replicas on disjoint depth maps
(weak scaling, no data-path collective).  Rank 0 prints ONE JSON line.

Without a launcher, ``--gpus N`` (N > 1) starts the N ranks itself (re-executes this file under
torch.distributed.run on 127.0.0.1 with a free port).  ``--stub`` replaces the GPU forward by a trivial CPU
step on the gloo backend: the launch / barrier / MAX-over-ranks / one-line protocol without a GPU
(tests/test_bench_cpu.py).

``--mode train`` times BASELINE.json configs[3] instead: one training step (forward, OT loss, backward, Adam) of the
512x640, 5-view, batch 2 per GPU workload per rank, the whole step captured in one hipGraph; with N > 1 ranks the
gradients go through ``shard.GradBucket`` (one 4.04 MB fp32 all-reduce over RCCL/xGMI per step, inside the graph: the
reference's DDP semantics, train_mvs4.py:389-392).  value = samples/s over all ranks.

Extra objects in the line:
  roofline      the dominant kernel instance of the forward, timed with HIP events on the launch
                stream in an instrumented eager pass inside this script (same inputs, same kernels)
  rooflines     the same for the dominant convolution kernel AND the fused warp/aggregation kernel of each of
                the four cascade stages (the north-star kernel; algorithmic bytes of SURVEY.md section 8d)
  cpu_baseline  the CPU oracle (pure PyTorch restatement of the reference, oracle/) timed on the
                host cores on a bounded sample of the same workload (rank 0, N=1 only): independent worker
                processes in parallel (depth maps are independent), the single-process figure beside it
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SHIPPED = dict(arch_mode="fpn", reg_net="reg2d", num_stage=4, fpn_base_channel=8, reg_channel=8,
               stage_splits=[8, 8, 4, 4], depth_interals_ratio=[0.5, 0.5, 0.5, 1], group_cor=True,
               group_cor_dim=[8, 8, 4, 4], inverse_depth=True, mono=True, attn_temp=2, attn_fuse_d=True)
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 (matrix), dense
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E peak


def load_weights():
    z = np.load(os.path.join(ROOT, "tests", "golden", "g7_checkpoint.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.keys()}


class KernelTimer:
    """HIP-event timing of every conv_mfma / warp_agg launch (instrumented eager pass)."""

    def __init__(self):
        self.records = []

    def install(self):
        import mvster_amd.conv_plan as cp
        import mvster_amd.ops as ops
        from mvster_amd import _lib
        timer = self
        self._orig_conv = cp.ConvLayer.__call__
        self._orig_warp = ops.warp_agg_fwd_cl
        self._orig_warp_sched = ops.warp_agg_fwd_sched_cl
        self._orig_sel = cp.fused_conv11_select
        self._orig_fpn = (ops.fpn_tail_fused, ops.fpn_tail_gather, ops.fpn_lateral_up)

        def conv_call(layer, x, skip=None, skip_mode=0, tiles=None):
            B, Di, Hi, Wi, _ = x.shape
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = timer._orig_conv(layer, x, skip, skip_mode, tiles)
            e1.record()
            bytes_ = 4 * (x.numel() + out.numel() + layer.wpk.numel() + (skip.numel() if skip is not None else 0))
            # the library reports what it dispatched (mvster_last_kernel): no second copy of the dispatch rules here
            timer.records.append((_lib.last_kernel(), e0, e1, layer.flops(B, Di, Hi, Wi), bytes_))
            return out

        def warp_call(ref_cl, src_cl, rt, hypo, G, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = timer._orig_warp(ref_cl, src_cl, rt, hypo, G, *a, **k)
            e1.record()
            bytes_ = 4 * (ref_cl.numel() + src_cl.numel() + hypo.numel() + hypo.numel() * G)
            timer.records.append((_lib.last_kernel(), e0, e1, 0, bytes_))
            return out

        def warp_sched_call(ref_cl, src_cl, rt, G, D, *a, **k):
            # (the stage's hypotheses are computed inside this launch: written instead of read, same byte count)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = timer._orig_warp_sched(ref_cl, src_cl, rt, G, D, *a, **k)
            e1.record()
            if out is not None:
                nh = out[1].numel()
                timer.records.append((_lib.last_kernel(), e0, e1, 0, 4 * (ref_cl.numel() + src_cl.numel() + nh + nh * G)))
            return out

        def select_call(L, t, c0, hypo, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = timer._orig_sel(L, t, c0, hypo, *a, **k)
            e1.record()
            # conv11 + prob + selection: read the 16-channel volume, the skip volume and the hypotheses, write the attention
            # volume and four maps (depth, confidence, inverse bounds)
            B, D, hi, wi, _ = t.shape
            bytes_ = 4 * (t.numel() + c0.numel() + 2 * hypo.numel() + 4 * B * 4 * hi * wi)
            timer.records.append((_lib.last_kernel(), e0, e1, 2 * t.numel() * 9 * 8, bytes_))
            return out

        def fpn_op(idx):
            # the FPN top-down kernels of the two fine levels: algorithmic bytes = every input read once + the output written
            def call(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out = timer._orig_fpn[idx](*a, **k)
                e1.record()
                if out is not None:
                    bytes_ = 4 * (sum(t.numel() for t in a if torch.is_tensor(t)) + out.numel())
                    timer.records.append((_lib.last_kernel() if idx == 0 else ("fpn_tail_gather_lds_kernel<%d>" % a[1].shape[1], "fpn_lateral_up_kernel<16, 72>")[idx - 1],
                                          e0, e1, 0, bytes_))
                return out
            return call

        cp.ConvLayer.__call__ = conv_call
        ops.warp_agg_fwd_cl = warp_call
        ops.warp_agg_fwd_sched_cl = warp_sched_call
        cp.fused_conv11_select = select_call
        ops.fpn_tail_fused, ops.fpn_tail_gather, ops.fpn_lateral_up = fpn_op(0), fpn_op(1), fpn_op(2)

    def remove(self):
        import mvster_amd.conv_plan as cp
        import mvster_amd.ops as ops
        cp.ConvLayer.__call__ = self._orig_conv
        ops.warp_agg_fwd_cl = self._orig_warp
        ops.warp_agg_fwd_sched_cl = self._orig_warp_sched
        cp.fused_conv11_select = self._orig_sel
        ops.fpn_tail_fused, ops.fpn_tail_gather, ops.fpn_lateral_up = self._orig_fpn

    def summary(self):
        agg = {}
        for name, e0, e1, flops, bytes_ in self.records:
            a = agg.setdefault(name, dict(ms=0.0, n=0, flops=0, bytes=0, shapes={}))
            ms = e0.elapsed_time(e1)
            a["ms"] += ms
            a["n"] += 1
            a["flops"] += flops
            a["bytes"] += bytes_
            sh = a["shapes"].setdefault((flops, bytes_), [0.0, 0])       # one entry per distinct layer shape
            sh[0] += ms
            sh[1] += 1
        return agg


def kernel_source_hash():
    """Fingerprint of everything that decides what the kernels do and which one runs: the HIP sources, the
    shared headers and the per-layer tuning table.  profiles/pmc_traffic.json carries the fingerprint it was
    measured at; a PMC figure measured on other kernels is not quoted."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "mvster_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "mvster_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "mvster_amd", "csrc", "*.hpp")) +
                   [os.path.join(ROOT, "mvster_amd", "tuning_gfx950.json")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: one process per GPU under torch.distributed.run on this
    node (the reference starts its ranks the same way, scripts/train_dtu.sh:20 / train_mvs4.py:321-326)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it here
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _cpu_worker(idx, threads, shape, seconds, barrier, queue):
    """One oracle process of the parallel CPU baseline: its own depth maps, ``threads`` intra-op threads."""
    torch.set_num_threads(threads)
    from mvster_amd.synthetic import make_inputs
    from oracle import mvs4_oracle as O
    H, W, N = shape
    oracle = O.OracleMVS4net(**SHIPPED)
    oracle.load_state_dict(load_weights(), strict=True)
    oracle.eval()
    imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=idx)
    try:
        with torch.no_grad():
            oracle(imgs, proj, dv)                          # warm-up
            # a worker that died before this point (out of memory, killed) must not hang the others: the wait times out,
            # which breaks the barrier for everyone, and each survivor measures on its own
            try:
                barrier.wait(timeout=240)
            except Exception:            # threading.BrokenBarrierError
                pass
            n, t0 = 0, time.perf_counter()
            while n == 0 or time.perf_counter() - t0 < seconds:
                oracle(imgs, proj, dv)
                n += 1
            queue.put((idx, n, t0, time.perf_counter()))
    except BaseException as e:           # report instead of leaving the parent to its queue timeout
        queue.put((idx, 0, 0.0, 0.0, repr(e)))
        raise


def cpu_baseline(H, W, N, seed):
    """The oracle on this host's cores: (a) one process, torch intra-op threads swept; (b) cores // threads
    independent worker processes side by side (what a CPU deployment of the reference would do: depth maps are
    independent).  (b) is the reported value."""
    import torch.multiprocessing as mp
    from mvster_amd.synthetic import make_inputs
    from oracle import mvs4_oracle as O
    cores = os.cpu_count() or 1
    oracle = O.OracleMVS4net(**SHIPPED)
    oracle.load_state_dict(load_weights(), strict=True)
    oracle.eval()
    cimgs, cproj, cdv = make_inputs(nviews=N, H=H, W=W, seed=seed)
    sweep = {}
    with torch.no_grad():
        # PyTorch's intra-op pool does not scale to 256 hardware threads on these small convolutions
        # (1 forward took 154 s with 256 threads): sweep a few pool sizes once, keep the fastest
        for t in sorted({min(cores, c) for c in (8, 16, 32)}):
            torch.set_num_threads(t)
            oracle(cimgs, cproj, cdv)                   # warm-up at this pool size
            c0 = time.perf_counter()
            oracle(cimgs, cproj, cdv)
            sweep[t] = time.perf_counter() - c0
        best_t = min(sweep, key=sweep.get)
        torch.set_num_threads(best_t)
        n, c0 = 0, time.perf_counter()
        while n < 8 and (time.perf_counter() - c0) < 6.0:
            oracle(cimgs, cproj, cdv)
            n += 1
        single = n / (time.perf_counter() - c0)
    del oracle
    # (b) process-parallel: throughput per thread is best at the smallest pool, so use 8-thread workers
    threads = min(8, cores)
    nproc = max(1, cores // threads)
    try:
        import psutil
        nproc = max(1, min(nproc, int(psutil.virtual_memory().available // (6 << 30))))     # ~3 GB peak per worker
    except ImportError:
        nproc = min(nproc, 16)
    nproc = min(nproc, 32)
    ctx = mp.get_context("spawn")
    barrier, queue = ctx.Barrier(nproc), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(i, threads, (H, W, N), 10.0, barrier, queue)) for i in range(nproc)]
    for p in procs:
        p.start()
    import queue as _queue
    res, failed, deadline = [], [], time.time() + 600
    try:
        while len(res) + len(failed) < nproc and time.time() < deadline:
            try:
                r = queue.get(timeout=5)
            except _queue.Empty:
                # a worker that was killed outright never reports: count it out once its process is gone
                dead = sum(1 for p in procs if not p.is_alive() and p.exitcode not in (0, None))
                if dead > len(failed):
                    failed += ["worker exited with a non-zero code"] * (dead - len(failed))
                continue
            if r[1] > 0:
                res.append(r)
            else:
                failed.append(r[4] if len(r) > 4 else "no forward completed")
    finally:
        for p in procs:
            p.join(30)
            if p.is_alive():
                p.kill()
    if not res:
        return {"value": round(single, 4), "unit": "depth-maps/s", "cores": best_t, "kind": "port", "processes": 1,
                "sample": "single process only: every worker of the process-parallel run failed (%s)" % "; ".join(failed[:3]),
                "single_process": {"value": round(single, 4), "cores": best_t,
                                   "sweep_s_per_forward": {str(k): round(v, 3) for k, v in sorted(sweep.items())}}}
    total = sum(r[1] for r in res)
    span = max(r[3] for r in res) - min(r[2] for r in res)
    nproc_ok = len(res)
    out_failed = {"workers_failed": len(failed) + (nproc - nproc_ok - len(failed))} if nproc_ok < nproc else {}
    nproc = nproc_ok
    return {"value": round(total / span, 4), "unit": "depth-maps/s", "cores": nproc * threads, "kind": "port",
            "processes": nproc, "threads_per_process": threads, **out_failed,
            "sample": "%d forwards of the same %dx%d %d-view 4-stage workload by %d independent oracle processes x %d "
                      "intra-op threads over %.1f s on a %d-thread host" % (total, H, W, N, nproc, threads, span, cores),
            "single_process": {"value": round(single, 4), "cores": best_t,
                               "sweep_s_per_forward": {str(k): round(v, 3) for k, v in sorted(sweep.items())}}}


def timed_windows(step, steps, shard, min_total_s=0.5, max_windows=41, sync=None):
    """The contract's timed region -- EXACTLY ``steps`` steps between barrier + synchronise pairs, MAX over ranks -- run as
    k >= 3 back-to-back windows (k odd, chosen from the first window so that the windows add up to >= ``min_total_s``; the
    reduced time is the same on every rank, so every rank picks the same k) and reported through the MEDIAN window: a
    20-step window of this workload is 20 ms, too short for one sample to be robust against a box's clock ramp.
    -> (median elapsed, every window's elapsed, every window's (min, max) rank-local elapsed)."""
    sync = sync or torch.cuda.synchronize
    out, spread = [], []
    k = 3
    i = 0
    while i < k:
        sync()
        shard.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        mine = time.perf_counter() - t0
        shard.barrier()
        sync()
        el = shard.max_over_ranks(time.perf_counter() - t0)
        out.append(el)
        spread.append((-shard.max_over_ranks(-mine), shard.max_over_ranks(mine)))
        if i == 0:
            k = int(min(max_windows, max(3, -(-1.1 * min_total_s // max(el, 1e-6)))))       # (10 % margin: the first window is the slowest)
            k += 1 - k % 2
        i += 1
    order = sorted(range(len(out)), key=lambda j: out[j])
    mid = order[len(out) // 2]
    return out[mid], out, spread[mid]


def api_call_measure(args, model, dev, seed, shard):
    """What the UNCHANGED reference driver gets: the plain ``model(imgs, proj_matrices, depth_values)`` call of
    test_mvs4.py:205, one stream, a different sample's tensors at every call, freshly allocated outputs.  Four loops, each
    EXACTLY ``steps`` calls between synchronisations, median window:
      value_api_call          device-resident samples, calls back to back (MVS4net.forward's transparent hipGraph cache)
      ..._eager               the same with ``model.graph_cache = False`` (every call issues its launches: the round-4 API path)
      ..._synced              a device synchronisation after every call (the reference's loop reads the outputs back per sample)
      ..._tocuda              the samples start in host memory and go through the reference's ``tocuda`` (utils.py:59-67:
                              ``.to(torch.device("cuda"))`` per tensor, pageable memory, default DataLoader) before each call
      with_outputs_to_host    device-resident samples, and every call followed by the reference's ``tensor2numpy(outputs)``
                              (utils.py:50-57, test_mvs4.py:208: ``.detach().cpu().numpy().copy()`` of every tensor of the
                              nested dict, 47 MB per 512x640 depth map, the last stage's twice) -- the D2H leg of the loop
      ..._packed              the same through ``mvster_amd.graph.outputs_to_numpy`` (one copy through a pinned buffer)
      ..._packed_needed_keys  ... keeping only what ``save_depth`` reads (depth + photometric_confidence, 6.5 MB)"""
    from mvster_amd.synthetic import make_inputs
    pool_n = 4
    host = [make_inputs(nviews=args.views, H=args.height, W=args.width, seed=seed + 31 * k, batch=args.batch) for k in range(pool_n)]
    pool = [([i.to(dev) for i in im], {k: v.to(dev) for k, v in pr.items()}, d.to(dev)) for im, pr, d in host]
    cnt = [0]
    keep = [None]

    def call():
        im, pr, d = pool[cnt[0] % pool_n]
        cnt[0] += 1
        keep[0] = model(im, pr, d)

    def call_synced():
        call()
        torch.cuda.synchronize()

    def call_tocuda():
        im, pr, d = host[cnt[0] % pool_n]
        cnt[0] += 1
        keep[0] = model([i.to(dev) for i in im], {k: v.to(dev) for k, v in pr.items()}, d.to(dev))

    def call_to_host():
        call()
        o = keep[0]
        keep[0] = {k: ({k2: v2.detach().cpu().numpy().copy() for k2, v2 in v.items()} if isinstance(v, dict)
                       else v.detach().cpu().numpy().copy()) for k, v in o.items()}

    def call_to_host_packed():
        from mvster_amd.graph import outputs_to_numpy
        call()
        keep[0] = outputs_to_numpy(keep[0])

    def call_to_host_needed():
        from mvster_amd.graph import outputs_to_numpy
        call()
        keep[0] = outputs_to_numpy(keep[0], keys=("depth", "photometric_confidence"))

    def run(fn, warm):
        for _ in range(warm):
            fn()
        el, win, _ = timed_windows(fn, args.steps, shard, min_total_s=0.3, max_windows=21)
        return {"value": round(args.steps * args.batch / el, 3), "ms_per_call": round(1e3 * el / args.steps, 4), "windows": len(win)}

    stats0 = dict(model._fwd_cache.stats)
    out = {"unit": "depth-maps/s", "calls_per_window": args.steps, "samples_in_rotation": pool_n}
    graphed = run(call, max(3, args.warmup))
    out.update(value=graphed["value"], ms_per_call=graphed["ms_per_call"], windows=graphed["windows"])
    out["synced_every_call"] = run(call_synced, 2)
    out["from_host_tocuda"] = run(call_tocuda, 2)
    out["from_host_tocuda"]["h2d_MB_per_call"] = round(sum(i.numel() for i in host[0][0]) * 4 / 1e6, 2)
    out["with_outputs_to_host"] = run(call_to_host, 2)
    o = keep[0]
    out["with_outputs_to_host"]["d2h_MB_per_call"] = round(sum(a.nbytes for k, v in o.items() for a in (v.values() if isinstance(v, dict) else [v])) / 1e6, 2)
    out["with_outputs_to_host_packed"] = run(call_to_host_packed, 2)
    out["with_outputs_to_host_packed_needed_keys"] = run(call_to_host_needed, 2)
    stats1 = dict(model._fwd_cache.stats)
    out["cache"] = {k: stats1[k] - stats0[k] for k in stats1}
    # one replayed call against the eager forward on the same sample: the cache must not change a bit
    im, pr, d = pool[0]
    a = model(im, pr, d)
    b = model.forward_eager(im, pr, d)
    out["bit_identical_to_eager"] = bool(all(torch.equal(a["stage%d" % s][k], b["stage%d" % s][k]) for s in range(1, 5)
                                             for k in ("depth", "attn_weight", "photometric_confidence")))
    model.graph_cache = False
    try:
        out["eager"] = run(call, 3)
    finally:
        model.graph_cache = True
    out["speedup_over_eager"] = round(out["value"] / out["eager"]["value"], 3)
    return out


def timing_note(steps, windows, rank_span):
    """How `value` was timed: the windows (each EXACTLY `steps` steps between barrier + synchronise pairs, MAX over ranks),
    which one is reported, and the fastest / slowest rank's own time for that window (stragglers show here)."""
    return {"windows": len(windows), "reported": "median window", "steps_per_window": steps,
            "window_ms": [round(1e3 * w, 3) for w in windows], "total_timed_s": round(sum(windows), 4),
            "rank_ms_per_step_min": round(1e3 * rank_span[0] / steps, 4), "rank_ms_per_step_max": round(1e3 * rank_span[1] / steps, 4)}


def rank_local_s(step, steps):
    """This rank's own wall time for ``steps`` more steps (collectives inside the step keep the ranks in lock-step, so
    with a gradient all-reduce the per-rank figures differ only by what each rank does outside it)."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def stub_main(args):
    """The launch / timing / one-line protocol on CPU tensors over gloo (no GPU, no model): what the multi-process
    CPU test drives.  A step is a small matmul."""
    from mvster_amd import shard
    rank, local_rank, world = shard.init_distributed(backend="gloo")
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d: launched with WORLD_SIZE=%d" % (args.gpus, world))
    x = torch.randn(64, 64)
    step = lambda: (x @ x).sum().item()     # noqa: E731
    if args.mode == "train":
        # the training protocol on a toy module: backward, ONE bucketed all-reduce (shard.GradBucket), SGD update
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 1))
        opt = torch.optim.SGD(net.parameters(), lr=1e-2)
        bucket = shard.GradBucket(net.parameters())
        xb = torch.randn(8, 16, generator=torch.Generator().manual_seed(rank))

        def step():
            opt.zero_grad(set_to_none=True)
            net(xb).square().mean().backward()
            bucket.sync()
            opt.step()
    for _ in range(args.warmup):
        step()
    elapsed, windows, rank_span = timed_windows(step, args.steps, shard, min_total_s=0.02, sync=lambda: None)
    ranks_seen = int(round(shard.sum_over_ranks(1.0)))
    if ranks_seen != world:
        raise SystemExit("bench.py: %d ranks answered the collective, WORLD_SIZE=%d" % (ranks_seen, world))
    # (same reporting as the GPU line: every rank's own rate in rank order, one kernel fingerprint on all ranks)
    r0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    per_rank = shard.gather_over_ranks(args.steps / max(time.perf_counter() - r0, 1e-9))
    rank_ids = shard.gather_over_ranks(float(rank))
    hashes = shard.gather_over_ranks(float(int(kernel_source_hash()[:12], 16)))
    if len(set(hashes)) != 1:
        raise SystemExit("bench.py: the ranks run different kernel sources")
    spread = None
    if args.mode == "train":
        # after the same number of averaged updates every rank holds the same parameters (collectives: every rank calls)
        digest = float(sum(p.detach().double().sum() for p in net.parameters()))
        spread = shard.max_over_ranks(digest) + shard.max_over_ranks(-digest)
    if rank == 0:
        line = {"metric": "stub steps/s", "value": round(args.steps * world / elapsed, 3), "unit": "steps/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
                "scaling": "weak", "ranks_seen": ranks_seen, "stub": True, "mode": args.mode,
                "per_rank_value": [round(v, 3) for v in per_rank], "rank_order": [int(r) for r in rank_ids],
                "timing": timing_note(args.steps, windows, rank_span)}
        if spread is not None:
            line["param_digest_spread"] = spread
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


class TrainKernelTimer:
    """HIP-event timing of the weight-gradient and warp-backward launches of an eager training step."""

    def __init__(self):
        self.records = []

    def install(self):
        import mvster_amd.ops as ops
        from mvster_amd import _lib
        timer = self
        self._orig = (ops.conv_wgrad, ops.warp_agg_bwd_cl)

        def wgrad(x_cl, gy_cl, kernel, stride, padding, **kw):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            out = timer._orig[0](x_cl, gy_cl, kernel, stride, padding, **kw)
            e1.record()
            B, Do, Ho, Wo, CO = gy_cl.shape
            flops = 2 * B * Do * Ho * Wo * kernel[0] * kernel[1] * kernel[2] * x_cl.shape[-1] * CO
            # (two launches: the slot kernel, whose name the library reported, and the finish; both inside the pair)
            timer.records.append((_lib.last_kernel(), e0, e1, flops, 4 * (x_cl.numel() + gy_cl.numel())))
            return out

        def warp_bwd(ref_cl, src_cl, rt, hypo, out, wsum, grad_out, G, *a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            res = timer._orig[1](ref_cl, src_cl, rt, hypo, out, wsum, grad_out, G, *a, **k)
            e1.record()
            # algorithmic bytes: read features, hypotheses, the forward's outputs and the incoming gradient; write both gradients
            bytes_ = 4 * (2 * ref_cl.numel() + 2 * src_cl.numel() + hypo.numel() + out.numel() + wsum.numel() + grad_out.numel())
            timer.records.append((_lib.last_kernel(), e0, e1, 0, bytes_))
            return res

        ops.conv_wgrad, ops.warp_agg_bwd_cl = wgrad, warp_bwd

    def remove(self):
        import mvster_amd.ops as ops
        ops.conv_wgrad, ops.warp_agg_bwd_cl = self._orig


def train_main(args):
    """BASELINE.json configs[3]: DDP training, batch 2 per GPU, 512x640, 5 views -- one captured step per rank."""
    from mvster_amd import shard
    rank, local_rank, world = shard.init_distributed()
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d: launched with WORLD_SIZE=%d" % (args.gpus, world))
    line = train_measure(args, rank, local_rank, world, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


def train_measure(args, rank, local_rank, world, steps, warmup, min_total_s=0.5):
    """One rank's share of the training measurement; rank 0 gets the line (a dict), the others None."""
    from mvster_amd import MVS4net, MVS4net_loss, shard
    from mvster_amd.graph import GraphedTrainStep
    from mvster_amd.synthetic import make_inputs

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        shard.pin_to_gpu_numa(local_rank)
    B = args.batch if args.batch > 1 else 2            # scripts/train_dtu.sh:20: batch 2 per GPU
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    if args.coherent:
        with torch.no_grad():
            for r in model.reg:
                r.prob.weight.zero_()
                r.prob.weight.requires_grad_(False)
    model.to(dev).train()
    if getattr(args, "train_side_stages", None):
        model.train_side_stages = tuple(int(v) for v in args.train_side_stages.lstrip("s").split(","))
        model.train_side_separate = args.train_side_stages.startswith("s")
    if getattr(args, "no_fpn_tail_stream", False):
        model.train_fpn_tail_stream = False
    params = [p for p in model.parameters() if p.requires_grad]
    H, W, N = args.height, args.width, args.views
    imgs, proj, dv = make_inputs(nviews=N, H=H, W=W, seed=100 + rank, device=dev, batch=B)      # every rank its own samples
    g = torch.Generator().manual_seed(rank)
    gt, mask = {}, {}
    for s in range(1, 5):
        hs, ws = H // 2 ** (4 - s), W // 2 ** (4 - s)
        gt["stage%d" % s] = (500 + 300 * torch.rand(B, hs, ws, generator=g)).to(dev)
        mask["stage%d" % s] = (torch.rand(B, hs, ws, generator=g) > 0.2).float().to(dev)

    def loss_fn(o, g_, m_):
        return MVS4net_loss(o, g_, m_, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1,
                            ot_continous=False, mono=True)

    bucket = shard.GradBucket(params) if world > 1 else None
    if getattr(args, "torch_adam", False):
        opt = torch.optim.Adam(params, lr=1e-4, capturable=not args.no_graph, fused=True)
    else:
        from mvster_amd.optim import FusedAdam          # torch.optim.Adam's update, 3 launches instead of 6 (0.2 ms per step)
        opt = FusedAdam(params, lr=1e-4)
    if args.no_graph:
        def step():
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(model(imgs, proj, dv), gt, mask)[0]
            loss.backward()
            if bucket is not None:
                bucket.sync()
            opt.step()
            return loss.detach()
    else:
        if getattr(args, "wgrad_streams", None):
            GraphedTrainStep.wgrad_streams = args.wgrad_streams
        if getattr(args, "wgrad_early", None) is not None:
            GraphedTrainStep.wgrad_early = bool(args.wgrad_early)
        if getattr(args, "wgrad_policy", None):
            GraphedTrainStep.wgrad_policy = args.wgrad_policy
        if getattr(args, "wgrad_overlap", None) is not None:
            GraphedTrainStep.wgrad_overlap = bool(args.wgrad_overlap)
        graphed = GraphedTrainStep(model, opt, loss_fn, imgs, proj, dv, gt, mask, warmup=3, grad_sync=bucket)
        step = lambda: graphed()               # noqa: E731

    for _ in range(warmup):
        step()
    last = [None]

    def timed_step():
        last[0] = step()
    elapsed, windows, rank_span = timed_windows(timed_step, steps, shard, min_total_s=min_total_s, max_windows=9)
    loss = last[0]
    ranks_seen = int(round(shard.sum_over_ranks(1.0)))
    if ranks_seen != world:
        raise SystemExit("bench.py: %d ranks answered the collective, WORLD_SIZE=%d" % (ranks_seen, world))
    per_rank = shard.gather_over_ranks(steps * B / max(rank_span[1] if world == 1 else rank_local_s(timed_step, steps), 1e-9))
    hashes = shard.gather_over_ranks(float(int(kernel_source_hash()[:12], 16)))
    if len(set(hashes)) != 1:
        raise SystemExit("bench.py: the ranks run different kernel sources")
    last_loss = float(loss.item())
    # every rank must hold the same parameters after the same averaged updates (collectives: every rank calls)
    digest = float(sum(p.detach().double().sum() for p in params))
    spread = shard.max_over_ranks(digest) + shard.max_over_ranks(-digest)

    rooflines = []
    if rank == 0:
        timer = TrainKernelTimer()
        timer.install()
        try:
            for i in range(3):
                if i == 1:
                    timer.records.clear()
                torch.cuda._sleep(20_000_000)          # keep the GPU behind the host: event pairs bracket kernel time
                model.zero_grad(set_to_none=True)
                loss_fn(model(imgs, proj, dv), gt, mask)[0].backward()
            torch.cuda.synchronize()
        finally:
            timer.remove()
        agg = {}
        for name, e0, e1, flops, bytes_ in timer.records:
            a = agg.setdefault(name, dict(ms=0.0, n=0, flops=0, bytes=0))
            a["ms"] += e0.elapsed_time(e1)
            a["n"] += 1
            a["flops"] += flops
            a["bytes"] += bytes_
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
            if name.startswith("conv_wgrad"):
                ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
                rooflines.append({"kernel": name, "bound": "mfma", "achieved": round(ach, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                  "avg_launch_us": round(a["ms"] / a["n"] * 1e3, 2), "launches_per_step": a["n"] // 2,
                                  "ms_per_step": round(a["ms"] / 2, 3), "note": "slot kernel + finish kernel per launch"})
            else:
                ach = a["bytes"] / (a["ms"] * 1e-3) / 1e9
                rooflines.append({"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                                  "avg_launch_us": round(a["ms"] / a["n"] * 1e3, 2), "launches_per_step": a["n"] // 2,
                                  "ms_per_step": round(a["ms"] / 2, 3)})
    if rank != 0:
        return None
    return {
        "metric": "training samples/sec (DTU %dx%d, %d-view, batch %d/GPU, DDP gradient all-reduce)" % (H, W, N, B),
        "value": round(steps * world * B / elapsed, 3), "unit": "samples/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "mode": "train",
        "config": {"workload": "DTU mid %dx%d, %d views, DDP training batch=%d/GPU, 4-stage cascade 8/8/4/4 hyp, OT loss "
                               "(10 Sinkhorn iterations), Adam" % (H, W, N, B),
                   "launch": "eager" if args.no_graph else "one hipGraph per step (forward + loss + backward + "
                                                           "gradient all-reduce + Adam)",
                   "parallelism": "dp%d" % world,
                   "optimizer": "torch.optim.Adam(fused)" if getattr(args, "torch_adam", False) else "mvster_amd.optim.FusedAdam",
                   "gradient_sync": ("none (one rank)" if bucket is None else
                                     "one %.2f MB fp32 bucket, one all-reduce per step (RCCL), averaged" % (bucket.flat.numel() * 4 / 1e6)),
                   "depth_regime": "smooth (prob heads zeroed)" if args.coherent else "random-weight winners"},
        "ranks_seen": ranks_seen, "per_rank_value": [round(v, 3) for v in per_rank],
        "timing": timing_note(steps, windows, rank_span),
        "loss_last": round(last_loss, 5), "param_digest_spread_over_ranks": spread,
        "roofline": rooflines[0] if rooflines else None, "rooflines": rooflines[:8], "cpu_baseline": None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--views", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="also print the per-kernel table to stderr")
    ap.add_argument("--inflight", type=int, default=2,
                    help="depth maps processed concurrently on one GPU: independent captured forwards replayed on "
                         "separate HIP streams (1 = strictly one after the other)")
    ap.add_argument("--batch", type=int, default=1,
                    help="depth maps per forward call (the B of MVS4net.forward); the reference's eval driver uses 1")
    ap.add_argument("--no-stream-inputs", action="store_true",
                    help="skip the second timed loop that feeds the inputs from pinned host memory (value_with_h2d)")
    ap.add_argument("--h2d-direct", action="store_true",
                    help="value_with_h2d: copy straight into the graphs' static inputs (no staging buffers); A/B switch")
    ap.add_argument("--no-coherent", action="store_true",
                    help="skip the second instrumented pass (warp kernels on smooth depth maps: rooflines_warp_smooth_depth)")
    ap.add_argument("--no-batched", action="store_true",
                    help="skip the 2- and 4-depth-maps-per-forward measurements (value_batched)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the few graph replays of the 1152x1600x5 and 1024x1920x7 workloads (other_configs)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="profiling passes: the fine FPN levels on the main stream too (no co-running kernels)")
    ap.add_argument("--no-fuse-hypotheses", action="store_true",
                    help="A/B switch: hypothesis scheduling as its own launch per stage (as before round 5)")
    ap.add_argument("--no-fuse-conv0", action="store_true",
                    help="A/B switch: FPN conv0[0] and conv0[1] as two launches (as before the one-launch form of round 6)")
    ap.add_argument("--pair-wpc", type=int, default=0,
                    help="A/B switch: workgroups per CU of the one-launch conv0 pair (0 = the kernel's default)")
    ap.add_argument("--no-merge-launches", action="store_true",
                    help="A/B switch: the forward's first three launches and the three confidence up-samplings separately")
    ap.add_argument("--no-api-call", action="store_true",
                    help="skip the plain model(imgs, proj, depth_values) loops (value_api_call: what the unchanged reference "
                         "driver gets)")
    ap.add_argument("--no-train", action="store_true",
                    help="eval mode, N = 1: skip the embedded training measurement (line['train']: ten captured steps of config 4)")
    ap.add_argument("--wgrad-streams", type=int, default=0,
                    help="train mode: streams the postponed weight-gradient kernels are spread over (0 = the default, 2)")
    ap.add_argument("--wgrad-early", type=int, default=None,
                    help="train mode: 1 = the weight gradients collected before the FPN's backward are launched there, 0 = all at the end")
    ap.add_argument("--wgrad-policy", choices=("rr", "lpt"), default=None,
                    help="train mode: how the postponed weight-gradient kernels are dealt to their streams")
    ap.add_argument("--wgrad-overlap", type=int, default=None,
                    help="train mode: 1 = weight-gradient kernels beside the backward chain on one side stream, 0 = after it (default: the class attribute)")
    ap.add_argument("--train-side-stages", default=None,
                    help="train mode (experiment): cascade stages (0-based, comma-separated) whose forward and backward run on a side stream")
    ap.add_argument("--no-fpn-tail-stream", action="store_true",
                    help="train mode (A/B): the FPN's two fine levels on the caller's stream instead of their own")
    ap.add_argument("--torch-adam", action="store_true",
                    help="train mode: torch.optim.Adam(fused=True) instead of mvster_amd.optim.FusedAdam (A/B)")
    ap.add_argument("--mode", choices=("eval", "train"), default="eval",
                    help="eval: depth-maps/s of the forward (the headline, BASELINE configs[1]); train: samples/s of the "
                         "captured training step with the bucketed RCCL gradient all-reduce (BASELINE configs[3])")
    ap.add_argument("--coherent", action="store_true",
                    help="train mode: zero the prob heads, i.e. smooth depth maps between stages as in a trained network "
                         "(the fixture weights are random: neighbouring pixels pick unrelated hypotheses)")
    ap.add_argument("--stub", action="store_true",
                    help="CPU/gloo dry run of the launch + timing protocol (no GPU, no model); used by the CPU tests")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))       # no launcher around us: start the ranks ourselves
    if args.stub:
        return stub_main(args)
    if args.mode == "train":
        return train_main(args)

    from mvster_amd import MVS4net, shard
    from mvster_amd.graph import GraphedForward
    from mvster_amd.synthetic import make_inputs

    rank, local_rank, world = shard.init_distributed()
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d: launched with WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    # N > 1: each rank's host thread on the cores of its GPU's NUMA node (N = 1 keeps every core: the CPU baseline uses them)
    numa = shard.pin_to_gpu_numa(local_rank) if world > 1 else None

    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    model.to(dev).eval()
    if args.no_overlap:
        model.overlap_streams = False      # profiling passes: one stream, no co-running kernels
    if args.no_fuse_hypotheses:
        model.fuse_hypotheses = False
    if args.no_merge_launches:
        model.merge_launches = False
    if args.no_fuse_conv0:
        from mvster_amd import conv_plan as _cp
        _cp.FUSE_CONV0 = False
    if args.pair_wpc:
        from mvster_amd import conv_plan as _cp
        _cp.NARROW_PAIR_WPC = args.pair_wpc
    # every rank works on its own depth maps: disjoint seeds = disjoint units of the shard
    units = shard.shard_units(world * (args.steps + args.warmup), rank, world)
    imgs, proj, dv = make_inputs(nviews=args.views, H=args.height, W=args.width, seed=units[0], device=dev, batch=args.batch)

    sequential = None
    if args.no_graph:
        step = lambda: model.forward_eager(imgs, proj, dv)   # noqa: E731
    elif args.inflight <= 1:
        graphed = GraphedForward(model, imgs, proj, dv)
        step = lambda: graphed()               # noqa: E731
    else:
        # reference point: one depth map at a time (latency of a single forward)
        g1 = GraphedForward(model, imgs, proj, dv)
        for _ in range(args.warmup):
            g1()
        nseq = max(10, min(args.steps, 50))
        seq_el, seq_windows, _ = timed_windows(lambda: g1(), nseq, shard, min_total_s=0.25)
        sequential = seq_el / nseq
        del g1
        # several independent depth maps in flight: one captured forward + one stream per slot
        slots = []
        for k in range(args.inflight):
            im, pr, d = make_inputs(nviews=args.views, H=args.height, W=args.width, seed=units[0] + 1000 * k, device=dev,
                                    batch=args.batch)
            slots.append((GraphedForward(model, im, pr, d, packed=True), torch.cuda.Stream(device=dev)))
        counter = [0]

        def step():
            g, st = slots[counter[0] % len(slots)]
            counter[0] += 1
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                g.graph.replay()

    for _ in range(args.warmup):
        step()
    elapsed, windows, rank_span = timed_windows(step, args.steps, shard)
    ranks_seen = int(round(shard.sum_over_ranks(1.0)))
    if ranks_seen != world:
        raise SystemExit("bench.py: %d ranks answered the collective, WORLD_SIZE=%d" % (ranks_seen, world))
    # every rank's own rate over one more window of the same loop (no MAX over ranks in it): a straggler GPU shows here;
    # and every rank must run the same kernels (fingerprint of the sources the in-tree library was built from)
    torch.cuda.synchronize()
    shard.barrier()
    r0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    per_rank = shard.gather_over_ranks(args.steps * args.batch / (time.perf_counter() - r0))
    hashes = shard.gather_over_ranks(float(int(kernel_source_hash()[:12], 16)))
    if len(set(hashes)) != 1:
        raise SystemExit("bench.py: the ranks run different kernel sources (%s)" % sorted(set(hashes)))

    # ---- the same loop with the inputs coming from the host: pinned buffers, one copy stream -----------------------
    # (the reference's loop moves every sample to the GPU first, test_mvs4.py:202-207).  `value` above keeps the inputs
    # resident in HBM as the metric is defined; this is the PCIe-inclusive rate measured, not computed.
    with_h2d = None
    if not args.no_graph and args.inflight > 1 and not args.no_stream_inputs:
        from mvster_amd.graph import pack_sample
        copy_stream = torch.cuda.Stream(device=dev)
        # One packed pinned buffer per sample and ONE copy per depth map (GraphedForward(packed=True): the static inputs are
        # views of one flat device buffer): 19.66 MB move at ~50 GB/s as one buffer against 38 GB/s as five images plus the
        # small tensors (profiles/r03_q_h2d_rate.txt).  With `inflight` slots a slot's copy waits for its previous replay.
        # Two more input slots measured SLOWER in round 3 (779 against 834 depth-maps/s, profiles/r03_q_bench_4slots.json).
        host = [pack_sample(g.imgs, g.proj, g.depth_values) for g, _ in slots]
        copied = [torch.cuda.Event() for _ in slots]
        done = [torch.cuda.Event() for _ in slots]
        for e in done:
            e.record()
        k_h2d = [0]

        # Staged (default): the host buffer lands in one of two device staging buffers of the slot on the copy stream -- it
        # only has to wait for that staging buffer, not for the slot's running forward -- and a device-to-device copy
        # (19.66 MB, ~10 us) moves it into the graph's static inputs on the slot's stream, behind the previous replay.
        # --h2d-direct: the copy goes straight into the static inputs and so waits for the slot's previous replay, which
        # keeps one of the two forwards from running for the length of a copy (the round-3 form: 985 against 1 055 depth-maps/s, same box).
        staging = [[torch.empty_like(g.flat) for _ in range(2)] for g, _ in slots]
        staged = [[torch.cuda.Event() for _ in range(2)] for _ in slots]
        freed = [[torch.cuda.Event() for _ in range(2)] for _ in slots]
        for pair in freed:
            for e in pair:
                e.record()

        def step_h2d_direct():
            i = k_h2d[0] % len(slots)
            k_h2d[0] += 1
            g, st = slots[i]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(done[i])            # the slot's previous replay has consumed its inputs
                g.load_packed(host[i])
                copied[i].record()
            with torch.cuda.stream(st):
                st.wait_event(copied[i])
                g.graph.replay()
                done[i].record()

        def step_h2d_staged():
            n = k_h2d[0]
            k_h2d[0] += 1
            i, j = n % len(slots), (n // len(slots)) % 2
            g, st = slots[i]
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(freed[i][j])        # the staging buffer's previous content has been consumed
                staging[i][j].copy_(host[i], non_blocking=True)
                staged[i][j].record()
            with torch.cuda.stream(st):
                st.wait_event(staged[i][j])
                g.flat.copy_(staging[i][j], non_blocking=True)     # (stream order: after the slot's previous replay)
                freed[i][j].record()
                g.graph.replay()

        step_h2d = step_h2d_direct if args.h2d_direct else step_h2d_staged

        # the pinned path is warmed on its own (first copies out of freshly pinned pages ran at a fraction of the rate on
        # some boxes: BENCH_r03 saw 6 GB/s in a 25-step run where the builder's boxes saw 16): plain copies first, then the loop
        with torch.cuda.stream(copy_stream):
            for _ in range(8):
                for (g, _), h in zip(slots, host):
                    g.load_packed(h)
            copy_stream.synchronize()
            plain_gbps, ncopy = 0.0, 16
            for _ in range(3):                          # best of three rounds (the first round after pinning runs slow on some boxes)
                c0 = time.perf_counter()
                for _ in range(ncopy):
                    for (g, _), h in zip(slots, host):
                        g.load_packed(h)
                copy_stream.synchronize()
                plain_gbps = max(plain_gbps, ncopy * sum(h.numel() for h in host) * 4 / (time.perf_counter() - c0) / 1e9)
        for _ in range(max(args.warmup, 10)):
            step_h2d()
        el2, win2, _ = timed_windows(step_h2d, args.steps, shard)
        mb = host[0].numel() * 4 / 1e6
        with_h2d = {"value": round(args.steps * world * args.batch / el2, 3), "unit": "depth-maps/s",
                    "ms_per_step": round(1e3 * el2 / args.steps, 4), "h2d_MB_per_depth_map": round(mb / args.batch, 2),
                    "how": ("one packed pinned buffer and ONE host-to-device copy per depth map on one copy stream, %s; "
                            "median of %d windows of %d steps"
                            % ("straight into the graph's static inputs (waits for the slot's previous forward)" if args.h2d_direct
                               else "into one of two staging buffers per slot, then a device-to-device copy into the graph's "
                                    "static inputs on the slot's stream", len(win2), args.steps)),
                    "h2d_GBps": round(mb * args.steps / el2 / 1e3, 2),
                    "h2d_GBps_plain_copy": round(plain_gbps, 2),
                    "window_ms": [round(1e3 * w, 3) for w in win2]}

    # ---- the reference driver's own loop: model(imgs, proj, depth_values), nothing else (test_mvs4.py:202-207) -------
    # (after the pinned-copy loop above: the pageable host-to-device copies of its `from_host_tocuda` variant leave the
    #  runtime's copy path in a state that slows a later pinned pipeline by 20 % -- 758 against 948 depth-maps/s, same box,
    #  alternating runs, gpurun_out/r5q -- while this leg measures the same after it as before it)
    api_call = None
    if rank == 0 and world == 1 and not args.no_graph and not args.no_api_call:      # (N = 1 only: its windows use the barrier / MAX protocol)
        api_call = api_call_measure(args, model, dev, units[0], shard)

    # ---- the same workload with several depth maps per forward call (the B of MVS4net.forward; the reference's eval driver
    # uses 1, its training 2): not the headline -- `value` stays one depth map per call, comparable with every earlier record --
    # but what a throughput-oriented caller gets: the coarse stages' short launches are shared by the maps of a batch.
    batched = []
    if (rank == 0 and world == 1 and not args.no_graph and args.inflight > 1 and not args.no_batched and args.batch == 1
            and (args.height, args.width, args.views) == (512, 640, 5)):
        for bsz in (2, 4):
            try:
                bslots = []
                for k in range(args.inflight):
                    im, pr, d = make_inputs(nviews=args.views, H=args.height, W=args.width, seed=units[0] + 77 * k, device=dev, batch=bsz)
                    bslots.append((GraphedForward(model, im, pr, d), torch.cuda.Stream(device=dev)))
                bk = [0]

                def bstep():
                    g, st = bslots[bk[0] % len(bslots)]
                    bk[0] += 1
                    st.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(st):
                        g.graph.replay()
                nb = max(4, args.steps)          # as many forwards per window as the headline has steps
                for _ in range(4):
                    bstep()
                bel, bwin, _ = timed_windows(bstep, nb, shard, min_total_s=0.2, max_windows=15)
                batched.append({"depth_maps_per_forward": bsz, "depth_maps_in_flight": bsz * len(bslots),
                                "value": round(nb * bsz / bel, 3), "unit": "depth-maps/s", "ms_per_forward": round(1e3 * bel / nb, 4),
                                "forwards_per_window": nb, "windows": len(bwin),
                                "finite": bool(torch.isfinite(bslots[0][0].outputs["depth"]).all().item())})
                del bslots
                torch.cuda.empty_cache()
            except RuntimeError as e:
                batched.append({"depth_maps_per_forward": bsz, "error": str(e)[:120]})

    # ---- the other inference configurations of BASELINE.json (runnable forms of configs[2] and configs[4]) ----------
    other_configs = []
    if rank == 0 and not args.no_graph and not args.no_other_configs and (args.height, args.width, args.views) == (512, 640, 5):
        # (published: the reference's README.md:74-75, one RTX 3090, timer without a device synchronisation; "mid" is 832x1152
        #  as test_mvs4.py:41-42 + the /64 rounding of general_eval4.py:92-100 run it, "raw" is 1152x1600)
        for (oh, ow, on, label, pub) in ((832, 1152, 5, "DTU mid as the reference's test script runs it (832x1152, 5 views)", 0.09),
                                         (1152, 1600, 5, "DTU raw as it runs (1152x1600, 5 views)", 0.17),
                                         (1024, 1920, 7, "Tanks&Temples as it runs (1024x1920, 7 views)", None)):
            try:
                im, pr, d = make_inputs(nviews=on, H=oh, W=ow, seed=7, device=dev)
                go = GraphedForward(model, im, pr, d)
                for _ in range(2):
                    go()
                torch.cuda.synchronize()
                reps = []
                for _ in range(3):                       # median of three windows of five replays
                    c0 = time.perf_counter()
                    for _ in range(5):
                        go()
                    torch.cuda.synchronize()
                    reps.append(1e3 * (time.perf_counter() - c0) / 5)
                ms = sorted(reps)[1]
                oc = {"workload": label + ", 4-stage cascade, B=1 eval, one depth map at a time",
                      "ms_per_depth_map": round(ms, 3), "depth_maps_per_s": round(1e3 / ms, 2),
                      "finite": bool(torch.isfinite(go.outputs["depth"]).all().item())}
                if pub is not None:
                    oc["published"] = {"s_per_depth_map": pub, "hardware": "1x RTX 3090", "source": "reference README.md:74-75",
                                       "speedup_over_published": round(pub * 1e3 / ms, 1)}
                other_configs.append(oc)
                del go, im, pr, d
                torch.cuda.empty_cache()
            except RuntimeError as e:          # (out of memory next to the resident slots: reported, not fatal)
                other_configs.append({"workload": label, "error": str(e)[:120]})

    # ---- instrumented eager pass: per-kernel HIP-event timing (rank 0 only) -------------------
    roofline = None
    rooflines = []
    rooflines_coherent = None
    table = None
    if rank == 0:
        timer = KernelTimer()
        timer.install()
        overlap = model.overlap_streams
        model.overlap_streams = False      # one stream: per-kernel durations without co-running kernels
        ninstr = max(1, min(args.steps, 20))
        try:
            for _ in range(3):
                model.forward_eager(imgs, proj, dv)
            timer.records.clear()
            for _ in range(ninstr):
                # keep the GPU behind the host: a ~2 ms spin kernel first, so that every launch of this forward is already
                # queued when the GPU reaches it and an event pair brackets kernel time, not the host's launch gaps
                torch.cuda._sleep(5_000_000)
                model.forward_eager(imgs, proj, dv)
            torch.cuda.synchronize()
            table = timer.summary()
        finally:
            timer.remove()
            model.overlap_streams = overlap
        # HBM bytes per launch from the PMC passes (scripts/gpu_pmc.sh: separate FETCH_SIZE / WRITE_SIZE runs of
        # this same command, FETCH_SIZE x2 on gfx950, KB -> bytes).  PMC cannot be collected from inside the
        # process: the committed summary is quoted ONLY if it was measured on these very kernels (source hash).
        pmc, pmc_note = {}, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            if pj.get("workload") != [args.height, args.width, args.views]:
                pmc_note = "profiles/pmc_traffic.json was measured on another workload"
            elif pj.get("kernel_source_hash") != kernel_source_hash():
                pmc_note = ("profiles/pmc_traffic.json was measured on other kernel sources (hash %s, now %s): not quoted"
                            % (pj.get("kernel_source_hash"), kernel_source_hash()))
            else:
                pmc = pj
        except (OSError, ValueError):
            pmc_note = "no profiles/pmc_traffic.json"

        def entry(name, a):
            avg_ms = a["ms"] / a["n"]
            # (the narrow full-resolution layers and conv11 + selection move 8-24 bytes per FLOP-pair: HBM is their roofline,
            #  whether they compute on the VALU or on MFMA tiles; their fraction of the fp32 MFMA peak is carried beside it)
            streaming = name.startswith(("conv_small", "conv_narrow", "deconv_se"))
            if a["flops"] > 0 and name.startswith("conv") and not streaming:      # (MFMA kernels)
                achieved = a["flops"] / (a["ms"] * 1e-3) / 1e12
                e = {"kernel": name, "bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
                     "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": a["n"] // ninstr,
                     "flops_per_launch": a["flops"] // a["n"]}
                if len(a["shapes"]) > 1:      # the kernel serves several layer shapes: the fraction of each
                    e["per_shape"] = [{"flops_per_launch": fl, "avg_launch_us": round(ms / n * 1e3, 2),
                                       "launches_per_step": n // ninstr,
                                       "frac": round(fl / (ms / n * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}
                                      for (fl, _), (ms, n) in sorted(a["shapes"].items(), key=lambda kv: -kv[0][0])]
                if name.startswith("conv_wino"):
                    # `achieved` counts the ALGORITHMIC (direct-form) FLOPs of the layer, as for every other kernel; the
                    # minimal-filtering form executes 16 multiplications per 2x2 output block and (cin, cout) instead of 36
                    e["algorithm"] = "Winograd F(2x2, 3x3): 2.25x fewer multiplications than the FLOPs counted in `achieved`"
                    e["frac_of_peak_executed_mfma"] = round(achieved / 2.25 / FP32_MFMA_PEAK_TFLOPS, 4)
            else:
                achieved = a["bytes"] / (a["ms"] * 1e-3) / 1e9
                e = {"kernel": name, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "avg_launch_us": round(avg_ms * 1e3, 2), "launches_per_step": a["n"] // ninstr,
                     "bytes_per_launch": a["bytes"] // a["n"]}
                if name.startswith("fpn_tail"):
                    e["limiter"] = ("fp32 VALU issue, not HBM: position-dependent bilinear weights (36 per output pixel) leave no "
                                    "shared operand for the matrix cores; ~650 VALU / LDS instructions per wave, half of them the gather-sum "
                                    "(DESIGN.md section 4.3)")
                if streaming and a["flops"] > 0:
                    e["frac_of_fp32_mfma_peak_algorithmic_flops"] = round(a["flops"] / (a["ms"] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
            # (the profiler spells the wave-local warp kernel with its fourth template argument, the library's
            #  mvster_last_kernel() without it when it is 0: same kernel)
            pname = name if name in pmc.get("kernels", {}) else name[:-1] + ", 0>"
            if pname in pmc.get("kernels", {}):
                e["traffic"] = int(pmc["kernels"][pname])
                e["traffic_unit"] = "bytes/launch"
                e["traffic_source"] = pmc.get("source", "profiles/pmc_traffic.json")
            elif pmc_note:
                e["traffic_note"] = pmc_note
            return e

        name, a = max(table.items(), key=lambda kv: kv[1]["ms"])
        total_ms = sum(v["ms"] for v in table.values())

        def entry_share(k, v):
            e = entry(k, v)
            e["share_of_timed_kernel_time"] = round(v["ms"] / total_ms, 4)
            return e
        roofline = entry_share(name, a)
        # the dominant kernel first, then the eight largest by time (the forward has no kernel above ~7 % any more: the
        # list shows the MFMA-bound and the HBM-bound ones side by side), then the fused warp kernel of every stage
        top = [k for k, _ in sorted(table.items(), key=lambda kv: -kv[1]["ms"])[:9] if k != name][:8]
        rooflines = ([roofline] + [entry_share(k, table[k]) for k in top] +
                     [entry_share(k, v) for k, v in sorted(table.items()) if k.startswith("warp_agg") and k != name and k not in top])
        # the same warp launches in the geometrically coherent regime (prob heads zeroed: every pixel keeps hypothesis 0,
        # so the depth maps handed from stage to stage are smooth, as for a trained network; the fixture's random weights
        # make neighbouring pixels pick unrelated hypotheses)
        if not args.no_coherent:
            smooth = MVS4net(**SHIPPED)
            smooth.load_state_dict(model.state_dict(), strict=True)
            with torch.no_grad():
                for r in smooth.reg:
                    r.prob.weight.zero_()
            smooth.to(dev).eval()
            smooth.overlap_streams = False
            timer2 = KernelTimer()
            timer2.install()
            try:
                for _ in range(3):
                    smooth.forward_eager(imgs, proj, dv)
                timer2.records.clear()
                for _ in range(ninstr):
                    torch.cuda._sleep(5_000_000)
                    smooth.forward_eager(imgs, proj, dv)
                torch.cuda.synchronize()
                t2 = timer2.summary()
            finally:
                timer2.remove()
            rooflines_coherent = [dict(entry(k, v), depth_regime="smooth (prob heads zeroed)") for k, v in sorted(t2.items())
                                  if k.startswith("warp_agg")]
            del smooth
        if args.kernel_table:
            tot = sum(v["ms"] for v in table.values())
            for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
                print("%-34s n=%4d  avg %8.2f us  %5.1f%%  %7.2f TFLOP/s  %8.1f GB/s" % (
                    k, v["n"], v["ms"] / v["n"] * 1e3, 100 * v["ms"] / tot, v["flops"] / (v["ms"] * 1e-3) / 1e12,
                    v["bytes"] / (v["ms"] * 1e-3) / 1e9), file=sys.stderr)

    # ---- BASELINE.json configs[3] on this rank: a few captured training steps (N = 1 only; `--mode train` is the full line) ---
    train = None
    if world == 1 and not args.no_train and not args.no_graph and (args.height, args.width, args.views, args.batch) == (512, 640, 5, 1):
        try:
            slots = graphed = step = None         # the eval graphs' memory goes back to the allocator first
            torch.cuda.empty_cache()
            targs = argparse.Namespace(**vars(args))
            targs.batch, targs.coherent = 2, False
            tl = train_measure(targs, rank, local_rank, world, steps=10, warmup=3, min_total_s=0.3)
            train = {k: tl[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "timing", "loss_last")}
            train["config"] = tl["config"]
            train["rooflines"] = [{k: r[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us",
                                                     "launches_per_step", "ms_per_step")} for r in tl["rooflines"][:3]]
        except RuntimeError as e:                 # (reported, not fatal: the headline is the eval line)
            train = {"error": str(e)[:200]}

    # ---- CPU baseline: the oracle on the host cores, bounded sample ----------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.height, args.width, args.views, units[0])

    if rank == 0:
        total_maps = args.steps * world * args.batch
        metric = "depth-maps/sec (DTU 512x640, 5-view, 4-stage)"
        try:        # BASELINE.json's own wording of the metric (the file ships with the repository)
            with open(os.path.join(ROOT, "BASELINE.json")) as f:
                metric = json.load(f).get("metric", metric)
        except (OSError, ValueError):
            pass
        line = {
            "metric": metric, "value": round(total_maps / elapsed, 3),
            "unit": "depth-maps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DTU mid %dx%d, %d views, 4-stage cascade 8/8/4/4 hyp, B=%d eval, %d depth map(s) per step per GPU"
                                   % (args.height, args.width, args.views, args.batch, args.batch),
                       "launch": "eager" if args.no_graph else "hipGraph replay", "parallelism": "replicas x%d" % world,
                       "depth_maps_in_flight_per_gpu": args.inflight,
                       "host_threads": "rank 0 pinned to NUMA node %s" % numa if numa is not None else "not pinned"},
            "ranks_seen": ranks_seen, "timing": timing_note(args.steps, windows, rank_span),
            "roofline": roofline, "rooflines": rooflines, "cpu_baseline": cpu,
        }
        if rooflines_coherent:
            line["rooflines_warp_smooth_depth"] = rooflines_coherent
        if sequential is not None:
            line["single_forward_ms"] = round(1e3 * sequential, 4)       # one depth map at a time (latency)
            line["value_one_in_flight"] = round(world * args.batch / sequential, 3)
        if api_call is not None:
            line["value_api_call"] = api_call         # the plain model(...) call of the reference's drivers, one stream
            if sequential is not None:
                line["value_api_call"]["frac_of_value_one_in_flight"] = round(api_call["value"] * sequential / (world * args.batch), 3)
        line["per_rank_value"] = [round(v, 3) for v in per_rank]
        if with_h2d is not None:
            line["value_with_h2d"] = with_h2d          # never `value`: the metric is defined on HBM-resident inputs
        if batched:
            line["value_batched"] = batched          # never `value`: see the comment at its measurement
        if other_configs:
            line["other_configs"] = other_configs
        if train is not None:
            line["train"] = train                      # BASELINE.json configs[3], one rank (python bench.py --mode train [--gpus N])
        if cpu:
            line["vs_cpu_baseline"] = round(line["value"] / cpu["value"], 2)
        print(json.dumps(line))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
