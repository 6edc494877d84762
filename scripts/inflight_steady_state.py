#!/usr/bin/env python3
"""Two depth maps in flight, steady state: N replays of two captured forwards alternating on two streams (what bench.py times),
run under `rocprofv3 --kernel-trace`; `--analyse <kernel_trace.csv>` then reads the trace: how much of the wall time has 0 / 1 /
2+ kernels running, and which kernels run ALONE for how long (those bound the throughput; co-running time is shared).
    rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python scripts/inflight_steady_state.py run
    python scripts/inflight_steady_state.py --analyse OUT/t_kernel_trace.csv"""
import collections
import csv
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NREP = 60


def run():
    import torch
    from bench import SHIPPED, load_weights
    from mvster_amd import MVS4net
    from mvster_amd.graph import GraphedForward
    from mvster_amd.synthetic import make_inputs
    dev = torch.device("cuda:0")
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    model.to(dev).eval()
    model.graph_cache = False          # instrumented eager launches (MVS4net.forward would replay a captured graph)
    slots = []
    for k in range(2):
        im, pr, d = make_inputs(nviews=5, H=512, W=640, seed=k, device=dev)
        slots.append((GraphedForward(model, im, pr, d, packed=True), torch.cuda.Stream(device=dev)))
    for rep in range(20 + NREP):
        g, st = slots[rep % 2]
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g.graph.replay()
        if rep == 19:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|mvconv::|void ", "", name)
    return re.sub(r"\(.*$", "", name)[:58]


def analyse(path):
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
    rows.sort()
    # the steady state = the last NREP replays, which follow the only synchronize() of the run: the largest launch-free gap
    # among the rows where that boundary can lie (70-90 kernels per forward)
    lo, hi = max(1, len(rows) - (NREP + 15) * 90), len(rows) - NREP * 60
    ends = 0
    best, cut = -1, lo
    for i in range(len(rows) - 1):
        ends = max(ends, rows[i][1])
        if lo <= i < hi and rows[i + 1][0] - ends > best:
            best, cut = rows[i + 1][0] - ends, i
    tail = rows[cut + 1:]
    per = len(tail) / NREP
    t0, t1 = tail[0][0], max(r[1] for r in tail)
    ev = []
    for s, e, n in tail:
        ev.append((s, 1, n))
        ev.append((e, -1, n))
    ev.sort(key=lambda x: (x[0], x[1]))
    running = collections.Counter()
    hist = collections.Counter()
    alone = collections.Counter()
    shared = collections.Counter()
    last = t0
    for t, d, n in ev:
        k = sum(running.values())
        dt = t - last
        if dt > 0:
            hist[min(k, 3)] += dt
            for name, c in running.items():
                if c > 0:
                    (alone if k == 1 else shared)[name] += dt * c
        running[n] += d
        last = t
    wall = t1 - t0
    busy_sum = sum(e - s for s, e, _ in tail)
    print("steady state: %d replays, %.1f kernels per forward, %.3f ms wall per depth map, %.3f ms of kernel time per depth map (%.2fx)"
          % (NREP, per, wall / NREP / 1e6, busy_sum / NREP / 1e6, busy_sum / wall))
    for k in range(4):
        print("  %s kernels running: %5.1f %% of the wall time" % (("3+" if k == 3 else str(k)), 100.0 * hist[k] / wall))
    print("kernels by time spent running ALONE (us per depth map; co-running time beside it):")
    for name, t in alone.most_common(22):
        print("  %8.1f alone  %8.1f co-running   %s" % (t / NREP / 1e3, shared[name] / NREP / 1e3, name))
    print("  %8.1f alone  %8.1f co-running   (all kernels)" % (sum(alone.values()) / NREP / 1e3, sum(shared.values()) / NREP / 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        run()
