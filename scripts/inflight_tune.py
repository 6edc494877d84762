#!/usr/bin/env python3
"""Re-tune kernel occupancy choices of the eval forward IN the regime the headline is measured in: two captured forwards in
flight on two streams.  The measured table (tuning_gfx950.json) was filled from isolated launches; a persistent kernel sized for
an empty chip (two or three workgroups per CU, most of the LDS) can keep the other depth map's kernels off the CUs, so the best
isolated choice is not always the best here (the one-launch conv0 pair: 2 workgroups per CU 1 156, 1 workgroup 1 169
depth-maps/s).  Greedy coordinate descent over the forward's layers: for each ConvLayer call the alternatives differ from the
current choice in workgroups per CU (persistent / 1x1 / Winograd / narrow kernels) or tile height (narrow); an alternative is
kept when the whole-forward rate improves by more than --gain (confirmed by a second measurement).  Prints the entries to merge
into the table.  GPU only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from bench import SHIPPED, load_weights  # noqa: E402
from mvster_amd import MVS4net  # noqa: E402
from mvster_amd.graph import GraphedForward  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gain", type=float, default=0.002)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--out", default="gpurun_out/inflight_tune.json")
ap.add_argument("--pair-wpc", type=int, default=0)
args = ap.parse_args()
H, W, N = 512, 640, 5
dev = torch.device("cuda:0")
if args.pair_wpc:
    cp.NARROW_PAIR_WPC = args.pair_wpc
model = MVS4net(**SHIPPED)
model.load_state_dict(load_weights(), strict=True)
model.to(dev).eval()
model.graph_cache = False
inputs = [make_inputs(N, H, W, seed=1000 * k, device=dev) for k in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


def measure(windows=3):
    """ms per depth map with two captured forwards in flight (bench.py's loop), median window."""
    slots = [GraphedForward(model, *inp, packed=True) for inp in inputs]
    cur = torch.cuda.current_stream()

    def run(n):
        for i in range(n):
            st = streams[i & 1]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                slots[i & 1].graph.replay()
    run(20)
    torch.cuda.synchronize()
    ts = []
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(args.steps)
        for st in streams:
            cur.wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / args.steps)
    del slots
    ts.sort()
    return ts[len(ts) // 2]


# the forward's ConvLayer calls and the choice the plan makes for each
calls = []
orig = cp.ConvLayer.__call__


def rec(layer, x, skip=None, skip_mode=0, tiles=None):
    B, Di, Hi, Wi, _ = x.shape
    sm = skip_mode if skip is not None else 0
    _, mt, nt, _, var = layer._geom(B, Di, Hi, Wi, sm)
    calls.append((cp.layer_signature(layer, B, Di, Hi, Wi, sm), [int(var), int(mt), int(nt)], B * Di * Hi * Wi))
    return orig(layer, x, skip, skip_mode, tiles)


cp.ConvLayer.__call__ = rec
model.forward_eager(*inputs[0])
cp.ConvLayer.__call__ = orig
torch.cuda.synchronize()
seen, layers = set(), []
for sig, choice, vox in calls:
    if sig not in seen:
        seen.add(sig)
        layers.append((sig, choice, vox))
layers.sort(key=lambda t: -t[2])                       # big maps first


def alternatives(choice):
    var, mt, nt = choice
    kind, wpc = var & 0xff, var >> 8
    out = []
    if kind in (5, 6, 8, 9):
        for w in (1, 2, 3):
            if w != wpc:
                out.append([kind | (w << 8), mt, nt])
    elif kind == 10:
        for m in (2, 4):
            for w in (1, 2):
                if (m, w) != (mt if mt in (2, 4) else 0, nt & 31):
                    out.append([10, m, w])
    return out


table = cp._tuning()


def set_entry(sig, val):
    if val is None:
        table.pop(sig, None)
    else:
        table[sig] = val
    model.invalidate_plans()


# sustained load lowers the clocks (first measurement 0.873 ms, after a minute 0.933 on the same choices): heat up until two
# consecutive measurements agree, compare every candidate with a baseline measured in the same state
best = measure()
for _ in range(12):
    t = measure()
    steady = abs(t - best) < 1e-3 * best
    best = t
    if steady:
        break
print("start (steady): %.4f ms per depth map (%.1f /s)" % (best, 1e3 / best), flush=True)
changed = {}
since_base = 0
for sig, choice, vox in layers:
    alts = alternatives(choice)
    if not alts:
        continue
    cur = table.get(sig)                                # entry in force for this layer (None = heuristics / family)
    for alt in alts:
        if since_base >= 8:
            set_entry(sig, cur)
            best = measure()
            since_base = 0
        set_entry(sig, alt)
        try:
            t = measure()
        except Exception as e:                           # (a kernel form that does not exist for this layer)
            print("  %-46s %s: %s" % (sig, alt, str(e)[:60]), flush=True)
            set_entry(sig, cur)
            continue
        since_base += 1
        mark = ""
        if t < best * (1 - args.gain):
            set_entry(sig, cur)                          # confirm: baseline, candidate, both again in this state
            b2 = measure()
            set_entry(sig, alt)
            t2 = measure()
            since_base = 0
            if t2 < b2 * (1 - args.gain):
                cur, choice, best, mark = alt, alt, t2, "  <-- kept (%.4f against %.4f)" % (t2, b2)
                changed[sig] = alt
            else:
                best = b2
                mark = "  (not confirmed: %.4f against %.4f)" % (t2, b2)
        print("  %-46s -> %s: %.4f ms (baseline %.4f)%s" % (sig, alt, t, best, mark), flush=True)
    set_entry(sig, cur)
final = measure()
print("end: %.4f ms per depth map (%.1f /s); %d entries changed" % (final, 1e3 / final, len(changed)))
os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
with open(args.out, "w") as f:
    json.dump(changed, f, indent=1)
print(json.dumps(changed))
