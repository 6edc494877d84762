#!/usr/bin/env python3
"""Per-layer timing of the convolution kernels on the shapes of one 512x640x5 forward.
For every ConvLayer call of the eval forward: time the direct kernel (variant 0) and, where it
applies, the LDS-staged kernel (variant 1) over a few tile shapes.  GPU only."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mvster_amd.conv_plan as cp  # noqa: E402
from bench import SHIPPED, load_weights  # noqa: E402
from mvster_amd import MVS4net  # noqa: E402
from mvster_amd.synthetic import make_inputs  # noqa: E402


def timeit(fn, n=20):
    """GPU time per call in us: n calls captured in one hipGraph (host launch overhead, ~12 us per
    call from Python, would otherwise hide every kernel shorter than that)."""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * n) * 1e3


def candidates(layer, B, Di, Hi, Wi, sm):
    """Every (variant, mt, nt) the kernels support for this layer."""
    out = []
    if layer.w_small is not None and sm in (0, 1):
        # narrow layers: the VALU kernel or the shift-packed MFMA kernel (mt = tile rows / 4, nt = workgroups per CU)
        return [("V", (0, 0, 3))] + [("N%d w%d" % (m, w), (m, w, 10)) for m in (2, 4) for w in (1, 2)]
    nts = [n for n in (1, 2, 3, 4, 5, 9) if n <= layer.ntile_total and layer.ntile_total % n == 0]
    for m in (1, 2, 4):
        for n in nts:
            if n in (3, 5, 9) and (layer.cin != 64 or (m, n) == (4, 9)):
                continue
            out.append(("d%d,%d" % (m, n), (m, n, 0)))
    if not layer.transposed and layer.cin % 16 == 0 and layer.kernel[2] in (3, 5) and sm in (0, 1):
        for m in (2, 4):
            patch = layer.kernel[0] * ((2 * m - 1) * layer.stride[1] + layer.kernel[1]) * (31 * layer.stride[2] + layer.kernel[2]) * 64
            if patch > cp.LDS_BUDGET:
                continue
            for n in (1, 2, 4):
                if n in nts:
                    out.append(("L%d,%d" % (m, n), (m, n, 1)))
    if layer.cin >= 16:
        for m, n in ((1, 1), (1, 2), (1, 4), (2, 1), (2, 2)):
            if n in nts:
                out.append(("S%d,%d" % (m, n), (m, n, 2)))
    if (not layer.transposed and layer.prob is None and sm == 2 and layer.kernel == (1, 1, 1) and layer.stride == (1, 1, 1)
            and layer.cin in (16, 32, 64)):
        for m in (1, 2, 4):                   # persistent 1x1 with the bilinear up-sampling add in its epilogue
            for wpc in (1, 2, 3):
                out.append(("Q%d w%d" % (m, wpc), (m, 1, 6 | (wpc << 8))))
    if not layer.transposed and layer.prob is None and sm in (0, 1):
        # persistent kernels (conv_pers.hip); workgroups per CU in bits 8.. of the variant.  What a family does not
        # cover raises and is skipped by the caller.
        if layer.kernel == (1, 1, 1) and layer.stride == (1, 1, 1) and layer.cin in (16, 32, 64):
            for m in (1, 2, 4):
                for wpc in (1, 2, 3):
                    out.append(("Q%d w%d" % (m, wpc), (m, 1, 6 | (wpc << 8))))
        if layer.cin == 8 and layer.cout == 16 and layer.kernel in ((1, 3, 3), (1, 5, 5)) and layer.stride == (1, 2, 2):
            for wpc in (1, 2, 3):             # conv_pers8_kernel
                out.append(("P8 w%d" % wpc, (2, 1, 5 | (wpc << 8))))
        if layer.kernel in ((1, 3, 3), (3, 3, 3), (1, 5, 5)) and layer.cin in (16, 32, 64) and layer.cout % 16 == 0:
            for n in nts:
                for wpc in (1, 2, 3):
                    out.append(("P2,%d w%d" % (n, wpc), (2, n, 5 | (wpc << 8))))
                if layer.stride[2] == 2:      # waves 4-7 issue the DMA (built for the stride-2 families)
                    for wpc in (33, 34):
                        out.append(("P2,%d L%d" % (n, wpc & 15), (2, n, 5 | (wpc << 8))))
    if (layer.transposed and layer.prob is None and sm in (0, 1) and layer.kernel == (1, 3, 3) and layer.stride == (1, 2, 2)
            and layer.cin in (32, 64) and layer.cout % 16 == 0):
        for wpc in (1, 2):                    # conv_tpers_kernel: all four output-parity classes from one staged tile
            out.append(("TP w%d" % wpc, (2, 1, 5 | (wpc << 8))))
    if not layer.transposed and layer.prob is None and sm in (0, 1):
        if layer.wino_eligible():
            # Winograd F(2x2,3x3) on the persistent frame (conv_wino.hip); not bit-identical to the others
            if layer.kernel[0] == 1 and layer.cin in (16, 32):
                for n in nts:
                    for wpc in (1, 2):
                        out.append(("W%d w%d" % (n, wpc), (2, n, 8 | (wpc << 8))))
            for n in (1, 2):                  # ring form: 1 = waves 4-7 only load, 2 = they compute a second N tile
                if n in nts:
                    out.append(("R%d" % n, (2, n, 9)))
            if 2 in nts:                      # ... or the compute waves hold both N tiles and waves 4-7 load
                out.append(("R2L", (2, 2, 9 | (1 << 8))))
    return out


def record_eval_calls(model, H, W, N, dev):
    """[(layer, input shape, skip shape or None, skip mode)] of one eval forward at H x W with N views (distinct signatures)."""
    calls, seen = [], set()
    orig = cp.ConvLayer.__call__

    def rec(layer, x, skip=None, skip_mode=0, tiles=None):
        sm = skip_mode if skip is not None else 0
        sig = cp.layer_signature(layer, *x.shape[:4], sm)
        if sig not in seen:
            seen.add(sig)
            calls.append((layer, tuple(x.shape), None if skip is None else tuple(skip.shape), sm))
        return orig(layer, x, skip, skip_mode, tiles)
    imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev)
    cp.ConvLayer.__call__ = rec
    try:
        model(imgs, proj, dv)
    finally:
        cp.ConvLayer.__call__ = orig
    torch.cuda.synchronize()
    return calls


def auto_vs_best(calls, dev, n=6):
    """Per recorded call: (signature, us of the plan's own choice, us of the best candidate, its name, how the choice was made)."""
    rows = []
    for layer, xs, ss, sm in calls:
        B, Di, Hi, Wi, _ = xs
        x = torch.randn(*xs, device=dev)
        skip = torch.randn(*ss, device=dev) if ss else None
        auto = timeit(lambda: layer(x, skip=skip, skip_mode=sm), n=n)
        best, best_name, best_tiles = auto, "auto", None
        for name, tiles in candidates(layer, B, Di, Hi, Wi, sm):
            try:
                us = timeit(lambda: layer(x, skip=skip, skip_mode=sm, tiles=tiles), n=n)
            except RuntimeError:
                continue
            if us < best:
                best, best_name, best_tiles = us, name, tiles
        if best_tiles is not None:
            # the winner of a noisy search is biased low: time the pair again, interleaved, and keep each side's minimum
            a2, b2 = [], []
            for _ in range(2):
                a2.append(timeit(lambda: layer(x, skip=skip, skip_mode=sm), n=n))
                b2.append(timeit(lambda: layer(x, skip=skip, skip_mode=sm, tiles=best_tiles), n=n))
            auto, best = min(a2), min(b2)
            if best > auto:
                best, best_name = auto, "auto"
        how = cp.tuned_choice(layer, B, Di, Hi, Wi, sm)[1] or "heuristic"
        rows.append((cp.layer_signature(layer, B, Di, Hi, Wi, sm), auto, best, best_name, how))
        del x, skip
    return rows


def retune_narrow(path, shapes):
    """Re-time the narrow-layer signatures (VALU kernel vs the shift-packed MFMA kernel, its tile heights and workgroups per CU)
    of the given eval workloads and write them into the table at ``path``; every other entry is left alone."""
    import json
    dev = torch.device("cuda:0")
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    model.to(dev).eval()
    model.graph_cache = False          # instrumented eager launches (MVS4net.forward would replay a captured graph)
    with open(path) as f:
        table = json.load(f)
    for (H, W, N) in shapes:
        calls = [c for c in record_eval_calls(model, H, W, N, dev) if c[0].w_small is not None]
        for c in calls:
            table.pop(cp.layer_signature(c[0], *c[1][:4], c[3]), None)
        _tune(calls, table, dev)
    with open(path, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print("wrote", path, len(table), "entries")


def _tune(calls, table, dev):
    """Time every candidate of every recorded call that is not in the table yet; keep the winners."""
    for layer, xs, ss, sm in calls:
        B, Di, Hi, Wi, _ = xs
        sig = cp.layer_signature(layer, B, Di, Hi, Wi, sm)
        if sig in table:
            continue
        x = torch.randn(*xs, device=dev)
        skip = torch.randn(*ss, device=dev) if ss else None
        best = None
        for name, tiles in candidates(layer, B, Di, Hi, Wi, sm):
            try:
                us = min(timeit(lambda: layer(x, skip=skip, skip_mode=sm, tiles=tiles), n=10) for _ in range(2))
            except RuntimeError:
                continue
            if best is None or us < best[0]:
                best = (us, tiles)
        table[sig] = [best[1][2], best[1][0], best[1][1]]
        print("%-48s -> v%d mt%d nt%d  %.1f us" % (sig, best[1][2], best[1][0], best[1][1], best[0]), flush=True)
        del x, skip


def emit_table(shapes, path, train_shapes=((512, 640, 5, 2),)):
    """Time every candidate on the layer shapes of the given workloads -- eval forwards, and the forward + input-gradient
    layers of a training step -- and write the winners as JSON."""
    import json
    dev = torch.device("cuda:0")
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    model.to(dev).eval()
    model.graph_cache = False          # instrumented eager launches (MVS4net.forward would replay a captured graph)
    table = {}
    orig = cp.ConvLayer.__call__
    cp.FORCE_VARIANT = None
    calls = []

    def rec(layer, x, skip=None, skip_mode=0, tiles=None):
        calls.append((layer, tuple(x.shape), None if skip is None else tuple(skip.shape), skip_mode if skip is not None else 0))
        return orig(layer, x, skip, skip_mode, tiles)
    for (H, W, N) in shapes:
        imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev)
        del calls[:]
        cp.ConvLayer.__call__ = rec
        model(imgs, proj, dv)
        cp.ConvLayer.__call__ = orig
        torch.cuda.synchronize()
        _tune(list(calls), table, dev)
    from mvster_amd import MVS4net_loss
    model.train()
    for (H, W, N, B) in train_shapes:
        imgs, proj, dv = make_inputs(N, H, W, seed=0, device=dev, batch=B)
        gt = {"stage%d" % s: 500 + 300 * torch.rand(B, H // 2 ** (4 - s), W // 2 ** (4 - s), device=dev) for s in range(1, 5)}
        mask = {k: torch.ones_like(v) for k, v in gt.items()}
        del calls[:]
        cp.ConvLayer.__call__ = rec
        out = model(imgs, proj, dv)
        MVS4net_loss(out, gt, mask, stage_lw=[1, 1, 1, 1], l1ot_lw=[0, 1], inverse_depth=True, ot_iter=10, ot_eps=1,
                     ot_continous=False, mono=True)[0].backward()
        cp.ConvLayer.__call__ = orig
        torch.cuda.synchronize()
        model.zero_grad(set_to_none=True)
        print("-- training step %dx%d N=%d B=%d: %d layer calls" % (H, W, N, B, len(calls)), flush=True)
        _tune(list(calls), table, dev)
    with open(path, "w") as f:
        json.dump(table, f, indent=0, sort_keys=True)
    print("wrote", path, len(table), "entries")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--retune-narrow":
        retune_narrow(sys.argv[2], [(512, 640, 5), (1152, 1600, 5), (1024, 1920, 7), (832, 1152, 5), (128, 192, 5), (128, 192, 3), (64, 128, 3)])
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--untuned":
        # an untuned resolution against the exhaustive per-layer search: what does the family fallback leave on the table?
        H, W, N = (int(v) for v in sys.argv[2:5])
        dev = torch.device("cuda:0")
        model = MVS4net(**SHIPPED)
        model.load_state_dict(load_weights(), strict=True)
        model.to(dev).eval()
        model.graph_cache = False          # instrumented eager launches (MVS4net.forward would replay a captured graph)
        rows = auto_vs_best(record_eval_calls(model, H, W, N, dev), dev)
        for sig, a, b, name, how in rows:
            print("%-52s auto %7.1f us  best %7.1f us (%s)  [%s]%s" % (sig, a, b, name, how, "   <-- %.0f %%" % (100 * (a / b - 1)) if a > 1.05 * b else ""))
        sa, sb = sum(r[1] for r in rows), sum(r[2] for r in rows)
        print("%dx%dx%d: sum of distinct layers auto %.1f us, best-of-candidates %.1f us (+%.1f %%)" % (H, W, N, sa, sb, 100 * (sa / sb - 1)))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--emit":
        shapes = [(512, 640, 5), (1152, 1600, 5), (1024, 1920, 7), (128, 192, 5), (128, 192, 3), (64, 128, 3)]
        emit_table(shapes, sys.argv[2])
        return
    dev = torch.device("cuda:0")
    model = MVS4net(**SHIPPED)
    model.load_state_dict(load_weights(), strict=True)
    model.to(dev).eval()
    model.graph_cache = False          # instrumented eager launches (MVS4net.forward would replay a captured graph)
    imgs, proj, dv = make_inputs(5, 512, 640, seed=0, device=dev)
    calls = []
    orig = cp.ConvLayer.__call__

    def rec(layer, x, skip=None, skip_mode=0, tiles=None):
        calls.append((layer, tuple(x.shape), None if skip is None else tuple(skip.shape), skip_mode))
        return orig(layer, x, skip, skip_mode, tiles)
    cp.ConvLayer.__call__ = rec
    model(imgs, proj, dv)
    cp.ConvLayer.__call__ = orig
    torch.cuda.synchronize()
    print("%-3s %-28s %-22s %9s | %-30s" % ("#", "layer", "input", "GFLOP", "us (TFLOP/s) per variant/tile"))
    tot = {}
    for i, (layer, xs, ss, sm) in enumerate(calls):
        x = torch.randn(*xs, device=dev)
        skip = torch.randn(*ss, device=dev) if ss else None
        B, Di, Hi, Wi, _ = xs
        fl = layer.flops(B, Di, Hi, Wi)
        _, mt, nt, _, var = layer._geom(B, Di, Hi, Wi, sm if skip is not None else 0)
        desc = "%s%dx%dx%d s%s %d->%d" % ("T" if layer.transposed else "", *layer.kernel, layer.stride[1], layer.cin, layer.cout)
        res = []
        cands = [("auto", None)]
        d_mt, d_nt = cp._tiles(B * Di * Hi * Wi // (layer.stride[1] * layer.stride[2]) if not layer.transposed else B * Di * Hi * Wi, layer.ntile_total, len(layer.classes))
        cands.append(("d%d,%d" % (d_mt, d_nt), (d_mt, d_nt, 0)))
        if not layer.transposed and layer.cin % 16 == 0 and layer.kernel[2] in (3, 5) and sm in (0, 1):
            for m in (2, 4):
                patch = layer.kernel[0] * ((2 * m - 1) * layer.stride[1] + layer.kernel[1]) * (31 * layer.stride[2] + layer.kernel[2]) * 64
                if patch > cp.LDS_BUDGET:
                    continue
                for n in (1, 2, 4):
                    if n <= layer.ntile_total and layer.ntile_total % n == 0:
                        cands.append(("L%d,%d" % (m, n), (m, n, 1)))
        if layer.cin >= 16 and B * Di * Hi * Wi <= 400000:
            for n in (1, 2, 4):
                if n <= layer.ntile_total and layer.ntile_total % n == 0:
                    cands.append(("S1,%d" % n, (1, n, 2)))
        best = None
        for name, tiles in cands:
            try:
                us = timeit(lambda: layer(x, skip=skip, skip_mode=sm, tiles=tiles))
            except RuntimeError as e:
                res.append("%s:ERR" % name)
                continue
            res.append("%s:%.0f(%.0f)" % (name, us, fl / us / 1e6))
            if name != "auto" and (best is None or us < best[1]):
                best = (name, us)
            if name == "auto":
                tot["auto"] = tot.get("auto", 0) + us
        tot["best"] = tot.get("best", 0) + (best[1] if best else 0)
        print("%-3d %-28s %-22s %9.2f | auto=v%d(%d,%d) %s" % (i, desc, "x".join(map(str, xs)), fl / 1e9, var, mt, nt, " ".join(res)))
    print("sum auto %.0f us, sum best-of-candidates %.0f us" % (tot["auto"], tot["best"]))


if __name__ == "__main__":
    main()
