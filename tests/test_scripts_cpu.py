"""The evidence tooling that turns rocprofv3 CSVs into the tables under profiles/: the steady-state difference of two training
profiles (scripts/train_categories.py) and the two-in-flight overlap analysis (scripts/inflight_steady_state.py), on synthetic
traces with known answers."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_stats(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs"])
        w.writerows(rows)


def test_train_categories_subtracts_the_short_run(tmp_path):
    """8-step profile minus 3-step profile: 5 steady steps; start-up fills (same count in both runs) vanish."""
    long_, short = tmp_path / "long.csv", tmp_path / "short.csv"
    per_step = [("conv_wgrad_kernel<1>", 2, 2_000_000), ("bn_stats_kernel", 4, 400_000), ("void at::native::fill<float>", 3, 30_000)]
    startup = [("void at::native::fill<float>", 400, 4_000_000), ("__amd_rocclr_copyBuffer", 130, 1_300_000)]
    for path, steps in ((long_, 8), (short, 3)):
        rows = {}
        for n, c, t in per_step:
            rows[n] = [c * steps, t * steps]
        for n, c, t in startup:
            a = rows.setdefault(n, [0, 0])
            a[0] += c
            a[1] += t
        _write_stats(path, [(n, c, t) for n, (c, t) in rows.items()])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_categories.py"), str(long_), "8", str(short), "3"],
                         capture_output=True, text=True, check=True).stdout
    assert "5 steps" in out
    lines = {l.split("launches")[1].strip(): l for l in out.splitlines() if "launches " in l and "total" not in l}
    assert lines["weight gradients (+finish)"].split()[0] == "2.00" and lines["weight gradients (+finish)"].split()[2] == "2"
    assert lines["BatchNorm kernels"].split()[0] == "0.40"
    assert lines["torch (aten) kernels"].split()[0] == "0.03" and lines["torch (aten) kernels"].split()[2] == "3"
    assert "total 2.43 ms, 9 launches per step" in out
    # without the short run the start-up work is smeared over the steps
    raw = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train_categories.py"), str(long_), "8"],
                         capture_output=True, text=True, check=True).stdout
    assert "total 3.09 ms" in raw


def test_inflight_overlap_analysis_on_a_synthetic_trace(tmp_path):
    """Two streams, 60 'forwards' of two kernels each after a synchronise gap: kernel A always overlaps B of the other stream for
    half of its length, kernel C runs alone."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import importlib
    mod = importlib.import_module("inflight_steady_state")
    rows = []
    t = 1_000_000
    for i in range(6000):                              # start-up noise: short serial kernels
        rows.append((t, t + 500, "warmup_kernel()"))
        t += 600
    t += 5_000_000                                     # the synchronize()
    for r in range(mod.NREP):
        for k in range(35):                            # 70 kernels per forward: A (20 us) then C (10 us), B overlaps A's second half
            rows.append((t, t + 20_000, "void (anonymous namespace)::kernel_a<1>(Args)"))
            rows.append((t + 10_000, t + 20_000, "kernel_b(Args)"))
            t += 20_000
    path = tmp_path / "t_kernel_trace.csv"
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for s, e, n in rows:
            w.writerow(["KERNEL_DISPATCH", n, s, e])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "inflight_steady_state.py"), "--analyse", str(path)],
                         capture_output=True, text=True, check=True).stdout
    assert "60 replays, 70.0 kernels per forward" in out
    assert "1 kernels running:  50.0 %" in out and "2 kernels running:  50.0 %" in out and "0 kernels running:   0.0 %" in out
    a_line = [l for l in out.splitlines() if l.strip().endswith("kernel_a<1>")][0].split()
    assert a_line[0] == "350.0" and a_line[2] == "350.0"          # per depth map: 35 x 10 us alone, 35 x 10 us co-running
