"""bench.py's launch protocol without a GPU: ``--gpus 2`` starts its own two ranks (no external launcher), they
rendezvous over gloo on 127.0.0.1, time a stub step between barriers, take the MAX over ranks, and rank 0 prints
exactly one JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "5", "--warmup", "2"] + extra,
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 5 and line["warmup"] == 2
    assert line["value"] > 0 and line["scaling"] == "weak" and line["higher_is_better"] is True
    # every rank's own rate, in rank order (a straggler GPU is visible in the driver's first scaling run)
    assert line["rank_order"] == [0, 1] and len(line["per_rank_value"]) == 2 and min(line["per_rank_value"]) > 0


def test_train_mode_protocol_two_ranks():
    """--mode train: the same launch / barrier / MAX protocol around a step that ends in the bucketed gradient all-reduce
    (shard.GradBucket); after the timed updates every rank holds the same parameters."""
    line = _run(["--gpus", "2", "--mode", "train"])
    assert line["mode"] == "train" and line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert abs(line["param_digest_spread"]) <= 1e-9
    assert line["value"] > 0
    # every rank's own rate in rank order, also in training mode (the driver's scaling run reads it)
    assert line["rank_order"] == [0, 1] and len(line["per_rank_value"]) == 2 and min(line["per_rank_value"]) > 0


def test_train_mode_under_the_drivers_launcher_two_ranks():
    """The driver's own launch line -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1
    ... bench.py --gpus 2 -- through the training protocol: one JSON line from rank 0, two per-rank values."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--stub", "--mode", "train",
                        "--gpus", "2", "--steps", "5", "--warmup", "2"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["mode"] == "train" and line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert line["rank_order"] == [0, 1] and len(line["per_rank_value"]) == 2 and min(line["per_rank_value"]) > 0
    assert abs(line["param_digest_spread"]) <= 1e-9


def test_four_ranks_train_protocol():
    """Four gloo ranks through the training protocol (bucketed all-reduce): every rank reports, the windows are timed on all
    of them (fastest / slowest rank of the reported window in the line), parameters agree afterwards."""
    line = _run(["--gpus", "4", "--mode", "train"])
    assert line["n_gpus"] == 4 and line["ranks_seen"] == 4 and abs(line["param_digest_spread"]) <= 1e-9
    assert line["rank_order"] == [0, 1, 2, 3] and len(line["per_rank_value"]) == 4
    t = line["timing"]
    assert t["windows"] >= 3 and t["windows"] % 2 == 1 and len(t["window_ms"]) == t["windows"] and t["steps_per_window"] == 5
    assert 0 < t["rank_ms_per_step_min"] <= t["rank_ms_per_step_max"]
    assert abs(line["ms_per_step"] * 5 - sorted(t["window_ms"])[t["windows"] // 2]) < 1e-2        # the median window is the one reported


def test_single_rank_needs_no_launcher():
    line = _run(["--gpus", "1"])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1


def test_world_size_mismatch_is_an_error():
    env = {k: v for k, v in os.environ.items()}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0


def test_kernel_source_hash_tracks_the_sources(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
