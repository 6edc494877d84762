"""Multi-process path on CPU: gloo, world_size 2 -- sharding, the bench timing protocol and the
DDP wiring (with the CPU oracle module tree standing in for the GPU model)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mvster_amd import shard


def test_round_robin_partition_is_disjoint_and_complete():
    for world in (1, 2, 4, 8):
        units = [shard.shard_units(49 * 22, r, world) for r in range(world)]
        flat = sorted(u for us in units for u in us)
        assert flat == list(range(49 * 22))
        assert max(len(u) for u in units) - min(len(u) for u in units) <= 1
    pairs = shard.shard_scans(["scan1", "scan4"], 49, 1, 8)
    assert pairs[0] == ("scan1", 1) and len(pairs) in (12, 13)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, lr, w = shard.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # timing protocol: MAX over ranks; throughput: SUM over ranks
    assert shard.max_over_ranks(1.0 + rank) == float(world)
    assert shard.sum_over_ranks(2.0) == 2.0 * world
    shard.barrier()
    # DDP wiring: one all-reduce of the gradients, BN statistics stay per rank (no SyncBN)
    from oracle import mvs4_oracle as O
    torch.manual_seed(0)
    net = O.Reg2d(input_channel=4, base_channel=8)
    ddp = shard.wrap_ddp(net)
    torch.manual_seed(100 + rank)
    x = torch.randn(2, 4, 4, 8, 8)
    ddp(x).square().mean().backward()
    g = net.prob.weight.grad.clone()
    gathered = [torch.zeros_like(g) for _ in range(world)]
    dist.all_gather(gathered, g)
    same = all(torch.equal(gathered[0], t) for t in gathered)
    rm = net.conv0.bn.running_mean.clone()
    rms = [torch.zeros_like(rm) for _ in range(world)]
    dist.all_gather(rms, rm)
    q.put((rank, same, bool(torch.equal(rms[0], rms[1]))))
    dist.destroy_process_group()


def test_two_process_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, same_grads, same_bn in res:
        assert same_grads            # gradients are averaged across ranks
        assert not same_bn           # BatchNorm running stats are NOT synchronised (reference: plain BatchNorm)


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    shard.init_distributed(backend="gloo")
    from oracle import mvs4_oracle as O

    def build():
        torch.manual_seed(0)
        return O.Reg2d(input_channel=4, base_channel=8)
    xs = [torch.randn(2, 4, 4, 8, 8, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    # (a) the bucketed all-reduce (what GraphedTrainStep captures): one flat bucket, one collective
    net = build()
    net(xs[rank]).square().mean().backward()
    own = {k: p.grad.clone() for k, p in net.named_parameters()}
    bucket = shard.GradBucket(net.parameters())
    flat = bucket.sync()
    assert flat.numel() == sum(p.numel() for p in net.parameters())
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))      # slices, no unpack pass
    got = {k: p.grad.clone() for k, p in net.named_parameters()}
    # (b) DistributedDataParallel on the same data: the reference's reduction
    ddp_net = build()
    ddp = shard.wrap_ddp(ddp_net)                   # (keep the wrapper alive through backward: its reducer owns the hooks)
    ddp(xs[rank]).square().mean().backward()
    want = {k: p.grad.clone() for k, p in ddp_net.named_parameters()}
    # (c) by hand: the mean over ranks of the per-rank gradients
    mean = {}
    for k, g in own.items():
        parts = [torch.zeros_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        mean[k] = torch.stack(parts).mean(0)
    # (relative to the largest gradient of the model: some tensors -- a bias in front of a softmax -- have gradients that
    #  are rounding noise only)
    scale = max(v.abs().max().item() for v in want.values())
    worst_ddp = max((got[k] - want[k]).abs().max().item() for k in got) / scale
    worst_mean = max((got[k] - mean[k]).abs().max().item() for k in got) / scale
    # every rank holds the same averaged gradients
    digest = torch.cat([g.reshape(-1) for g in got.values()])
    parts = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(parts, digest)
    # a second step: gradients now live in the bucket; zero_grad(set_to_none=True) + backward + sync again
    net.zero_grad(set_to_none=True)
    net(xs[rank]).square().mean().backward()
    bucket.sync()
    again = max(((p.grad - got[k]).abs().max()).item() for k, p in net.named_parameters())
    q.put((rank, worst_ddp, worst_mean, all(torch.equal(parts[0], t) for t in parts), again))
    dist.destroy_process_group()


def test_bucketed_all_reduce_equals_ddp_two_ranks():
    """shard.GradBucket (one flat bucket + one all-reduce, the capturable form of the gradient exchange) against
    DistributedDataParallel and against the hand-computed mean, 2 gloo ranks on different samples."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, worst_ddp, worst_mean, same_everywhere, again in res:
        assert worst_ddp <= 1e-6, (rank, worst_ddp)
        assert worst_mean <= 1e-6, (rank, worst_mean)
        assert same_everywhere
        assert again <= 1e-7


def test_grad_bucket_single_process_is_a_pack():
    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    lin(torch.randn(5, 4)).sum().backward()
    want = [p.grad.clone() for p in lin.parameters()]
    b = shard.GradBucket(lin.parameters())
    flat = b.sync()
    assert flat.numel() == 4 * 3 + 3 + 3 * 2 + 2 and b.world() == 1
    for p, w in zip(lin.parameters(), want):
        assert torch.equal(p.grad, w)
    # a parameter that received no gradient contributes zeros
    lin.zero_grad(set_to_none=True)
    lin[1](torch.randn(5, 3)).sum().backward()
    b.sync()
    assert lin[0].weight.grad.abs().max() == 0 and lin[1].weight.grad.abs().max() > 0


def test_cpulist_parser_and_numa_pin_is_best_effort(tmp_path):
    from mvster_amd import shard
    assert shard.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert shard.parse_cpulist("5") == [5] and shard.parse_cpulist("") == []
    before = os.sched_getaffinity(0)
    assert shard.pin_to_gpu_numa(0, sysfs=str(tmp_path)) is None           # no GPU / no sysfs entry: nothing changes
    assert os.sched_getaffinity(0) == before


def test_grad_bucket_mixed_and_aliased_gradients():
    """GradBucket.sync outside the captured step: a gradient that already is its bucket slice survives a later step in which
    another parameter has no gradient (only that parameter's slice is zeroed), and partially aliased gradients never feed a
    copy that overlaps them."""
    torch.manual_seed(0)
    a, b, c = (torch.nn.Parameter(torch.randn(n)) for n in (3, 4, 2))
    bucket = shard.GradBucket([a, b, c])
    a.grad, b.grad, c.grad = torch.ones(3), 2 * torch.ones(4), 3 * torch.ones(2)
    bucket.sync()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip((a, b, c), bucket.views))
    # next eager step: a accumulates in place (aliased), b gets a fresh tensor, c has no gradient this time
    a.grad.add_(1.0)
    b.grad = 5 * torch.ones(4)
    c.grad = None
    flat = bucket.sync()
    assert torch.equal(flat, torch.tensor([2., 2, 2, 5, 5, 5, 5, 0, 0]))
    assert torch.equal(a.grad, 2 * torch.ones(3)) and torch.equal(c.grad, torch.zeros(2))


class _ToyNet(torch.nn.Module):
    """A module with MVS4net's call signature (imgs list, proj dict, depth_values) -> dict, small enough for the CPU."""

    def __init__(self):
        super().__init__()
        self.conv = torch.nn.Conv2d(6, 4, 3, padding=1)
        self.bn = torch.nn.BatchNorm2d(4)
        self.head = torch.nn.Conv2d(4, 1, 1)

    def forward(self, imgs, proj_matrices, depth_values):
        x = torch.cat(imgs, 1) * proj_matrices["stage1"].mean() + depth_values.mean()
        return {"depth": self.head(torch.relu(self.bn(self.conv(x)))).squeeze(1)}


def _graphed_step_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    shard.init_distributed(backend="gloo")
    from mvster_amd.graph import GraphedTrainStep

    def sample(seed):
        g = torch.Generator().manual_seed(seed)
        imgs = [torch.rand(2, 3, 8, 12, generator=g) for _ in range(2)]
        proj = {"stage1": torch.rand(2, 2, 2, 4, 4, generator=g)}
        dv = torch.rand(2, 2, generator=g)
        gt = {"stage1": torch.rand(2, 8, 12, generator=g)}
        mask = {"stage1": (torch.rand(2, 8, 12, generator=g) > 0.3).float()}
        return imgs, proj, dv, gt, mask

    def loss_fn(out, gt, mask):
        return (((out["depth"] - gt["stage1"]) ** 2) * mask["stage1"]).mean(), None

    torch.manual_seed(0)
    bare = _ToyNet().train()
    torch.manual_seed(0)
    ref = _ToyNet().train()
    ddp = shard.wrap_ddp(ref)
    opt_a = torch.optim.Adam(bare.parameters(), lr=1e-2)
    opt_b = torch.optim.Adam(ref.parameters(), lr=1e-2)
    # the captured step's object and call sequence, without the capture (no GPU here): static buffers, zero_grad, forward,
    # loss, backward, ONE bucketed all-reduce, optimizer step
    step = GraphedTrainStep(bare, opt_a, loss_fn, *sample(1000 + rank), grad_sync=shard.GradBucket(bare.parameters()), capture=False)
    losses = []
    for it in range(4):
        imgs, proj, dv, gt, mask = sample(10 * it + rank)              # every rank its own samples, new ones every step
        la = step(imgs, proj, dv, gt, mask)
        opt_b.zero_grad(set_to_none=True)
        lb = loss_fn(ddp(imgs, proj, dv), gt, mask)[0]
        lb.backward()
        opt_b.step()
        losses.append((float(la), float(lb)))
    worst = max((pa.detach() - pb.detach()).abs().max().item() for pa, pb in zip(bare.parameters(), ref.parameters()))
    flat = torch.cat([p.detach().reshape(-1) for p in bare.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    bn_same = [torch.zeros_like(bare.bn.running_mean) for _ in range(world)]
    dist.all_gather(bn_same, bare.bn.running_mean)
    q.put((rank, losses, worst, all(torch.equal(gathered[0], t) for t in gathered), bool(torch.equal(bn_same[0], bn_same[1]))))
    dist.destroy_process_group()


def test_graphed_train_step_call_sequence_two_ranks_equals_ddp():
    """``GraphedTrainStep(..., grad_sync=GradBucket, capture=False)`` on two gloo ranks, end to end: the step object the GPU
    captures (static input buffers fed per call, one bucketed all-reduce between backward and the optimizer step) trains
    exactly like the reference's DistributedDataParallel loop (train_mvs4.py:389-392, :207-218) -- same losses, same
    parameters after four steps on per-rank data, identical parameters on both ranks, BatchNorm statistics per rank.
    (The capture itself and the RCCL collective inside it: tests/test_gpu_train.py on one rank; > 1 rank needs the driver's node.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_graphed_step_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, losses, worst, same_params, same_bn in res:
        for la, lb in losses:
            assert abs(la - lb) <= 1e-6 * max(1.0, abs(lb)), (rank, losses)
        assert worst <= 1e-6, (rank, worst)
        assert same_params and not same_bn
