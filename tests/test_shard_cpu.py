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
