"""Multi-GPU: independent replicas over disjoint depth maps (inference) and DDP (training).

The path shards naturally: each depth map (one reference view and its
source views) is an independent unit, so inference uses one process per GPU, weights replicated
(4 MB), round-robin assignment of units to ranks and NO data-path collective -- exactly what the
reference's scan loop does on one GPU (test_mvs4.py:161-164).  Training is the reference's DDP
(train_mvs4.py:321-326, :389-392): one all-reduce of 4.04 MB of fp32 gradients per step, which
``torch.distributed`` backend "nccl" runs on RCCL over xGMI; a single default 25 MB bucket.
``GradBucket`` is the same reduction without DDP's autograd hooks: one flat bucket, ONE all-reduce issued from the
training step itself -- which makes the step capturable in a hipGraph on every rank (``graph.GraphedTrainStep``).
"""
import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_distributed(backend=None):
    """Join the process group described by the environment; returns (rank, local_rank, world)."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, local_rank, world


def shard_units(num_units, rank, world):
    """Round-robin partition of unit indices (depth maps / reference views) for one rank."""
    return list(range(rank, num_units, world))


def shard_scans(scans, views_per_scan, rank, world):
    """Flatten (scan, ref_view) units in the reference's iteration order (test_mvs4.py:161-164,
    datasets/general_eval4.py:111) and keep this rank's share."""
    units = [(s, v) for s in scans for v in range(views_per_scan)]
    return [units[i] for i in shard_units(len(units), rank, world)]


def parse_cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def pin_to_gpu_numa(device_index, sysfs="/sys"):
    """Restrict this process to the cores of the NUMA node its GPU hangs off (one process per GPU: launches and pinned
    copies then never cross the socket interconnect).  Best effort: returns the node, or None when the topology cannot be
    read (no sysfs entry, single-node host, node -1) -- in which case nothing is changed."""
    try:
        prop = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
        with open(os.path.join(sysfs, "bus/pci/devices", addr, "numa_node")) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as f:
            cpus = parse_cpulist(f.read())
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return node
    except (OSError, ValueError, AttributeError, RuntimeError, AssertionError):
        return None


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (timing protocol of bench.py)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(value, device=None):
    """Every rank's python float, in rank order, on every rank (bench.py: per-rank rates, kernel fingerprints)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [float(value)]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def wrap_ddp(model, local_rank=None):
    """DistributedDataParallel exactly as the reference wraps it (train_mvs4.py:389-392).
    No SyncBN: BatchNorm statistics stay per rank, as in the reference."""
    if not (dist.is_available() and dist.is_initialized()):
        return model
    if next(model.parameters()).is_cuda:
        lr = torch.cuda.current_device() if local_rank is None else local_rank
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[lr], output_device=lr)
    return torch.nn.parallel.DistributedDataParallel(model)


class GradBucket:
    """All gradients of a model in ONE flat fp32 bucket (4.04 MB for the shipped network: far below the size where a ring
    over xGMI would be bandwidth-bound, so one collective per step is the right granularity) and one all-reduce per step.

    ``sync()`` packs the gradients that backward produced (``torch.cat``: a few batched-copy launches), all-reduces the
    bucket over the process group (RCCL on the GPU, gloo in the CPU tests), averages, and re-points every ``p.grad`` at its
    slice of the bucket, so the optimizer reads the averaged values without an unpack pass.  No autograd hooks and no host
    synchronisation: unlike DistributedDataParallel's reducer the sequence is a fixed list of launches and can be captured
    in a hipGraph together with forward, backward and the optimizer step.  The result equals DDP's (mean over ranks of the
    per-rank gradients, train_mvs4.py:389-392); BatchNorm statistics stay per rank, as there."""

    def __init__(self, params, group=None, always_reduce=False):
        # always_reduce: issue the collective on a one-rank group too (tests: capturing an RCCL all-reduce on one GPU)
        self.always_reduce = always_reduce
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise RuntimeError("GradBucket: no parameter requires a gradient")
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise RuntimeError("GradBucket: parameters must share one device and dtype")
        self.group = group
        self.sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(self.sizes), device=dev, dtype=dt)
        self.views, o = [], 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[o:o + n].view_as(p))
            o += n

    def world(self):
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def sync(self):
        grads = [p.grad for p in self.params]
        aliased = [g is not None and g.data_ptr() == v.data_ptr() for g, v in zip(grads, self.views)]
        if all(g is not None for g in grads) and not any(aliased):
            # the captured step: every gradient freshly produced by backward -> one batched copy into the bucket
            torch.cat([g.reshape(-1) for g in grads], out=self.flat)
        else:
            # an eager loop: gradients that already live in the bucket (p.grad still is its slice from the last step, i.e.
            # no zero_grad(set_to_none=True) in between) stay where they are -- never an input of a copy into the bucket
            # that overlaps them; a parameter WITHOUT a gradient contributes zeros to the average (another rank may have
            # one: the collective is over the whole bucket), and only ITS slice is zeroed
            for g, v, same in zip(grads, self.views, aliased):
                if g is None:
                    v.zero_()
                elif not same:
                    v.copy_(g)
        w = self.world()
        if w > 1 or (self.always_reduce and dist.is_available() and dist.is_initialized()):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if w > 1:
                self.flat.div_(w)
        # (every parameter ends up with a gradient, zeros where no rank produced one -- DistributedDataParallel leaves those
        #  at None instead; the difference is visible only to optimizers that skip parameters without a gradient, and
        #  keeping the list of launches fixed is what makes the step capturable)
        for p, v in zip(self.params, self.views):
            p.grad = v
        return self.flat
