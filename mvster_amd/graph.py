"""HIP-graph replay of the eval forward.

Per-stage tensors are small (the fused warp/aggregation moves ~138 MB for a whole 512x640
depth map), so the ~90 kernel launches of one forward are launch-latency-bound when issued
eagerly.  ``GraphedForward`` captures one ``MVS4net`` eval forward on static input buffers
into a hipGraph (``torch.cuda.CUDAGraph`` is the hipGraph front-end on ROCm; our kernels are
launched on the capturing stream, so they are recorded like any other node) and replays it.
"""
import torch


def _sample_layout(imgs, proj_matrices, depth_values):
    """Element offsets of a sample's tensors inside one flat fp32 buffer (each segment 256-byte aligned)."""
    segs, off = [], 0
    for name, t in ([("img%d" % i, t) for i, t in enumerate(imgs)] + [("proj_" + k, proj_matrices[k]) for k in sorted(proj_matrices)]
                    + [("depth_values", depth_values)]):
        segs.append((name, off, tuple(t.shape)))
        off += (t.numel() + 63) // 64 * 64
    return segs, off


def pack_sample(imgs, proj_matrices, depth_values, pin=True):
    """One flat (pinned) host buffer holding a sample in the layout of ``GraphedForward(..., packed=True).flat``:
    a whole sample then moves host -> device as ONE copy (``load_packed``) instead of one per tensor."""
    segs, total = _sample_layout(imgs, proj_matrices, depth_values)
    flat = torch.zeros(total, dtype=torch.float32)
    if pin:
        flat = flat.pin_memory()
    src = {("img%d" % i): t for i, t in enumerate(imgs)}
    src.update({"proj_" + k: v for k, v in proj_matrices.items()})
    src["depth_values"] = depth_values
    for name, off, shape in segs:
        flat[off:off + src[name].numel()].copy_(src[name].reshape(-1).float().cpu())
    return flat


class GraphedForward:
    def __init__(self, model, imgs, proj_matrices, depth_values, warmup=2, packed=False):
        """``packed=True``: the static inputs are views of ONE flat device buffer (``self.flat``, layout of
        ``pack_sample``), so that a new sample arrives with a single host -> device copy (``load_packed``)."""
        if model.training:
            raise RuntimeError("GraphedForward captures the eval forward")
        self.model = model
        self.flat = None
        if packed:
            segs, total = _sample_layout(imgs, proj_matrices, depth_values)
            self.flat = torch.zeros(total, dtype=torch.float32, device=depth_values.device)
            views = {name: self.flat[off:off + int(torch.tensor(shape).prod())].view(shape) for name, off, shape in segs}
            self.imgs = [views["img%d" % i] for i in range(len(imgs))]
            self.proj = {k: views["proj_" + k] for k in proj_matrices}
            self.depth_values = views["depth_values"]
            for dst, src_ in zip(self.imgs, imgs):
                dst.copy_(src_)
            for k in self.proj:
                self.proj[k].copy_(proj_matrices[k])
            self.depth_values.copy_(depth_values)
        else:
            self.imgs = [i.clone() for i in imgs]
            self.proj = {k: v.clone() for k, v in proj_matrices.items()}
            self.depth_values = depth_values.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                    # builds the plans, warms the allocator
                model(self.imgs, self.proj, self.depth_values)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = model(self.imgs, self.proj, self.depth_values)

    def load_packed(self, host_flat):
        """Queue ONE non-blocking copy of a ``pack_sample`` buffer into the static inputs (current stream)."""
        if self.flat is None:
            raise RuntimeError("GraphedForward.load_packed: build the graph with packed=True")
        self.flat.copy_(host_flat, non_blocking=True)

    def __call__(self, imgs=None, proj_matrices=None, depth_values=None):
        """Copy new inputs (same shapes) into the static buffers and replay; returns the static
        output dict (overwritten by the next replay)."""
        if imgs is not None:
            for dst, src in zip(self.imgs, imgs):
                dst.copy_(src, non_blocking=True)
        if proj_matrices is not None:
            for k in self.proj:
                self.proj[k].copy_(proj_matrices[k], non_blocking=True)
        if depth_values is not None:
            self.depth_values.copy_(depth_values, non_blocking=True)
        self.graph.replay()
        return self.outputs


class GraphedTrainStep:
    """One whole training step -- forward, loss, backward, optimizer update -- captured in a hipGraph.

    The native training path has no device synchronisation and no host-dependent control flow left in it, so a step is
    a fixed sequence of ~3 000 launches; issued eagerly it is launch-bound on the host (41 ms of enqueue for ~35 ms
    of kernels at 512x640x5, B=2).  Captured once on static input buffers, a step is one ``replay()``.

    Multi-GPU: DistributedDataParallel issues its collectives from autograd hooks and cannot be captured this way; pass
    ``grad_sync=shard.GradBucket(model.parameters())`` instead (the bare model, not the DDP wrapper): the step then packs
    the gradients into one 4 MB bucket and issues ONE all-reduce (RCCL, capturable) between backward and the optimizer
    update, inside the graph -- every rank replays the same captured step (the reference's DDP semantics,
    train_mvs4.py:389-392: averaged gradients, per-rank BatchNorm statistics).  The optimizer must be built with
    ``capturable=True`` (torch.optim.Adam / AdamW); the loss function takes ``(outputs, depth_gt_ms, mask_ms)`` and
    returns the scalar to minimise first, like ``MVS4net_loss``.
    """

    def __init__(self, model, optimizer, loss_fn, imgs, proj_matrices, depth_values, depth_gt_ms, mask_ms, warmup=3,
                 grad_sync=None):
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures a training step: call model.train() first")
        for group in optimizer.param_groups:
            if not group.get("capturable", False):
                raise RuntimeError("GraphedTrainStep: build the optimizer with capturable=True")
        if isinstance(model, torch.nn.parallel.DistributedDataParallel):
            raise RuntimeError("GraphedTrainStep: pass the bare model and grad_sync=shard.GradBucket(model.parameters()) "
                               "(DistributedDataParallel's hook-driven reducer cannot be captured)")
        self.model, self.optimizer, self.loss_fn, self.grad_sync = model, optimizer, loss_fn, grad_sync
        self.imgs = [i.clone() for i in imgs]
        self.proj = {k: v.clone() for k, v in proj_matrices.items()}
        self.depth_values = depth_values.clone()
        self.gt = {k: v.clone() for k, v in depth_gt_ms.items()}
        self.mask = {k: v.clone() for k, v in mask_ms.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                    # builds the cached layers and the optimizer state, warms the allocator
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the ~130 per-layer weight refreshes a captured step would record become one launch (train_ops._LayerCache)
        from . import train_ops
        self._cache = train_ops.CACHE
        self._batch = self._cache.build_batch(list(model.parameters()))      # (kept alive here: the graph reads its tables)
        self.graph = torch.cuda.CUDAGraph()
        # (with a collective in the step, RCCL's watchdog thread touches the device during the capture: relaxed mode)
        collective = grad_sync is not None and (grad_sync.world() > 1 or grad_sync.always_reduce)
        mode = {"capture_error_mode": "thread_local"} if collective else {}
        with torch.cuda.graph(self.graph, **mode):
            self.loss = self._step()

    def _step(self):
        # grads are re-created by every backward: inside the capture they come from the graph's private pool, so a replay
        # writes them in place and no zero-fill / accumulate kernels are recorded
        self.optimizer.zero_grad(set_to_none=True)
        batched = getattr(self, "_batch", None) is not None
        if batched:
            self._cache.run_batch(self._batch)
        try:
            out = self.model(self.imgs, self.proj, self.depth_values)
            res = self.loss_fn(out, self.gt, self.mask)
            loss = res[0] if isinstance(res, (tuple, list)) else res
            loss.backward()
        finally:
            if batched:
                self._cache.end_batch()
        if self.grad_sync is not None:
            self.grad_sync.sync()          # pack -> one all-reduce -> p.grad = slices of the averaged bucket
        self.optimizer.step()
        return loss.detach()

    def __call__(self, imgs=None, proj_matrices=None, depth_values=None, depth_gt_ms=None, mask_ms=None):
        """Copy a new sample (same shapes) into the static buffers and run the captured step; returns the static loss
        tensor (overwritten by the next call)."""
        if imgs is not None:
            for dst, src in zip(self.imgs, imgs):
                dst.copy_(src, non_blocking=True)
        for static, new in ((self.proj, proj_matrices), (self.gt, depth_gt_ms), (self.mask, mask_ms)):
            if new is not None:
                for k in static:
                    static[k].copy_(new[k], non_blocking=True)
        if depth_values is not None:
            self.depth_values.copy_(depth_values, non_blocking=True)
        self.graph.replay()
        # the replayed optimizer update moved the parameters without touching their version counters: everything folded or
        # packed from them outside this graph (the eval plans, the cached training layers of an eager step) is stale now
        self._cache.epoch += 1
        return self.loss
