"""HIP-graph replay of the eval forward.

Per-stage tensors are small (the fused warp/aggregation moves ~138 MB for a whole 512x640
depth map), so the ~90 kernel launches of one forward are launch-latency-bound when issued
eagerly.  ``GraphedForward`` captures one ``MVS4net`` eval forward on static input buffers
into a hipGraph (``torch.cuda.CUDAGraph`` is the hipGraph front-end on ROCm; our kernels are
launched on the capturing stream, so they are recorded like any other node) and replays it.

``ForwardCache`` is the same thing behind the reference's own call: ``MVS4net.forward`` keeps one and routes every
eval call through it, so the unchanged ``model(imgs, proj_matrices, depth_values)`` loop of the reference's drivers
(test_mvs4.py:202-207, train_mvs4.py:268) replays a graph from the second call of a shape on.
"""
import collections

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


def _is_dense(t):
    """True when ``t`` covers a gap-free, non-overlapping block of its storage in some dimension order (contiguous tensors
    and their permutations): such a tensor is cloned as one flat run of ``numel`` elements from ``storage_offset``."""
    expect = 1
    for size, stride in sorted(((sz, st) for sz, st in zip(t.shape, t.stride()) if sz != 1), key=lambda p: p[1]):
        if stride != expect:
            return False
        expect *= size
    return True


_HOST_STAGING = {}


def outputs_to_numpy(outputs, keys=None):
    """The reference's ``tensor2numpy(outputs)`` (utils.py:50-57; test_mvs4.py:208) for the dict ``MVS4net`` returns, with
    ONE device -> host copy where that is possible: a ``ForwardCache`` call hands out all its tensors as views of one
    fresh device buffer, which goes through one pinned staging buffer (kept per size) instead of ~40 blocking pageable
    copies of 0.3-10 MB (the flattened last-stage entries are the same tensors as ``stage4``'s: the reference's recursion
    copies them twice).  ``keys``: keep only these entries per stage (e.g. ``("depth", "photometric_confidence")`` -- all
    ``save_depth`` reads, test_mvs4.py:213-264: 6.5 of the 47 MB per 512x640 depth map).  Returns freshly allocated numpy
    arrays in the reference's structure.  Anything else (eager results, other devices) falls back to per-tensor copies."""
    import numpy as np

    def want(k):
        return keys is None or k in keys
    flat = []
    for k, v in outputs.items():
        if isinstance(v, dict):
            flat += [((k, k2), t) for k2, t in v.items() if want(k2) and torch.is_tensor(t)]
        elif torch.is_tensor(v) and want(k):
            flat.append(((k,), v))
    out = {}

    def put(path, arr):
        if len(path) == 1:
            out[path[0]] = arr
        else:
            out.setdefault(path[0], {})[path[1]] = arr
    tensors = [t for _, t in flat]
    one = (len(tensors) > 0 and all(t.is_cuda and t.dtype == torch.float32 for t in tensors)
           and len({t.untyped_storage().data_ptr() for t in tensors}) == 1)
    if not one:
        for path, t in flat:
            put(path, t.detach().cpu().numpy().copy())
        return out
    st = tensors[0].untyped_storage()
    n = st.nbytes() // 4
    dev_flat = torch.empty(0, dtype=torch.float32, device=tensors[0].device).set_(st, 0, (n,), (1,))
    # only the span the wanted tensors cover
    lo = min(t.storage_offset() for t in tensors)
    hi = max(t.storage_offset() + (0 if t.numel() == 0 else 1 + sum((sz - 1) * sd for sz, sd in zip(t.shape, t.stride()))) for t in tensors)
    stage = _HOST_STAGING.get(n)
    if stage is None:
        if len(_HOST_STAGING) > 4:
            _HOST_STAGING.clear()
        stage = _HOST_STAGING[n] = torch.empty(n, dtype=torch.float32).pin_memory()
    stage[lo:hi].copy_(dev_flat[lo:hi], non_blocking=True)
    torch.cuda.current_stream().synchronize()
    seen = {}
    for path, t in flat:
        key = (t.storage_offset(), tuple(t.shape), tuple(t.stride()))
        arr = seen.get(key)
        if arr is None:
            arr = seen[key] = np.array(stage.as_strided(t.shape, t.stride(), t.storage_offset()).numpy(), copy=True)
        put(path, arr)
    return out


class _CachedForward:
    """One captured eval forward of a ``ForwardCache``: static inputs (views of one flat buffer), the graph, the static
    outputs and the recipe that clones them into one fresh allocation per call."""

    def __init__(self, model, imgs, proj_matrices, depth_values):
        dev = imgs[0].device
        segs, total = _sample_layout(imgs, proj_matrices, depth_values)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        views = {}
        for name, off, shape in segs:
            n = 1
            for d in shape:
                n *= d
            views[name] = self.flat[off:off + n].view(shape)
        self.imgs = [views["img%d" % i] for i in range(len(imgs))]
        self.proj = {k: views["proj_" + k] for k in proj_matrices}
        self.depth_values = views["depth_values"]
        self._dst = self.imgs + [self.proj[k] for k in sorted(self.proj)] + [self.depth_values]
        self._proj_keys = sorted(self.proj)
        self.load(imgs, proj_matrices, depth_values)
        # the packed-weight plans the capture records pointers of: kept alive here, whatever happens to the model's own
        self.plans = model._get_plans()
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture: this fires inside the reference's unchanged drivers, where other threads touch the device
        # too (the DataLoader's pin-memory thread, RCCL's watchdog in a DDP run): in the default global mode their calls
        # would invalidate the capture -- and raise in THEIR thread
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = model._forward_eval(self.imgs, self.proj, self.depth_values)
        # ---- clone recipe: every distinct output tensor once, laid out back to back in one fresh buffer per call ---------
        self._slots, self._paths, seen, off = [], [], {}, 0
        for path, t in self._walk(self.outputs):
            j = seen.get(id(t))
            if j is None:
                j = seen[id(t)] = len(self._slots)
                if not _is_dense(t):
                    t_src = t.contiguous()                      # (no output of the forward is like this today)
                    raise RuntimeError("ForwardCache: output %r is not a dense tensor (%s / %s)" % (path, tuple(t.shape), t_src.stride()))
                if t.dtype != torch.float32:
                    raise RuntimeError("ForwardCache: output %r is %s (the clone buffer is float32)" % (path, t.dtype))
                self._slots.append((t, off, tuple(t.shape), tuple(t.stride())))
                off += (t.numel() + 63) // 64 * 64
            self._paths.append((path, j))
        self._out_total = off
        self._src = [t for t, _, _, _ in self._slots]

    @staticmethod
    def _walk(outputs):
        for k, v in outputs.items():
            if isinstance(v, dict):
                for k2, v2 in v.items():
                    yield (k, k2), v2
            else:
                yield (k,), v

    def load(self, imgs, proj_matrices, depth_values):
        """The caller's sample into the static inputs: one multi-tensor copy launch (device fp32 tensors; anything else --
        host tensors, other dtypes -- goes through ``copy_``'s conversions tensor by tensor)."""
        src = list(imgs) + [proj_matrices[k] for k in self._proj_keys] + [depth_values]
        torch._foreach_copy_(self._dst, src)

    def replay(self):
        """Replay on the current stream; -> a dict of the reference's structure whose tensors live in ONE freshly
        allocated buffer (the caller owns them: the next replay does not touch them)."""
        self.graph.replay()
        fresh = torch.empty(self._out_total, dtype=torch.float32, device=self.flat.device)
        new = [fresh.as_strided(shape, stride, off) for _, off, shape, stride in self._slots]
        torch._foreach_copy_(new, self._src)
        out = {}
        for path, j in self._paths:
            if len(path) == 1:
                out[path[0]] = new[j]
            else:
                out.setdefault(path[0], {})[path[1]] = new[j]
        # the reference's dict order: stageN sub-dict first, then its entries flattened (MVS4Net.py:104-105)
        return {k: out[k] for k in self.outputs}


class ForwardCache:
    """Transparent hipGraph cache of ``MVS4net``'s eval forward, keyed on what decides the launch sequence: device, batch,
    views, image size, width of ``depth_values``.  First call of a key (or first call after the model's parameters
    changed): eager, like before.  Second call: capture + replay.  From then on: copy the caller's tensors into the
    static inputs (one launch), replay (one launch), clone the outputs into a fresh buffer (one launch) -- the caller
    gets freshly allocated tensors every time, which is the reference's contract.

    A captured entry is only ever replayed while ``model._state_stamp()`` equals the stamp it was captured at (in-place
    weight updates, ``p.data = ...``, ``load_state_dict``, ``.to()``, a captured training step's replay all change it);
    it holds its own reference to the packed-weight plans it recorded, so rebuilding the model's plans in between (a
    ``train()`` / ``eval()`` round trip without a parameter change) does not invalidate it.
    One host thread per model and device, as in the reference's drivers; least-recently-used entries beyond ``capacity``
    are dropped (each keeps the activations of one forward resident: ~0.7 GB at 512x640x5, ~4 GB at 1152x1600x5)."""

    def __init__(self, capacity=4):
        self.capacity = capacity
        self.entries = collections.OrderedDict()       # key -> [stamp, _CachedForward or None]
        self.stats = {"eager": 0, "captured": 0, "replayed": 0, "capture_failed": 0}
        self.disabled_keys = set()

    @staticmethod
    def key(model, imgs, proj_matrices, depth_values):
        im = imgs[0]
        # (the attributes of the model that pick kernels or streams: flipping one between two calls is a different graph)
        from . import conv_plan, ops
        # (+ the module-level switches that pick kernels: flipped between two calls they are a different launch sequence)
        glob = (conv_plan.FORCE_VARIANT, conv_plan.FUSE_TAIL, conv_plan.FUSE_CONV0, conv_plan.NARROW_PAIR_WPC, conv_plan.FUSE_SELECT, conv_plan.LDS_BUDGET, conv_plan.NARROW_MIN_VOXELS,
                id(conv_plan._TUNING), ops.WGRAD_MAX_SLOTS)
        cfg = (glob, model.warp_variant, getattr(model, "fuse_hypotheses", None), getattr(model, "merge_launches", None), model.overlap_streams, float(model.attn_temp), model.attn_fuse_d, model.num_stage,
               tuple(model.stage_splits), tuple(model.depth_interals_ratio), tuple(model.group_cor_dim))
        return (im.device.index, len(imgs), tuple(im.shape), int(depth_values.shape[1]), tuple(sorted(proj_matrices.keys())), cfg)

    def clear(self):
        self.entries.clear()

    def __call__(self, model, imgs, proj_matrices, depth_values):
        key = self.key(model, imgs, proj_matrices, depth_values)
        if key in self.disabled_keys:
            self.stats["eager"] += 1
            return model._forward_eval(imgs, proj_matrices, depth_values)
        stamp = model._state_stamp()
        hit = self.entries.get(key)
        if hit is not None and hit[0] == stamp:
            self.entries.move_to_end(key)
            if hit[1] is None:
                try:
                    hit[1] = _CachedForward(model, imgs, proj_matrices, depth_values)
                    self.stats["captured"] += 1
                except RuntimeError as e:
                    # a forward that cannot be captured keeps working eagerly; said once, not silently
                    import warnings
                    warnings.warn("mvster_amd: hipGraph capture of the eval forward failed for %r (%s): this shape stays on "
                                  "eager launches" % (key, str(e)[:200]))
                    self.stats["capture_failed"] += 1
                    self.disabled_keys.add(key)
                    del self.entries[key]
                    return model._forward_eval(imgs, proj_matrices, depth_values)
            else:
                hit[1].load(imgs, proj_matrices, depth_values)
            self.stats["replayed"] += 1
            return hit[1].replay()
        # first sight of this shape, or the parameters moved since: eager now, capture when the same state comes back
        self.entries[key] = [stamp, None]
        self.entries.move_to_end(key)
        while len(self.entries) > self.capacity:
            self.entries.popitem(last=False)
        self.stats["eager"] += 1
        return model._forward_eval(imgs, proj_matrices, depth_values)


class GraphedForward:
    def __init__(self, model, imgs, proj_matrices, depth_values, warmup=2, packed=False):
        """``packed=True``: the static inputs are views of ONE flat device buffer (``self.flat``, layout of
        ``pack_sample``), so that a new sample arrives with a single host -> device copy (``load_packed``).
        The captured graph reads the weights as they were folded at capture: ``__call__`` compares the model's state
        stamp with the one recorded here and raises if the parameters have changed since (``check_state=False`` skips
        the ~60 us check; ``self.graph.replay()`` is the raw replay)."""
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
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.plans = model._get_plans()                # (kept alive: the graph holds raw pointers into them)
        self.stamp = model._state_stamp()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = self._eager()

    def _eager(self):
        # (plain launches: MVS4net.forward itself would route the warm-up calls through its own graph cache)
        fn = getattr(self.model, "forward_eager", self.model)
        return fn(self.imgs, self.proj, self.depth_values)

    def load_packed(self, host_flat):
        """Queue ONE non-blocking copy of a ``pack_sample`` buffer into the static inputs (current stream)."""
        if self.flat is None:
            raise RuntimeError("GraphedForward.load_packed: build the graph with packed=True")
        self.flat.copy_(host_flat, non_blocking=True)

    def __call__(self, imgs=None, proj_matrices=None, depth_values=None, check_state=True):
        """Copy new inputs (same shapes) into the static buffers and replay; returns the static
        output dict (overwritten by the next replay)."""
        if check_state and self.model._state_stamp() != self.stamp:
            raise RuntimeError("GraphedForward: the model's parameters or buffers changed after the capture (the graph "
                               "replays the weights folded at capture time): build a new GraphedForward")
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

    # HIP streams the postponed weight-gradient kernels are spread over (train_ops.deferred_wgrad_finish).  They are persistent
    # grids sized for an empty chip: beside each other (or beside the backward chain: wgrad_overlap) they lose more than the
    # overlap gives.  Measured on the config-4 step at the end of round 6: 1 stream 11.85 ms, 2 11.95, 3 12.90, 4 14.13; on one
    # side stream beside the backward chain instead of after it 12.69 (mid-round, before the kernels' rewrite: 13.25 / 13.12 /
    # 13.61 for 1 / 2 / 4).  With the cascade stages' backward on a side stream (MVS4net.train_side_stages) the balance moved:
    # 1 stream 10.87 ms, 2 10.60, 3 10.69, 4 10.82 (same box, alternating) -- two is the default.  Dealing the kernels by
    # estimated duration (wgrad_policy = "lpt") instead of round-robin in autograd's order: the same 10.6.
    wgrad_streams = 2
    wgrad_policy = "rr"
    wgrad_early = True              # the kernels collected before the FPN's backward begins are launched there (train_ops.wgrad_flush_point)
    wgrad_overlap = False           # True: the kernels run on one side stream beside the backward chain, not after it

    def __init__(self, model, optimizer, loss_fn, imgs, proj_matrices, depth_values, depth_gt_ms, mask_ms, warmup=3,
                 grad_sync=None, capture=True):
        """``capture=False``: the same object without the hipGraph -- every call runs the identical sequence (static input
        buffers, zero_grad, forward, loss, backward, ``grad_sync.sync()``, optimizer step) eagerly.  That is what the
        multi-process CPU test drives over gloo (tests/test_shard_cpu.py: two ranks end to end against DistributedDataParallel);
        it is also the fallback when a step cannot be captured."""
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures a training step: call model.train() first")
        for group in optimizer.param_groups:
            if capture and not group.get("capturable", False):
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
        self.graph = None
        from . import train_ops
        self._cache = train_ops.CACHE
        self._cell = [0]
        if not capture:
            self.loss = None
            return
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                    # builds the cached layers and the optimizer state, warms the allocator
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the ~130 per-layer weight refreshes a captured step would record become one launch (train_ops._LayerCache)
        # the captured optimizer update moves the parameters (and BatchNorm's running statistics) without touching their
        # version counters: one epoch cell for everything this step owns, bumped after every replay
        bare = model.module if hasattr(model, "module") and isinstance(model.module, torch.nn.Module) else model
        owned = {id(p): p for g in optimizer.param_groups for p in g["params"]}
        for t in list(bare.parameters()) + list(bare.buffers()):
            owned.setdefault(id(t), t)
        for t in owned.values():
            self._cache.cells.set(t, self._cell)
        self._batch = self._cache.build_batch(list(model.parameters()))      # (kept alive here: the graph reads its tables)
        self.graph = torch.cuda.CUDAGraph()
        # (with a collective in the step, RCCL's watchdog thread touches the device during the capture: relaxed mode)
        collective = grad_sync is not None and (grad_sync.world() > 1 or grad_sync.always_reduce)
        mode = {"capture_error_mode": "thread_local"} if collective else {}
        with torch.cuda.graph(self.graph, **mode):
            self.loss = self._step()

    def close(self):
        """Unregister this step's epoch cell (the model's tensors are then stamped by version / address alone again)."""
        self._cache.cells.drop(self._cell)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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
            # (nothing reads a weight gradient before the backward pass is over -- the bucketed all-reduce and the optimizer
            #  come after it -- so the 64 finishing launches of the weight-gradient kernels are issued as one)
            from .train_ops import deferred_wgrad_finish
            with deferred_wgrad_finish(streams=self.wgrad_streams, overlap=self.wgrad_overlap, policy=self.wgrad_policy, early=self.wgrad_early):
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
        if self.graph is None:
            self.loss = self._step()                   # (capture=False: the same sequence, eagerly)
            return self.loss
        sync = getattr(self.optimizer, "sync_hyperparameters", None)
        if sync is not None:
            sync()                                     # (optim.FusedAdam: a scheduler's learning rate into the device cell)
        self.graph.replay()
        # the replayed optimizer update moved the parameters without touching their version counters: everything folded or
        # packed from them outside this graph (the eval plans, the cached training layers of an eager step) is stale now
        self._cell[0] += 1
        return self.loss
