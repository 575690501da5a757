"""Data-parallel training step: one process per GPU, batch sharded across ranks, gradients
all-reduced per bucket while backward is still running.

The reference has no distributed code; this is the new exchange step around its per-batch
training body (Trainer::train_epoch, src/model/Trainer.cpp:201-227).  Samples are independent
through forward/backward, so the only cross-rank data are the parameter gradients:

* every rank divides its cross-entropy gradient by the GLOBAL batch, so the SUM of the ranks'
  gradients is the gradient of the global mean loss -- exactly what one device computes on the
  whole batch;
* backward runs in stages (head + final norm, encoder blocks last to first, embeddings), each of
  which finalises one contiguous slice of the flat gradient arena; consecutive slices are merged
  into buckets of at least ``bucket_bytes`` and each bucket is all-reduced (sum) asynchronously as
  soon as its last stage has been enqueued -- with the NCCL (=RCCL) backend the collective runs on
  the process group's own stream behind an event, overlapping the remaining backward kernels;
* Adam runs once all buckets have landed.  All ranks start from rank 0's parameters (broadcast)
  and apply identical updates, so replicas stay bit-identical.

``engine`` is any object with ``num_stages``, ``stage_grad_range(s) -> (offset, count)``,
``step_forward(x, y, grad_batch)``, ``backward_stage(s)``, ``update(lr, grad_scale)`` and the
flat ``grads`` / ``params`` torch tensors: visiontransformer_amd.host via :class:`HostEngine` on a
GPU, or a CPU stand-in in the gloo tests.
"""
import torch
import torch.distributed as dist


class _DeviceSpan:
    """Zero-copy view of raw device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def device_span(ptr, count):
    return torch.as_tensor(_DeviceSpan(ptr, count), device="cuda")


class HostEngine:
    """Adapter from visiontransformer_amd.host (Model + Trainer) to the engine protocol."""

    def __init__(self, model, trainer):
        self.model, self.trainer = model, trainer
        self.num_stages = trainer.num_stages
        self.params = device_span(model.params_device(), model.n)
        self.grads = device_span(model.grads_device(), model.n)

    def stage_grad_range(self, s):
        return self.trainer.stage_grad_range(s)

    def step_forward(self, x, y, grad_batch):
        self.trainer.step_forward(x, y, grad_batch)

    def backward_stage(self, s):
        self.trainer.backward_stage(s)

    def update(self, lr, grad_scale):
        self.trainer.update(lr, grad_scale)


def plan_buckets(ranges, bucket_bytes):
    """Merge the per-stage gradient slices (in the order backward finalises them) into buckets.

    ``ranges[s] = (offset, count)``.  Stages are visited in order; a bucket is closed once it holds
    at least ``bucket_bytes``.  Returns ``[(last_stage, offset, count), ...]``: the bucket may be
    reduced after ``last_stage`` has been enqueued.  Slices of consecutive stages are adjacent in
    the arena (each stage's slice ends where the previous one begins), so a bucket is one span."""
    buckets, cur = [], None
    for s, (off, cnt) in enumerate(ranges):
        if cnt == 0:
            continue
        if cur is None:
            cur = [s, off, cnt]
        else:
            assert off + cnt == cur[1], "stage gradient slices must be adjacent and descending"
            cur = [s, off, cur[2] + cnt]
        if cur[2] * 4 >= bucket_bytes:
            buckets.append(tuple(cur))
            cur = None
    if cur is not None:
        buckets.append(tuple(cur))
    return buckets


class DataParallel:
    def __init__(self, engine, group=None, bucket_bytes=25 << 20, always_reduce=False):
        """always_reduce: issue the collectives even with a single rank (they are then identities);
        lets a one-GPU box exercise the same RCCL calls, streams and waits as a multi-GPU run."""
        self.engine, self.group = engine, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.reduce = self.world > 1 or (always_reduce and dist.is_initialized())
        ranges = [engine.stage_grad_range(s) for s in range(engine.num_stages)]
        self.buckets = plan_buckets(ranges, bucket_bytes)
        self._by_stage = {b[0]: b for b in self.buckets}

    def broadcast_parameters(self, src=0):
        """Start every replica from rank ``src``'s parameters."""
        if self.reduce:
            dist.broadcast(self.engine.params, src=src, group=self.group)

    def step(self, x, y, lr):
        """One optimisation step on this rank's shard (x, y): returns nothing; losses stay on the
        device until the caller asks the engine for them."""
        e = self.engine
        global_batch = float(x.shape[0] * self.world)
        e.step_forward(x, y, global_batch)
        pending = []
        for s in range(e.num_stages):
            e.backward_stage(s)
            b = self._by_stage.get(s)
            if b is not None and self.reduce:
                _, off, cnt = b
                pending.append(dist.all_reduce(e.grads[off:off + cnt], op=dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))
        for w in pending:
            w.wait()
        e.update(lr, 1.0)
