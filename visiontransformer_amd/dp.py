"""Data-parallel training: one process per GPU, batch sharded across ranks, gradients all-reduced
per bucket while backward is still running.

The reference has no distributed code; this is the new exchange step around its per-batch
training body (Trainer::train_epoch, src/model/Trainer.cpp:163-236).  Samples are independent
through forward/backward, so the only cross-rank data are the parameter gradients:

* every rank divides its cross-entropy gradient by the GLOBAL batch, so the SUM of the ranks'
  gradients is the gradient of the global mean loss -- exactly what one device computes on the
  whole batch;
* backward runs in stages (head + final norm, encoder blocks last to first, embeddings), each of
  which finalises one contiguous slice of the flat gradient arena; consecutive slices are merged
  into buckets of at least ``bucket_bytes`` and each bucket is all-reduced (sum) as soon as its
  last stage has been enqueued;
* Adam runs once all buckets have landed.  All ranks start from rank 0's parameters (broadcast)
  and apply identical updates, so replicas stay bit-identical.

The exchange itself lives in the C++ host layer (core/Comm.hpp: RCCL on a library-owned side
stream gated by HIP events; ``vith_comm_*`` in include/vit_host.h), so a C++ application gets it
from ``Trainer`` alone.  This module is the launcher side for Python callers:

* :func:`init_rccl` -- rank 0 creates the RCCL id, ``torch.distributed`` (any backend, already
  initialised by the launcher) carries it to the other ranks, every rank joins;
* :func:`init_host_staged` -- a transport for tests: ranks that SHARE one GPU (which RCCL refuses)
  exchange through host memory and a CPU process group, one collective at a time;
* :class:`NativeDataParallel` -- the step driver on top of either.

:class:`DataParallel` is the same bucketing and ordering written against an abstract engine with
HOST tensors; it is what the CPU-only (gloo, world size 2) tests exercise with a CPU stand-in
engine.  It refuses device tensors: several asynchronous gloo collectives on device memory, with
two processes sharing one GPU, is a combination that has hung, and the device path above
(RCCL on the library's side stream) replaces it.
"""
import torch
import torch.distributed as dist


class _DeviceSpan:
    """Zero-copy view of raw device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def device_span(ptr, count):
    return torch.as_tensor(_DeviceSpan(ptr, count), device="cuda")


# ------------------------------------------------------------------ native (C++ / RCCL) path
def init_rccl(group=None):
    """Join the host layer's RCCL communicator with the ranks of an initialised torch process
    group (used only to carry the 128-byte id from rank 0 to the others)."""
    from . import host
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [host.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    host.comm_init(rank, world, box[0])
    return rank, world


def init_host_staged(group=None):
    """Transport for ranks that share one GPU: each collective waits for the compute stream, copies
    the span to host memory, reduces it over a CPU process group (gloo) and copies the result
    back before returning -- one collective in flight, nothing but host tensors ever reaches
    gloo.  Returns the list that collects exceptions raised inside the transport."""
    from . import host
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def allreduce(dev, count, stream):
        span = device_span(dev, count)
        torch.cuda.synchronize()
        staged = span.cpu()
        dist.all_reduce(staged, op=dist.ReduceOp.SUM, group=group)
        span.copy_(staged)
        torch.cuda.synchronize()

    def broadcast(dev, count, root, stream):
        span = device_span(dev, count)
        torch.cuda.synchronize()
        staged = span.cpu()
        dist.broadcast(staged, src=root, group=group)
        span.copy_(staged)
        torch.cuda.synchronize()

    return host.comm_init_external(rank, world, allreduce, broadcast, stream_ordered=False)


class NativeDataParallel:
    """Step driver over visiontransformer_amd.host with the exchange done by the C++ layer."""

    def __init__(self, model, trainer, bucket_bytes=25 << 20):
        from . import host
        self.model, self.trainer = model, trainer
        self.world = host.comm_world()
        trainer.set_bucket_bytes(bucket_bytes)

    def broadcast_parameters(self, src=0):
        self.model.broadcast_params(src)

    def step(self, x, y, lr):
        """One optimisation step on this rank's shard; losses stay on the device until asked for."""
        t = self.trainer
        t.step_forward(x, y, float(x.shape[0] * self.world))
        for s in range(t.num_stages):
            t.backward_stage(s)  # + one all-reduce per bucket this stage closes
        t.update(lr, 1.0)        # waits for the buckets on the compute stream, then Adam


# ------------------------------------------------- the same rule against an abstract engine
def plan_buckets(ranges, bucket_bytes):
    """Merge the per-stage gradient slices (in the order backward finalises them) into buckets.

    ``ranges[s] = (offset, count)``.  Stages are visited in order; a bucket is closed once it holds
    at least ``bucket_bytes``.  Returns ``[(last_stage, offset, count), ...]``: the bucket may be
    reduced after ``last_stage`` has been enqueued.  Slices of consecutive stages are adjacent in
    the arena (each stage's slice ends where the previous one begins), so a bucket is one span.
    (The same rule as vit::GradientReducer::planBuckets in the C++ layer.)"""
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
    """``engine`` is any object with ``num_stages``, ``stage_grad_range(s) -> (offset, count)``,
    ``step_forward(x, y, grad_batch)``, ``backward_stage(s)``, ``update(lr, grad_scale)`` and the
    flat ``grads`` / ``params`` torch tensors in host memory."""

    def __init__(self, engine, group=None, bucket_bytes=25 << 20, always_reduce=False):
        """always_reduce: issue the collectives even with a single rank (they are then identities)."""
        if engine.grads.is_cuda or engine.params.is_cuda:
            raise ValueError("dp.DataParallel works on host tensors; device models use NativeDataParallel")
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
