#!/usr/bin/env python3
"""Training-throughput benchmark of the MI355X-native ViT training step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json): ViT-B/16, 224x224x3 synthetic images, 1000 classes, batch 256 PER GPU
(weak scaling; global batch 2048 on 8 GPUs), fp32 end to end, dropout 0, Adam(L2) lr 3e-4 /
wd 1e-4 under the reference's cosine/warm-up schedule.  One step = forward + softmax-CE +
backward (+ bucketed RCCL gradient all-reduce overlapped with backward when N > 1) + Adam, with
the batch already resident in HBM.  Weights are random-init N(0, 0.02) (same on every rank).

Prints ONE JSON line on rank 0: images/s over all GPUs, plus
  roofline     -- the dominant kernel (the fp32 matrix-core GEMM instantiation with the largest
                  total time), timed live with HIP events around every launch inside the timed
                  region, against the 157.3 TFLOP/s dense fp32 MFMA peak;
  cpu_baseline -- (N = 1 only) the reference CPU path (oracle/_ref, the reference's own sources)
                  or its restatement (oracle/) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VIT_B16 = dict(image_size=224, patch_size=16, in_channels=3, num_classes=1000, embedding_dim=768, num_heads=12,
               num_layers=12, mlp_hidden_dim=3072)
WORKLOADS = {
    "vit_b16": (VIT_B16, 256),
    "vit_s16": (dict(VIT_B16, embedding_dim=384, num_heads=6, mlp_hidden_dim=1536), 256),
    "vit_ti16": (dict(VIT_B16, embedding_dim=192, num_heads=3, mlp_hidden_dim=768), 256),
    "blood_p7": (dict(image_size=28, patch_size=7, in_channels=3, num_classes=8, embedding_dim=192, num_heads=6,
                      num_layers=6, mlp_hidden_dim=768), 512),
    "mnist_p14": (dict(image_size=28, patch_size=14, in_channels=1, num_classes=10, embedding_dim=96, num_heads=8,
                       num_layers=8, mlp_hidden_dim=384), 128),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # dense fp32 matrix peak of MI355X (v_mfma_f32_32x32x2_f32)


def train_flops_per_image(c):
    """Algorithmic FLOPs of one training step per image: 3 x forward GEMM FLOPs (2 FLOP/MAC)."""
    n_p = (c["image_size"] // c["patch_size"]) ** 2
    n, d, h = n_p + 1, c["embedding_dim"], c["mlp_hidden_dim"]
    pd = c["patch_size"] ** 2 * c["in_channels"]
    fwd = 2 * (n_p * pd * d + c["num_layers"] * (4 * n * d * d + 2 * n * n * d + 2 * n * d * h) + d * c["num_classes"])
    return 3 * fwd


def executed_train_flops_per_image(c):
    """The same count when the last encoder block and the final norm work on the CLS token only (the
    default; vith_set_first_token_tail): its keys / values still come from all tokens, its query,
    attention row, output projection and feed-forward from one token."""
    n_p = (c["image_size"] // c["patch_size"]) ** 2
    n, d, h = n_p + 1, c["embedding_dim"], c["mlp_hidden_dim"]
    dense_block = 4 * n * d * d + 2 * n * n * d + 2 * n * d * h
    tail_block = 2 * n * d * d + 2 * d * d + 2 * n * d + 2 * d * h
    return train_flops_per_image(c) - 3 * 2 * (dense_block - tail_block)


def cpu_baseline(cfgd, seconds_budget=30.0):
    """Time the reference CPU path on the host: one training step on a tiny batch of the same model."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import vitlibs as V
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    cfg = V.Cfg.make(**cfgd)
    B = 2
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((B, cfg.in_channels, cfg.image_size, cfg.image_size)).astype(np.float32)
    y = V.onehot(np.random.default_rng(4321).integers(0, cfg.num_classes, B), cfg.num_classes)
    logits = V.f32(B, cfg.num_classes)
    R = V.ref()
    if R is not None:
        s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)  # the reference's own random init
        step = lambda: R.vref_train_step(s, x, y, B, 3e-4, logits, None)
        kind = "reference"
    else:
        O = V.oracle()
        s = O.vo_session_create(cfg, 0.9, 0.999, 1e-8, 1e-4)
        O.vo_session_set_params(s, V.init_params(cfg, 42))
        step = lambda: O.vo_session_train_step(s, x, y, B, 3e-4, logits, None)
        kind = "port"
    t0 = time.perf_counter()
    steps = 0
    while True:
        step()
        steps += 1
        el = time.perf_counter() - t0
        if el * (steps + 1) / steps > seconds_budget or steps >= 5:
            break
    return {"value": round(B * steps / el, 4), "unit": "images/s", "cores": cores, "kind": kind,
            "sample": f"{steps} training step(s) of the same model at batch {B} "
                      f"(forward + CE + backward + Adam), {el:.1f} s of host time, OpenMP threads = {cores}"}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC profile of this same command
    (rocprofv3 cannot collect counters from inside a timed run); None if there is no profile."""
    try:
        with open(os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def main():
    # stdout carries exactly one line, the result: everything libraries print on the way (RCCL's
    # version banner, for one) goes to stderr, and the JSON line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="vit_b16", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=25.0)
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise RCCL and issue the bucket all-reduces even with one rank (rehearsal of the "
                         "multi-GPU code path on a one-GPU box)")
    ap.add_argument("--dense-tail", action="store_true",
                    help="compute every token of the last encoder block as the reference does, instead of only the "
                         "CLS token that the classification head reads (same results; see include/vit_host.h)")
    ap.add_argument("--eval", action="store_true",
                    help="time the evaluation path instead (Trainer::evaluate's per-batch body: forward with dropout "
                         "off + arg-max on the device; its own metric name, no roofline object)")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="dropout rate (0 = the headline/parity configuration; 0.1 adds the device RNG cost)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from visiontransformer_amd import dp, host, vitk

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    host.init(local_rank)
    host.use_torch_stream()
    host.set_first_token_tail(not args.dense_tail)

    cfgd, batch = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    cfg = host.Config.make(**cfgd, dropout_rate=args.dropout)
    host.seed(42)  # identical N(0, 0.02) init on every rank (then broadcast from rank 0 anyway)
    model = host.Model(cfg, patch_mode=vitk.PATCH_IM2COL)
    trainer = host.Trainer(model, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-4)
    if world > 1 or args.force_collectives:
        dp.init_rccl()  # the host layer's own RCCL communicator; torch's group only carries the id
    ddp = dp.NativeDataParallel(model, trainer, bucket_bytes=int(args.bucket_mb * (1 << 20)))
    ddp.broadcast_parameters(0)

    # synthetic batch, resident in HBM: images N(0,1) (the scale of normalised MNIST), uniform labels
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    x = torch.randn(batch, cfg.in_channels, cfg.image_size, cfg.image_size, device="cuda", generator=g)
    labels = torch.randint(0, cfg.num_classes, (batch,), device="cuda", generator=g)
    y = torch.nn.functional.one_hot(labels, cfg.num_classes).to(torch.float32).contiguous()

    total_steps = args.warmup + args.steps
    warm = max(1, int(0.1 * total_steps))
    lr = lambda s: host.cosine_warmup_lr(s, warm, total_steps, 3e-4)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.eval:
        # forward only, one rank's batch per step, nothing exchanged between ranks
        for s in range(args.warmup):
            pred = model.predict(x)
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            pred = model.predict(x)
        barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0:
            f_fwd = train_flops_per_image(cfgd) / 3
            value = batch * world * args.steps / elapsed
            out = {"metric": "eval images/sec (forward + arg-max, dropout off)", "value": round(value, 2),
                   "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": args.workload, "batch_per_gpu": batch, "global_batch": batch * world,
                              "last_block": "all tokens" if args.dense_tail else "CLS token only after its keys/values",
                              "forward_gflop_per_image": round(f_fwd / 1e9, 2),
                              "frac_of_fp32_mfma_peak": round(value / world * f_fwd / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                              "classes_predicted": int(pred.unique().numel())}}
            os.write(json_fd, (json.dumps(out) + "\n").encode())
        if dist.is_initialized():
            dist.barrier()
            host.comm_destroy()
            dist.destroy_process_group()
        return

    for s in range(args.warmup):
        ddp.step(x, y, lr(s))
    barrier()
    # Per-launch GEMM timing (two HIP events around every contraction, on its stream) runs inside the
    # timed region for the large models, where a launch lasts a millisecond.  The small models'
    # contractions last microseconds -- two more packets per launch would be a quarter of the step
    # -- so theirs are timed in a pass of the same steps after the timed region.
    events_in_region = cfgd["image_size"] >= 224
    vitk.gemm_profile(events_in_region)
    t0 = time.perf_counter()
    for s in range(args.steps):
        ddp.step(x, y, lr(args.warmup + s))
    enqueued = time.perf_counter() - t0  # the host has issued every launch; the GPU may still be working
    barrier()
    elapsed = time.perf_counter() - t0
    loss, _ = trainer.result()
    if not events_in_region:
        vitk.gemm_profile(True)
        for s in range(args.steps):
            ddp.step(x, y, lr(args.warmup + s))
        barrier()
    vitk.gemm_profile(False)
    prof = vitk.gemm_profile_read()

    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = batch * world * args.steps / elapsed

    out = None
    if rank == 0:
        dom = max(prof, key=lambda r: r["ms"])
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        gemm_ms = sum(r["ms"] for r in prof)
        f_img = train_flops_per_image(cfgd)
        out = {
            "metric": "train images/sec (ViT-B/16 224\u00b2, 1/2/4/8 MI355X) + % MFMA roofline",
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: ViT {cfgd['image_size']}x{cfgd['image_size']}x{cfgd['in_channels']}"
                                   f" patch {cfgd['patch_size']} d={cfgd['embedding_dim']} L={cfgd['num_layers']}"
                                   f" h={cfgd['num_heads']} mlp={cfgd['mlp_hidden_dim']} K={cfgd['num_classes']}"
                                   + (" (K=1000 assumed: BASELINE.json does not give the class count)"
                                      if cfgd["num_classes"] == 1000 else "") + ", one training"
                                   " step = fwd + softmax-CE + bwd + Adam(L2), dropout " + (f"{args.dropout:g}") + ", random-init weights",
                       "batch_per_gpu": batch, "global_batch": batch * world,
                       "parallelism": f"dp{world}" + (f" ({trainer.num_buckets()} gradient buckets, ncclAllReduce on a side"
                                                      " HIP stream overlapped with backward)"
                                                      if (world > 1 or args.force_collectives) else ""),
                       "patch_mode": "im2col", "params": int(model.n), "final_loss": round(float(loss), 6),
                       "last_block": "all tokens (as the reference)" if args.dense_tail else
                                     "CLS token only after its keys/values (the head reads nothing else; same logits, "
                                     "loss, gradients and updates)"},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 2),
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": measured_traffic(dom["name"]),
                         "traffic_note": "HBM bytes per launch, rocprofv3 PMC passes of this command "
                                         "(profiles/r02_hbm_traffic.json)",
                         "timing": "HIP events around every GEMM launch, " +
                                   ("inside the timed region" if events_in_region else
                                    f"in a separate pass of {args.steps} more steps after the timed region"),
                         "host_enqueue_ms_per_step": round(1e3 * enqueued / args.steps, 3),
                         "launches": dom["launches"], "avg_launch_ms": round(dom["ms"] / dom["launches"], 4),
                         "flops_per_launch": dom["flops"] / dom["launches"],
                         "all_gemm_ms_per_step": round(gemm_ms / args.steps, 3),
                         "step_frac_of_peak": round(value / world * f_img / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                         "train_gflop_per_image": round(f_img / 1e9, 2),
                         "train_gflop_per_image_executed": round((f_img if args.dense_tail else
                                                                  executed_train_flops_per_image(cfgd)) / 1e9, 2),
                         "step_frac_of_peak_executed": round(
                             value / world * (f_img if args.dense_tail else
                                              executed_train_flops_per_image(cfgd)) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                         "gemm_variants": [{"kernel": r["name"], "launches": r["launches"],
                                            "avg_ms": round(r["ms"] / r["launches"], 4),
                                            "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2)} for r in prof]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfgd)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        host.comm_destroy()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
