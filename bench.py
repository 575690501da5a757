#!/usr/bin/env python3
"""Training-throughput benchmark of the MI355X-native ViT training step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher (WORLD_SIZE unset) and asked for more than one GPU -- or for the
collectives on one -- the command starts its N ranks itself, one child process per GPU, before
anything touches a GPU, prints rank 0's result line and fails if any rank fails or the run exceeds
--launch-timeout.  Under a launcher that has set RANK / LOCAL_RANK / WORLD_SIZE it is one rank.

Workload (BASELINE.json): ViT-B/16, 224x224x3 synthetic images, 1000 classes, batch 256 PER GPU
(weak scaling; global batch 2048 on 8 GPUs), fp32 end to end, dropout 0, Adam(L2) lr 3e-4 /
wd 1e-4 under the reference's cosine/warm-up schedule.  One step = forward + softmax-CE +
backward (+ bucketed RCCL gradient all-reduce overlapped with backward when N > 1) + Adam, with
the batch already resident in HBM.  Weights are random-init N(0, 0.02) (same on every rank).

Prints ONE JSON line on rank 0: images/s over all GPUs (with N > 1 or --force-collectives also every
rank's own step time, the number of ranks the communicator holds and the part of the gradient
exchange that backward does not hide), plus
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
import zlib

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


def cpu_baseline(cfgd, batch, seconds_budget=30.0):
    """Time the reference CPU path on the host: whole training steps of the same model -- at the
    workload's own batch when one such step fits the time budget (the 28 x 28 models), else at batch 2."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import vitlibs as V
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    cfg = V.Cfg.make(**cfgd)
    R = V.ref()

    def session(B):
        rng = np.random.default_rng(1234)
        x = rng.standard_normal((B, cfg.in_channels, cfg.image_size, cfg.image_size)).astype(np.float32)
        y = V.onehot(np.random.default_rng(4321).integers(0, cfg.num_classes, B), cfg.num_classes)
        logits = V.f32(B, cfg.num_classes)
        if R is not None:
            s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)  # the reference's own random init
            return lambda: R.vref_train_step(s, x, y, B, 3e-4, logits, None)
        O = V.oracle()
        s = O.vo_session_create(cfg, 0.9, 0.999, 1e-8, 1e-4)
        O.vo_session_set_params(s, V.init_params(cfg, 42))
        return lambda: O.vo_session_train_step(s, x, y, B, 3e-4, logits, None)

    kind = "reference" if R is not None else "port"
    t_begin = time.perf_counter()
    B = 2
    step = session(B)
    t0 = time.perf_counter()
    step()
    first = time.perf_counter() - t0
    steps, el = 1, first
    if batch > B and first * batch / B <= 0.4 * seconds_budget:  # a full-batch step fits several times over
        B = batch
        step = session(B)
        steps, el = 0, 0.0
    t0 = time.perf_counter()
    while steps < 5:
        if steps and (time.perf_counter() - t_begin) + el / steps > seconds_budget:
            break
        step()
        steps += 1
        el = time.perf_counter() - t0 + (first if B == 2 else 0.0)
    return {"value": round(B * steps / el, 4), "unit": "images/s", "cores": cores, "kind": kind,
            "sample": f"{steps} training step(s) of the same model at batch {B}"
                      f"{' (the workload batch)' if B == batch else ''} "
                      f"(forward + CE + backward + Adam), {el:.1f} s of host time, OpenMP threads = {cores}"}


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC profile of this same command
    (rocprofv3 cannot collect counters from inside a timed run); None if there is no profile."""
    try:
        for name in TRAFFIC_PROFILES:
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                with open(path) as f:
                    return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"], name
    except (OSError, KeyError, ValueError):
        pass
    return None, None


TRAFFIC_PROFILES = ("r03_hbm_traffic.json", "r02_hbm_traffic.json")


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args, argv):
    """Parent of a self-started run: one child per GPU with the variables a launcher would set.
    Touches no GPU (imports neither torch nor the libraries).  Returns the exit status."""
    import subprocess
    import uuid
    world = args.gpus
    base = dict(os.environ, WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                VIT_COMM_RUN_ID=uuid.uuid4().hex)
    base.setdefault("GLOO_SOCKET_IFNAME", "lo")       # container host names may not resolve
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this driver
    children = []
    for r in range(world):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the result line; with --dry-launch every child reports
        out = subprocess.PIPE if (r == 0 or args.dry_launch) else sys.stderr
        children.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + argv, env=env,
                                         stdout=out, cwd=os.getcwd()))

    def stop_all():
        for c in children:
            if c.poll() is None:
                c.terminate()
        t_end = time.monotonic() + 10
        for c in children:
            try:
                c.wait(max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                c.kill()
                c.wait()

    deadline = time.monotonic() + args.launch_timeout
    status = 0
    try:
        while any(c.poll() is None for c in children):
            failed = [(r, c.returncode) for r, c in enumerate(children) if c.poll() not in (None, 0)]
            if failed:
                print(f"bench.py: rank {failed[0][0]} exited with status {failed[0][1]}; stopping the others",
                      file=sys.stderr)
                status = failed[0][1] if failed[0][1] > 0 else 1
                break
            if time.monotonic() > deadline:
                print(f"bench.py: the ranks did not finish within {args.launch_timeout} s; stopping them", file=sys.stderr)
                status = 124
                break
            time.sleep(0.1)
    finally:
        stop_all()
    if status == 0:
        bad = [(r, c.returncode) for r, c in enumerate(children) if c.returncode != 0]
        if bad:
            print(f"bench.py: rank {bad[0][0]} exited with status {bad[0][1]}", file=sys.stderr)
            status = bad[0][1] if bad[0][1] > 0 else 1
    if status == 0:
        if args.dry_launch:
            # (libraries may print to the children's stdout as well: each child's report is its last JSON line)
            reports = [[l for l in c.stdout.read().decode().splitlines() if l.startswith("{")][-1] for c in children]
            print(json.dumps({"dry_launch": [json.loads(r) for r in reports]}), flush=True)
        else:
            sys.stdout.write(children[0].stdout.read().decode())
            sys.stdout.flush()
    return status


def init_group(dist, rank, world):
    """The CPU (gloo) process group that carries the communicator id, the barriers and the timings.
    Rendezvous through the launcher's variables (env://): under torch.distributed.run the ranks are
    clients of the store its agent already serves on MASTER_PORT; started by launch() above, rank 0
    serves it."""
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))


def parse_args(argv):
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
    ap.add_argument("--through-trainer", action="store_true",
                    help="time the reference's own loop instead (Trainer::train on a data set of 8 batches resident in "
                         "HBM: per-epoch shuffle, device gather, per-step loss read-back for the progress line); "
                         "its own metric name, one GPU")
    ap.add_argument("--dropout", type=float, default=0.0,
                    help="dropout rate (0 = the headline/parity configuration; 0.1 adds the device RNG cost)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="seconds after which a self-started multi-rank run is stopped")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses GPU 0 and the gradient exchange goes through host "
                         "memory and the gloo group (RCCL refuses two ranks on one device); numbers are not throughput")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, have each report the variables it was given, and stop (no GPU needed)")
    return ap.parse_args(argv)


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_collectives or args.dry_launch):
        sys.exit(launch(args, argv))
    if args.dry_launch:
        # what this rank was given, and -- through the same rendezvous the benchmark uses -- who else is there
        me = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                             "VIT_COMM_RUN_ID")}
        import torch.distributed as dist
        init_group(dist, int(me["RANK"]), int(me["WORLD_SIZE"]))
        seen = [None] * int(me["WORLD_SIZE"])
        dist.all_gather_object(seen, int(me["RANK"]))
        dist.barrier()
        dist.destroy_process_group()
        me["group_ranks"] = seen
        os.write(1, (json.dumps(me) + "\n").encode())
        return
    # stdout carries exactly one line, the result: everything libraries print on the way (RCCL's
    # version banner, for one) goes to stderr, and the JSON line is written to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    collectives = world > 1 or args.force_collectives
    if args.through_trainer and world > 1:
        raise SystemExit("--through-trainer measures one GPU")
    if collectives:
        # a CPU (gloo) group carries the 128-byte communicator id, the barriers and the timings, so the
        # only RCCL communicator of the process is the host layer's own
        init_group(dist, rank, world)
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
    if collectives and args.share_gpu:
        transport_errors = dp.init_host_staged()
    elif collectives:
        dp.init_rccl()
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
        """This rank's device idle, then every rank's; returns the time this rank got idle."""
        torch.cuda.synchronize()
        t_idle = time.perf_counter()
        if world > 1:
            dist.barrier()
        return t_idle

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def every_rank(v):
        t = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        if world > 1:
            dist.all_gather(t, torch.tensor([v], dtype=torch.float64))
        else:
            t[0][0] = v
        return [float(x.item()) for x in t]

    def finish():
        if dist.is_initialized():
            dist.barrier()
            host.comm_destroy()
            dist.destroy_process_group()

    if args.eval:
        # forward only, one rank's batch per step, nothing exchanged between ranks
        for s in range(args.warmup):
            pred = model.predict(x)
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            pred = model.predict(x)
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
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
        finish()
        return

    if args.through_trainer:
        # The reference's own loop (app/main.cpp:60 -> src/model/Trainer.cpp:68-128, 163-236): a data set of
        # 8 batches handed over in host memory, uploaded once, shuffled per epoch, gathered per step on the
        # device, loss and accuracy read back every step.  Epoch 0 is the warm-up; the rest is timed by the
        # Trainer itself (train_epoch until the device is idle; the evaluation pass is not part of it).
        n = 8 * batch
        epochs = 1 + max(1, -(-args.steps // 8))
        xs = torch.randn(n, cfg.in_channels, cfg.image_size, cfg.image_size, generator=torch.Generator().manual_seed(1234)).numpy()
        ys = torch.nn.functional.one_hot(torch.randint(0, cfg.num_classes, (n,), generator=torch.Generator().manual_seed(4321)),
                                         cfg.num_classes).to(torch.float32).numpy()
        import tempfile
        with tempfile.TemporaryDirectory() as scratch:  # the Trainer writes its vit_results.csv where it runs
            here = os.getcwd()
            os.chdir(scratch)
            try:
                model.fit(xs, ys, xs[:8], ys[:8], epochs, batch, lr_init=3e-4, weight_decay=1e-4, warmup_frac=0.1)
            finally:
                os.chdir(here)
        ms = host.last_fit_epoch_ms()[1:]
        steps = 8 * len(ms)
        value = batch * steps / (sum(ms) * 1e-3)
        f_img = train_flops_per_image(cfgd)
        out = {"metric": "train images/sec through Trainer::train (the reference's epoch loop)", "value": round(value, 2),
               "unit": "images/s", "n_gpus": 1, "steps": steps, "warmup": 8, "ms_per_step": round(sum(ms) / steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": args.workload, "batch_per_gpu": batch, "data_set": n, "epochs_timed": len(ms),
                          "epoch_ms": [round(m, 2) for m in ms], "patch_mode": "im2col", "augmentation": "off",
                          "step_frac_of_peak": round(value * f_img / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                          "step_frac_of_peak_executed": round(value * executed_train_flops_per_image(cfgd) / 1e12 /
                                                              PEAK_FP32_MFMA_TFLOPS, 4)}}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
        finish()
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
    timing = collectives and not args.share_gpu  # (the exposed-exchange events need the stream-ordered transport)
    if timing:
        host.comm_timing(True)
    t0 = time.perf_counter()
    for s in range(args.steps):
        ddp.step(x, y, lr(args.warmup + s))
    enqueued = time.perf_counter() - t0  # the host has issued every launch; the GPU may still be working
    own = barrier() - t0
    elapsed = time.perf_counter() - t0
    exposed_ms, exposed_n = host.comm_exposed_ms() if timing else (0.0, 0)
    if timing:
        host.comm_timing(False)
    if collectives and args.share_gpu and transport_errors:
        raise SystemExit(f"gradient exchange failed: {transport_errors[0]!r}")
    loss, _ = trainer.result()
    if not events_in_region:
        vitk.gemm_profile(True)
        for s in range(args.steps):
            ddp.step(x, y, lr(args.warmup + s))
        barrier()
    vitk.gemm_profile(False)
    prof = vitk.gemm_profile_read()

    elapsed = max_over_ranks(elapsed)
    # (after the timed region) every rank trained on its own images: the parameters only stay the
    # same on all of them if every gradient went through the exchange
    params_crc = [int(v) for v in every_rank(zlib.crc32(model.get_params().tobytes()))] if collectives else []
    per_rank_ms = [round(1e3 * v / args.steps, 3) for v in every_rank(own)]
    exposed_per_rank = [round(v, 3) for v in every_rank(exposed_ms / max(exposed_n, 1))]
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
                                                      if collectives else ""),
                       "patch_mode": "im2col", "params": int(model.n), "final_loss": round(float(loss), 6),
                       "last_block": "all tokens (as the reference)" if args.dense_tail else
                                     "CLS token only after its keys/values (the head reads nothing else; same logits, "
                                     "loss, gradients and updates)"},
            "roofline": {"bound": "mfma", "kernel": dom["name"], "achieved": round(achieved, 2),
                         "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4),
                         "traffic": measured_traffic(dom["name"])[0],
                         "traffic_note": "HBM bytes per launch, rocprofv3 PMC passes of this command "
                                         f"(profiles/{measured_traffic(dom['name'])[1]})",
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
        if collectives:
            # every rank's own time from the first launch of the timed steps until its device was idle;
            # the ranks the host layer's communicator holds; HIP-event time per step from the last
            # backward kernel to the end of the last bucket's all-reduce (what backward did not hide)
            out["per_rank_ms_per_step"] = per_rank_ms
            out["n_ranks_seen"] = host.comm_world()
            out["exposed_comm_ms_per_step"] = max(exposed_per_rank)
            out["exposed_comm_ms_per_step_per_rank"] = exposed_per_rank
            out["ranks_hold_identical_parameters"] = len(set(params_crc)) == 1
            out["config"]["bucket_mb"] = args.bucket_mb
            out["config"]["communicators"] = ("rehearsal: ranks share GPU 0, gradients exchanged through host memory over gloo"
                                              if args.share_gpu else "one RCCL communicator (the host layer's); id, "
                                              "barriers and timings over a CPU gloo group")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfgd, batch)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    finish()


if __name__ == "__main__":
    main()
