"""Data parallelism behind the C boundary (include/vit_host.h: vith_comm_*; core/Comm.hpp) on a GPU.

* one rank through the real RCCL calls, side stream and events: the step and the whole Trainer loop
  must equal the plain ones bit for bit (a one-rank all-reduce is the identity);
* two ranks SHARING the one GPU of the test box: RCCL refuses two ranks on one device, so the ranks
  exchange through host memory and a CPU (gloo) process group -- a transport plugged in through
  vith_comm_init_external -- while sharding, global-batch loss scaling, bucket order, the zero
  contribution of an empty shard and the update order are the C++ layer's.  Expected: both ranks
  end bit-identical and equal to one process training on the whole batch.
"""
import datetime
import os
import socket
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import vitlibs as V  # noqa: E402

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CFGD = dict(V.TINY, num_layers=3)
B_RANK, STEPS = 6, 3
FIT = dict(n=70, nt=32, epochs=2, batch=16, lr=1e-3, wd=1e-4, wf=0.1)  # nt: whole evaluation batches at both batch sizes (evaluate averages per-batch means)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def step_inputs():
    ocfg = V.Cfg.make(**CFGD)
    K = ocfg.num_classes
    xs = [V.rng_normal(10 + s, 2 * B_RANK, 1, 28, 28) for s in range(STEPS)]
    ys = [V.onehot(np.random.default_rng(20 + s).integers(0, K, 2 * B_RANK), K) for s in range(STEPS)]
    return ocfg, V.init_params(ocfg, 5), xs, ys


def fit_inputs():
    ocfg = V.Cfg.make(**CFGD)
    K, f = ocfg.num_classes, FIT
    x = V.rng_normal(41, f["n"], 1, 28, 28)
    y = V.onehot(np.random.default_rng(42).integers(0, K, f["n"]), K)
    xt = V.rng_normal(43, f["nt"], 1, 28, 28)
    yt = V.onehot(np.random.default_rng(44).integers(0, K, f["nt"]), K)
    return x, y, xt, yt


def run_steps(H, ddp_factory, prm, xs, ys, lo, hi):
    m = H.Model(H.Config.make(**CFGD))
    m.set_params(prm)
    t = H.Trainer(m, weight_decay=1e-4)
    drv = ddp_factory(m, t)
    for s in range(STEPS):
        x, y, lr = dev(xs[s][lo:hi]), dev(ys[s][lo:hi]), 1e-3 * (s + 1)
        if drv is None:
            t.train_step(x, y, lr, want_result=False)
        else:
            drv.step(x, y, lr)
    return m, t, m.get_params()


@pytest.fixture()
def H():
    from visiontransformer_amd import host
    host.init(0)
    host.use_torch_stream()
    return host


def test_one_rank_through_rccl_equals_the_plain_step_bit_for_bit(H, tmp_path, monkeypatch):
    from visiontransformer_amd import dp
    monkeypatch.chdir(tmp_path)
    ocfg, prm, xs, ys = step_inputs()
    _, _, want = run_steps(H, lambda m, t: None, prm, xs, ys, 0, 2 * B_RANK)
    x, y, xt, yt = fit_inputs()
    f = FIT
    m0 = H.Model(H.Config.make(**CFGD))
    m0.set_params(prm)
    hist0 = m0.fit(x, y, xt, yt, f["epochs"], f["batch"], f["lr"], f["wd"], f["wf"])
    H.comm_init(0, 1, H.comm_unique_id())
    try:
        assert (H.comm_rank(), H.comm_world()) == (0, 1)
        m, t, got = run_steps(H, lambda m, t: dp.NativeDataParallel(m, t, bucket_bytes=64 << 10), prm, xs, ys, 0,
                              2 * B_RANK)
        assert t.num_buckets() >= 2, t.num_buckets()  # several collectives in flight on the side stream
        np.testing.assert_array_equal(got, want)
        # the C++ Trainer loop under a communicator (sharded batches, staged backward, reducer)
        m1 = H.Model(H.Config.make(**CFGD))
        m1.set_params(prm)
        hist1 = m1.fit(x, y, xt, yt, f["epochs"], f["batch"], f["lr"], f["wd"], f["wf"])
        np.testing.assert_array_equal(m1.get_params(), m0.get_params())
        np.testing.assert_allclose(hist1, hist0, rtol=2e-6)
    finally:
        H.comm_destroy()
    assert H.comm_world() == 1


def _rank_main(rank, world, port, outdir):
    import torch.distributed as dist
    from visiontransformer_amd import dp, host
    os.chdir(outdir)
    torch.cuda.set_device(0)  # both ranks on the one GPU
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=60))
    host.init(0)
    host.use_torch_stream()
    errors = dp.init_host_staged()
    ocfg, prm, xs, ys = step_inputs()
    # each rank starts from its own (different) parameters: the broadcast must make them rank 0's
    mine = prm if rank == 0 else V.init_params(ocfg, 99)

    def make(m, t):
        d = dp.NativeDataParallel(m, t, bucket_bytes=64 << 10)
        d.broadcast_parameters(0)
        return d

    m, t, got = run_steps(host, make, mine, xs, ys, rank * B_RANK, (rank + 1) * B_RANK)
    assert t.num_buckets() >= 2
    np.save(os.path.join(outdir, f"steps_r{rank}.npy"), got)
    # a training state written by rank 0 and loaded by every rank: each keeps a dropout stream of its own
    state = os.path.join(outdir, "dp_state.bin")
    if rank == 0:
        t.save_state(state)
    dist.barrier()
    t.load_state(state)
    np.save(os.path.join(outdir, f"drop_r{rank}.npy"), np.array(host.dropout_state(), dtype=np.uint64))
    # Trainer::train, every rank holding the whole data set and taking its slice of each batch
    x, y, xt, yt = fit_inputs()
    f = FIT
    m2 = host.Model(host.Config.make(**CFGD))
    m2.set_params(mine)
    os.makedirs(f"r{rank}", exist_ok=True)
    os.chdir(f"r{rank}")
    hist = m2.fit(x, y, xt, yt, f["epochs"], f["batch"] // 2, f["lr"], f["wd"], f["wf"])
    np.save(os.path.join(outdir, f"fit_r{rank}.npy"), m2.get_params())
    np.save(os.path.join(outdir, f"hist_r{rank}.npy"), hist)
    assert os.path.exists("vit_results.csv") == (rank == 0)  # rank 0 alone writes the log
    assert not errors, errors
    host.comm_destroy()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_on_one_gpu_match_one_rank_on_the_whole_batch(H, tmp_path, monkeypatch):
    import torch.multiprocessing as mp
    monkeypatch.chdir(tmp_path)
    ctx = mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=False)
    deadline = time.time() + 100
    try:
        while not ctx.join(timeout=2):
            assert time.time() < deadline, "data-parallel ranks did not finish in 100 s"
    finally:
        for p in ctx.processes:
            if p.is_alive():
                p.kill()
    ocfg, prm, xs, ys = step_inputs()
    shapes = V.param_shapes(ocfg)
    a, b = np.load(tmp_path / "steps_r0.npy"), np.load(tmp_path / "steps_r1.npy")
    np.testing.assert_array_equal(a, b)  # replicas stay bit-identical
    _, _, want = run_steps(H, lambda m, t: None, prm, xs, ys, 0, 2 * B_RANK)

    def same_training(got, ref, lr_sum, what):
        off = 0
        for i, shp in enumerate(shapes):
            k = int(np.prod(shp))
            # reduction order differs (sum of two half-batch gradients vs one pass over the batch);
            # see test_dropout_device_stream_masks_against_the_oracle for why the bound is Adam's
            assert np.abs(got[off:off + k] - ref[off:off + k]).max() <= 0.1 * lr_sum, (what, i, shp)
            assert V.relerr(got[off:off + k], ref[off:off + k]) <= 2e-5, (what, i, shp)
            off += k

    same_training(a, want, 6e-3, "steps")
    d0, d1 = np.load(tmp_path / "drop_r0.npy"), np.load(tmp_path / "drop_r1.npy")
    assert d0[0] != d1[0] and d0[1] == d1[1], (d0, d1)  # resumed ranks still draw different masks, same position
    fa, fb = np.load(tmp_path / "fit_r0.npy"), np.load(tmp_path / "fit_r1.npy")
    np.testing.assert_array_equal(fa, fb)
    np.testing.assert_array_equal(np.load(tmp_path / "hist_r0.npy"), np.load(tmp_path / "hist_r1.npy"))
    x, y, xt, yt = fit_inputs()
    f = FIT
    m = H.Model(H.Config.make(**CFGD))
    m.set_params(prm)
    hist = m.fit(x, y, xt, yt, f["epochs"], f["batch"], f["lr"], f["wd"], f["wf"])  # one process, batch 16
    same_training(fa, m.get_params(), 6 * f["lr"], "fit")
    got_hist = np.load(tmp_path / "hist_r0.npy")
    np.testing.assert_allclose(got_hist[:, [0, 2]], hist[:, [0, 2]], rtol=1e-4)
    np.testing.assert_allclose(got_hist[:, [1, 3]], hist[:, [1, 3]], atol=1e-6)


def test_bench_starts_its_ranks_itself_and_reports_the_exchange(tmp_path):
    """`python bench.py --gpus 1 --force-collectives` with no launcher around it: the parent starts the
    rank as a child before touching the GPU; the child joins the host layer's RCCL communicator
    (its only one; a CPU gloo group carries the id) and issues every bucket's all-reduce."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-collectives",
                        "--workload", "blood_p7", "--batch", "64", "--steps", "4", "--warmup", "2", "--bucket-mb", "0.25",
                        "--no-cpu-baseline", "--launch-timeout", "100"],
                       env=env, capture_output=True, text=True, timeout=110, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1
    assert len(out["per_rank_ms_per_step"]) == 1 and 0 < out["per_rank_ms_per_step"][0] <= out["ms_per_step"] * 1.001
    assert out["exposed_comm_ms_per_step"] >= 0 and out["exposed_comm_ms_per_step"] < out["ms_per_step"]
    assert "gradient buckets" in out["config"]["parallelism"] and out["ranks_hold_identical_parameters"] is True
    assert np.isfinite(out["config"]["final_loss"]) and out["value"] > 0


def test_bench_with_two_ranks_end_to_end(tmp_path):
    """The N > 1 path of bench.py itself on the one GPU of the test box: the parent starts two ranks,
    they meet over gloo, shard nothing but exchange every gradient bucket (through host memory: RCCL
    refuses two ranks on one device), and rank 0's line carries both ranks' timings."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--share-gpu", "--workload", "mnist_p14",
                        "--batch", "32", "--steps", "4", "--warmup", "2", "--bucket-mb", "0.25", "--launch-timeout", "100"],
                       env=env, capture_output=True, text=True, timeout=110, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["config"]["global_batch"] == 64
    assert len(out["per_rank_ms_per_step"]) == 2 and all(0 < t <= out["ms_per_step"] * 1.001 for t in out["per_rank_ms_per_step"])
    assert "cpu_baseline" not in out and np.isfinite(out["config"]["final_loss"])
    # the two ranks trained on different images: equal parameters mean every gradient was exchanged
    assert out["ranks_hold_identical_parameters"] is True
