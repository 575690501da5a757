"""bench.py as its own launcher, and the file rendezvous of the host layer's communicator -- the
two pieces that start a multi-GPU run without an external launcher.  Host code only: no GPU."""
import json
import os
import subprocess
import sys
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visiontransformer_amd import host  # noqa: E402


def run_bench(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                          text=True, timeout=120)


@pytest.mark.parametrize("world", [1, 2, 4])
def test_bench_starts_its_own_ranks(world):
    r = run_bench("--gpus", str(world), "--dry-launch")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1  # stdout carries one line
    kids = json.loads(lines[0])["dry_launch"]
    assert [k["RANK"] for k in kids] == [str(i) for i in range(world)]
    assert [k["LOCAL_RANK"] for k in kids] == [str(i) for i in range(world)]
    assert {k["WORLD_SIZE"] for k in kids} == {str(world)}
    assert {k["MASTER_ADDR"] for k in kids} == {"127.0.0.1"}
    assert len({k["MASTER_PORT"] for k in kids}) == 1 and int(kids[0]["MASTER_PORT"]) > 0
    assert len({k["VIT_COMM_RUN_ID"] for k in kids}) == 1 and len(kids[0]["VIT_COMM_RUN_ID"]) == 32
    # the ranks met in the process group the benchmark uses (gloo over the launcher's variables)
    assert all(k["group_ranks"] == list(range(world)) for k in kids)


def test_bench_under_torch_distributed_run_is_one_rank_per_process():
    """The driver's command for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`.  With the launcher's variables
    present bench.py starts nothing itself, and its ranks meet through the store the launcher's agent
    serves on MASTER_PORT (a second server on that port would fail to bind)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--dry-launch"], env=e, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-3000:]
    kids = sorted((json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")), key=lambda k: int(k["RANK"]))
    assert [k["RANK"] for k in kids] == ["0", "1"] and {k["WORLD_SIZE"] for k in kids} == {"2"}
    assert {k["MASTER_PORT"] for k in kids} == {str(port)}
    assert all(k["group_ranks"] == [0, 1] for k in kids)


def test_a_failing_rank_fails_the_launch():
    """Without a GPU every real rank exits non-zero at once: the parent must report that, not hang."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (the ranks would run the benchmark)")
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-timeout", "100")
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "exited with status" in r.stderr


# ------------------------------------------------------------------ id exchange through files
def exchange(world, path, ident, results, errors, timeout_s=20):
    def one(rank):
        try:
            results[rank] = host.comm_exchange_id_file(rank, world, path, ident if rank == 0 else None, timeout_s)
        except Exception as e:  # noqa: BLE001
            errors[rank] = e
    threads = [threading.Thread(target=one, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()


@pytest.mark.parametrize("world", [2, 4])
def test_id_exchange_reaches_every_rank_and_cleans_up(tmp_path, world):
    path = str(tmp_path / "comm.id")
    ident = bytes(range(128))
    results, errors = {}, {}
    exchange(world, path, ident, results, errors)
    assert not errors, errors
    assert all(results[r] == ident for r in range(world))
    assert os.listdir(tmp_path) == []  # rank 0 removes every file of the exchange


def test_id_exchange_ignores_what_an_earlier_launch_left_behind(tmp_path):
    """Stale files of every kind, freshly dated, at the same path: a published id of a launch that
    died while joining, an announcement and a confirmation of its rank 1."""
    path = str(tmp_path / "comm.id")
    stale_id = b"\xee" * 128
    world = 2
    pub = b"VITCOMM2" + (world).to_bytes(8, "little") + stale_id + (0).to_bytes(8, "little") + (77).to_bytes(8, "little")
    for name, data in ((path, pub), (path + ".req1", (77).to_bytes(8, "little")), (path + ".ack1", (77).to_bytes(8, "little") + (5).to_bytes(8, "little"))):
        with open(name, "wb") as f:
            f.write(data)
        os.chmod(name, 0o600)
    ident = bytes(reversed(range(128)))
    results, errors = {}, {}
    exchange(world, path, ident, results, errors)
    assert not errors, errors
    assert results[0] == ident and results[1] == ident


def test_id_exchange_refuses_symbolic_links(tmp_path):
    """A link planted at the announcement's temporary or final name is never written through."""
    path = str(tmp_path / "comm.id")
    victim = tmp_path / "victim"
    victim.write_bytes(b"untouched")
    os.symlink(victim, path)  # rank 0 unlinks the name and creates its own file beside it
    results, errors = {}, {}
    exchange(2, path, b"\x01" * 128, results, errors)
    assert not errors, errors
    assert victim.read_bytes() == b"untouched"


def test_id_exchange_times_out_alone(tmp_path):
    with pytest.raises(host.HostError, match="timed out"):
        host.comm_exchange_id_file(1, 2, str(tmp_path / "nobody.id"), None, timeout_s=1)
