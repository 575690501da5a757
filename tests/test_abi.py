"""CPU-side checks of the C-ABI: the libraries load and export every symbol the headers declare,
the ctypes signature tables cover them, and argument validation fails with a message (no GPU use)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_vitk_exports_every_declared_symbol():
    from visiontransformer_amd import vitk
    names = declared("vitk.h", "vitk_")
    assert len(names) >= 25
    d = vitk.dll()
    for n in names:
        assert hasattr(d, n), f"libvitk.so does not export {n}"
        assert n in vitk.SIGNATURES, f"no ctypes signature for {n}"
    assert sorted(vitk.SIGNATURES) == names
    assert d.vitk_version() >= 100


def test_vitk_argument_validation_reports_errors():
    from visiontransformer_amd import vitk
    d = vitk.dll()
    a = vitk.GemmArgs()
    a.M, a.N, a.K = 4, 4, 4  # null pointers
    assert d.vitk_gemm_f32(C.byref(a), None) != 0
    assert b"null operand" in d.vitk_last_error()
    with pytest.raises(vitk.VitkError):
        vitk.call("vitk_layernorm_fwd", None, 4, 8, 4, None, None, 1e-5, None, None, None, None)  # stride < D
    with pytest.raises(vitk.VitkError):
        vitk.call("vitk_dropout_fwd", None, 4, 1.5, 0, 0, None, None, None)
    assert C.sizeof(vitk.GemmArgs) == 208


def test_vit_host_exports_every_declared_symbol():
    from visiontransformer_amd import host
    names = declared("vit_host.h", "vith_")
    assert len(names) >= 40
    d = host.dll()
    for n in names:
        assert hasattr(d, n), f"libvit_host.so does not export {n}"
        assert n in host.SIGNATURES, f"no ctypes signature for {n}"
    assert sorted(host.SIGNATURES) == names
    # pure-host entry points work without a GPU
    assert host.cosine_warmup_lr(0, 10, 100, 3e-4) == 0.0
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vitlibs as V
    g = V.golden("ops.npz")
    import numpy as np
    sched = np.array([host.cosine_warmup_lr(s, 10, 100, 3e-4) for s in range(100)], dtype=np.float32)
    np.testing.assert_array_equal(sched, g["sched"])  # bit-identical learning rates
    for ep in (0, 1, 7):
        np.testing.assert_array_equal(host.epoch_order(50, ep), g[f"order_{ep}"])  # Trainer's per-epoch order


def test_config_json_round_trip_matches_reference_file(tmp_path):
    from visiontransformer_amd import host
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vitlibs as V
    ref_json = os.path.join(V.GOLDEN, "vit_mnist_4_8.json")
    cfg = host.load_config(ref_json)
    assert cfg.as_dict() == dict(V.CONFIGS["CK"], dropout_rate=0.0)
    out = tmp_path / "c.json"
    host.save_config(cfg, str(out))
    # same content as the shipped file (which an older writer indented by 2) ...
    assert out.read_text().split() == open(ref_json).read().split()
    # ... and the same BYTES as the reference's current writer (4-space indent), where it is built
    R = V.ref()
    if R is not None:
        rcfg = V.Cfg.make(**V.CONFIGS["CK"])
        ref_out = tmp_path / "r.json"
        assert R.vref_save_config(rcfg, str(ref_out).encode()) == 0
        assert out.read_bytes() == ref_out.read_bytes()
        rcfg = V.Cfg.make(**V.CONFIGS["C1"], dropout_rate=0.2)
        assert R.vref_save_config(rcfg, str(ref_out).encode()) == 0
        host.save_config(host.Config.make(**V.CONFIGS["C1"], dropout_rate=0.2), str(out))
        assert out.read_bytes() == ref_out.read_bytes()
    cfg2 = host.Config.make(**V.CONFIGS["C1"], dropout_rate=0.2)
    host.save_config(cfg2, str(out))
    back = host.load_config(str(out))
    assert back.as_dict() == cfg2.as_dict()
    with pytest.raises(host.HostError):
        host.load_config(str(tmp_path / "missing.json"))


def test_bucket_plan_of_the_host_layer_matches_the_python_driver():
    """The gradient-bucket rule exists twice (vit::GradientReducer in C++, dp.plan_buckets for the
    generic Python driver): same plan for ViT-B-like stage sizes at several bucket sizes."""
    from visiontransformer_amd import dp, host
    emb, blk, tail, L = 742_656, 7_087_872, 770_536, 12
    offs = [0, emb] + [emb + blk * (i + 1) for i in range(L)] + [emb + blk * L + tail]
    ranges = [(offs[-2], tail)] + [(offs[i], blk) for i in range(L, 0, -1)] + [(0, emb)]
    for bb in (1, 1 << 20, 25 << 20, 50 << 20, 1 << 40):
        want = dp.plan_buckets(ranges, bb)
        assert host.plan_buckets(ranges, bb) == [tuple(b) for b in want], bb
    assert len(host.plan_buckets(ranges, 25 << 20)) == 13  # head+last block, 11 blocks, embeddings
    with pytest.raises(host.HostError):
        host.plan_buckets([(0, 4), (100, 4)], 1 << 20)  # slices of one bucket must be adjacent
