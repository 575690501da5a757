"""Pins oracle/libvit_oracle.so against the committed golden vectors (tests/golden/*.npz), which
were produced by the compiled reference (tests/golden/make_golden.py).  Needs neither a GPU nor
/root/reference.  Tolerance 5e-6 relative to each tensor's scale (the oracle keeps the
reference's accumulation order; the slack covers libm/OpenMP-order differences only)."""
import os

import numpy as np
import pytest

import vitlibs as V

O = V.oracle()
TOL = 5e-6


def close(a, b, tol=TOL):
    assert a.shape == b.shape
    e = V.max_relerr(a, b)
    assert e <= tol, e


def test_ops_golden():
    g = V.golden("ops.npz")
    c = V.op_case("dense")
    y, dW, db, dx = V.f32(c["rows"], c["o"]), V.f32(c["i"], c["o"]), V.f32(1, c["o"]), V.f32(c["rows"], c["i"])
    O.vo_dense_forward(c["x"], c["rows"], c["i"], c["o"], c["W"], c["b"], y)
    O.vo_dense_backward(c["x"], c["dy"], c["rows"], c["i"], c["o"], c["W"], dW, db, dx)
    close(y, g["dense_y"]); close(dW, g["dense_dW"]); close(db, g["dense_db"]); close(dx, g["dense_dx"])
    c = V.op_case("ln")
    rows, D = c["rows"], c["D"]
    y, xh, inv, dg, dbe, dx = V.f32(rows, D), V.f32(rows, D), V.f32(rows), V.f32(1, D), V.f32(1, D), V.f32(rows, D)
    O.vo_layernorm_forward(c["x"], rows, D, c["g"], c["b"], 1e-5, y, xh, inv)
    O.vo_layernorm_backward(c["dy"], xh, inv, c["g"], rows, D, dg, dbe, dx)
    close(y, g["ln_y"]); close(dg, g["ln_dgamma"]); close(dbe, g["ln_dbeta"]); close(dx, g["ln_dx"])
    c = V.op_case("gelu")
    y, dx = V.f32(c["n"]), V.f32(c["n"])
    O.vo_gelu_forward(c["x"], c["n"], y)
    O.vo_gelu_backward(c["x"], c["dy"], c["n"], dx)
    close(y, g["gelu_y"]); close(dx, g["gelu_dx"])
    c = V.op_case("softmax")
    p, d = V.f32(6, 17, 17), V.f32(6, 17, 17)
    O.vo_softmax_rows(c["x"], 6 * 17, 17, p)
    O.vo_softmax_rows_backward(c["g"], p, 6 * 17, 17, d)
    close(p, g["softmax_p"]); close(d, g["softmax_dx"])
    for tag in ("mha_a", "mha_b"):
        c = V.op_case(tag)
        y, g_, dx = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"]), V.f32(c["B"], c["N"], c["D"])
        O.vo_mha(c["D"], c["h"], c["prm"], c["x"], c["B"], c["N"], c["dy"], y, g_, dx)
        close(y, g[f"{tag}_y"]); close(g_, g[f"{tag}_g"]); close(dx, g[f"{tag}_dx"])
    c = V.op_case("ffn")
    y, g_, dx = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"]), V.f32(c["B"], c["N"], c["D"])
    O.vo_ffn(c["D"], c["H"], c["prm"], c["x"], c["B"], c["N"], c["dy"], y, g_, dx)
    close(y, g["ffn_y"]); close(g_, g["ffn_g"]); close(dx, g["ffn_dx"])
    c = V.op_case("block")
    y, g_, dx = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"]), V.f32(c["B"], c["N"], c["D"])
    O.vo_block(c["D"], c["h"], c["H"], c["prm"], c["x"], c["B"], c["N"], c["dy"], y, g_, dx)
    close(y, g["block_y"]); close(g_, g["block_g"]); close(dx, g["block_dx"])
    for tag in ("emb_a", "emb_b", "emb_c"):
        c = V.op_case(tag)
        y, g_ = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"])
        O.vo_embeddings(c["img"], c["P"], c["C"], c["D"], c["prm"], c["x"], c["B"], c["dy"], y, g_)
        close(y, g[f"{tag}_y"]); close(g_, g[f"{tag}_g"])
    c = V.op_case("ce")
    g_ = V.f32(c["B"], c["K"])
    lw = O.vo_cross_entropy(c["lg"], c["y"], c["B"], c["K"], c["w"], c["K"], g_)
    assert abs(lw - g["ce_loss_w"]) <= TOL * abs(g["ce_loss_w"])
    close(g_, g["ce_grad"])
    l0 = O.vo_cross_entropy(c["lg"], c["y"], c["B"], c["K"], None, 0, g_)
    assert abs(l0 - g["ce_loss"]) <= TOL * abs(g["ce_loss"])
    c = V.op_case("adam")
    p, m, v = c["p"].copy(), V.f32(c["r"], c["c"]), V.f32(c["r"], c["c"])
    for s in range(c["steps"]):
        b1t = np.float32(np.float64(np.float32(0.9)) ** (s + 1))
        b2t = np.float32(np.float64(np.float32(0.999)) ** (s + 1))
        O.vo_adam_update(p, c["grads"][s], m, v, c["r"] * c["c"], float(c["lrs"][s]), 0.9, 0.999, float(b1t),
                         float(b2t), 1e-8, 1e-4)
    close(p, g["adam_p"])
    sched = np.array([O.vo_cosine_warmup_lr(s, 10, 100, 3e-4) for s in range(100)], dtype=np.float32)
    np.testing.assert_array_equal(sched, g["sched"])
    assert sched[0] == 0.0  # warm-up starts from lr = 0


def test_patch_gather_is_contiguous_run_not_im2col():
    # 28x28, P=14, C=1: patches are flat ranges [0,196) [14,210) [392,588) [406,602)
    img = np.arange(2 * 784, dtype=np.float32).reshape(2, 1, 28, 28)
    out = V.f32(8, 196)
    O.vo_patch_gather(img, 2, 1, 28, 14, out)
    for b in range(2):
        for k, start in enumerate((0, 14, 392, 406)):
            np.testing.assert_array_equal(out[b * 4 + k], np.arange(start, start + 196) + b * 784)
    # C=3: only channel 0 is ever read
    img = np.zeros((1, 3, 28, 28), np.float32)
    img[0, 1:] = 7.0
    out = V.f32(16, 147)
    O.vo_patch_gather(img, 1, 3, 28, 7, out)
    assert out.max() == 0.0


@pytest.mark.parametrize("name", list(V.MODEL_CASES))
def test_model_golden(name):
    cfgd, B, seed, steps, full = V.MODEL_CASES[name]
    c = V.model_case_inputs(cfgd, B, seed, steps)
    g = V.golden(f"model_{name}.npz")
    cfg, prm = c["cfg"], c["prm"]
    cs = np.array([prm.astype(np.float64).sum(), np.abs(prm).astype(np.float64).sum()])
    np.testing.assert_allclose(cs, g["param_checksum"], rtol=1e-12)  # the seeded weights are the stored ones
    K, L, N, D = cfg.num_classes, cfg.num_layers, cfg.tokens, cfg.embedding_dim
    shapes = V.param_shapes(cfg)
    n = len(prm)
    acts, lg = V.f32(L + 2, B, N, D), V.f32(B, K)
    O.vo_forward(cfg, prm, c["x"], B, lg, acts)
    for i in range(L + 2):
        close(acts[i], g["acts"][i], 2e-5)
    close(lg, g["logits"], 2e-5); close(lg, g["logits_eval"], 2e-5)
    grads, dacts = V.f32(n), V.f32(L + 1, B, N, D)
    loss = O.vo_forward_backward(cfg, prm, c["x"], c["y"], B, c["w"], K, lg, grads, dacts)
    assert abs(loss - g["loss"]) <= 1e-5 * abs(g["loss"])
    for i in range(L + 1):
        close(dacts[i], g["dacts"][i], 2e-5)
    V.check_digest(grads, shapes, g["grad_sums"], g["grad_norms"], g["grad_samples"], 2e-5, "grad")
    if full:
        off = 0
        for shp in shapes:
            k = int(np.prod(shp))
            close(grads[off:off + k], g["grads"][off:off + k], 2e-5)
            off += k
    s = O.vo_session_create(cfg, 0.9, 0.999, 1e-8, 1e-4)
    O.vo_session_set_params(s, prm); O.vo_session_set_class_weights(s, c["w"], K)
    for t in range(steps):
        lr = O.vo_cosine_warmup_lr(t + 1, 2, steps + 2, 3e-4)
        lo = O.vo_session_train_step(s, c["xs"][t], c["ys"][t], B, lr, lg, None)
        assert abs(lo - g["step_losses"][t]) <= 2e-5 * abs(g["step_losses"][t])
    p2 = V.f32(n)
    O.vo_session_get_params(s, p2)
    V.check_digest(p2, shapes, g["p_sums"], g["p_norms"], g["p_samples"], 2e-5, "param")
    O.vo_session_destroy(s)


def test_shipped_checkpoint_top1_identical():
    g = V.golden("ck_top1.npz")
    cfg = V.Cfg.make(**V.CONFIGS["CK"])
    prm, _ = V.read_weights_file(V.GOLDEN + "/vit_mnist_4_8.weights", V.param_shapes(cfg))
    assert len(prm) == 899466
    x = g["x"]
    B = len(x)
    lg = V.f32(B, 10)
    O.vo_forward(cfg, prm, x, B, lg, None)
    am = np.zeros(B, np.int32)
    O.vo_argmax_rows(lg, B, 10, am)
    np.testing.assert_array_equal(am, g["argmax"])
    close(lg, g["logits"], 2e-5)
    assert g["min_margin"] > 1e-3  # identity is not decided by rounding on this set


def test_loss_curve_golden():
    c = V.loss_curve_inputs()
    g = V.golden("loss_curve_c1_b32.npz")
    s = O.vo_session_create(c["cfg"], 0.9, 0.999, 1e-8, 1e-4)
    O.vo_session_set_params(s, c["prm"]); O.vo_session_set_class_weights(s, c["w"], 10)
    lg = V.f32(c["B"], 10)
    for t in range(c["steps"]):
        lr = O.vo_cosine_warmup_lr(t, 2, c["steps"], 3e-4)
        assert lr == g["lrs"][t]
        lo = O.vo_session_train_step(s, c["xs"][t], c["ys"][t], c["B"], lr, lg, None)
        assert abs(lo - g["losses"][t]) <= 1e-4 * abs(g["losses"][t]), (t, lo, g["losses"][t])
    O.vo_session_destroy(s)
    assert g["losses"][-1] < g["losses"][0]


@pytest.mark.parametrize("name", list(V.AUGMENT_CASES))
def test_augmentation_golden(name):
    """tests/golden/augment.npz: the reference's DataAugmentation::apply with a seeded engine."""
    c = V.augment_case(name)
    o = V.f32(c["B"], c["C"], c["H"], c["W"])
    O.vo_augment(c["x"], c["B"], c["C"], c["H"], c["W"], c["seed"], c["cfg"], o, None)
    np.testing.assert_array_equal(o, V.golden("augment.npz")[name])


def test_dropout_golden():
    """The reference at dropout 0.2 with the masks it drew (tests/golden/dropout.npz): a block and
    a whole C1 model, forward, backward (attention-branch mask applied, feed-forward mask not,
    src/layers/FeedForward.cpp:30-31) and two Adam steps."""
    g = V.golden("dropout.npz")
    c = V.dropout_block_case()
    D, H, h, B, N, rate = c["D"], c["H"], c["h"], c["B"], c["N"], c["rate"]
    masks = V.unpack_masks(g["block_masks"], (2, B * N * D), rate)
    assert 0.1 < (masks == 0).mean() < 0.3
    y, g_, dx = V.f32(B, N, D), V.f32(c["n"]), V.f32(B, N, D)
    O.vo_block_masked(D, h, H, c["prm"], c["x"], B, N, masks, c["dy"], y, g_, dx)
    close(y, g["block_y"]); close(g_, g["block_g"]); close(dx, g["block_dx"])

    cfgd, B, seed, steps = V.DROPOUT_MODEL_CASE
    c = V.model_case_inputs(cfgd, B, seed, steps)
    cfg, K = c["cfg"], c["cfg"].num_classes
    L, N, D = cfg.num_layers, cfg.tokens, cfg.embedding_dim
    shapes = V.param_shapes(cfg)
    n = len(c["prm"])
    masks = V.unpack_masks(g["model_masks"], (L, 2, B * N * D), cfg.dropout_rate)
    acts, lg, gr, dacts = V.f32(L + 2, B, N, D), V.f32(B, K), V.f32(n), V.f32(L + 1, B, N, D)
    loss = O.vo_forward_backward_masked(cfg, c["prm"], c["x"], c["y"], B, c["w"], K, masks, lg, gr, acts, dacts)
    assert abs(loss - g["model_loss"]) <= 1e-5 * abs(g["model_loss"])
    close(acts, g["model_acts"]); close(lg, g["model_logits"])
    for i in range(L + 1):
        close(dacts[i], g["model_dacts"][i], 2e-5)
    V.check_digest(gr, shapes, g["grad_sums"], g["grad_norms"], g["grad_samples"], 2e-5, "grad")
    s = O.vo_session_create(cfg, 0.9, 0.999, 1e-8, 1e-4)
    O.vo_session_set_params(s, c["prm"]); O.vo_session_set_class_weights(s, c["w"], K)
    for t in range(steps):
        lr = O.vo_cosine_warmup_lr(t + 1, 2, steps + 2, 3e-4)
        masks = V.unpack_masks(g["step_masks"][t], (L, 2, B * N * D), cfg.dropout_rate)
        lo = O.vo_session_train_step_masked(s, c["xs"][t], c["ys"][t], B, lr, masks, lg, None)
        assert abs(lo - g["step_losses"][t]) <= 1e-5 * abs(g["step_losses"][t])
    p = V.f32(n)
    O.vo_session_get_params(s, p)
    V.check_digest(p, shapes, g["p_sums"], g["p_norms"], g["p_samples"], 1e-5, "param after Adam")
    O.vo_session_destroy(s)


def test_embeddings_image_gradient_golden():
    g = V.golden("embed_input_grad.npz")
    for tag in ("emb_a", "emb_b", "emb_c"):
        c = V.op_case(tag)
        dimg = V.f32(c["B"], c["C"], c["img"], c["img"])
        O.vo_embeddings_input_grad(c["img"], c["P"], c["C"], c["D"], c["prm"], c["B"], c["dy"], dimg)
        np.testing.assert_array_equal(dimg, g[tag])


def test_oracle_runs_clean_under_address_and_undefined_behaviour_sanitizers():
    """`make -C oracle sanitize`: the C restatement driven through its model-level and layer-level
    entry points in a -fsanitize=address,undefined build (host code; runtime linked statically)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or shutil.which("make") is None:
        pytest.skip("needs gcc and make")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "oracle"), "sanitize"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "clean" in r.stdout
