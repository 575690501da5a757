"""Generates the golden vectors in this directory from the REFERENCE ITSELF
(oracle/_ref/libvit_ref.so = /root/reference/src compiled in place by oracle/Makefile).

Run where /root/reference exists:   python tests/golden/make_golden.py
Outputs are data only: inputs are regenerated from seeds by tests/vitlibs.py (op_case,
model_case_inputs, loss_curve_inputs); the reference's outputs are stored here.
The two shipped-checkpoint files vit_mnist_4_8.{json,weights} are data files copied verbatim
from the reference's models/ directory (the known-answer artefact for top-1 identity).
"""
import os
import shutil
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vitlibs as V  # noqa: E402

R = V.ref()
assert R is not None, "build oracle/_ref first (make -C oracle ref)"
G = V.GOLDEN
REFDIR = "/root/reference"


def ops():
    out = {}
    c = V.op_case("dense")
    y, dW, db, dx = V.f32(c["rows"], c["o"]), V.f32(c["i"], c["o"]), V.f32(1, c["o"]), V.f32(c["rows"], c["i"])
    R.vref_dense(c["i"], c["o"], c["W"], c["b"], c["x"], c["rows"], c["dy"], y, dW, db, dx)
    out.update(dense_y=y, dense_dW=dW, dense_db=db, dense_dx=dx)
    c = V.op_case("ln")
    rows, D = c["rows"], c["D"]
    y, dg, dbe, dx = V.f32(rows, D), V.f32(1, D), V.f32(1, D), V.f32(rows, D)
    R.vref_layernorm(D, 1e-5, c["g"], c["b"], c["x"], rows, c["dy"], y, dg, dbe, dx)
    out.update(ln_y=y, ln_dgamma=dg, ln_dbeta=dbe, ln_dx=dx)
    c = V.op_case("gelu")
    y, dx = V.f32(c["n"]), V.f32(c["n"])
    R.vref_gelu(c["x"], c["n"], c["dy"], y, dx)
    out.update(gelu_y=y, gelu_dx=dx)
    c = V.op_case("softmax")
    p, d = V.f32(6, 17, 17), V.f32(6, 17, 17)
    R.vref_softmax3d(c["x"], 6, 17, 17, p)
    R.vref_softmax3d_backward(c["g"], p, 6, 17, 17, d)
    out.update(softmax_p=p, softmax_dx=d)
    for tag in ("mha_a", "mha_b"):
        c = V.op_case(tag)
        y, g_, dx = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"]), V.f32(c["B"], c["N"], c["D"])
        R.vref_mha(c["D"], c["h"], c["prm"], c["x"], c["B"], c["N"], c["dy"], y, g_, dx)
        out.update({f"{tag}_y": y, f"{tag}_g": g_, f"{tag}_dx": dx})
    c = V.op_case("ffn")
    y, g_, dx = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"]), V.f32(c["B"], c["N"], c["D"])
    R.vref_ffn(c["D"], c["H"], c["prm"], c["x"], c["B"], c["N"], c["dy"], y, g_, dx)
    out.update(ffn_y=y, ffn_g=g_, ffn_dx=dx)
    c = V.op_case("block")
    y, g_, dx = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"]), V.f32(c["B"], c["N"], c["D"])
    R.vref_block(c["D"], c["h"], c["H"], c["prm"], c["x"], c["B"], c["N"], c["dy"], y, g_, dx)
    out.update(block_y=y, block_g=g_, block_dx=dx)
    for tag in ("emb_a", "emb_b", "emb_c"):
        c = V.op_case(tag)
        y, g_ = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"])
        dimg = V.f32(c["B"], c["C"], c["img"], c["img"])
        R.vref_embeddings(c["img"], c["P"], c["C"], c["D"], c["prm"], c["x"], c["B"], c["dy"], y, g_, dimg)
        out.update({f"{tag}_y": y, f"{tag}_g": g_})
    c = V.op_case("ce")
    g_ = V.f32(c["B"], c["K"])
    out["ce_loss_w"] = np.float32(R.vref_cross_entropy(c["lg"], c["y"], c["B"], c["K"], c["w"], c["K"], g_))
    out["ce_grad"] = g_.copy()
    out["ce_loss"] = np.float32(R.vref_cross_entropy(c["lg"], c["y"], c["B"], c["K"], None, 0, g_))
    c = V.op_case("adam")
    p = c["p"].copy()
    R.vref_adam(c["r"], c["c"], 0.9, 0.999, 1e-8, 1e-4, c["steps"], c["lrs"], c["grads"], p)
    out["adam_p"] = p
    out["sched"] = np.array([R.vref_cosine_warmup_lr(s, 10, 100, 3e-4) for s in range(100)], dtype=np.float32)
    for ep in (0, 1, 7):
        o_ = np.zeros(50, np.uint64)
        R.vref_epoch_order(50, ep, o_)
        out[f"order_{ep}"] = o_
    np.savez_compressed(os.path.join(G, "ops.npz"), **out)


def model_case(name, cfgd, B, seed, steps, full_grads):
    c = V.model_case_inputs(cfgd, B, seed, steps)
    cfg, prm, w, x, y = c["cfg"], c["prm"], c["w"], c["x"], c["y"]
    K, L, N, D = cfg.num_classes, cfg.num_layers, cfg.tokens, cfg.embedding_dim
    shapes = V.param_shapes(cfg)
    n = len(prm)
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    assert R.vref_param_count(s) == n
    R.vref_set_params(s, prm)
    R.vref_set_class_weights(s, w, K)
    out = dict(param_checksum=np.array([prm.astype(np.float64).sum(), np.abs(prm).astype(np.float64).sum()]))
    ev = V.f32(B, K)
    R.vref_forward(s, x, B, 0, ev)  # eval pass first: it clobbers the layers' saved training state
    acts, lg = V.f32(L + 2, B, N, D), V.f32(B, K)
    R.vref_forward_trace(s, x, B, acts, lg)
    out.update(acts=acts, logits=lg.copy(), logits_eval=ev)
    dl = V.f32(B, K)
    out["loss"] = np.float32(R.vref_cross_entropy(lg, y, B, K, w, K, dl))
    dacts = V.f32(L + 1, B, N, D)
    R.vref_backward_trace(s, dl, B, dacts)
    g = V.f32(n)
    R.vref_get_grads(s, g)
    out.update(dlogits=dl, dacts=dacts)
    if full_grads:
        out["grads"] = g
    gs, gn, gsm = V.tensor_digest(g, shapes)
    out.update(grad_sums=gs, grad_norms=gn, grad_samples=gsm)
    losses = []
    for t in range(steps):
        lr = R.vref_cosine_warmup_lr(t + 1, 2, steps + 2, 3e-4)
        losses.append(R.vref_train_step(s, c["xs"][t], c["ys"][t], B, lr, lg, None))
    p2 = V.f32(n)
    R.vref_get_params(s, p2)
    ps, pn, psm = V.tensor_digest(p2, shapes)
    out.update(step_losses=np.array(losses, np.float32), p_sums=ps, p_norms=pn, p_samples=psm)
    if full_grads:
        out["params_after"] = p2
    R.vref_destroy(s)
    np.savez_compressed(os.path.join(G, f"model_{name}.npz"), **out)


def ck_inputs():
    """256 inputs for the shipped checkpoint: 128 synthetic stroke images on the MNIST-normalised
    background + 128 N(0,1) noise images (the MNIST CSVs are a download and are not available)."""
    B = 256
    rng = np.random.default_rng(1234)
    x = rng.standard_normal((B, 1, 28, 28)).astype(np.float32)
    bg = np.float32((0 - 0.1307) / 0.3081)
    fg = np.float32((1 - 0.1307) / 0.3081)
    for i in range(B // 2):
        img = np.full((28, 28), bg, np.float32)
        for _ in range(3):
            r0, c0 = rng.integers(4, 24, 2)
            dr, dc = rng.integers(-1, 2, 2)
            for t in range(rng.integers(5, 14)):
                rr, cc = np.clip(r0 + dr * t, 0, 27), np.clip(c0 + dc * t, 0, 27)
                img[max(rr - 1, 0):rr + 2, max(cc - 1, 0):cc + 2] = fg
        x[i, 0] = img
    return x


def shipped_checkpoint():
    for ext in ("json", "weights"):
        shutil.copyfile(f"{REFDIR}/models/vit_mnist_4_8.{ext}", os.path.join(G, f"vit_mnist_4_8.{ext}"))
    cfg = V.Cfg()
    assert R.vref_load_config(f"{REFDIR}/models/vit_mnist_4_8.json".encode(), cfg) == 0
    assert cfg.as_dict() == dict(V.CONFIGS["CK"], dropout_rate=0.0), cfg.as_dict()
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    assert R.vref_load_weights(s, f"{REFDIR}/models/vit_mnist_4_8.weights".encode()) == 0
    n = R.vref_param_count(s)
    p = V.f32(n)
    R.vref_get_params(s, p)
    flat, shapes = V.read_weights_file(os.path.join(G, "vit_mnist_4_8.weights"), V.param_shapes(cfg))
    assert np.array_equal(flat, p)
    x = ck_inputs()
    B = len(x)
    lg = V.f32(B, 10)
    R.vref_forward(s, x, B, 0, lg)
    am = np.zeros(B, np.int32)
    R.vref_argmax(lg, B, 10, am)
    srt = np.sort(lg, axis=1)
    np.savez_compressed(os.path.join(G, "ck_top1.npz"), x=x, logits=lg, argmax=am,
                        min_margin=np.float32((srt[:, -1] - srt[:, -2]).min()))
    tmp = os.path.join(G, "_tmp.weights")
    assert R.vref_save_weights(s, tmp.encode()) == 0
    assert open(tmp, "rb").read() == open(os.path.join(G, "vit_mnist_4_8.weights"), "rb").read()
    os.remove(tmp)
    R.vref_destroy(s)


def loss_curve():
    c = V.loss_curve_inputs()
    cfg, B, steps = c["cfg"], c["B"], c["steps"]
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    R.vref_set_params(s, c["prm"])
    R.vref_set_class_weights(s, c["w"], 10)
    losses, lrs = [], []
    lg = V.f32(B, 10)
    for t in range(steps):
        lr = R.vref_cosine_warmup_lr(t, 2, steps, 3e-4)
        lrs.append(lr)
        losses.append(R.vref_train_step(s, c["xs"][t], c["ys"][t], B, lr, lg, None))
    np.savez_compressed(os.path.join(G, "loss_curve_c1_b32.npz"), losses=np.array(losses, np.float32),
                        lrs=np.array(lrs, np.float32))
    R.vref_destroy(s)


def augment():
    """DataAugmentation::apply of the reference with its engine re-seeded (vref_augment)."""
    out = {}
    for name in V.AUGMENT_CASES:
        c = V.augment_case(name)
        y = V.f32(c["B"], c["C"], c["H"], c["W"])
        R.vref_augment(c["x"], c["B"], c["C"], c["H"], c["W"], c["seed"], c["cfg"], y)
        out[name] = y
    np.savez_compressed(os.path.join(G, "augment.npz"), **out)


def embed_input_grad():
    """The image gradient Embeddings::backward returns for the three embedding cases of ops.npz."""
    out = {}
    for tag in ("emb_a", "emb_b", "emb_c"):
        c = V.op_case(tag)
        y, g_ = V.f32(c["B"], c["N"], c["D"]), V.f32(c["n"])
        dimg = V.f32(c["B"], c["C"], c["img"], c["img"])
        R.vref_embeddings(c["img"], c["P"], c["C"], c["D"], c["prm"], c["x"], c["B"], c["dy"], y, g_, dimg)
        out[tag] = dimg
    np.savez_compressed(os.path.join(G, "embed_input_grad.npz"), **out)


def tensor_surface():
    """The Tensor-API scenarios of tests/tensor_surface.inc run on the reference's own Tensor class."""
    out = {}
    for sc in range(V.TENSOR_SCENARIOS):
        out[f"s{sc}"] = V.run_tensor_scenario(R.dll.vref_tensor_scenario, sc)
    np.savez_compressed(os.path.join(G, "tensor_surface.npz"), **out)


def apps():
    """stdout of the reference's evaluation app (app/test.cpp built against the reference's own
    sources: oracle/Makefile ref_apps) on a synthetic MNIST-format test set and the shipped
    vit_mnist_4_8 checkpoint: accuracy, per-class precision / recall / F1, confusion matrix."""
    import subprocess
    import tempfile
    exe = os.path.join(V.ROOT, "oracle", "_ref", "ref_app_test")
    assert os.path.exists(exe), "make -C oracle ref_apps"
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "data"))
        os.makedirs(os.path.join(d, "models"))
        V.write_digit_csv(os.path.join(d, "data", "mnist_test.csv"), V.APP_TEST_ROWS, V.APP_TEST_SEED)
        for ext in ("json", "weights"):
            shutil.copyfile(os.path.join(G, f"vit_mnist_4_8.{ext}"), os.path.join(d, "models", f"vit_mnist_4_8.{ext}"))
        r = subprocess.run([exe], cwd=d, capture_output=True, text=True, check=True)
    with open(os.path.join(G, "app_test_stdout.txt"), "w") as f:
        f.write(r.stdout)
    print(r.stdout)


def dropout():
    """The reference at dropout 0.2 (its own default, app/main.cpp:19).  Its masks come from a
    random_device-seeded engine, so each run below is unrepeatable; the masks it drew are read
    back (vref_block_dropout / vref_get_dropout_masks) and stored, one bit per element, next to
    what it computed with them."""
    out = {}
    c = V.dropout_block_case()
    D, H, h, B, N, rate = c["D"], c["H"], c["h"], c["B"], c["N"], c["rate"]
    y, g_, dx, masks = V.f32(B, N, D), V.f32(c["n"]), V.f32(B, N, D), V.f32(2, B * N * D)
    R.vref_block_dropout(D, h, H, rate, c["prm"], c["x"], B, N, c["dy"], y, g_, dx, masks)
    out.update(block_y=y, block_g=g_, block_dx=dx, block_masks=V.pack_masks(masks))

    cfgd, B, seed, steps = V.DROPOUT_MODEL_CASE
    c = V.model_case_inputs(cfgd, B, seed, steps)
    cfg, prm, w, x, y = c["cfg"], c["prm"], c["w"], c["x"], c["y"]
    K, L, N, D = cfg.num_classes, cfg.num_layers, cfg.tokens, cfg.embedding_dim
    shapes = V.param_shapes(cfg)
    n = len(prm)
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    R.vref_set_params(s, prm)
    R.vref_set_class_weights(s, w, K)
    acts, lg, masks = V.f32(L + 2, B, N, D), V.f32(B, K), V.f32(L, 2, B * N * D)
    R.vref_forward_trace(s, x, B, acts, lg)
    R.vref_get_dropout_masks(s, B, masks)
    dl = V.f32(B, K)
    out["model_loss"] = np.float32(R.vref_cross_entropy(lg, y, B, K, w, K, dl))
    dacts = V.f32(L + 1, B, N, D)
    R.vref_backward_trace(s, dl, B, dacts)
    g = V.f32(n)
    R.vref_get_grads(s, g)
    gs, gn, gsm = V.tensor_digest(g, shapes)
    out.update(model_acts=acts, model_logits=lg.copy(), model_dlogits=dl, model_dacts=dacts,
               model_masks=V.pack_masks(masks), grad_sums=gs, grad_norms=gn, grad_samples=gsm)
    losses, step_masks = [], []
    for t in range(steps):
        lr = R.vref_cosine_warmup_lr(t + 1, 2, steps + 2, 3e-4)
        losses.append(R.vref_train_step(s, c["xs"][t], c["ys"][t], B, lr, lg, None))
        R.vref_get_dropout_masks(s, B, masks)
        step_masks.append(V.pack_masks(masks))
    p2 = V.f32(n)
    R.vref_get_params(s, p2)
    ps, pn, psm = V.tensor_digest(p2, shapes)
    out.update(step_losses=np.array(losses, np.float32), step_masks=np.stack(step_masks), p_sums=ps, p_norms=pn,
               p_samples=psm)
    R.vref_destroy(s)
    np.savez_compressed(os.path.join(G, "dropout.npz"), **out)


if __name__ == "__main__":
    if sys.argv[1:] == ["augment"]:
        augment()
        sys.exit(0)
    if sys.argv[1:] == ["embed_input_grad"]:
        embed_input_grad()
        sys.exit(0)
    if sys.argv[1:] == ["tensor_surface"]:
        tensor_surface()
        sys.exit(0)
    if sys.argv[1:] == ["apps"]:
        apps()
        sys.exit(0)
    if sys.argv[1:] == ["dropout"]:
        dropout()
        sys.exit(0)
    ops()
    augment()
    dropout()
    embed_input_grad()
    tensor_surface()
    apps()
    for name, (cfgd, B, seed, steps, full) in V.MODEL_CASES.items():
        model_case(name, cfgd, B, seed, steps, full)
    shipped_checkpoint()
    loss_curve()
    for f in sorted(os.listdir(G)):
        print(f, os.path.getsize(os.path.join(G, f)))
