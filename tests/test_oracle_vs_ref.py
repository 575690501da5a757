"""Pins oracle/libvit_oracle.so (this repo's C restatement) against oracle/_ref/libvit_ref.so
(the reference's own sources compiled in place).  Skipped where the reference build is absent
(the golden-vector tests in test_oracle_golden.py pin the oracle there instead).

Tolerances: exact (0) wherever the reference is deterministic and the restatement keeps its
operation order; 2e-6 relative where the reference itself reduces in OpenMP thread order
(the CE loss value) or where libm calls vectorise differently.
"""
import numpy as np
import pytest

import vitlibs as V

R = V.ref()
O = V.oracle()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref not built (no /root/reference here)")


def same(a, b, tol=0.0):
    if tol == 0.0:
        np.testing.assert_array_equal(a, b)
    else:
        assert V.max_relerr(a, b) <= tol, V.max_relerr(a, b)


def test_matmul_bmm():
    a, b = V.rng_normal(1, 37, 53), V.rng_normal(2, 53, 29)
    o, r = V.f32(37, 29), V.f32(37, 29)
    O.vo_matmul(a, b, 37, 53, 29, o)
    R.vref_matmul(a, b, 37, 53, 29, r)
    same(o, r)
    a, b = V.rng_normal(3, 6, 17, 12), V.rng_normal(4, 6, 12, 17)
    o, r = V.f32(6, 17, 17), V.f32(6, 17, 17)
    O.vo_bmm(a, b, 6, 17, 12, 17, o)
    R.vref_bmm(a, b, 6, 17, 12, 17, r)
    same(o, r)


@pytest.mark.parametrize("rows,i,o_", [(40, 96, 384), (17, 147, 192), (8, 64, 10)])
def test_dense(rows, i, o_):
    W, b = V.rng_normal(1, i, o_, scale=0.02), V.rng_normal(2, 1, o_, scale=0.02)
    x, dy = V.rng_normal(3, rows, i), V.rng_normal(4, rows, o_)
    y, dW, db, dx = V.f32(rows, o_), V.f32(i, o_), V.f32(1, o_), V.f32(rows, i)
    y2, dW2, db2, dx2 = V.f32(rows, o_), V.f32(i, o_), V.f32(1, o_), V.f32(rows, i)
    O.vo_dense_forward(x, rows, i, o_, W, b, y)
    O.vo_dense_backward(x, dy, rows, i, o_, W, dW, db, dx)
    R.vref_dense(i, o_, W, b, x, rows, dy, y2, dW2, db2, dx2)
    same(y, y2); same(dW, dW2); same(db, db2); same(dx, dx2)


@pytest.mark.parametrize("rows,D", [(85, 96), (640, 64), (33, 192)])
def test_layernorm(rows, D):
    g, b = 1 + V.rng_normal(1, 1, D, scale=0.1), V.rng_normal(2, 1, D, scale=0.1)
    x, dy = V.rng_normal(3, rows, D) * 2 + 0.3, V.rng_normal(4, rows, D)
    y, xh, inv, dg, db, dx = V.f32(rows, D), V.f32(rows, D), V.f32(rows), V.f32(1, D), V.f32(1, D), V.f32(rows, D)
    y2, dg2, db2, dx2 = V.f32(rows, D), V.f32(1, D), V.f32(1, D), V.f32(rows, D)
    O.vo_layernorm_forward(x, rows, D, g, b, 1e-5, y, xh, inv)
    O.vo_layernorm_backward(dy, xh, inv, g, rows, D, dg, db, dx)
    R.vref_layernorm(D, 1e-5, g, b, x, rows, dy, y2, dg2, db2, dx2)
    same(y, y2); same(dg, dg2); same(db, db2); same(dx, dx2)


def test_gelu():
    x, dy = V.rng_normal(1, 5000) * 3, V.rng_normal(2, 5000)
    y, dx, y2, dx2 = V.f32(5000), V.f32(5000), V.f32(5000), V.f32(5000)
    O.vo_gelu_forward(x, 5000, y)
    O.vo_gelu_backward(x, dy, 5000, dx)
    R.vref_gelu(x, 5000, dy, y2, dx2)
    same(y, y2, 2e-6); same(dx, dx2, 2e-6)


def test_softmax():
    x, g = V.rng_normal(1, 12, 17, 17) * 3, V.rng_normal(2, 12, 17, 17)
    p, p2, d, d2 = V.f32(12, 17, 17), V.f32(12, 17, 17), V.f32(12, 17, 17), V.f32(12, 17, 17)
    O.vo_softmax_rows(x, 12 * 17, 17, p)
    R.vref_softmax3d(x, 12, 17, 17, p2)
    same(p, p2, 2e-6)
    O.vo_softmax_rows_backward(g, p2, 12 * 17, 17, d)
    R.vref_softmax3d_backward(g, p2, 12, 17, 17, d2)
    same(d, d2)


@pytest.mark.parametrize("D,h,B,N", [(96, 8, 3, 5), (64, 8, 2, 17), (48, 3, 2, 9)])
def test_mha(D, h, B, N):
    n = 4 * (D * D + D)
    prm = V.rng_normal(1, n, scale=0.2)
    x, dy = V.rng_normal(2, B, N, D), V.rng_normal(3, B, N, D)
    y, g, dx = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    y2, g2, dx2 = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    O.vo_mha(D, h, prm, x, B, N, dy, y, g, dx)
    R.vref_mha(D, h, prm, x, B, N, dy, y2, g2, dx2)
    same(y, y2, 2e-6); same(g, g2, 2e-6); same(dx, dx2, 2e-6)


def test_ffn_and_block():
    D, H, h, B, N = 64, 160, 8, 3, 5
    n = D * H + H + H * D + D
    prm = V.rng_normal(1, n, scale=0.1)
    x, dy = V.rng_normal(2, B, N, D), V.rng_normal(3, B, N, D)
    y, g, dx = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    y2, g2, dx2 = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    O.vo_ffn(D, H, prm, x, B, N, dy, y, g, dx)
    R.vref_ffn(D, H, prm, x, B, N, dy, y2, g2, dx2)
    same(y, y2, 2e-6); same(g, g2, 2e-6); same(dx, dx2, 2e-6)
    n = 4 * D + 4 * (D * D + D) + n
    prm = V.rng_normal(4, n, scale=0.1)
    prm[:D] += 1; prm[2 * D + 4 * (D * D + D):3 * D + 4 * (D * D + D)] += 1
    y, g, dx = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    y2, g2, dx2 = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    O.vo_block(D, h, H, prm, x, B, N, dy, y, g, dx)
    R.vref_block(D, h, H, prm, x, B, N, dy, y2, g2, dx2)
    same(y, y2, 2e-6); same(g, g2, 5e-6); same(dx, dx2, 5e-6)


@pytest.mark.parametrize("img,P,C,D,B", [(28, 14, 1, 96, 4), (28, 7, 3, 48, 3), (32, 16, 3, 40, 2)])
def test_embeddings(img, P, C, D, B):
    np_ = (img // P) ** 2
    N, pd = np_ + 1, P * P * C
    n = pd * D + D + D + N * D
    prm = V.rng_normal(1, n, scale=0.05)
    x, dy = V.rng_normal(2, B, C, img, img), V.rng_normal(3, B, N, D)
    y, g = V.f32(B, N, D), V.f32(n)
    y2, g2, dimg = V.f32(B, N, D), V.f32(n), V.f32(B, C, img, img)
    O.vo_embeddings(img, P, C, D, prm, x, B, dy, y, g)
    R.vref_embeddings(img, P, C, D, prm, x, B, dy, y2, g2, dimg)
    same(y, y2); same(g, g2)
    # the image gradient the reference returns (and its Trainer drops): patch 0 of image 0 everywhere
    dimg_o = V.f32(B, C, img, img)
    O.vo_embeddings_input_grad(img, P, C, D, prm, B, dy, dimg_o)
    same(dimg_o, dimg)
    assert np.abs(dimg).max() > 0
    np.testing.assert_array_equal(dimg[0, :, :P, :P], dimg[B - 1, :, img - P:, img - P:])


def test_cross_entropy_adam_schedule_argmax():
    B, K = 37, 10
    lg = V.rng_normal(1, B, K) * 2
    y = V.onehot(np.random.default_rng(2).integers(0, K, B), K)
    w = np.abs(V.rng_normal(3, K)) + 0.1
    w /= w.sum()
    for ww, n in ((w, K), (None, 0)):
        g, g2 = V.f32(B, K), V.f32(B, K)
        lo = O.vo_cross_entropy(lg, y, B, K, ww, n, g)
        lr = R.vref_cross_entropy(lg, y, B, K, ww, n, g2)
        assert abs(lo - lr) <= 2e-6 * abs(lr)
        same(g, g2, 2e-6)
    # Adam: 5 steps on one {7,9} tensor
    r, c, steps = 7, 9, 5
    p0 = V.rng_normal(4, r, c)
    grads = V.rng_normal(5, steps, r, c)
    lrs = np.array([0.0, 1e-4, 3e-4, 3e-4, 2e-4], dtype=np.float32)
    pr = p0.copy()
    R.vref_adam(r, c, 0.9, 0.999, 1e-8, 1e-4, steps, lrs, grads, pr)
    po, m, v = p0.copy(), V.f32(r, c), V.f32(r, c)
    for s in range(steps):
        b1t = np.float32(np.float64(np.float32(0.9)) ** (s + 1))
        b2t = np.float32(np.float64(np.float32(0.999)) ** (s + 1))
        O.vo_adam_update(po, grads[s], m, v, r * c, float(lrs[s]), 0.9, 0.999, float(b1t), float(b2t), 1e-8, 1e-4)
    same(po, pr)
    for step, wu, tot in [(0, 10, 100), (5, 10, 100), (10, 10, 100), (55, 10, 100), (99, 10, 100), (3, 0, 7)]:
        assert O.vo_cosine_warmup_lr(step, wu, tot, 3e-4) == R.vref_cosine_warmup_lr(step, wu, tot, 3e-4)
    lg[3, 4] = lg[3, 7] = lg[3].max() + 1  # tie: first index wins
    a, a2 = np.zeros(B, np.int32), np.zeros(B, np.int32)
    O.vo_argmax_rows(lg, B, K, a)
    R.vref_argmax(lg, B, K, a2)
    same(a, a2)
    assert a[3] == 4


@pytest.mark.parametrize("name,B", [("C1", 4), ("CK", 3), ("C2", 2)])
def test_model_forward_backward(name, B):
    cfg = V.Cfg.make(**V.CONFIGS[name])
    n = O.vo_param_count(cfg)
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    assert n == R.vref_param_count(s)
    assert O.vo_param_tensors(cfg) == R.vref_param_tensors(s)
    import ctypes as C
    for i in range(O.vo_param_tensors(cfg)):
        d1, d2 = (C.c_uint64 * 4)(), (C.c_uint64 * 4)()
        assert O.vo_param_shape(cfg, i, d1) == R.vref_param_shape(s, i, d2)
        assert list(d1)[:3] == list(d2)[:3]
    assert [tuple(x) for x in V.param_shapes(cfg)] and sum(int(np.prod(x)) for x in V.param_shapes(cfg)) == n
    prm = V.init_params(cfg, 7)
    R.vref_set_params(s, prm)
    x = V.rng_normal(8, B, cfg.in_channels, cfg.image_size, cfg.image_size)
    y = V.onehot(np.random.default_rng(9).integers(0, cfg.num_classes, B), cfg.num_classes)
    L, N, D, K = cfg.num_layers, cfg.tokens, cfg.embedding_dim, cfg.num_classes
    acts, acts2 = V.f32(L + 2, B, N, D), V.f32(L + 2, B, N, D)
    lg, lg2 = V.f32(B, K), V.f32(B, K)
    O.vo_forward(cfg, prm, x, B, lg, acts)
    R.vref_forward_trace(s, x, B, acts2, lg2)
    same(acts, acts2, 5e-6); same(lg, lg2, 5e-6)
    lg3 = V.f32(B, K)
    R.vref_forward(s, x, B, 0, lg3)
    same(lg3, lg2)
    g, g2 = V.f32(n), V.f32(n)
    dacts, dacts2 = V.f32(L + 1, B, N, D), V.f32(L + 1, B, N, D)
    lo = O.vo_forward_backward(cfg, prm, x, y, B, None, 0, lg, g, dacts)
    lr = R.vref_forward_backward(s, x, y, B, lg2)
    R.vref_get_grads(s, g2)
    dl = V.f32(B, K)
    R.vref_cross_entropy(lg2, y, B, K, None, 0, dl)
    R.vref_forward_trace(s, x, B, acts2, lg2)
    R.vref_backward_trace(s, dl, B, dacts2)
    assert abs(lo - lr) <= 2e-6 * abs(lr)
    # per-tensor gradient comparison, each relative to its own scale
    off = 0
    for shp in V.param_shapes(cfg):
        k = int(np.prod(shp))
        assert V.max_relerr(g[off:off + k], g2[off:off + k]) <= 2e-5, shp
        off += k
    for i in range(L + 1):
        assert V.max_relerr(dacts[i], dacts2[i]) <= 2e-5
    R.vref_destroy(s)


def test_train_steps_match_reference():
    cfg = V.Cfg.make(**V.CONFIGS["C1"])
    B, steps = 6, 4
    n = O.vo_param_count(cfg)
    prm = V.init_params(cfg, 11)
    w = np.full(10, 0.1, np.float32)
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    R.vref_set_params(s, prm); R.vref_set_class_weights(s, w, 10)
    o = O.vo_session_create(cfg, 0.9, 0.999, 1e-8, 1e-4)
    O.vo_session_set_params(o, prm); O.vo_session_set_class_weights(o, w, 10)
    for t in range(steps):
        x = V.rng_normal(100 + t, B, 1, 28, 28)
        y = V.onehot(np.random.default_rng(200 + t).integers(0, 10, B), 10)
        lr = O.vo_cosine_warmup_lr(t + 1, 2, 10, 3e-4)
        lg, lg2 = V.f32(B, 10), V.f32(B, 10)
        lo = O.vo_session_train_step(o, x, y, B, lr, lg, None)
        lrf = R.vref_train_step(s, x, y, B, lr, lg2, None)
        assert abs(lo - lrf) <= 1e-5 * abs(lrf)
    p1, p2 = V.f32(n), V.f32(n)
    O.vo_session_get_params(o, p1); R.vref_get_params(s, p2)
    assert V.max_relerr(p1, p2) <= 1e-5
    O.vo_session_destroy(o); R.vref_destroy(s)


@pytest.mark.parametrize("name", list(V.AUGMENT_CASES))
def test_augmentation_matches_reference_bit_for_bit(name):
    """Random draws (mt19937 + libstdc++ uniform_real_distribution<float>) and the three
    resampling stages of src/utils/DataAugmentation.cpp, against the reference's own class with
    its engine re-seeded."""
    c = V.augment_case(name)
    o, r = V.f32(c["B"], c["C"], c["H"], c["W"]), V.f32(c["B"], c["C"], c["H"], c["W"])
    draws = V.f32(c["B"], 6)
    O.vo_augment(c["x"], c["B"], c["C"], c["H"], c["W"], c["seed"], c["cfg"], o, draws)
    R.vref_augment(c["x"], c["B"], c["C"], c["H"], c["W"], c["seed"], c["cfg"], r)
    same(o, r)
    if name == "all_stages_rgb":
        assert (draws[:, 0] == 7).all() and (o != c["x"]).any()
    if name == "none":
        assert (draws[:, 0] == 0).all()
        same(o, c["x"])


def test_augmentation_many_seeds():
    c = V.augment_case("all_stages_rgb")
    for seed in range(40):
        o, r = V.f32(*c["x"].shape), V.f32(*c["x"].shape)
        O.vo_augment(c["x"], c["B"], c["C"], c["H"], c["W"], seed, c["cfg"], o, None)
        R.vref_augment(c["x"], c["B"], c["C"], c["H"], c["W"], seed, c["cfg"], r)
        same(o, r)


# ------------------------------------------------------------------------------ dropout > 0
# The reference draws its masks from a random_device-seeded engine (src/layers/Dropout.cpp:31-33),
# so a run cannot be repeated -- but the masks it drew stay in its Dropout objects.  They are read
# back and handed to the oracle, which must then reproduce that run: forward with both masks,
# backward with the attention-branch mask only (src/layers/FeedForward.cpp:30-31 discards the
# masked gradient; src/model/TransformerEncoderBlock.cpp:58 keeps it).
def _check_masks(masks, rate):
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    vals = np.unique(masks)
    assert set(vals.tolist()) <= {0.0, float(scale)}, vals
    frac = float((masks == 0).mean())
    assert abs(frac - rate) < 0.05, frac


def test_block_with_dropout_matches_reference_given_its_masks():
    D, H, h, B, N, rate = 64, 160, 8, 3, 5, 0.2
    n = 4 * D + 4 * (D * D + D) + D * H + H + H * D + D
    prm = V.rng_normal(4, n, scale=0.1)
    prm[:D] += 1; prm[2 * D + 4 * (D * D + D):3 * D + 4 * (D * D + D)] += 1
    x, dy = V.rng_normal(2, B, N, D), V.rng_normal(3, B, N, D)
    y, g, dx = V.f32(B, N, D), V.f32(n), V.f32(B, N, D)
    y2, g2, dx2, masks = V.f32(B, N, D), V.f32(n), V.f32(B, N, D), V.f32(2, B * N * D)
    R.vref_block_dropout(D, h, H, rate, prm, x, B, N, dy, y2, g2, dx2, masks)
    _check_masks(masks, rate)
    O.vo_block_masked(D, h, H, prm, x, B, N, masks, dy, y, g, dx)
    same(y, y2, 2e-6); same(g, g2, 5e-6); same(dx, dx2, 5e-6)
    # the masks matter: without them (or with the feed-forward mask also applied in the
    # backward, i.e. the textbook gradient) the result is far from the reference's
    O.vo_block(D, h, H, prm, x, B, N, dy, y, g, dx)
    assert V.max_relerr(y, y2) > 1e-2 and V.max_relerr(g, g2) > 1e-2


def test_model_with_dropout_matches_reference_given_its_masks():
    cfg = V.Cfg.make(**dict(V.CONFIGS["C1"], dropout_rate=0.2))
    B, steps = 4, 3
    L, N, D, K = cfg.num_layers, cfg.tokens, cfg.embedding_dim, cfg.num_classes
    n = O.vo_param_count(cfg)
    prm = V.init_params(cfg, 21)
    w = np.full(K, 1.0 / K, np.float32)
    s = R.vref_create(cfg, 3e-4, 0.9, 0.999, 1e-8, 1e-4)
    R.vref_set_params(s, prm); R.vref_set_class_weights(s, w, K)
    o = O.vo_session_create(cfg, 0.9, 0.999, 1e-8, 1e-4)
    O.vo_session_set_params(o, prm); O.vo_session_set_class_weights(o, w, K)
    shapes = V.param_shapes(cfg)
    for t in range(steps):
        x = V.rng_normal(300 + t, B, 1, 28, 28)
        y = V.onehot(np.random.default_rng(400 + t).integers(0, K, B), K)
        lr = 3e-4 * (t + 1) / steps
        lg, lg2, g, g2 = V.f32(B, K), V.f32(B, K), V.f32(n), V.f32(n)
        masks = V.f32(L, 2, B * N * D)
        lrf = R.vref_train_step(s, x, y, B, lr, lg2, g2)
        R.vref_get_dropout_masks(s, B, masks)
        _check_masks(masks, 0.2)
        lo = O.vo_session_train_step_masked(o, x, y, B, lr, masks, lg, g)
        assert abs(lo - lrf) <= 1e-5 * abs(lrf)
        same(lg, lg2, 1e-5)
        off = 0
        for shp in shapes:
            k = int(np.prod(shp))
            assert V.max_relerr(g[off:off + k], g2[off:off + k]) <= 2e-5, (t, shp)
            off += k
    p1, p2 = V.f32(n), V.f32(n)
    O.vo_session_get_params(o, p1); R.vref_get_params(s, p2)
    assert V.max_relerr(p1, p2) <= 1e-5
    O.vo_session_destroy(o); R.vref_destroy(s)
