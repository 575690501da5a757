"""GPU parity tests of every libvitk kernel, called through the C-ABI (ctypes) and checked against
the CPU oracle (oracle/libvit_oracle.so), the committed golden vectors (tests/golden/ops.npz, made by
the compiled reference) and float64 numpy.

Tolerances (fp32 everywhere; stated relative to each tensor's max magnitude):
  * element-wise / normalisation / optimiser kernels: 2e-6 (same formulas, correctly rounded
    sqrt/div; only libm ulp differences and reduction order remain);
  * matrix-core contractions: 1e-5 (k-ordered fp32 FMA chain vs the oracle's mul+add chain);
  * composed layers (attention block): 2e-5.  The bar named by the task is 1e-4.
"""
import numpy as np
import pytest

import vitlibs as V

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def K():
    from visiontransformer_amd import vitk
    vitk.init(0)
    return vitk


@pytest.fixture
def gemm_path(request, K):
    """Every test_gemm_* runs three times: with small problems routed to the one-wave-per-tile kernel
    (the default: four waves per tile where the contraction is long), with that kernel kept to one
    wave per tile, and with everything on the tiled kernel."""
    K.gemm_small({"small_kernel_on": 1, "small_kernel_one_wave": 2, "tiled_only": 0}[request.param])
    yield request.param
    K.gemm_small(1)


@pytest.fixture
def sdpa_path(request, K):
    """Every test_sdpa_* runs three times: head dimensions 33..64 on the 16-row-tile kernels (the
    default; the backward then gives every workgroup one head at these sizes), on the 32-row-tile
    kernels that serve the other head sizes, and with the 16-row backward dealing the key blocks
    of three heads to a workgroup (the large-batch configuration: rounds that span two heads, dQ
    accumulated across rounds)."""
    K.sdpa_tile16({"tile16_on": 1, "tile32_only": 0, "tile16_bwd_3_heads_per_workgroup": 3}[request.param])
    # (tile32_only also keeps sequences of at most 32 tokens off the one-wave-per-head kernels)
    K.check(K.dll().vitk_sdpa_head_kernels_enable(int(request.param != "tile32_only")))
    yield request.param
    K.sdpa_tile16(True)
    K.check(K.dll().vitk_sdpa_head_kernels_enable(1))


@pytest.fixture
def ln_path(request, K):
    """Every test_layernorm_* runs twice: few rows in one launch (the default) and on the two-stage
    kernels that serve larger inputs."""
    K.check(K.dll().vitk_layernorm_small_enable(int(request.param == "one_launch")))
    yield request.param
    K.check(K.dll().vitk_layernorm_small_enable(1))


def pytest_generate_tests(metafunc):
    name = metafunc.function.__name__
    if name.startswith("test_layernorm_"):
        if "ln_path" not in metafunc.fixturenames:
            metafunc.fixturenames.append("ln_path")
        metafunc.parametrize("ln_path", ["one_launch", "two_stage"], indirect=True)
    if name.startswith("test_gemm_"):
        if "gemm_path" not in metafunc.fixturenames:
            metafunc.fixturenames.append("gemm_path")
        metafunc.parametrize("gemm_path", ["small_kernel_on", "small_kernel_one_wave", "tiled_only"], indirect=True)
    if name.startswith("test_sdpa_") and "row" not in name:
        if "sdpa_path" not in metafunc.fixturenames:
            metafunc.fixturenames.append("sdpa_path")
        metafunc.parametrize("sdpa_path", ["tile16_on", "tile32_only", "tile16_bwd_3_heads_per_workgroup"], indirect=True)


def dev(a):
    a = np.ascontiguousarray(a)
    assert a.dtype in (np.float32, np.int32), a.dtype
    return torch.from_numpy(a).cuda()


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def empty(*shape):
    return torch.empty(shape, dtype=torch.float32, device="cuda")


def close(got, want, tol, floor=None):
    V.assert_close(got, want, tol, "", floor)


# ----------------------------------------------------------------------------------- GEMM
def ref_gemm(a, b, ta, tb):
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    return (a64.T if ta else a64) @ (b64.T if tb else b64)


def test_gemm_layout_identity_with_asymmetric_b(K):
    # A = I, B asymmetric: catches a transposed accumulator map or swapped operands exactly
    n = 96
    a = np.eye(n, dtype=np.float32)
    b = ((np.arange(n * n).reshape(n, n) % 251) - 17 * (np.arange(n)[:, None] % 3)).astype(np.float32)
    c = empty(n, n)
    K.gemm(dev(a), dev(b), c, n, n, n)
    np.testing.assert_array_equal(host(c), b)
    K.gemm(dev(a), dev(b), c, n, n, n, trans_b=True)
    np.testing.assert_array_equal(host(c), b.T)
    K.gemm(dev(b), dev(a), c, n, n, n, trans_a=True)
    np.testing.assert_array_equal(host(c), b.T)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,Kd", [(640, 96, 96), (257, 130, 75), (128, 128, 16), (33, 10, 147), (512, 384, 200),
                                    (1000, 768, 49), (5, 5, 12), (197, 197, 64)])
def test_gemm_shapes(K, ta, tb, M, N, Kd):
    a = V.rng_normal(1, *((Kd, M) if ta else (M, Kd)))
    b = V.rng_normal(2, *((N, Kd) if tb else (Kd, N)))
    c = empty(M, N)
    K.gemm(dev(a), dev(b), c, M, N, Kd, trans_a=bool(ta), trans_b=bool(tb))
    close(host(c), ref_gemm(a, b, ta, tb), 1e-5)


def test_gemm_matches_oracle_matmul(K):
    O = V.oracle()
    a, b = V.rng_normal(3, 300, 192), V.rng_normal(4, 192, 768) * 0.02
    want = V.f32(300, 768)
    O.vo_matmul(a, b, 300, 192, 768, want)
    c = empty(300, 768)
    K.gemm(dev(a), dev(b), c, 300, 768, 192)
    close(host(c), want, 1e-5)


def test_gemm_epilogues(K):
    O = V.oracle()
    M, N, Kd = 200, 160, 64
    a, b = V.rng_normal(5, M, Kd), V.rng_normal(6, Kd, N) * 0.2
    bias, res = V.rng_normal(7, N), V.rng_normal(8, M, N)
    acc = ref_gemm(a, b, 0, 0)
    c = empty(M, N)
    K.gemm(dev(a), dev(b), c, M, N, Kd, bias=dev(bias), residual=dev(res), alpha=0.5)
    close(host(c), 0.5 * acc + bias + res, 1e-5)
    # GELU forward: aux receives the pre-activation, C its GELU
    pre, act = empty(M, N), empty(M, N)
    K.gemm(dev(a), dev(b), act, M, N, Kd, bias=dev(bias), act=K.ACT_GELU_FWD, aux=pre)
    want_pre = (acc + bias).astype(np.float32)
    close(host(pre), want_pre, 1e-5)
    g = V.f32(M, N)
    O.vo_gelu_forward(host(pre), M * N, g)
    close(host(act), g, 2e-6)
    # GELU backward: C = acc * gelu'(aux)
    dx = empty(M, N)
    K.gemm(dev(a), dev(b), dx, M, N, Kd, act=K.ACT_GELU_BWD, aux=pre)
    gd = V.f32(M, N)
    O.vo_gelu_backward(host(pre), acc.astype(np.float32), M * N, gd)
    close(host(dx), gd, 1e-5)
    # pitched operands and output (sub-matrices of larger buffers)
    big_a, big_b, big_c = V.rng_normal(9, M, Kd + 12), V.rng_normal(10, Kd, N + 8), np.zeros((M, N + 4), np.float32)
    dc = dev(big_c)
    K.gemm(dev(big_a), dev(big_b), dc, M, N, Kd, lda=Kd + 12, ldb=N + 8, ldc=N + 4)
    out = host(dc)
    close(out[:, :N], ref_gemm(big_a[:, :Kd], big_b[:, :N], 0, 0), 1e-5)
    assert (out[:, N:] == 0).all()


@pytest.mark.parametrize("split", [0, 2, 7])
def test_gemm_split_k_weight_gradient_shape(K, split):
    # dW = X^T dY with a long contraction (rows) and a small output
    rows, i, o = 5000, 96, 200
    x, dy = V.rng_normal(11, rows, i), V.rng_normal(12, rows, o)
    c = empty(i, o)
    K.gemm(dev(x), dev(dy), c, i, o, rows, trans_a=True, split_k=split)
    close(host(c), ref_gemm(x, dy, 1, 0), 1e-5)
    c2 = empty(i, o)
    K.gemm(dev(x), dev(dy), c2, i, o, rows, trans_a=True, split_k=split)
    np.testing.assert_array_equal(host(c), host(c2))  # fixed reduction order: bitwise reproducible


def test_gemm_batched_strided_heads(K):
    # scores[b,h] = q[b,:,h,:] k[b,:,h,:]^T read straight out of a packed [B*N][3D] buffer
    B, h, N, dh = 3, 4, 17, 8
    D = h * dh
    qkv = V.rng_normal(13, B * N, 3 * D)
    dq = dev(qkv)
    ld = 20
    s = torch.zeros(B * h, N, ld, device="cuda")
    K.gemm(dq, dq[:, D:], s, N, N, dh, trans_b=True, lda=3 * D, ldb=3 * D, ldc=ld, alpha=0.25, batch_outer=B,
           batch_inner=h, stride_a=(N * 3 * D, dh), stride_b=(N * 3 * D, dh), stride_c=(h * N * ld, N * ld))
    q = qkv[:, :D].reshape(B, N, h, dh).transpose(0, 2, 1, 3).astype(np.float64)
    k = qkv[:, D:2 * D].reshape(B, N, h, dh).transpose(0, 2, 1, 3).astype(np.float64)
    want = 0.25 * np.einsum("bhnd,bhmd->bhnm", q, k).reshape(B * h, N, N)
    out = host(s)
    close(out[:, :, :N], want, 1e-5)
    assert (out[:, :, N:] == 0).all()


# ----------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("rows,D", [(50, 96), (640, 64), (333, 768), (77, 192), (19, 1000), (9, 100), (6, 1030), (1024, 96),
                                    (1500, 384)])
def test_layernorm_fwd_bwd_vs_oracle(K, rows, D):
    O = V.oracle()
    g, b = 1 + V.rng_normal(1, D) * 0.1, V.rng_normal(2, D) * 0.1
    x, dy, res = V.rng_normal(3, rows, D) * 2 + 0.3, V.rng_normal(4, rows, D), V.rng_normal(5, rows, D)
    y, xh, inv = V.f32(rows, D), V.f32(rows, D), V.f32(rows)
    dg, db, dx = V.f32(D), V.f32(D), V.f32(rows, D)
    O.vo_layernorm_forward(x, rows, D, g, b, 1e-5, y, xh, inv)
    O.vo_layernorm_backward(dy, xh, inv, g, rows, D, dg, db, dx)
    dX, dG, dB = dev(x), dev(g), dev(b)
    Y, MU, INV = empty(rows, D), empty(rows), empty(rows)
    K.call("vitk_layernorm_fwd", dX, rows, D, D, dG, dB, 1e-5, Y, MU, INV, K.stream())
    close(host(Y), y, 2e-6)
    close(host(INV), inv, 2e-6)
    ws_bytes = K.dll().vitk_layernorm_bwd_workspace_bytes(rows, D)
    ws = empty(ws_bytes // 4)
    DX, DG, DB = empty(rows, D), empty(D), empty(D)
    K.call("vitk_layernorm_bwd", dev(dy), dX, MU, INV, dG, rows, D, D, None, DX, DG, DB, ws, ws_bytes, K.stream())
    close(host(DX), dx, 5e-6)
    close(host(DG), dg, 5e-6)
    close(host(DB), db, 5e-6)
    K.call("vitk_layernorm_bwd", dev(dy), dX, MU, INV, dG, rows, D, D, dev(res), DX, DG, DB, ws, ws_bytes, K.stream())
    close(host(DX), dx + res, 5e-6)


def test_layernorm_golden_and_strided_rows(K):
    g = V.golden("ops.npz")
    c = V.op_case("ln")
    rows, D = c["rows"], c["D"]
    Y, MU, INV = empty(rows, D), empty(rows), empty(rows)
    K.call("vitk_layernorm_fwd", dev(c["x"]), rows, D, D, dev(c["g"]), dev(c["b"]), 1e-5, Y, MU, INV, K.stream())
    close(host(Y), g["ln_y"], 2e-6)
    ws_bytes = K.dll().vitk_layernorm_bwd_workspace_bytes(rows, D)
    ws, DX, DG, DB = empty(ws_bytes // 4), empty(rows, D), empty(1, D), empty(1, D)
    K.call("vitk_layernorm_bwd", dev(c["dy"]), dev(c["x"]), MU, INV, dev(c["g"]), rows, D, D, None, DX, DG, DB, ws,
           ws_bytes, K.stream())
    close(host(DX), g["ln_dx"], 5e-6); close(host(DG), g["ln_dgamma"], 5e-6); close(host(DB), g["ln_dbeta"], 5e-6)
    # rows 3*D apart (token 0 of each image in a [B][3][D] stream)
    x3 = V.rng_normal(9, rows, 3, D)
    y3 = torch.zeros(rows, 3, D, device="cuda")
    K.call("vitk_layernorm_fwd", dev(x3), rows, D, 3 * D, dev(c["g"]), dev(c["b"]), 1e-5, y3, None, None, K.stream())
    O = V.oracle()
    want = V.f32(rows, D)
    O.vo_layernorm_forward(np.ascontiguousarray(x3[:, 0]), rows, D, c["g"], c["b"], 1e-5, want, None, None)
    out = host(y3)
    close(out[:, 0], want, 2e-6)
    assert (out[:, 1:] == 0).all()


def test_gemm_operand_of_more_than_4_gib(K):
    """A k-contiguous operand whose rows span more than 4 GiB (M * lda * 4 bytes > 2^32): the buffer
    descriptors of the unguarded path are 32-bit, so they must cover one tile, not the rest of
    the operand.  Checked on the first and last row tiles against float64."""
    M, Kc, N, lda = (1 << 20) + 128, 64, 128, 1040
    assert M * lda * 4 > 1 << 32
    g = torch.Generator(device="cuda").manual_seed(3)
    A = torch.zeros(M * lda, dtype=torch.float32, device="cuda").view(M, lda)
    A[:256, :Kc] = torch.randn(256, Kc, device="cuda", generator=g)
    A[-256:, :Kc] = torch.randn(256, Kc, device="cuda", generator=g)
    Bm = torch.randn(Kc, N, device="cuda", generator=g)
    Cm = torch.full((M, N), 7.0, device="cuda")
    K.gemm(A, Bm, Cm, M, N, Kc, lda=lda)
    torch.cuda.synchronize()
    for sl in (slice(0, 256), slice(M - 256, M)):
        want = A[sl, :Kc].double() @ Bm.double()
        close(host(Cm[sl]), want.cpu().numpy(), 2e-6)
    assert float(Cm[300:M - 300].abs().max()) == 0.0
    # the same rows as the transposed B operand (C = X A^T): N side of the descriptor
    X = torch.randn(128, Kc, device="cuda", generator=g)
    Ct = torch.empty(128, M, device="cuda")
    K.gemm(X, A, Ct, 128, M, Kc, trans_b=True, ldb=lda)
    torch.cuda.synchronize()
    want = X.double() @ A[M - 256:, :Kc].double().T
    close(host(Ct[:, M - 256:]), want.cpu().numpy(), 2e-6)
    del A, Cm, Ct


# ------------------------------------------------------------------ element-wise kernels
@pytest.mark.parametrize("n", [1, 7, 2048, 100003])
def test_gelu_vs_oracle(K, n):
    O = V.oracle()
    x, dy = V.rng_normal(1, n) * 3, V.rng_normal(2, n)
    y, dx = V.f32(n), V.f32(n)
    O.vo_gelu_forward(x, n, y)
    O.vo_gelu_backward(x, dy, n, dx)
    Y, DX = empty(n), empty(n)
    K.call("vitk_gelu_fwd", dev(x), n, Y, K.stream())
    K.call("vitk_gelu_bwd", dev(x), dev(dy), n, DX, K.stream())
    # the logistic form x / (1 + e^(-2u)) against the reference's 0.5 x (1 + tanh u) in float:
    # 2e-6 of the largest output, and per element 3e-6 (of |value| + rms)
    assert V.max_relerr(host(Y), y) <= 2e-6 and V.max_relerr(host(DX), dx) <= 2e-6
    close(host(Y), y, 3e-6)
    close(host(DX), dx, 3e-6)
    if n == 2048:
        g = V.golden("ops.npz")
        c = V.op_case("gelu")
        K.call("vitk_gelu_fwd", dev(c["x"]), n, Y, K.stream())
        K.call("vitk_gelu_bwd", dev(c["x"]), dev(c["dy"]), n, DX, K.stream())
        close(host(Y), g["gelu_y"], 3e-6); close(host(DX), g["gelu_dx"], 3e-6)


def test_dropout(K):
    n, rate = 1_000_003, 0.2
    x = V.rng_normal(1, n)
    Y, Mk = empty(n), empty(n)
    K.call("vitk_dropout_fwd", dev(x), n, rate, 1234, 0, Y, Mk, K.stream())
    m, y = host(Mk), host(Y)
    scale = np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    assert set(np.unique(m)) == {np.float32(0), scale}
    assert abs((m == 0).mean() - rate) < 4 * np.sqrt(rate * (1 - rate) / n)
    np.testing.assert_array_equal(y, x * m)
    Y2, M2 = empty(n), empty(n)
    K.call("vitk_dropout_fwd", dev(x), n, rate, 1234, 0, Y2, M2, K.stream())
    np.testing.assert_array_equal(host(M2), m)  # same (seed, offset) => same mask
    K.call("vitk_dropout_fwd", dev(x), n, rate, 1234, n, Y2, M2, K.stream())
    assert (host(M2) != m).mean() > 0.2  # different offset => different stream
    dy = V.rng_normal(2, n)
    DX = empty(n)
    K.call("vitk_dropout_bwd", dev(dy), Mk, n, DX, K.stream())
    np.testing.assert_array_equal(host(DX), dy * m)
    # misaligned, odd-length view takes the scalar path and gives the same mask prefix
    Y3, M3 = empty(1001), empty(1001)
    K.call("vitk_dropout_fwd", dev(x)[1:], 1001, rate, 1234, 0, Y3, M3, K.stream())
    np.testing.assert_array_equal(host(M3), m[:1001])
    # skip connection folded in, mask not stored; backward regenerates the factors from the stream
    res = V.rng_normal(3, n)
    K.call("vitk_dropout_residual_fwd", dev(x), dev(res), n, rate, 1234, 0, Y2, None, K.stream())
    np.testing.assert_array_equal(host(Y2), res + x * m)
    K.call("vitk_dropout_residual_fwd", dev(x), None, n, rate, 1234, 0, Y2, None, K.stream())
    np.testing.assert_array_equal(host(Y2), y)
    K.call("vitk_dropout_stream_bwd", dev(dy), n, rate, 1234, 0, DX, K.stream())
    np.testing.assert_array_equal(host(DX), dy * m)
    K.call("vitk_dropout_stream_bwd", dev(dy)[1:], 1001, rate, 1234, 0, Y3, K.stream())  # scalar path
    np.testing.assert_array_equal(host(Y3), dy[1:1002] * m[:1001])


def test_add_scale_fill_copy(K):
    n = 70001
    a, b, c = V.rng_normal(1, n), V.rng_normal(2, n), V.rng_normal(3, n)
    out = empty(n)
    K.call("vitk_add_f32", dev(a), dev(b), n, out, K.stream())
    np.testing.assert_array_equal(host(out), a + b)
    K.call("vitk_add3_f32", dev(a), dev(b), dev(c), n, out, K.stream())
    np.testing.assert_array_equal(host(out), (a + b) + c)
    K.call("vitk_scale_f32", dev(a), 0.125, n, out, K.stream())
    np.testing.assert_array_equal(host(out), a * np.float32(0.125))
    K.call("vitk_fill_f32", out, 3.5, n, K.stream())
    assert (host(out) == 3.5).all()
    src = V.rng_normal(4, 37, 52)
    dst = torch.zeros(37, 44, device="cuda")
    K.call("vitk_copy2d_f32", dev(src), 52, dst, 44, 37, 40, K.stream())
    o = host(dst)
    np.testing.assert_array_equal(o[:, :40], src[:, :40])
    assert (o[:, 40:] == 0).all()
    K.call("vitk_add_f32", dev(a), dev(b), 0, out, K.stream())  # empty input is a no-op


@pytest.mark.parametrize("rows,cols", [(5000, 768), (256, 3840), (1, 10), (37, 1000), (50432, 96)])
def test_colsum(K, rows, cols):
    x = V.rng_normal(1, rows, cols)
    nb = K.dll().vitk_colsum_workspace_bytes(rows, cols)
    ws, out = empty(max(nb // 4, 1)), empty(cols)
    K.call("vitk_colsum_f32", dev(x), rows, cols, cols, out, ws, nb, K.stream())
    want = x.astype(np.float64).sum(0)
    assert np.max(np.abs(host(out) - want)) <= 2e-6 * np.abs(x).astype(np.float64).sum(0).max()


# -------------------------------------------------------------------------------- softmax
@pytest.mark.parametrize("rows,n,ld", [(6 * 17, 17, 17), (40, 197, 200), (13, 5, 8), (3, 1000, 1000)])
def test_softmax_rows(K, rows, n, ld):
    O = V.oracle()
    x, g = V.rng_normal(1, rows, n) * 3, V.rng_normal(2, rows, n)
    p, d = V.f32(rows, n), V.f32(rows, n)
    O.vo_softmax_rows(x, rows, n, p)
    O.vo_softmax_rows_backward(g, p, rows, n, d)
    xp, gp = np.full((rows, ld), 9.0, np.float32), np.full((rows, ld), 9.0, np.float32)
    xp[:, :n], gp[:, :n] = x, g
    P, D = empty(rows, ld), empty(rows, ld)
    K.call("vitk_softmax_rows", dev(xp), rows, n, ld, P, K.stream())
    out = host(P)
    close(out[:, :n], p, 2e-6)
    assert (out[:, n:] == 0).all()
    K.call("vitk_softmax_rows_bwd", dev(gp), P, rows, n, ld, 0.125, D, K.stream())
    close(host(D)[:, :n], d * np.float32(0.125), 5e-6)
    if (rows, n) == (102, 17):
        gd = V.golden("ops.npz")
        c = V.op_case("softmax")
        K.call("vitk_softmax_rows", dev(c["x"]), rows, n, n, P, K.stream())
        close(host(P), gd["softmax_p"].reshape(rows, n), 2e-6)
        K.call("vitk_softmax_rows_bwd", dev(c["g"]), P, rows, n, n, 1.0, D, K.stream())
        close(host(D), gd["softmax_dx"].reshape(rows, n), 5e-6)


@pytest.mark.parametrize("B,Kc", [(37, 10), (256, 1000), (1, 8)])
def test_softmax_ce_vs_oracle(K, B, Kc):
    O = V.oracle()
    lg = V.rng_normal(1, B, Kc) * 2
    lab = np.random.default_rng(2).integers(0, Kc, B)
    y = V.onehot(lab, Kc)
    w = np.abs(V.rng_normal(3, Kc)) + 0.1
    w = (w / w.sum()).astype(np.float32)
    ws = empty(2 * B)
    for ww in (w, None):
        grad = V.f32(B, Kc)
        loss = O.vo_cross_entropy(lg, y, B, Kc, ww, Kc if ww is not None else 0, grad)
        DL, PR, out2 = empty(B, Kc), empty(B, Kc), empty(2)
        K.call("vitk_softmax_ce", dev(lg), dev(y), B, Kc, None if ww is None else dev(ww), float(B), DL, PR, out2, ws,
               2 * B * 4, K.stream())
        o2 = host(out2)
        assert abs(o2[0] - loss) <= 5e-6 * abs(loss)
        close(host(DL), grad, 2e-6)
        assert o2[1] == (lg.argmax(1) == lab).sum()
        # a different gradient batch (data parallel: global batch) only rescales the gradient
        K.call("vitk_softmax_ce", dev(lg), dev(y), B, Kc, None, float(4 * B), DL, None, out2, ws, 2 * B * 4, K.stream())
        close(host(DL) * 4, grad, 2e-6)
    if (B, Kc) == (37, 10):
        g = V.golden("ops.npz")
        c = V.op_case("ce")
        DL, out2 = empty(B, Kc), empty(2)
        K.call("vitk_softmax_ce", dev(c["lg"]), dev(c["y"]), B, Kc, dev(c["w"]), float(B), DL, None, out2, ws, 2 * B * 4,
               K.stream())
        assert abs(host(out2)[0] - g["ce_loss_w"]) <= 5e-6 * abs(g["ce_loss_w"])
        close(host(DL), g["ce_grad"], 2e-6)


def test_argmax_first_maximum(K):
    B, Kc = 300, 1000
    x = V.rng_normal(1, B, Kc)
    x[3, 700] = x[3, 20] = 50.0  # tie: lowest index wins (strict '>' scan of the reference)
    x[4, :] = 1.0
    out = torch.zeros(B, dtype=torch.int32, device="cuda")
    K.call("vitk_argmax_rows", dev(x), B, Kc, out, K.stream())
    o = host(out)
    np.testing.assert_array_equal(o, x.argmax(1))
    assert o[3] == 20 and o[4] == 0


# ----------------------------------------------------------------------------------- Adam
def test_adam_vs_oracle_and_golden(K):
    O = V.oracle()
    n = 100003
    p0, m0, v0 = V.rng_normal(1, n), V.rng_normal(2, n) * 0.01, np.abs(V.rng_normal(3, n)) * 1e-4
    g = V.rng_normal(4, n) * 0.1
    b1t, b2t = float(np.float32(0.9 ** 3)), float(np.float32(0.999 ** 3))
    p, m, v = p0.copy(), m0.copy(), v0.copy()
    O.vo_adam_update(p, g, m, v, n, 3e-4, 0.9, 0.999, b1t, b2t, 1e-8, 1e-4)
    P, M, Vv = dev(p0), dev(m0), dev(v0)
    K.call("vitk_adam_l2", P, dev(g), M, Vv, n, 3e-4, 0.9, 0.999, b1t, b2t, 1e-8, 1e-4, 1.0, K.stream())
    close(host(M), m, 1e-6); close(host(Vv), v, 1e-6)
    assert np.max(np.abs(host(P) - p)) <= 1e-6 * 3e-4 * 10 + 1e-7 * np.abs(p).max()
    gd = V.golden("ops.npz")
    c = V.op_case("adam")
    r, cc = c["r"], c["c"]
    P, M, Vv = dev(c["p"]), torch.zeros(r, cc, device="cuda"), torch.zeros(r, cc, device="cuda")
    for s in range(c["steps"]):
        b1t = float(np.float32(np.float64(np.float32(0.9)) ** (s + 1)))
        b2t = float(np.float32(np.float64(np.float32(0.999)) ** (s + 1)))
        K.call("vitk_adam_l2", P, dev(c["grads"][s]), M, Vv, r * cc, float(c["lrs"][s]), 0.9, 0.999, b1t, b2t, 1e-8,
               1e-4, 1.0, K.stream())
    close(host(P), gd["adam_p"], 2e-6)


# ------------------------------------------------------------------- embeddings plumbing
def test_patch_gather_modes(K):
    O = V.oracle()
    for (B, Cc, HW, P) in [(3, 1, 28, 14), (2, 3, 28, 7), (2, 3, 32, 16), (1, 3, 224, 16)]:
        img = V.rng_normal(1, B, Cc, HW, HW)
        np_ = HW // P
        pd = P * P * Cc
        want = V.f32(B * np_ * np_, pd)
        O.vo_patch_gather(img, B, Cc, HW, P, want)
        out = empty(B * np_ * np_, pd)
        K.call("vitk_patch_gather", dev(img), B, Cc, HW, P, K.PATCH_REFERENCE, out, K.stream())
        np.testing.assert_array_equal(host(out), want)
        K.call("vitk_patch_gather", dev(img), B, Cc, HW, P, K.PATCH_IM2COL, out, K.stream())
        true = img.reshape(B, Cc, np_, P, np_, P).transpose(0, 2, 4, 1, 3, 5).reshape(B * np_ * np_, pd)
        np.testing.assert_array_equal(host(out), true)


def test_embed_assemble_split_cls(K):
    B, Np, D = 5, 16, 48
    N = Np + 1
    proj, cls, pos = V.rng_normal(1, B * Np, D), V.rng_normal(2, D), V.rng_normal(3, N, D)
    out = empty(B, N, D)
    K.call("vitk_embed_assemble", dev(proj), dev(cls), dev(pos), B, Np, D, out, K.stream())
    want = np.concatenate([np.broadcast_to(cls, (B, 1, D)), proj.reshape(B, Np, D)], 1) + pos
    np.testing.assert_array_equal(host(out), want)
    dy = V.rng_normal(4, B, N, D)
    dproj = empty(B * Np, D)
    K.call("vitk_embed_split_bwd", dev(dy), B, Np, D, dproj, K.stream())
    np.testing.assert_array_equal(host(dproj), dy[:, 1:].reshape(B * Np, D))
    rows = empty(B, D)
    K.call("vitk_cls_gather", dev(dy), B, N, D, rows, K.stream())
    np.testing.assert_array_equal(host(rows), dy[:, 0])
    back = empty(B, N, D)
    K.call("vitk_cls_scatter", rows, B, N, D, back, K.stream())
    want = np.zeros_like(dy)
    want[:, 0] = dy[:, 0]
    np.testing.assert_array_equal(host(back), want)


def test_dk_reference_scatter(K):
    B, h, N, dh = 2, 3, 5, 4
    t = V.rng_normal(1, B * h, dh, N)
    out = torch.zeros(B * N, h * dh + 2, device="cuda")
    K.call("vitk_dk_reference_scatter", dev(t), B, h, N, dh, out, h * dh + 2, K.stream())
    want = t.reshape(B, h, N, dh).transpose(0, 2, 1, 3).reshape(B * N, h * dh)  # [dh][N] bytes re-read as [N][dh]
    np.testing.assert_array_equal(host(out)[:, :h * dh], want)


@pytest.mark.parametrize("M,N,Kd,split", [(96, 384, 640, 1), (96, 384, 640, 0), (768, 768, 6400, 0), (75, 130, 257, 1),
                                          (192, 64, 8704, 0), (128, 128, 16, 1), (40, 10, 33, 4),
                                          # the library's own split count in its three regimes: three whole rounds of
                                          # workgroups (144 tiles x 16 splits), more splits than tiles x 64 (9 tiles),
                                          # one workgroup per CU on a short contraction
                                          (768, 3072, 12800, 0), (384, 384, 12800, 0), (192, 192, 8704, 0)])
def test_gemm_fused_column_sums_of_b(K, M, N, Kd, split):
    """dW = X^T dY with db = dY.sum(0) from the same launch (Dense::backward, src/layers/Dense.cpp:90-95)."""
    x, dy = V.rng_normal(5, Kd, M), V.rng_normal(6, Kd, N)
    dw, db = empty(M, N), torch.full((N,), np.nan, device="cuda")
    K.gemm(dev(x), dev(dy), dw, M, N, Kd, trans_a=True, split_k=split, colsum_b=db)
    close(host(dw), ref_gemm(x, dy, 1, 0), 1e-5)
    close(host(db), dy.astype(np.float64).sum(0), 1e-5)
    # same numbers without the fused sum
    dw2 = empty(M, N)
    K.gemm(dev(x), dev(dy), dw2, M, N, Kd, trans_a=True, split_k=split)
    np.testing.assert_array_equal(host(dw), host(dw2))


def test_gemm_column_sums_need_untransposed_b(K):
    a, c, s = empty(8, 8), empty(8, 8), empty(8)
    with pytest.raises(K.VitkError):
        K.gemm(a, a, c, 8, 8, 8, trans_b=True, colsum_b=s)


@pytest.mark.parametrize("ta", [0, 1])
def test_gemm_tail_round_split_keeps_every_epilogue_operand_aligned(K, ta):
    """773 x 1 tiles of 128 = one full round of 768 workgroups + 5: the library runs the last 5 tile
    rows as a second launch.  Bias, pre-activation output, GELU and residual must land on the
    right rows of both parts."""
    M, N, Kd = 773 * 128, 128, 32
    a = V.rng_normal(7, *((Kd, M) if ta else (M, Kd)))
    b, bias, res = V.rng_normal(8, Kd, N), V.rng_normal(9, N), V.rng_normal(10, M, N)
    c, aux = empty(M, N), empty(M, N)
    K.gemm(dev(a), dev(b), c, M, N, Kd, trans_a=bool(ta), bias=dev(bias), act=K.ACT_GELU_FWD, aux=aux,
           residual=dev(res))
    pre = ref_gemm(a, b, ta, 0) + bias
    close(host(aux), pre, 1e-5)
    gelu = 0.5 * pre * (1.0 + np.tanh(0.7978845608028654 * (pre + 0.044715 * pre ** 3)))
    close(host(c), gelu + res, 1e-5)


@pytest.mark.parametrize("ta,rem", [(0, 64), (0, 37), (1, 32), (0, 1)])
def test_gemm_ragged_last_tile_row_on_the_unguarded_path(K, ta, rem):
    """M = 5 x 128 + rem rows (a batch whose token count is not a tile multiple) with everything
    else aligned: the last tile row reads zeros beyond the operand's extent and stores only its
    valid rows; bias, pre-activation output, GELU and residual must follow the rows."""
    M, N, Kd = 5 * 128 + rem, 256, 48
    a = V.rng_normal(17, *((Kd, M) if ta else (M, Kd)))
    b, bias, res = V.rng_normal(18, Kd, N), V.rng_normal(19, N), V.rng_normal(20, M, N)
    c, aux = empty(M, N), empty(M, N)
    K.gemm(dev(a), dev(b), c, M, N, Kd, trans_a=bool(ta), bias=dev(bias), act=K.ACT_GELU_FWD, aux=aux,
           residual=dev(res))
    pre = ref_gemm(a, b, ta, 0) + bias
    close(host(aux), pre, 1e-5)
    gelu = 0.5 * pre * (1.0 + np.tanh(0.7978845608028654 * (pre + 0.044715 * pre ** 3)))
    close(host(c), gelu + res, 1e-5)
    plain = empty(M, N)
    K.gemm(dev(a), dev(b), plain, M, N, Kd, trans_a=bool(ta))
    close(host(plain), ref_gemm(a, b, ta, 0), 1e-5)

@pytest.mark.parametrize("Kd,split", [(16 * 7 + 5, 1), (4096 + 37, 0), (16 * 40 + 1, 3)])
def test_gemm_ragged_contraction_of_a_weight_gradient(K, Kd, split):
    """dW = X^T dY with a token count that is not a multiple of the slab depth (batch 100 x 197):
    the k beyond the end read as zeros through the descriptor extents.  The operands sit at the end
    of buffers whose remainder is NaN, so an unclipped read would poison the result."""
    M, N = 256, 384
    x, dy = V.rng_normal(31, Kd, M), V.rng_normal(32, Kd, N)
    pad = 64 * max(M, N)
    bx = torch.full((Kd * M + pad,), float("nan"), device="cuda")
    bdy = torch.full((Kd * N + pad,), float("nan"), device="cuda")
    bx[:Kd * M] = dev(x).reshape(-1)
    bdy[:Kd * N] = dev(dy).reshape(-1)
    dw, bias = empty(M, N), empty(N)
    K.gemm(bx, bdy, dw, M, N, Kd, trans_a=True, split_k=split, colsum_b=bias)
    close(host(dw), x.astype(np.float64).T @ dy.astype(np.float64), 1e-5)
    close(host(bias), dy.astype(np.float64).sum(0), 1e-5)

@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_every_slab_count_through_prologue_unrolled_loop_and_tail(K, ta, tb):
    """K = 16 .. 25 slabs of 16 (+ a ragged one where the layout allows): the unguarded kernel runs
    a straight-line prologue, a loop unrolled six slabs deep and a guarded tail; every split of a
    contraction between the three must give the same product."""
    M, N = 256, 128
    for nslab in list(range(1, 26)) + [31, 48]:
        for extra in ((0, 5) if (ta and not tb) else (0,)):
            Kd = 16 * nslab + extra
            a = V.rng_normal(100 + nslab, *((Kd, M) if ta else (M, Kd)))
            b = V.rng_normal(200 + nslab, *((N, Kd) if tb else (Kd, N)))
            c = empty(M, N)
            K.gemm(dev(a), dev(b), c, M, N, Kd, trans_a=bool(ta), trans_b=bool(tb))
            close(host(c), ref_gemm(a, b, ta, tb), 1e-5)

@pytest.mark.parametrize("tile", ["128x64", "64x128"])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_rectangular_tiles(K, tile, ta, tb):
    """The 128 x 64 and 64 x 128 output tiles (unguarded kernels), forced: ragged last tile row, short
    and long contractions, every epilogue operand, and -- for weight gradients -- split-K with the
    fused column sums and a ragged contraction."""
    for M, N, Kd in [(5 * 128 + 36, 384, 16 * 25), (256, 128, 16 * 3), (192, 768, 16 * 9)]:
        a = V.rng_normal(41, *((Kd, M) if ta else (M, Kd)))
        b, bias, res = V.rng_normal(42, *((N, Kd) if tb else (Kd, N))), V.rng_normal(43, N), V.rng_normal(44, M, N)
        c, aux = empty(M, N), empty(M, N)
        K.gemm_profile(True)
        K.gemm(dev(a), dev(b), c, M, N, Kd, trans_a=bool(ta), trans_b=bool(tb), bias=dev(bias), act=K.ACT_GELU_FWD,
               aux=aux, residual=dev(res), tile=tile)
        K.gemm_profile(False)
        names = [r["name"] for r in K.gemm_profile_read()]
        assert names == [f"gemm_kernel<{tile.replace('x', ', ')}, {'true' if ta else 'false'}, {'true' if tb else 'false'}, true, false>"], names
        pre = ref_gemm(a, b, ta, tb) + bias
        close(host(aux), pre, 1e-5)
        gelu = 0.5 * pre * (1.0 + np.tanh(0.7978845608028654 * (pre + 0.044715 * pre ** 3)))
        close(host(c), gelu + res, 1e-5)
    if ta and not tb:
        for M, N, Kd, split in [(192, 768, 4096 + 37, 0), (384, 128, 16 * 40 + 1, 3), (128, 64, 300, 1)]:
            x, dy = V.rng_normal(45, Kd, M), V.rng_normal(46, Kd, N)
            dw, db = empty(M, N), torch.full((N,), np.nan, device="cuda")
            K.gemm(dev(x), dev(dy), dw, M, N, Kd, trans_a=True, split_k=split, colsum_b=db, tile=tile)
            close(host(dw), ref_gemm(x, dy, 1, 0), 1e-5)
            close(host(db), dy.astype(np.float64).sum(0), 1e-5)


@pytest.mark.parametrize("rows,i,o,gelu", [(640, 96, 384, False), (640, 384, 96, True), (85, 64, 64, False), (320, 96, 288, False),
                                            (1300, 96, 96, False), (50, 10, 64, False), (9000, 192, 192, False)])
def test_gemm_pair_of_a_dense_backward_equals_the_two_calls(K, gemm_path, rows, i, o, gelu):
    """vitk_gemm_pair_f32 (dX = dY W^T with its epilogue, dW = X^T dY with db): bitwise the results
    of two vitk_gemm_f32 calls, whether it ran as one launch of the small-problem kernel (the first
    shapes: the 5- and 17-token models' layers) or fell back to two (contractions over more than
    1024 rows, operands that cannot be read 16 bytes at a time, large outputs)."""
    x, dy, w = V.rng_normal(51, rows, i), V.rng_normal(52, rows, o), V.rng_normal(53, i, o) * 0.1
    pre = V.rng_normal(54, rows, i)
    dx, dw, db = empty(rows, i), empty(i, o), empty(o)
    dx2, dw2, db2 = empty(rows, i), empty(i, o), empty(o)
    dX = dict(A=dev(dy), B=dev(w), Cout=dx, M=rows, N=i, K=o, trans_b=True)
    if gelu:
        dX.update(act=K.ACT_GELU_BWD, aux=dev(pre))
    dW = dict(A=dev(x), B=dev(dy), Cout=dw, M=i, N=o, K=rows, trans_a=True, split_k=0, colsum_b=db)
    K.gemm_profile(True)
    K.gemm_pair(dX, dW)
    K.gemm_profile(False)
    names = [r["name"] for r in K.gemm_profile_read()]
    K.gemm(**dict(dX, Cout=dx2))
    K.gemm(**dict(dW, Cout=dw2, colsum_b=db2))
    np.testing.assert_array_equal(host(dx), host(dx2))
    np.testing.assert_array_equal(host(dw), host(dw2))
    np.testing.assert_array_equal(host(db), host(db2))
    close(host(dw), ref_gemm(x, dy, 1, 0), 1e-5)
    want_dx = ref_gemm(dy, w, 0, 1)
    if gelu:
        O = V.oracle()
        g = V.f32(rows, i)
        O.vo_gelu_backward(pre, want_dx.astype(np.float32), rows * i, g)
        want_dx = g
    close(host(dx), want_dx, 1e-5)
    one_launch = rows <= 1024 and o % 4 == 0 and (rows + 15) // 16 * ((i + 15) // 16) <= 8 * 256 and gemm_path != "tiled_only"
    assert (names == ["gemm_small_pair_kernel (dX NT + dW TN)"]) == one_launch, (names, one_launch)


def test_gemm_plan_takes_the_rectangular_tile_for_short_contractions_into_wide_outputs(K):
    """ViT-S's K = 384 into N = 1152: the plan launches 64 x 128 tiles; same product as the square ones."""
    M, N, Kd = 4096 + 197, 1152, 384
    a, b = V.rng_normal(47, M, Kd), V.rng_normal(48, Kd, N)
    c, c64 = empty(M, N), empty(M, N)
    K.gemm_profile(True)
    K.gemm(dev(a), dev(b), c, M, N, Kd)
    K.gemm_profile(False)
    assert [r["name"] for r in K.gemm_profile_read()] == ["gemm_kernel<64, 128, false, false, true, false>"]
    close(host(c), ref_gemm(a, b, 0, 0), 1e-5)
    K.gemm(dev(a), dev(b), c64, M, N, Kd, tile="64x64")
    np.testing.assert_array_equal(host(c), host(c64))  # the k order of a product does not depend on the tile


@pytest.mark.parametrize("B,N,D", [(5, 7, 12), (3, 197, 64), (4, 5, 10)])
def test_dropout_rows_takes_the_factors_of_the_dense_call(K, B, N, D):
    """vitk_dropout_rows on the first token of every sequence = those rows of vitk_dropout_residual_fwd
    over the whole {B,N,D} tensor at the same stream position, with and without a recorded mask."""
    rate, seed, off = 0.3, 0x1234567, 77
    x, res = V.rng_normal(1, B * N * D), V.rng_normal(2, B * N * D)
    dense, mask = empty(B * N * D), empty(B * N * D)
    K.call("vitk_dropout_residual_fwd", dev(x), dev(res), B * N * D, rate, seed, off, dense, mask, K.stream())
    want = host(dense).reshape(B, N * D)[:, :D]
    x0 = np.ascontiguousarray(x.reshape(B, N * D)[:, :D])
    r0 = np.ascontiguousarray(res.reshape(B, N * D)[:, :D])
    got = empty(B, D)
    K.call("vitk_dropout_rows", dev(x0), dev(r0), None, B, D, N * D, rate, seed, off, got, K.stream())
    np.testing.assert_array_equal(host(got), want)
    K.call("vitk_dropout_rows", dev(x0), dev(r0), mask, B, D, N * D, rate, 0, 0, got, K.stream())
    np.testing.assert_array_equal(host(got), want)
    # backward form: no residual
    K.call("vitk_dropout_rows", dev(x0), None, None, B, D, N * D, rate, seed, off, got, K.stream())
    np.testing.assert_array_equal(host(got), x0 * host(mask).reshape(B, N * D)[:, :D])


def test_gather_rows_assembles_a_shuffled_batch(K):
    src = V.rng_normal(3, 50, 37)
    idx = np.random.default_rng(4).permutation(50)[:21].astype(np.int32)
    out = empty(21, 37)
    K.call("vitk_gather_rows_f32", dev(src), 50, dev(idx), 21, 37, out, K.stream())
    np.testing.assert_array_equal(host(out), src[idx])
    K.call("vitk_gather_rows_f32", dev(src), 50, dev(idx), 0, 37, out, K.stream())  # empty batch: no-op


# ---------------------------------------- data augmentation (vitk_augment_f32)
def aug_records(draws):
    """Oracle draws (flags, angle, tx, ty, zoom, 0) -> the kernel's 32-byte records; cos/sin from
    the C library's cosf/sinf, which is what std::cos(float) calls."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.cosf.restype = libm.sinf.restype = ctypes.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [ctypes.c_float]
    rec = np.zeros((len(draws), 8), np.int32)
    f = rec.view(np.float32)
    for i, (flags, angle, tx, ty, zoom, _) in enumerate(draws):
        rad = np.float32(np.float64(np.float32(angle)) * np.pi / np.float64(np.float32(180.0)))
        rec[i, 0] = int(flags)
        f[i, 1], f[i, 2] = libm.cosf(float(rad)), libm.sinf(float(rad))
        rec[i, 3], rec[i, 4] = int(tx), int(ty)  # truncation toward zero, as static_cast<int>
        f[i, 5] = zoom
    return rec


@pytest.mark.parametrize("name", list(V.AUGMENT_CASES))
def test_augment_kernel_is_the_three_stage_resampling_bit_for_bit(K, name):
    """One gather per output pixel == rotation, translation and zoom applied one after another
    (oracle: vo_augment, pinned to the reference's DataAugmentation); also folded with a batch
    gather from a larger data set."""
    O = V.oracle()
    c = V.augment_case(name)
    B, Cc, Hh, W = c["B"], c["C"], c["H"], c["W"]
    want, draws = V.f32(B, Cc, Hh, W), V.f32(B, 6)
    O.vo_augment(c["x"], B, Cc, Hh, W, c["seed"], c["cfg"], want, draws)
    rec = dev(aug_records(draws).view(np.float32))
    out = empty(B, Cc, Hh, W)
    K.call("vitk_augment_f32", dev(c["x"]), B, None, rec, B, Cc, Hh, W, out, K.stream())
    np.testing.assert_array_equal(host(out), want)
    # folded gather: images of the batch are rows `idx` of a larger set
    big = V.rng_normal(77, 3 * B, Cc, Hh, W)
    idx = np.random.default_rng(5).permutation(3 * B)[:B].astype(np.int32)
    O.vo_augment(np.ascontiguousarray(big[idx]), B, Cc, Hh, W, c["seed"], c["cfg"], want, draws)
    K.call("vitk_augment_f32", dev(big), 3 * B, dev(idx), rec, B, Cc, Hh, W, out, K.stream())
    np.testing.assert_array_equal(host(out), want)
    K.call("vitk_augment_f32", dev(big), 3 * B, dev(idx), rec, 0, Cc, Hh, W, out, K.stream())  # empty batch


def test_augment_kernel_full_size_properties(K):
    """224x224x3 images, batch 256 (beyond what the oracle finishes quickly): no stage selected is
    the identity; a pure integer translation is an exact shift with zero fill."""
    B, Cc, S = 256, 3, 224
    x = torch.randn(B, Cc, S, S, device="cuda")
    rec = np.zeros((B, 8), np.int32)
    rec.view(np.float32)[:, 5] = 1.0
    out = torch.empty_like(x)
    K.call("vitk_augment_f32", x, B, None, dev(rec.view(np.float32)), B, Cc, S, S, out, K.stream())
    assert torch.equal(out, x)
    rec[:, 0] = 2
    rec[:, 3], rec[:, 4] = 5, -7  # tx, ty
    K.call("vitk_augment_f32", x, B, None, dev(rec.view(np.float32)), B, Cc, S, S, out, K.stream())
    want = torch.zeros_like(x)
    want[:, :, :S - 7, 5:] = x[:, :, 7:, :S - 5]  # out(h, w) = x(h + 7, w - 5)
    assert torch.equal(out, want)


# ---------------------------------------- fused attention (vitk_sdpa_fwd / vitk_sdpa_bwd)
def sdpa_numpy(q, k, v, do, scale):
    """float64 attention per (batch, head): q, k, v, do are [B][H][N][dh]."""
    s = np.einsum("bhqd,bhkd->bhqk", q, k) * scale
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    l = e.sum(-1, keepdims=True)
    p = e / l
    o = np.einsum("bhqk,bhkd->bhqd", p, v)
    dp = np.einsum("bhqd,bhkd->bhqk", do, v)
    ds = p * (dp - (dp * p).sum(-1, keepdims=True)) * scale
    dq = np.einsum("bhqk,bhkd->bhqd", ds, k)
    dk = np.einsum("bhqk,bhqd->bhkd", ds, q)
    dv = np.einsum("bhqk,bhqd->bhkd", p, do)
    return o, m[..., 0], l[..., 0], dq, dk, dv


def heads_of(x, B, N, H, dh, col0):
    """[B*N][ld] packed rows -> [B][H][N][dh] starting at column col0."""
    return x[:, col0:col0 + H * dh].reshape(B, N, H, dh).transpose(0, 2, 1, 3)


@pytest.mark.parametrize("B,H,N,dh,pad", [(2, 3, 197, 64, 0), (3, 8, 5, 12, 0), (2, 6, 17, 32, 0), (1, 2, 256, 64, 0),
                                          (2, 2, 33, 20, 1), (1, 1, 1, 4, 0), (2, 4, 50, 16, 0), (1, 3, 129, 48, 3),
                                          (2, 2, 64, 7, 0), (2, 2, 100, 48, 0), (1, 4, 33, 64, 0), (1, 1, 1, 64, 0),
                                          (3, 2, 224, 40, 0), (2, 7, 197, 64, 0), (5, 1, 49, 36, 0),
                                          # one wave per head: a full tile, a ragged head count, tiny heads
                                          (5, 3, 32, 24, 0), (7, 5, 31, 8, 0), (1, 2, 2, 32, 0)])
@pytest.mark.parametrize("dk_mode", ["reference", "exact"])
def test_sdpa_fused_matches_float64(K, B, H, N, dh, pad, dk_mode):
    D, M = H * dh, B * N
    ld, ldo = 3 * D + pad, D + pad  # pad != 0: pitches that rule out 16-byte accesses
    qkv = np.zeros((M, ld), np.float32)
    qkv[:, :3 * D] = V.rng_normal(11, M, 3 * D)
    do = np.zeros((M, ldo), np.float32)
    do[:, :D] = V.rng_normal(12, M, D)
    scale = float(np.float32(1.0) / np.sqrt(np.float32(dh)))
    q64, k64, v64 = (heads_of(qkv.astype(np.float64), B, N, H, dh, i * D) for i in range(3))
    o, m, l, dq, dk, dv = sdpa_numpy(q64, k64, v64, heads_of(do.astype(np.float64), B, N, H, dh, 0), scale)

    assert K.dll().vitk_sdpa_supported(N, dh) == 1
    dqkv_t = dev(qkv)
    out, stats = torch.zeros(M, ldo, device="cuda"), empty(B * H, N, 2)
    K.call("vitk_sdpa_fwd", dqkv_t, dqkv_t[:, D:], dqkv_t[:, 2 * D:], ld, B, H, N, dh, scale, out, ldo, stats, K.stream())
    got_o = host(out)
    close(heads_of(got_o, B, N, H, dh, 0), o, 1e-5)
    assert not got_o[:, D:].any()  # nothing written past the head columns
    st = host(stats)
    close(st[..., 0].reshape(B, H, N), m, 1e-5, floor=1.0)  # scores are O(1) sums of dh products
    close(st[..., 1].reshape(B, H, N), l, 1e-5)

    mode = K.DK_REFERENCE if dk_mode == "reference" else K.DK_EXACT
    nb = K.dll().vitk_sdpa_bwd_workspace_bytes(B, H, N)
    ws, grads = empty(nb // 4), torch.zeros(M, ld, device="cuda")
    K.call("vitk_sdpa_bwd", dqkv_t, dqkv_t[:, D:], dqkv_t[:, 2 * D:], ld, out, dev(do), ldo, stats, B, H, N, dh, scale,
           mode, grads, grads[:, D:], grads[:, 2 * D:], ld, ws, nb, K.stream())
    g = host(grads)
    close(heads_of(g, B, N, H, dh, 0), dq, 1e-5)
    close(heads_of(g, B, N, H, dh, 2 * D), dv, 1e-5)
    if dk_mode == "exact":
        want_dk = dk
    else:  # the reference's [dh][N] buffer re-read as [N][dh] (src/layers/MultiHeadAttention.cpp:207-222)
        want_dk = dk.transpose(0, 1, 3, 2).reshape(B, H, N, dh)
    close(heads_of(g, B, N, H, dh, D), want_dk, 1e-5)
    assert not g[:, 3 * D:].any()


@pytest.mark.parametrize("B,H,N,dh,pad", [(3, 12, 197, 64, 0), (4, 8, 5, 12, 0), (2, 6, 17, 32, 0), (1, 2, 256, 64, 0),
                                          (2, 2, 33, 20, 1), (1, 1, 1, 4, 0), (2, 3, 100, 7, 0)])
@pytest.mark.parametrize("dk_mode", ["reference", "exact"])
def test_sdpa_single_query_row_matches_float64(K, B, H, N, dh, pad, dk_mode):
    """vitk_sdpa_row_fwd / _bwd: attention of the first token only (what the last encoder block needs),
    against the float64 attention with an output gradient that is zero for every other token."""
    D, M = H * dh, B * N
    ld = 3 * D + pad
    qkv = np.zeros((M, ld), np.float32)
    qkv[:, :3 * D] = V.rng_normal(21, M, 3 * D)
    do_row = V.rng_normal(22, B, D)
    do_full = np.zeros((M, D))
    do_full[::N] = do_row
    scale = float(np.float32(1.0) / np.sqrt(np.float32(dh)))
    q64, k64, v64 = (heads_of(qkv.astype(np.float64), B, N, H, dh, i * D) for i in range(3))
    o, m, l, dq, dk, dv = sdpa_numpy(q64, k64, v64, heads_of(do_full, B, N, H, dh, 0), scale)
    assert K.dll().vitk_sdpa_row_supported(N, dh) == 1
    t = dev(qkv)
    qrow = t[::N, :D].contiguous()  # [B][D]: the query token of every batch entry
    out, stats = empty(B, D), empty(B * H, 2)
    K.call("vitk_sdpa_row_fwd", qrow, D, t[:, D:], t[:, 2 * D:], ld, B, H, N, dh, scale, out, D, stats, K.stream())
    close(host(out).reshape(B, H, dh), o[:, :, 0], 1e-5)
    st = host(stats).reshape(B, H, 2)
    close(st[..., 0], m[:, :, 0], 1e-5, floor=1.0)
    close(st[..., 1], l[:, :, 0], 1e-5)
    mode = K.DK_REFERENCE if dk_mode == "reference" else K.DK_EXACT
    grads, dqrow = torch.zeros(M, ld, device="cuda"), empty(B, D)
    K.call("vitk_sdpa_row_bwd", qrow, D, t[:, D:], t[:, 2 * D:], ld, out, dev(do_row), D, stats, B, H, N, dh, scale, mode,
           dqrow, D, grads[:, D:], grads[:, 2 * D:], ld, K.stream())
    g = host(grads)
    close(host(dqrow).reshape(B, H, dh), dq[:, :, 0], 1e-5)
    assert np.abs(dq[:, :, 1:]).max() == 0 if N > 1 else True  # the other query rows get no gradient
    close(heads_of(g, B, N, H, dh, 2 * D), dv, 1e-5)
    want_dk = dk if dk_mode == "exact" else dk.transpose(0, 1, 3, 2).reshape(B, H, N, dh)
    close(heads_of(g, B, N, H, dh, D), want_dk, 1e-5)
    assert not g[:, :D].any() and not g[:, 3 * D:].any()


def test_sdpa_rejects_unsupported_shapes(K):
    assert K.dll().vitk_sdpa_supported(257, 64) == 0 and K.dll().vitk_sdpa_supported(197, 65) == 0
    x = empty(4, 4)
    with pytest.raises(K.VitkError):
        K.call("vitk_sdpa_fwd", x, x, x, 4, 1, 1, 300, 4, 1.0, x, 4, None, K.stream())


# ---------------------------------------- attention block composed from the C-ABI calls
def mha_via_abi(K, c, dk_mode, fused=False):
    """q/k/v projection (packed [D][3D] weights), scores, softmax, PV, output projection and the
    whole backward, exactly as the host layer sequences the kernels."""
    D, h, B, N = c["D"], c["h"], c["B"], c["N"]
    dh, M, BH = D // h, B * N, B * h
    prm = c["prm"]
    blk = D * D + D
    Wq, bq = prm[:D * D].reshape(D, D), prm[D * D:blk]
    Wk, bk = prm[blk:blk + D * D].reshape(D, D), prm[blk + D * D:2 * blk]
    Wv, bv = prm[2 * blk:2 * blk + D * D].reshape(D, D), prm[2 * blk + D * D:3 * blk]
    Wo, bo = prm[3 * blk:3 * blk + D * D].reshape(D, D), prm[3 * blk + D * D:]
    Wqkv, bqkv = dev(np.concatenate([Wq, Wk, Wv], 1)), dev(np.concatenate([bq, bk, bv]))
    x, dy = dev(c["x"].reshape(M, D)), dev(c["dy"].reshape(M, D))
    qkv = empty(M, 3 * D)
    K.gemm(x, Wqkv, qkv, M, 3 * D, D, bias=bqkv)
    ld = (N + 3) // 4 * 4
    S = torch.zeros(BH, N, ld, device="cuda")
    scale = float(np.float32(1.0) / np.sqrt(np.float32(dh)))
    hs = dict(batch_outer=B, batch_inner=h)
    tok = (N * 3 * D, dh)
    ctx = empty(M, D)
    if fused:
        stats = empty(BH, N, 2)
        K.call("vitk_sdpa_fwd", qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B, h, N, dh, scale, ctx, D, stats, K.stream())
    else:
        K.gemm(qkv, qkv[:, D:], S, N, N, dh, trans_b=True, lda=3 * D, ldb=3 * D, ldc=ld, alpha=scale, stride_a=tok,
               stride_b=tok, stride_c=(h * N * ld, N * ld), **hs)
        K.call("vitk_softmax_rows", S, BH * N, N, ld, S, K.stream())
        K.gemm(S, qkv[:, 2 * D:], ctx, N, dh, N, lda=ld, ldb=3 * D, ldc=D, stride_a=(h * N * ld, N * ld),
               stride_b=tok, stride_c=(N * D, dh), **hs)
    y = empty(M, D)
    K.gemm(ctx, dev(Wo), y, M, D, D, bias=dev(bo))
    # backward
    dWo, dctx = empty(D, D), empty(M, D)
    K.gemm(ctx, dy, dWo, D, D, M, trans_a=True, split_k=0)
    K.gemm(dy, dev(Wo), dctx, M, D, D, trans_b=True)
    dqkv = torch.zeros(M, 3 * D, device="cuda")
    dP = torch.zeros(BH, N, ld, device="cuda")
    if fused:
        nb = K.dll().vitk_sdpa_bwd_workspace_bytes(B, h, N)
        K.call("vitk_sdpa_bwd", qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, ctx, dctx, D, stats, B, h, N, dh, scale, dk_mode,
               dqkv, dqkv[:, D:], dqkv[:, 2 * D:], 3 * D, empty(nb // 4), nb, K.stream())
        return mha_tail(K, c, x, dy, dqkv, Wqkv, dWo, y)
    K.gemm(dctx, qkv[:, 2 * D:], dP, N, N, dh, trans_b=True, lda=D, ldb=3 * D, ldc=ld, stride_a=(N * D, dh),
           stride_b=tok, stride_c=(h * N * ld, N * ld), **hs)
    K.gemm(S, dctx, dqkv[:, 2 * D:], N, dh, N, trans_a=True, lda=ld, ldb=D, ldc=3 * D, stride_a=(h * N * ld, N * ld),
           stride_b=(N * D, dh), stride_c=tok, **hs)  # dV = P^T dctx
    K.call("vitk_softmax_rows_bwd", dP, S, BH * N, N, ld, scale, dP, K.stream())  # dS in place
    K.gemm(dP, qkv[:, D:], dqkv, N, dh, N, lda=ld, ldb=3 * D, ldc=3 * D, stride_a=(h * N * ld, N * ld), stride_b=tok,
           stride_c=tok, **hs)  # dQ = dS K
    if dk_mode == K.DK_EXACT:
        K.gemm(dP, qkv, dqkv[:, D:], N, dh, N, trans_a=True, lda=ld, ldb=3 * D, ldc=3 * D,
               stride_a=(h * N * ld, N * ld), stride_b=tok, stride_c=tok, **hs)  # dK = dS^T Q
    else:
        T = empty(BH, dh, N)
        K.gemm(qkv, dP, T, dh, N, N, trans_a=True, lda=3 * D, ldb=ld, ldc=N, stride_a=tok,
               stride_b=(h * N * ld, N * ld), stride_c=(h * dh * N, dh * N), **hs)  # Q^T dS
        K.call("vitk_dk_reference_scatter", T, B, h, N, dh, dqkv[:, D:], 3 * D, K.stream())
    return mha_tail(K, c, x, dy, dqkv, Wqkv, dWo, y)


def mha_tail(K, c, x, dy, dqkv, Wqkv, dWo, y):
    """Packed projection backward shared by the fused and the unfused attention paths."""
    D, B, N = c["D"], c["B"], c["N"]
    M = B * N
    dWqkv, dx = empty(D, 3 * D), empty(M, D)
    K.gemm(x, dqkv, dWqkv, D, 3 * D, M, trans_a=True, split_k=0)
    K.gemm(dqkv, Wqkv, dx, M, D, 3 * D, trans_b=True)
    nb = K.dll().vitk_colsum_workspace_bytes(M, 3 * D)
    ws, dbqkv, dbo = empty(max(nb // 4, 1)), empty(3 * D), empty(D)
    K.call("vitk_colsum_f32", dqkv, M, 3 * D, 3 * D, dbqkv, ws, nb, K.stream())
    K.call("vitk_colsum_f32", dy, M, D, D, dbo, ws, nb, K.stream())
    gW, gb = host(dWqkv), host(dbqkv)
    grads = np.concatenate([gW[:, :D].ravel(), gb[:D], gW[:, D:2 * D].ravel(), gb[D:2 * D], gW[:, 2 * D:].ravel(),
                            gb[2 * D:], host(dWo).ravel(), host(dbo)])
    return host(y).reshape(B, N, D), grads, host(dx).reshape(B, N, D)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("tag", ["mha_a", "mha_b"])
def test_attention_block_matches_reference_golden(K, tag, fused):
    g = V.golden("ops.npz")
    c = V.op_case(tag)
    y, grads, dx = mha_via_abi(K, c, K.DK_REFERENCE, fused)
    close(y, g[f"{tag}_y"], 2e-5)
    close(dx, g[f"{tag}_dx"], 2e-5)
    D = c["D"]
    blk = D * D + D
    for i, name in enumerate(("q", "k", "v", "out")):
        close(grads[i * blk:(i + 1) * blk], g[f"{tag}_g"][i * blk:(i + 1) * blk], 2e-5)


def test_attention_exact_key_gradient_differs_only_in_k(K):
    c = V.op_case("mha_b")
    y0, g0, dx0 = mha_via_abi(K, c, K.DK_REFERENCE)
    y1, g1, dx1 = mha_via_abi(K, c, K.DK_EXACT)
    D = c["D"]
    blk = D * D + D
    np.testing.assert_array_equal(y0, y1)
    np.testing.assert_array_equal(g0[:blk], g1[:blk])          # q
    np.testing.assert_array_equal(g0[2 * blk:], g1[2 * blk:])  # v, out
    assert V.max_relerr(g0[blk:2 * blk], g1[blk:2 * blk]) > 1e-2  # the reference's dK is a permutation
    # exact mode agrees with a float64 autograd of the same attention
    xt = torch.tensor(c["x"], dtype=torch.float64, requires_grad=True)
    prm = torch.tensor(c["prm"], dtype=torch.float64)
    Wk = prm[blk:blk + D * D].reshape(D, D).clone().requires_grad_(True)
    Wq, bq = prm[:D * D].reshape(D, D), prm[D * D:blk]
    bk = prm[blk + D * D:2 * blk]
    Wv, bv = prm[2 * blk:2 * blk + D * D].reshape(D, D), prm[2 * blk + D * D:3 * blk]
    Wo, bo = prm[3 * blk:3 * blk + D * D].reshape(D, D), prm[3 * blk + D * D:]
    B, N, h = c["B"], c["N"], c["h"]
    dh = D // h
    sp = lambda t: t.reshape(B, N, h, dh).transpose(1, 2)
    q, k, v = sp(xt @ Wq + bq), sp(xt @ Wk + bk), sp(xt @ Wv + bv)
    p = torch.softmax(q @ k.transpose(-1, -2) / np.sqrt(dh), -1)
    out = (p @ v).transpose(1, 2).reshape(B, N, D) @ Wo + bo
    out.backward(torch.tensor(c["dy"], dtype=torch.float64))
    close(g1[blk:blk + D * D], Wk.grad.numpy().ravel(), 2e-5)
    close(dx1, xt.grad.numpy(), 2e-5)
