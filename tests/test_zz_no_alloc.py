"""The entry point that allocates nothing: vitk_layernorm_bwd_static keeps its partial sums in an
array built into the library's device code.  It must give exactly what vitk_layernorm_bwd gives
with a caller-provided workspace, call after call, including a smaller call after a larger one
(the array keeps the previous call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run_both(K, torch, rows, D, stride, seed, with_residual):
    g = torch.Generator(device="cuda").manual_seed(seed)
    dy = torch.randn(rows, stride, device="cuda", generator=g)
    x = torch.randn(rows, stride, device="cuda", generator=g) * 2 + 0.3
    gamma = 1 + 0.1 * torch.randn(D, device="cuda", generator=g)
    beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    res = torch.randn(rows, stride, device="cuda", generator=g) if with_residual else None
    y = torch.empty_like(x)
    mean = torch.empty(rows, device="cuda")
    inv = torch.empty(rows, device="cuda")
    st = K.stream()
    K.call("vitk_layernorm_fwd", x, rows, D, stride, gamma, beta, 1e-5, y, mean, inv, st)
    need = K.dll().vitk_layernorm_bwd_workspace_bytes(rows, D)
    ws = torch.empty(max(need // 4, 1), device="cuda")
    out = []
    for static in (False, True):
        dx = torch.full((rows, stride), 7.0, device="cuda")
        dg = torch.full((D,), 7.0, device="cuda")
        db = torch.full((D,), 7.0, device="cuda")
        if static:
            K.call("vitk_layernorm_bwd_static", dy, x, mean, inv, gamma, rows, D, stride, res, dx, dg, db, st)
        else:
            K.call("vitk_layernorm_bwd", dy, x, mean, inv, gamma, rows, D, stride, res, dx, dg, db, ws, need, st)
        torch.cuda.synchronize()
        out.append((dx[:, :D].cpu().numpy(), dg.cpu().numpy(), db.cpu().numpy()))
    return out


def test_static_scratch_layernorm_backward_matches_the_workspace_entry_point():
    import torch
    from visiontransformer_amd import vitk as K
    K.init(0)
    assert K.dll().vitk_layernorm_bwd_static_max_d() == 1024
    # a large call first (1024 workgroups of partials, D = 768), then smaller ones that leave most
    # of the array holding stale partials (2000 and 1100 rows; the calls of at most 1024 rows take
    # the one-launch path that needs no partials at all), then the generic (unaligned D) path
    for rows, D, stride, res in ((50432, 768, 768, True), (2000, 192, 192, True), (640, 96, 96, False),
                                 (1100, 96, 100, False), (37, 100, 104, True), (5, 30, 30, False),
                                 (4100, 1024, 1024, True)):
        (dx0, dg0, db0), (dx1, dg1, db1) = _run_both(K, torch, rows, D, stride, 11 + rows, res)
        assert np.array_equal(dx0, dx1), (rows, D)
        assert np.array_equal(dg0, dg1), (rows, D)
        assert np.array_equal(db0, db1), (rows, D)
        assert np.isfinite(dg1).all() and np.abs(dg1).max() > 0


def test_static_scratch_rejects_sizes_beyond_the_array():
    import torch
    from visiontransformer_amd import vitk as K
    K.init(0)
    t = torch.zeros(8, 2048, device="cuda")
    v = torch.zeros(2048, device="cuda")
    r = torch.zeros(8, device="cuda")
    with pytest.raises(K.VitkError, match="beyond the built-in scratch"):
        K.call("vitk_layernorm_bwd_static", t, t, r, r, v, 8, 2048, 2048, None, t, v, v, K.stream())
