"""LayerNorm backward at the sizes of the 5-token model (640 x 96, 128 x 96 rows x features): the
one-launch form against the two-stage form, microseconds per call back to back.
    PYTHONPATH=. python tools/bench_ln_small.py"""
import torch

from visiontransformer_amd import vitk as K

K.init(0)
for rows, D in ((640, 96), (128, 96), (1024, 192), (512, 192)):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(rows, D, device="cuda", generator=g)
    dy = torch.randn(rows, D, device="cuda", generator=g)
    gamma, beta = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, mean, inv = torch.empty_like(x), torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    st = K.stream()
    K.call("vitk_layernorm_fwd", x, rows, D, D, gamma, beta, 1e-5, y, mean, inv, st)
    need = K.dll().vitk_layernorm_bwd_workspace_bytes(rows, D)
    ws = torch.empty(need // 4, device="cuda")
    dx, dg, db = torch.empty_like(x), torch.empty(D, device="cuda"), torch.empty(D, device="cuda")
    out = []
    for small in (1, 0):
        K.check(K.dll().vitk_layernorm_small_enable(small))
        ts = []
        for it in range(8):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                K.call("vitk_layernorm_bwd", dy, x, mean, inv, gamma, rows, D, D, None, dx, dg, db, ws, need, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        ts.sort()
        out.append(ts[len(ts) // 2])
    K.check(K.dll().vitk_layernorm_small_enable(1))
    print(f"{rows} x {D}: one launch {out[0]:.2f} us, two stages {out[1]:.2f} us", flush=True)
