"""How much of the gap between the GEMM inside a step (138-143 TFLOP/s) and a bare MFMA + LDS-read
loop (148.5, tools/mfma_ceiling.hip) is the memory side?  The same kernel, tile count, LDS traffic
and output stream, once on the real operands and once as a batch of row blocks that all read ONE
128-row A panel (every global read is then an L2 hit; nothing but the output goes to HBM)."""
import sys
import torch
from visiontransformer_amd import vitk

vitk.init(0)
M, TM = 50432, 128
_w = torch.randn(8192, 1024, device="cuda")
_c = torch.empty(8192, 8192, device="cuda")
for _ in range(60):
    vitk.gemm(_w, _w, _c, 8192, 8192, 1024, trans_b=True)
torch.cuda.synchronize()


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for n, k in ((768, 768), (3072, 768), (768, 3072), (2304, 768)):
    a = torch.randn(M, k, device="cuda")
    b = torch.randn(k, n, device="cuda")
    c = torch.empty(M, n, device="cuda")
    real = timed(lambda: vitk.gemm(a, b, c, M, n, k))
    nb = M // TM
    blocks = timed(lambda: vitk.gemm(a, b, c, TM, n, k, batch_outer=nb, stride_a=(TM * k, 0), stride_b=(0, 0),
                                     stride_c=(TM * n, 0)))
    one_panel = timed(lambda: vitk.gemm(a, b, c, TM, n, k, batch_outer=nb, stride_a=(0, 0), stride_b=(0, 0),
                                        stride_c=(TM * n, 0)))
    tf = lambda ms: 2.0 * M * n * k / ms / 1e9
    print(f"NN M={M} N={n} K={k}: one call {tf(real):6.1f} TF/s | {nb} row blocks as a batch {tf(blocks):6.1f} | "
          f"the same, every block reading A panel 0 {tf(one_panel):6.1f}", flush=True)
