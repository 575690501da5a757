"""How the 64- and 128-wide tile kernels' throughput grows with the contraction depth K at
M = 256 x 197 rows (what a longer K loop per workgroup would buy on the K = 192 / 384 shapes).
    PYTHONPATH=. python tools/bench_gemm_depth.py"""
import ctypes as C

import torch

from visiontransformer_amd import vitk

vitk.init(0)
M = 50432
_w = torch.randn(8192, 1024, device="cuda")
_c = torch.empty(8192, 8192, device="cuda")
for _ in range(40):
    vitk.gemm(_w, _w, _c, 8192, 8192, 1024, trans_b=True)
torch.cuda.synchronize()
for n in (192, 576, 768, 384, 1536):
    for k in (96, 192, 384, 768, 1536, 3072):
        a = torch.randn(M, k, device="cuda")
        b = torch.randn(k, n, device="cuda")
        c = torch.empty(M, n, device="cuda")
        out = []
        for forced in (0, 1):
            g = vitk.GemmArgs()
            g.trans_a, g.trans_b, g.M, g.N, g.K = 0, 0, M, n, k
            g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
            g.lda, g.ldb, g.ldc, g.ldr, g.ldaux = k, n, n, n, n
            g.alpha, g.split_k, g.reserved = 1.0, 1, forced
            ts = []
            for it in range(15):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                vitk.check(vitk.dll().vitk_gemm_f32(C.byref(g), vitk.stream()))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            out.append(2.0 * M * n * k / ts[len(ts) // 2] / 1e9)
        print(f"N={n:5d} K={k:5d}: plan {out[0]:6.1f} TF/s   64-wide forced {out[1]:6.1f} TF/s", flush=True)
