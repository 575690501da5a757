"""Tile-width choice on the small-K contractions of ViT-Ti / ViT-S (M = 256 x 197 rows): the
library's plan against 64-wide tiles forced (vitk_gemm_args.reserved = 1), plain epilogue.
    PYTHONPATH=. python tools/bench_gemm_tiles.py"""
import ctypes as C

import torch

from visiontransformer_amd import vitk

vitk.init(0)
M = 50432
shapes = [("Ti qkv", 576, 192, 0), ("Ti out", 192, 192, 0), ("Ti ffn1", 768, 192, 0), ("Ti ffn2", 192, 768, 0),
          ("Ti dx ffn1", 192, 768, 1), ("Ti dx ffn2", 768, 192, 1), ("Ti dx qkv", 192, 576, 1),
          ("S qkv", 1152, 384, 0), ("S out", 384, 384, 0), ("S ffn1", 1536, 384, 0), ("S ffn2", 384, 1536, 0),
          ("S dx ffn2", 1536, 384, 1), ("S dx ffn1", 384, 1536, 1)]
_w = torch.randn(8192, 1024, device="cuda")
_c = torch.empty(8192, 8192, device="cuda")
for _ in range(40):
    vitk.gemm(_w, _w, _c, 8192, 8192, 1024, trans_b=True)
torch.cuda.synchronize()
for name, n, k, tb in shapes:
    a = torch.randn(M, k, device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    c = torch.empty(M, n, device="cuda")
    out = []
    for forced in (0, 1):
        g = vitk.GemmArgs()
        g.trans_a, g.trans_b, g.M, g.N, g.K = 0, tb, M, n, k
        g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
        g.lda, g.ldb, g.ldc, g.ldr, g.ldaux = k, (k if tb else n), n, n, n
        g.alpha, g.split_k, g.reserved = 1.0, 1, forced
        ts = []
        for it in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vitk.check(vitk.dll().vitk_gemm_f32(C.byref(g), vitk.stream()))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        out.append(2.0 * M * n * k / ts[len(ts) // 2] / 1e9)
    print(f"{name:12s} N={n:5d} K={k:5d}: plan {out[0]:6.1f} TF/s   64-wide forced {out[1]:6.1f} TF/s", flush=True)
