"""Tile-shape choice on the contractions of ViT-Ti / ViT-S (M = 256 x 197 rows): the library's plan
against each tile forced (vitk_gemm_args.reserved: 1 = 64 x 64, 2 = 128 x 64, 3 = 64 x 128), plain
epilogue.  Weight gradients (TN, K = 50 432) keep the plan's split count.
    PYTHONPATH=. python tools/bench_gemm_tiles.py [substring of the shape names] [--rows M]
(--rows 8704: the same contractions at the row count of the 17-token model, batch 512)"""
import ctypes as C
import sys

import torch

from visiontransformer_amd import vitk

vitk.init(0)
M = 50432
if "--rows" in sys.argv:
    i = sys.argv.index("--rows")
    M = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
# (name, m, n, k, trans_a, trans_b)
shapes = [("Ti qkv", M, 576, 192, 0, 0), ("Ti out", M, 192, 192, 0, 0), ("Ti ffn1", M, 768, 192, 0, 0),
          ("Ti ffn2", M, 192, 768, 0, 0), ("Ti dx ffn1", M, 192, 768, 0, 1), ("Ti dx ffn2", M, 768, 192, 0, 1),
          ("Ti dx qkv", M, 192, 576, 0, 1), ("Ti dx out", M, 192, 192, 0, 1),
          ("Ti dW qkv", 192, 576, M, 1, 0), ("Ti dW out", 192, 192, M, 1, 0), ("Ti dW ffn1", 192, 768, M, 1, 0),
          ("Ti dW ffn2", 768, 192, M, 1, 0),
          ("S qkv", M, 1152, 384, 0, 0), ("S out", M, 384, 384, 0, 0), ("S ffn1", M, 1536, 384, 0, 0),
          ("S ffn2", M, 384, 1536, 0, 0), ("S dx ffn2", M, 1536, 384, 0, 1), ("S dx ffn1", M, 384, 1536, 0, 1),
          ("S dx qkv", M, 384, 1152, 0, 1), ("S dx out", M, 384, 384, 0, 1),
          ("S dW qkv", 384, 1152, M, 1, 0), ("S dW out", 384, 384, M, 1, 0), ("S dW ffn1", 384, 1536, M, 1, 0),
          ("S dW ffn2", 1536, 384, M, 1, 0),
          ("B out", M, 768, 768, 0, 0), ("B ffn1", M, 3072, 768, 0, 0)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
ws = torch.empty(96 << 20, dtype=torch.float32, device="cuda")
_w = torch.randn(8192, 1024, device="cuda")
_c = torch.empty(8192, 8192, device="cuda")
for _ in range(40):
    vitk.gemm(_w, _w, _c, 8192, 8192, 1024, trans_b=True)
torch.cuda.synchronize()
print(f"{'':14s} {'M':>6s} {'N':>5s} {'K':>6s}   plan   64x64  128x64  64x128   (TFLOP/s, median of 25)")
for name, m, n, k, ta, tb in shapes:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    c = torch.empty(m, n, device="cuda")
    out = []
    for forced in (0, 1, 2, 3):
        g = vitk.GemmArgs()
        g.trans_a, g.trans_b, g.M, g.N, g.K = ta, tb, m, n, k
        g.A, g.B, g.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
        g.lda, g.ldb, g.ldc, g.ldr, g.ldaux = (m if ta else k), (k if tb else n), n, n, n
        g.alpha, g.split_k, g.reserved = 1.0, (0 if ta else 1), forced
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if vitk.dll().vitk_gemm_workspace_bytes(C.byref(g)) > ws.numel() * 4:
            out.append(float("nan"))
            continue
        ts = []
        for it in range(25):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            vitk.check(vitk.dll().vitk_gemm_f32(C.byref(g), vitk.stream()))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        out.append(2.0 * m * n * k / ts[len(ts) // 2] / 1e9)
    print(f"{name:14s} {m:6d} {n:5d} {k:6d}  " + "  ".join(f"{v:6.1f}" for v in out), flush=True)
