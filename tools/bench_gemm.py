"""Micro-benchmark of vitk_gemm_f32 on the ViT GEMM shapes (TFLOP/s vs the 157.3 TF fp32 matrix peak)."""
import os
import sys
import torch

from visiontransformer_amd import vitk

if len(sys.argv) > 2 and sys.argv[1] == "--lib":  # --lib <file name in visiontransformer_amd/lib> (diagnostic builds)
    vitk.LIB_PATH = os.path.join(os.path.dirname(vitk.LIB_PATH), sys.argv[2])
    del sys.argv[1:3]
vitk.init(0)
M = 50432
shapes = [
    ("qkv  NN", M, 2304, 768, False, False, 1),
    ("proj NN", M, 768, 768, False, False, 1),
    ("mlp1 NN", M, 3072, 768, False, False, 1),
    ("mlp2 NN", M, 768, 3072, False, False, 1),
    ("dx1  NT", M, 768, 3072, False, True, 1),
    ("dx2  NT", M, 3072, 768, False, True, 1),
    ("dxqkvNT", M, 768, 2304, False, True, 1),
    ("dW1  TN", 768, 3072, M, True, False, 0),
    ("dW2  TN", 3072, 768, M, True, False, 0),
    ("dWo  TN", 768, 768, M, True, False, 0),
    ("dWqkvTN", 768, 2304, M, True, False, 0),
]
if len(sys.argv) > 1 and sys.argv[1] == "--shape":  # --shape M N K [ta tb split_k] ...
    v = [int(t) for t in sys.argv[2:]]
    shapes = [("custom ", v[i], v[i + 1], v[i + 2], bool(v[i + 3]), bool(v[i + 4]), v[i + 5]) for i in range(0, len(v), 6)]
elif len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
# bring the chip to its loaded clock / power state first: the first shape timed after an idle
# period otherwise reads ~15 % low
_w = torch.randn(8192, 1024, device="cuda")
_c = torch.empty(8192, 8192, device="cuda")
for _ in range(60):
    vitk.gemm(_w, _w, _c, 8192, 8192, 1024, trans_b=True)
torch.cuda.synchronize()
for name, m, n, k, ta, tb, sk in shapes:
    a = torch.randn((k, m) if ta else (m, k), device="cuda")
    b = torch.randn((n, k) if tb else (k, n), device="cuda")
    c = torch.empty(m, n, device="cuda")
    for _ in range(2):
        vitk.gemm(a, b, c, m, n, k, trans_a=ta, trans_b=tb, split_k=sk, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record()
    for _ in range(iters):
        vitk.gemm(a, b, c, m, n, k, trans_a=ta, trans_b=tb, split_k=sk, workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * m * n * k / ms / 1e9
    print(f"{name}  M={m} N={n} K={k}: {ms:8.3f} ms  {tf:7.1f} TF/s  ({100 * tf / 157.3:.1f}% of fp32 MFMA peak)", flush=True)
