"""Fused attention forward + backward alone, at a ViT shape (default ViT-B/16, batch 256): a short
program to put under rocprofv3 (--kernel-trace --stats, or --pmc counters) without the rest of the
training step.    python tools/bench_sdpa.py [B H N dh [iters [heads per workgroup of the backward [of the forward]]]]"""
import sys

import torch

from visiontransformer_amd import vitk

vitk.init(0)
B, H, N, dh = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (256, 12, 197, 64)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 30
D, M = H * dh, B * N
if len(sys.argv) > 6:
    vitk.check(vitk.dll().vitk_sdpa_bwd_tile16_enable(int(sys.argv[6])))
if len(sys.argv) > 7:
    vitk.check(vitk.dll().vitk_sdpa_tile16_enable(int(sys.argv[7])))
g = torch.Generator(device="cuda").manual_seed(1)
qkv = torch.randn(M, 3 * D, device="cuda", generator=g)
do = torch.randn(M, D, device="cuda", generator=g)
out, stats = torch.empty(M, D, device="cuda"), torch.empty(B * H, N, 2, device="cuda")
grads = torch.empty(M, 3 * D, device="cuda")
nb = vitk.dll().vitk_sdpa_bwd_workspace_bytes(B, H, N)
ws = torch.empty(nb // 4, device="cuda")
scale = dh ** -0.5
st = vitk.stream()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf, tb = [], []
for it in range(iters + 1):
    e[0].record()
    vitk.call("vitk_sdpa_fwd", qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, B, H, N, dh, scale, out, D, stats, st)
    e[1].record()
    vitk.call("vitk_sdpa_bwd", qkv, qkv[:, D:], qkv[:, 2 * D:], 3 * D, out, do, D, stats, B, H, N, dh, scale,
              vitk.DK_REFERENCE, grads, grads[:, D:], grads[:, 2 * D:], 3 * D, ws, nb, st)
    e[2].record()
    torch.cuda.synchronize()
    if it:  # the first iteration warms up
        tf.append(e[0].elapsed_time(e[1]))
        tb.append(e[1].elapsed_time(e[2]))
flops = 4.0 * B * H * N * N * dh
tf.sort(), tb.sort()
mf, mb = tf[len(tf) // 2], tb[len(tb) // 2]
print(f"B={B} H={H} N={N} dh={dh}: forward median {mf:.3f} ms (min {tf[0]:.3f}; {flops / mf / 1e9:.1f} TFLOP/s), "
      f"backward median {mb:.3f} ms (min {tb[0]:.3f}; {2.5 * flops / mb / 1e9:.1f} TFLOP/s)")
