"""vitk_adam_l2 alone on a ViT-B sized parameter arena (28 bytes of traffic per parameter)."""
import os, sys, torch
from visiontransformer_amd import vitk
vitk.init(0)
n = 86567656
p, g, m, v = (torch.randn(n, device="cuda") for _ in range(4))
v.abs_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for it in range(12):
    e0.record()
    vitk.call("vitk_adam_l2", p, g, m, v, n, 1e-3, 0.9, 0.999, 0.9 ** (it + 1), 0.999 ** (it + 1), 1e-8, 1e-4, 1.0, vitk.stream())
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"adam, {n} parameters: median {ts[len(ts)//2]:.3f} ms  {28*n/ts[len(ts)//2]/1e9:.2f} TB/s")
