"""Diagnostic: where a GEMM workgroup's cycles go (needs `make -C visiontransformer_amd diag`).

Loads the stamped kernel library (never the product one), runs one GEMM shape and prints, over
all workgroups, the cycles spent before the loop, in the loop (per K slab) and in the epilogue,
the time parked at the per-slab barrier, and how the workgroups' lifetimes tile the kernel.
    PYTHONPATH=. python tools/gemm_stamps.py M N K [ta] [tb] [split_k]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

from visiontransformer_amd import vitk

vitk.LIB_PATH = os.path.join(os.path.dirname(vitk.LIB_PATH), "libvitk_stamps.so")
vitk.init(0)
m, n, k = (int(v) for v in sys.argv[1:4])
ta, tb, sk = (int(v) for v in (sys.argv[4:7] + ["0", "0", "1"][len(sys.argv[4:7]):]))
a = torch.randn((k, m) if ta else (m, k), device="cuda")
b = torch.randn((n, k) if tb else (k, n), device="cuda")
c = torch.empty(m, n, device="cuda")
ws = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
nblk_max = 1 << 16
stamps = torch.zeros(nblk_max * 16, dtype=torch.int64, device="cuda")
d = vitk.dll()
d.vitk_gemm_set_stamps.argtypes = [C.c_void_p]
d.vitk_gemm_set_stamps.restype = None
run = lambda: vitk.gemm(a, b, c, m, n, k, trans_a=bool(ta), trans_b=bool(tb), split_k=sk, workspace=ws)
for _ in range(3):
    run()
torch.cuda.synchronize()
# the clock reading is only meaningful at the state the chip settles in under this load: keep the
# kernel running back to back for STAMPS_SUSTAIN_S seconds (default 2.5) before the stamped launch
import time
t_end = time.perf_counter() + float(os.environ.get("STAMPS_SUSTAIN_S", "2.5"))
while time.perf_counter() < t_end:
    for _ in range(20):
        run()
    torch.cuda.synchronize()
d.vitk_gemm_set_stamps(stamps.data_ptr())
stamps.zero_()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
s = stamps.cpu().numpy().reshape(-1, 16)
s = s[s[:, 0] != 0]
nb = len(s)
t0, t1, t2, t3, hw, rt, bw, rt1 = (s[:, i].astype(np.int64) for i in range(8))
print(f"shape M={m} N={n} K={k} ta={ta} tb={tb}: {nb} workgroups, {ms:.3f} ms, {2.0*m*n*k/ms/1e9:.1f} TF/s")
span_rt = rt1.max() - rt.min()   # 100 MHz ticks
print(f"kernel span {span_rt/100.0:.1f} us by the 100 MHz clock; in-kernel shader clock "
      f"{np.median((t3 - t0) / np.maximum(rt1 - rt, 1)) * 0.1:.3f} GHz (median over workgroups)")
print("slot occupancy: sum of lifetimes / (span * 768) = %.3f" % ((rt1 - rt).sum() / (span_rt * 768.0)))
q = lambda x: "min %d  p50 %d  p90 %d  max %d  mean %.0f" % (x.min(), np.median(x), np.percentile(x, 90), x.max(), x.mean())
print("prologue  ", q(t1 - t0))
print("loop      ", q(t2 - t1))
print("epilogue  ", q(t3 - t2))
print("total     ", q(t3 - t0))
print("loop share of lifetime %.3f, prologue %.3f, epilogue %.3f" % ((t2 - t1).sum() / (t3 - t0).sum(), (t1 - t0).sum() / (t3 - t0).sum(), (t3 - t2).sum() / (t3 - t0).sum()))
wave_id = hw & 0xF
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
print("wave slots seen:", np.bincount(wave_id.astype(int)))
# ideal loop cycles: MFMAs per wave * 64 cycles * co-resident waves (3)
nwg_x = int(os.environ.get("STAMPS_GRID_X", "0")) or nb
ids = np.nonzero(stamps.cpu().numpy().reshape(-1, 16)[:, 0])[0]
bx = ids % nwg_x
life = (t3 - t0)
def group(label, key):
    print(label)
    for v in np.unique(key):
        sel = key == v
        print(f"   {v:4d}: n={sel.sum():5d}  lifetime mean {life[sel].mean():9.0f}  min {life[sel].min():8d}  max {life[sel].max():8d}  clock {np.median((t3-t0)[sel]/np.maximum(rt1-rt,1)[sel])*0.1:.3f} GHz")
group("by blockIdx.x % 8 (XCD)", bx % 8)
group("by HW_ID.se_id", ((hw >> 13) & 7).astype(int))
group("by HW_ID.cu_id", ((hw >> 8) & 15).astype(int))
group("by HW_ID.simd_id", ((hw >> 4) & 3).astype(int))
group("by wave slot", (hw & 15).astype(int))
# co-residency: number of workgroups alive at each workgroup's midpoint on the same (xcd, se, cu)
