"""Diagnostic: phase times of the fused attention backward (needs lib/libvitk_sdstamps.so, built with
-DVITK_SDPA_STAMPS).  Stamps of query-tile iteration 3, lane 0 of every wave."""
import ctypes as C, os, numpy as np, torch
from visiontransformer_amd import vitk
vitk.LIB_PATH = os.path.join(os.path.dirname(vitk.LIB_PATH), "libvitk_sdstamps.so")
vitk.init(0)
B, H, N, dh = 256, 12, 197, 64
D = H * dh
qkv = torch.randn(B * N, 3 * D, device="cuda"); dqkv = torch.empty_like(qkv)
out = torch.empty(B * N, D, device="cuda"); dout = torch.randn(B * N, D, device="cuda")
stats = torch.empty(B * H * N * 2, device="cuda")
q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
dq, dk, dv = dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:]
vitk.call("vitk_sdpa_fwd", q, k, v, 3 * D, B, H, N, dh, 0.125, out, D, stats, vitk.stream())
wsb = vitk.dll().vitk_sdpa_bwd_workspace_bytes(B, H, N)
ws = torch.empty(max(wsb // 4, 1), device="cuda")
def bwd():
    vitk.call("vitk_sdpa_bwd", q, k, v, 3 * D, out, dout, D, stats, B, H, N, dh, 0.125, 0, dq, dk, dv, 3 * D, ws, wsb, vitk.stream())
for _ in range(10): bwd()
torch.cuda.synchronize()
st = torch.zeros(B * H * 8 * 16, dtype=torch.int64, device="cuda")
d = vitk.dll(); d.vitk_sdpa_set_stamps.argtypes = [C.c_void_p]; d.vitk_sdpa_set_stamps.restype = None
d.vitk_sdpa_set_stamps(st.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); bwd(); e1.record(); torch.cuda.synchronize()
print("kernel %.1f us" % (e0.elapsed_time(e1) * 1000))
s = st.cpu().numpy().reshape(B * H, 8, 16)
names = ["load next tile + S,dP (64 MFMA)", "softmax/dS + transpose writes", "dV,dK (64 MFMA)", "dQ partial (32 MFMA)",
         "partial dQ -> LDS", "wait at barrier 1", "stage next tile + reduce dQ", "wait at barrier 2"]
for w in (0, 3, 6):
    x = s[:, w, :9].astype(np.int64)
    x = x[x[:, 0] != 0]
    print(f"wave {w}: iteration total median {np.median(x[:, 8] - x[:, 0]):.0f} cycles")
    for i, nm in enumerate(names):
        dlt = x[:, i + 1] - x[:, i]
        print(f"   {nm:34s} median {np.median(dlt):7.0f}  p90 {np.percentile(dlt, 90):7.0f}")
x7 = s[:, 7, :9].astype(np.int64); x7 = x7[x7[:, 0] != 0]
print("idle wave 7: iteration", np.median(x7[:, 8] - x7[:, 0]), " barrier1 wait", np.median(x7[:, 6] - x7[:, 5]))
