"""Weight-gradient contractions (A^T B over the token rows, split-K + reduction launch): time of
the pair of launches against the split count, next to the library's own choice (split 0).
    PYTHONPATH=. python tools/bench_gemm_split.py"""
import torch

from visiontransformer_amd import vitk

vitk.init(0)
cases = [("blood qkv", 8704, 192, 576), ("blood out", 8704, 192, 192), ("blood ffn1", 8704, 192, 768),
         ("blood ffn2", 8704, 768, 192), ("Ti qkv", 50432, 192, 576), ("Ti out", 50432, 192, 192),
         ("Ti ffn1", 50432, 192, 768), ("Ti ffn2", 50432, 768, 192), ("S qkv", 50432, 384, 1152),
         ("S out", 50432, 384, 384), ("S ffn1", 50432, 384, 1536), ("S ffn2", 50432, 1536, 384),
         ("B qkv", 50432, 768, 2304), ("B out", 50432, 768, 768), ("B ffn1", 50432, 768, 3072),
         ("B ffn2", 50432, 3072, 768)]
REPS = 8  # launches back to back per timing (a lone launch after a synchronize runs at a lower clock)
_w = torch.randn(8192, 1024, device="cuda")
_c = torch.empty(8192, 8192, device="cuda")
for _ in range(60):
    vitk.gemm(_w, _w, _c, 8192, 8192, 1024, trans_b=True)
torch.cuda.synchronize()
for name, rows, m, n in cases:
    x = torch.randn(rows, m, device="cuda")
    dy = torch.randn(rows, n, device="cuda")
    dw = torch.empty(m, n, device="cuda")
    db = torch.empty(n, device="cuda")
    line = f"{name:11s} K={rows:6d} {m:4d}x{n:4d}:"
    tile = 128 if (m >= 128 and n >= 128 and m % 128 == 0 and n % 128 == 0) else 64
    tiles = ((m + tile - 1) // tile) * ((n + tile - 1) // tile)
    gens = sorted({max(1, min(128, (g * 768) // tiles)) for g in (1, 2, 3, 4, 5, 6)} | {64})
    line += f" ({tiles} tiles of {tile})"
    import sys
    for split in ([0] if len(sys.argv) > 1 and sys.argv[1] == "--auto" else [0] + gens):
        ts = []
        try:
            for it in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                keep = vitk.gemm(x, dy, dw, m, n, rows, trans_a=True, split_k=split, colsum_b=db)
                e0.record()
                for _ in range(REPS):
                    keep = vitk.gemm(x, dy, dw, m, n, rows, trans_a=True, split_k=split, colsum_b=db)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / REPS)
        except Exception as ex:  # a split the library refuses
            line += f"  s{split}: n/a"
            continue
        ts.sort()
        line += f"  s{split}: {1e3 * ts[len(ts) // 2]:5.1f}"
    print(line + "  (us)", flush=True)
