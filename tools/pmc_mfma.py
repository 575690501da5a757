"""Matrix-pipe utilisation per kernel from one rocprofv3 --pmc pass of bench.py
(SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE) and the
--kernel-trace --stats summary of the same command without counters (for the launch durations).
SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs and GRBM_GUI_ACTIVE over the 8 XCDs.
    python tools/pmc_mfma.py <dir of the counter pass> <kernel_stats.csv> [out.json]"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return m.group(1) if m else name


acc = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0])):
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    disp[k].add(r["Dispatch_Id"])
dur = {short(r["Name"]): float(r["AverageNs"]) * 1e-9 for r in csv.DictReader(open(sys.argv[2]))}
out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -- "
                 "python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (ViT-B/16, batch 256); per launch. "
                 "mfma_busy_frac = (MFMA_BUSY / 1024 SIMDs) / (GUI_ACTIVE / 8 XCDs): a lower bound, the counter pass adds "
                 "per-dispatch time that GUI_ACTIVE includes.  clock_ghz_if_mfma_pipe_never_idle = MFMA cycles per SIMD / "
                 "launch duration WITHOUT a profiler attached (kernel-trace pass of the same command).",
       "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_VALU_MFMA_BUSY_CYCLES"]):
    if v["SQ_VALU_MFMA_BUSY_CYCLES"] == 0:
        continue
    n = len(disp[k])
    e = {"launches": n, **{c: x / n for c, x in v.items()}}
    e["mfma_cycles_per_simd"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
    e["mfma_busy_frac"] = round(e["mfma_cycles_per_simd"] / (e["GRBM_GUI_ACTIVE"] / 8), 4)
    if k in dur:
        e["launch_s_without_profiler"] = dur[k]
        e["clock_ghz_if_mfma_pipe_never_idle"] = round(e["mfma_cycles_per_simd"] / dur[k] / 1e9, 3)
    out["kernels"][k] = e
    print(f"{k[:50]:50s} n={n:4d} busy {e['mfma_busy_frac']:.3f}  {e.get('clock_ghz_if_mfma_pipe_never_idle', '')}")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
