#!/bin/bash
# Every number the design notes quote, from one box: bench lines of the five configurations (with
# the CPU baseline), evaluation throughput, the side variants of the headline configuration, then
# the profiler passes (kernel stats of ViT-B/S/Ti, HBM traffic, matrix-pipe counters).
#   bash tools/measure_round.sh <tag, e.g. r03> [bench|prof|pmc]  (from the repository root, on a GPU box;
#   pmc = the counter passes of prof alone, which need that tag's ViT-B kernel stats in gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r03}
WHAT=${2:-all}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
run() { echo "== $*" >&2; timeout -k 10 "$@" || { echo "FAILED ($?): $*"; exit 1; }; }
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  run 300 python bench.py > "$OUT/${TAG}_bench_vit_b16.json" 2> "$OUT/err.log" && tail -c 300 "$OUT/${TAG}_bench_vit_b16.json"
  run 300 python bench.py --workload vit_s16 > "$OUT/${TAG}_bench_vit_s16.json" 2>> "$OUT/err.log"
  run 300 python bench.py --workload vit_ti16 > "$OUT/${TAG}_bench_vit_ti16.json" 2>> "$OUT/err.log"
  run 300 python bench.py --workload blood_p7 --steps 50 --warmup 10 > "$OUT/${TAG}_bench_blood_p7.json" 2>> "$OUT/err.log"
  run 300 python bench.py --workload mnist_p14 --steps 100 --warmup 20 > "$OUT/${TAG}_bench_mnist_p14.json" 2>> "$OUT/err.log"
  echo "-- side variants"
  run 200 python bench.py --no-cpu-baseline --dense-tail > "$OUT/${TAG}_bench_vit_b16_dense_tail.json" 2>> "$OUT/err.log"
  run 200 python bench.py --no-cpu-baseline --dropout 0.1 > "$OUT/${TAG}_bench_vit_b16_dropout01.json" 2>> "$OUT/err.log"
  run 200 python bench.py --no-cpu-baseline --gpus 1 --force-collectives > "$OUT/${TAG}_bench_vit_b16_force_collectives.json" 2>> "$OUT/err.log"
  run 200 python bench.py --through-trainer --steps 24 > "$OUT/${TAG}_bench_vit_b16_through_trainer.json" 2>> "$OUT/err.log"
  : > "$OUT/${TAG}_bench_eval_all_configs.jsonl"
  for w in vit_b16 vit_s16 vit_ti16 blood_p7 mnist_p14; do
    run 200 python bench.py --eval --workload $w --steps 20 --warmup 5 >> "$OUT/${TAG}_bench_eval_all_configs.jsonl" 2>> "$OUT/err.log"
  done
  run 200 python bench.py --eval --dense-tail --steps 20 --warmup 5 >> "$OUT/${TAG}_bench_eval_all_configs.jsonl" 2>> "$OUT/err.log"
  echo "-- bench lines done"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
  export TMPDIR=/tmp
  for w in vit_b16 vit_s16 vit_ti16 blood_p7 mnist_p14; do
    rm -rf "/tmp/kt_$w"
    steps="--steps 5 --warmup 2"
    [ $w = blood_p7 ] && steps="--steps 50 --warmup 10"
    [ $w = mnist_p14 ] && steps="--steps 100 --warmup 20"
    run 300 rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/kt_$w" -o kt -- python3 "$ROOT/bench.py" --workload $w --no-cpu-baseline $steps \
        > "$OUT/${TAG}_bench_${w}_under_rocprof.json" 2>> "$OUT/err.log"
    cp "$(find /tmp/kt_$w -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_${w}_kernel_stats.csv" || exit 1
    echo "   kernel stats of $w copied"
  done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ] || [ "$WHAT" = pmc ]; then
  export TMPDIR=/tmp
  rm -rf /tmp/pmc_f /tmp/pmc_w /tmp/pmc_m
  run 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> "$OUT/err.log"
  echo "   fetch pass done"
  run 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_w -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> "$OUT/err.log"
  echo "   write pass done"
  run 60 python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w "$OUT/${TAG}_hbm_traffic.json" > "$OUT/${TAG}_hbm_traffic.txt"
  run 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -o p -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>> "$OUT/err.log"
  run 60 python tools/pmc_mfma.py /tmp/pmc_m "$OUT/${TAG}_bench_vit_b16_kernel_stats.csv" "$OUT/${TAG}_mfma_busy.json" > "$OUT/${TAG}_mfma_busy.txt"
  echo "-- profiler passes done"
fi
