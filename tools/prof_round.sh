#!/bin/bash
# Per-round profiles on a GPU box (run from the repo root through gpurun): kernel stats of the bench command and of
# four more BASELINE cells, plus PMC counters in separate passes (MI355X_MICROARCH.md: --pmc never together with
# tracing domains other than the kernel trace; FETCH_SIZE / WRITE_SIZE in passes of their own).
#   tools/prof_round.sh [out dir]
set -u
OUT=${1:-gpurun_out/prof_r03}
mkdir -p "$OUT"
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
run_cell() {   # name scene strategy
  local name=$1 sid=$2 kid=$3
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/stats" -- python3 tools/prof_target.py $sid $kid 1920 1080 10 > "$OUT/$name/target.json" 2> "$OUT/$name/stats.err"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
      --output-format csv -d "$OUT/$name/pmc1" -- python3 tools/prof_target.py $sid $kid 1920 1080 6 > /dev/null 2> "$OUT/$name/pmc1.err"
  rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/$name/pmc2" -- python3 tools/prof_target.py $sid $kid 1920 1080 6 > /dev/null 2> "$OUT/$name/pmc2.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$name/pmc3" -- python3 tools/prof_target.py $sid $kid 1920 1080 6 > /dev/null 2> "$OUT/$name/pmc3.err"
  rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d "$OUT/$name/pmc4" -- python3 tools/prof_target.py $sid $kid 1920 1080 6 > /dev/null 2> "$OUT/$name/pmc4.err"
  python3 tools/prof_collect.py "$OUT/$name/pmc_per_launch.json" "$OUT/$name/pmc1" "$OUT/$name/pmc2" "$OUT/$name/pmc3" "$OUT/$name/pmc4" > "$OUT/$name/collect.log" 2>&1
  # keep the summaries only (the merged copy-back is capped)
  find "$OUT/$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/$name/kernel_stats.csv" \; 2>/dev/null
  rm -rf "$OUT/$name/stats" "$OUT/$name/pmc1" "$OUT/$name/pmc2" "$OUT/$name/pmc3" "$OUT/$name/pmc4"
  echo "$name done: $(cat $OUT/$name/collect.log | cut -c1-200)"
}
mkdir -p "$OUT/bench"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench/stats" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/bench/bench_line_under_rocprof.json" 2> "$OUT/bench/stats.err"
find "$OUT/bench" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench/kernel_stats_bench.csv" \;
rm -rf "$OUT/bench/stats"
echo "bench done"
for cell in "mandelbulb_standard 10 0" "sphere_standard 0 0" "cube_standard 2 0" "menger_standard 9 0" "pillars_standard 12 0"; do
  set -- $cell
  mkdir -p "$OUT/$1"
  run_cell $1 $2 $3
done
