mkdir -p gpurun_out/prof_r03
echo "== gpu tests"; timeout -k 10 700 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== full matrix 1080p"; timeout -k 10 400 python tests/full_matrix_parity.py 2>&1 | tail -3
echo "== fuzz"; timeout -k 10 200 python tests/fuzz_parity.py 100000 777 2>&1 | tail -2
echo "== bench"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err; python -c "
import json;d=json.loads(open('gpurun_out/r3_bench3.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d.get('batched_frames'),d['temporal_order']['value'],d['chain_latency']['idle_team']['us_per_evaluation'])"
echo "== viewpoints"; timeout -k 10 400 python tools/viewpoint_orders.py > gpurun_out/prof_r03/viewpoint_orders.jsonl 2> gpurun_out/prof_r03/viewpoint_orders.err; tail -1 gpurun_out/prof_r03/viewpoint_orders.jsonl
echo "== baseline cells"; timeout -k 10 300 python tools/baseline_cells.py > gpurun_out/prof_r03/baseline_cells.jsonl 2>&1; tail -1 gpurun_out/prof_r03/baseline_cells.jsonl
echo "== trace"; timeout -k 10 100 python tools/trace_pipeline.py --frames 4 --tag r03_default > gpurun_out/prof_r03/trace_default.json 2>&1; head -c 300 gpurun_out/prof_r03/trace_default.json
echo "== profiles"; cd /tmp && export TMPDIR=/tmp && cd - > /dev/null && timeout -k 10 700 bash tools/prof_round.sh gpurun_out/prof_r03 2>&1 | tail -7
