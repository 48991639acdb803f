mkdir -p gpurun_out
echo "== pipeline tests"; timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -4
echo "== fuzz"; timeout -k 10 420 python tests/fuzz_parity.py 100000 20261005 2>&1 | tail -4
echo "== shards"; timeout -k 10 200 python tools/shard_sizes.py 2>&1
