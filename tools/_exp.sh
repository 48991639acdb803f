mkdir -p gpurun_out
echo "== pipeline tests"; timeout -k 10 400 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -4
echo "== fuzz"; timeout -k 10 1000 python tests/fuzz_parity.py 1000000 31337 2>&1 | tail -3
