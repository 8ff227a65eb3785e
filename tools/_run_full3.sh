mkdir -p gpurun_out/f3
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert" | tail -6 > gpurun_out/f3/pytest.txt
cat gpurun_out/f3/pytest.txt
bash tools/_run_bench2.sh
