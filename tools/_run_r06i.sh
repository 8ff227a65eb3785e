mkdir -p gpurun_out/r06i
python -m pytest tests/test_gpu_policy.py tests/test_gpu_rollout.py tests/test_gpu_edges.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > gpurun_out/r06i/pytest.txt
cat gpurun_out/r06i/pytest.txt
python tools/ab_rollout.py c3 --dropout 0.1 tools/probes/ab/r06c.so - > gpurun_out/r06i/ab_c3_drop.txt 2>&1
cat gpurun_out/r06i/ab_*.txt
