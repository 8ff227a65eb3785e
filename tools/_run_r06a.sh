mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_learn.py tests/test_gpu_rollout.py tests/test_gpu_policy.py tests/test_gpu_tracker.py tests/test_gpu_edges.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/r06a/pytest.txt
cat gpurun_out/r06a/pytest.txt
python tools/ab_rollout.py c3 tools/probes/ab/r05.so - > gpurun_out/r06a/ab_c3.txt 2>&1
python tools/ab_rollout.py c3 --dropout 0.1 tools/probes/ab/r05.so - > gpurun_out/r06a/ab_c3_drop.txt 2>&1
python tools/ab_rollout.py c2 tools/probes/ab/r05.so - > gpurun_out/r06a/ab_c2.txt 2>&1
cat gpurun_out/r06a/ab_*.txt
