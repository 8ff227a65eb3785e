mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_policy.py tests/test_gpu_rollout.py tests/test_gpu_edges.py tests/test_gpu_learn.py::test_lost_handoff_is_loud -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06b/pytest.txt
cat gpurun_out/r06b/pytest.txt
python tools/ab_rollout.py c3 --dropout 0.1 tools/probes/ab/r06a.so - > gpurun_out/r06b/ab_c3_drop.txt 2>&1
python tools/ab_rollout.py c2 --dropout 0.1 tools/probes/ab/r06a.so - > gpurun_out/r06b/ab_c2.txt 2>&1
cat gpurun_out/r06b/ab_*.txt
