mkdir -p gpurun_out/r06k
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_tracker.py tests/test_gpu_dropout.py tests/test_gpu_edges.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r06k/pytest.txt
cat gpurun_out/r06k/pytest.txt
python tools/ab_rollout.py c3 --dropout 0.1 --rounds 1 tools/probes/ab/r06d.so - > gpurun_out/r06k/ab_c3_drop.txt 2>&1
python tools/ab_rollout.py c2 --dropout 0.1 --rounds 1 tools/probes/ab/r06d.so - > gpurun_out/r06k/ab_c2.txt 2>&1
cat gpurun_out/r06k/ab_*.txt
