mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_rollout.py tests/test_gpu_dropout.py tests/test_gpu_tracker_bwd.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r06c/pytest.txt
cat gpurun_out/r06c/pytest.txt
python tools/ab_rollout.py c3 --dropout 0.1 tools/probes/ab/r06b.so - > gpurun_out/r06c/ab_c3_drop.txt 2>&1
python tools/ab_rollout.py c3 tools/probes/ab/r06b.so - > gpurun_out/r06c/ab_c3.txt 2>&1
python tools/ab_rollout.py c2 --dropout 0.1 tools/probes/ab/r06b.so - > gpurun_out/r06c/ab_c2.txt 2>&1
cat gpurun_out/r06c/ab_*.txt
python tools/probes/trk_prof.py c3 > gpurun_out/r06c/trk_prof_c3.txt 2>&1
tail -45 gpurun_out/r06c/trk_prof_c3.txt
