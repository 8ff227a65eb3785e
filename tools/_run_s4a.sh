mkdir -p gpurun_out/s4a
python -m pytest tests/test_gpu_learn.py tests/test_gpu_engine_dp.py tests/test_gpu_tp_learner.py tests/test_gpu_sharded.py tests/test_gpu_head_precision.py tests/test_gpu_tracker_bwd.py tests/test_gpu_entrypoint.py tests/test_gpu_plugin_surface.py -q -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert" | tail -6 > gpurun_out/s4a/pytest.txt
cat gpurun_out/s4a/pytest.txt
python tools/parity_margins.py > gpurun_out/s4a/margins.txt 2>&1; tail -12 gpurun_out/s4a/margins.txt
bash tools/_run_bench2.sh
