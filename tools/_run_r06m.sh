mkdir -p gpurun_out/r06m
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_tracker.py tests/test_gpu_dropout.py -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/r06m/pytest.txt; cat gpurun_out/r06m/pytest.txt
python tools/emulate_world.py --steps 12 --warmup 6 > gpurun_out/r06m/emulated_world.json 2> gpurun_out/r06m/emu.err
tail -14 gpurun_out/r06m/emulated_world.json | cut -c1-330
python tools/probe_dp_step.py > gpurun_out/r06m/dp_step.txt 2>&1; tail -8 gpurun_out/r06m/dp_step.txt
