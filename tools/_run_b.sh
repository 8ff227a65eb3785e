mkdir -p gpurun_out/r03b
python -m pytest tests/test_gpu_learn.py tests/test_gpu_head_precision.py tests/test_gpu_engine_dp.py tests/test_gpu_deepfm.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r03b/pytest.txt
python bench.py --steps 50 --warmup 15 --no-cpu-baseline --no-probes > gpurun_out/r03b/bench_c3.json 2> gpurun_out/r03b/bench_c3.err
tail -2 gpurun_out/r03b/pytest.txt
python -c "
import json;z=json.load(open('gpurun_out/r03b/bench_c3.json'));print({k:z[k] for k in ('value','ms_per_step','rollout_only_env_steps_per_s','update_only_ms')}, 'frac',z['roofline']['frac'], 'bwd',z['roofline']['seconds_per_launch'], 'mbstep',z['minibatch_step']['seconds'], 'stats', z['minibatch_step']['head_stats_kernel_seconds'])"
