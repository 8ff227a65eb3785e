mkdir -p gpurun_out/s2j
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_plugin_surface.py tests/test_gpu_entrypoint.py tests/test_gpu_dropout.py tests/test_gpu_engine_dp.py tests/test_abi.py -q -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert" | tail -8 > gpurun_out/s2j/pytest.txt
cat gpurun_out/s2j/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probes > gpurun_out/s2j/a.json 2> gpurun_out/s2j/a.err
python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 100 --warmup 150 > gpurun_out/s2j/c2.json 2>> gpurun_out/s2j/a.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2j/a.json").read().strip().splitlines()[-1])
print("drv", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["minibatch_step"]["seconds"], d.get("rollout_only_ms_per_collect"))
d=json.loads(open("gpurun_out/s2j/c2.json").read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], d["value"])
P
