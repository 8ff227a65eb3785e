mkdir -p gpurun_out/s2h
python -m pytest tests/test_gpu_dropout.py tests/test_gpu_tracker_bwd.py tests/test_gpu_learn.py tests/test_gpu_engine_dp.py -q -x 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8 > gpurun_out/s2h/pytest.txt
cat gpurun_out/s2h/pytest.txt
python tools/probes/attn_prof.py c3 2>&1 | tail -8
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probes > gpurun_out/s2h/a.json 2> gpurun_out/s2h/a.err
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/s2h/rd.json 2> gpurun_out/s2h/rd.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2h/a.json").read().strip().splitlines()[-1])
print("drv", d["ms_per_step"], d["value"], d.get("c2",{}).get("value"), d["roofline"]["frac"], d["minibatch_step"]["seconds"])
d=json.loads(open("gpurun_out/s2h/rd.json").read().strip().splitlines()[-1])
print("redraw", d["ms_per_step"], d["value"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
P
