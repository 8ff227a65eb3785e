mkdir -p gpurun_out/s2g
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8 > gpurun_out/s2g/pytest.txt
cat gpurun_out/s2g/pytest.txt
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s2g/a.json 2> gpurun_out/s2g/a.err
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/s2g/rd.json 2> gpurun_out/s2g/rd.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2g/a.json").read().strip().splitlines()[-1])
print("drv", d["ms_per_step"], d["value"], d.get("c2",{}).get("value"), d["roofline"]["frac"], d["minibatch_step"]["seconds"])
d=json.loads(open("gpurun_out/s2g/rd.json").read().strip().splitlines()[-1])
print("redraw", d["ms_per_step"], d["value"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
P
