mkdir -p gpurun_out/s2a
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8 > gpurun_out/s2a/pytest.txt
cat gpurun_out/s2a/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2a/a.json 2> gpurun_out/s2a/a.err
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/s2a/rd.json 2> gpurun_out/s2a/rd.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2a/a.json").read().strip().splitlines()[-1])
print("drv", d["ms_per_step"], d["value"], d.get("c2",{}).get("value"), d["roofline"]["frac"], d["minibatch_step"]["seconds"])
print({k:(v if not isinstance(v,dict) else '...') for k,v in d.items()}.keys())
d=json.loads(open("gpurun_out/s2a/rd.json").read().strip().splitlines()[-1])
print("redraw", d["ms_per_step"], d["value"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
P
