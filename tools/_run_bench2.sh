mkdir -p gpurun_out/b2
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probes > gpurun_out/b2/a.json 2> gpurun_out/b2/a.err
python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 100 --warmup 150 > gpurun_out/b2/c2.json 2>> gpurun_out/b2/a.err
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/b2/rd.json 2>> gpurun_out/b2/a.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/b2/a.json").read().strip().splitlines()[-1]); print("drv", d["ms_per_step"], d["value"], d["minibatch_step"]["seconds"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
d=json.loads(open("gpurun_out/b2/c2.json").read().strip().splitlines()[-1]); print("c2", d["ms_per_step"], d["value"])
d=json.loads(open("gpurun_out/b2/rd.json").read().strip().splitlines()[-1]); print("rd", d["ms_per_step"], d["value"])
P
