mkdir -p gpurun_out/drv
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv/a.json 2> gpurun_out/drv/a.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-probes > gpurun_out/drv/b.json 2> gpurun_out/drv/b.err
python bench.py --gpus 1 --steps 100 --warmup 30 --no-probes > gpurun_out/drv/c.json 2> gpurun_out/drv/c.err
python - <<'P'
import json
for f in "abc":
    d=json.loads(open(f"gpurun_out/drv/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d.get("fresh_policy",{}).get("ms_per_step"), d.get("c2",{}).get("value"), d["roofline"]["frac"], d["minibatch_step"]["seconds"])
P
