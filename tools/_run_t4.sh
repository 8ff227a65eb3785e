mkdir -p gpurun_out/t4
python -m pytest tests/test_gpu_dropout.py tests/test_gpu_tracker_bwd.py tests/test_gpu_learn.py -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -8
python tools/probes/prefix_prof.py 30 2>&1 | grep -v amdgpu.ids
for f in "" "--dropout-redraw"; do
python bench.py --steps 20 --warmup 5 $f --no-probes --no-cpu-baseline > gpurun_out/t4/b.json 2> gpurun_out/t4/b.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/t4/b.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["rollout_only_ms_per_collect"], d["update_only_ms"])
P
done
