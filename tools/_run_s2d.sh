mkdir -p gpurun_out/s2d
python -m pytest tests/test_gpu_dropout.py tests/test_gpu_tracker_bwd.py tests/test_gpu_rollout.py -q -x 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -5 > gpurun_out/s2d/pytest.txt
cat gpurun_out/s2d/pytest.txt
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/s2d/rd.json 2> gpurun_out/s2d/rd.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2d/rd.json").read().strip().splitlines()[-1])
print("redraw", d["ms_per_step"], d["value"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
P
for r in 1 8 16 30; do python tools/probes/prefix_prof.py $r 2>&1 | grep -v "launches" ; done > gpurun_out/s2d/prefix_prof.txt
cat gpurun_out/s2d/prefix_prof.txt
