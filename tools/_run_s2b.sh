mkdir -p gpurun_out/s2b
python -m pytest tests/test_gpu_dropout.py tests/test_gpu_tracker_bwd.py tests/test_abi.py -q -x 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -30 > gpurun_out/s2b/pytest.txt
cat gpurun_out/s2b/pytest.txt
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/s2b/rd.json 2> gpurun_out/s2b/rd.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2b/rd.json").read().strip().splitlines()[-1])
print("redraw", d["ms_per_step"], d["value"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
P
