mkdir -p gpurun_out/t1
python -m pytest tests/test_gpu_dropout.py -q -x 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -30 > gpurun_out/t1/pytest.txt
cat gpurun_out/t1/pytest.txt
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes > gpurun_out/t1/rd.json 2> gpurun_out/t1/rd.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/t1/rd.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["rollout_only_ms_per_collect"], d["update_only_ms"])
P
CIRS_TRACKER_PREFIX_LAUNCHES=1 python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes > gpurun_out/t1/rd0.json 2> gpurun_out/t1/rd0.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/t1/rd0.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["rollout_only_ms_per_collect"], d["update_only_ms"])
P
