mkdir -p gpurun_out/s5a
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_tracker.py tests/test_gpu_dropout.py -q -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert" | tail -5
for rep in 1 2; do for w in 4 1; do
  CIRS_STEP_WAVES=$w python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 150 --warmup 150 > gpurun_out/s5a/c2_${w}_$rep.json 2>> gpurun_out/s5a/err.txt
done; done
python - <<'P'
import json
for rep in (1,2):
  for w in (4,1):
    c=json.loads(open(f"gpurun_out/s5a/c2_{w}_{rep}.json").read().strip().splitlines()[-1])
    print(rep, "waves", w, "c2", round(c["ms_per_step"],4), round(c["value"]), c.get("rollout_only_ms_per_collect"))
P
