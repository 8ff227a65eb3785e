mkdir -p gpurun_out/s5c
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_policy.py -q -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert" | tail -5
for rep in 1 2; do for m in new old; do
  if [ $m = old ]; then export CIRS_HIP_LIB=$(pwd)/tools/probes/ab/base.so; else unset CIRS_HIP_LIB; fi
  python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 150 --warmup 150 > gpurun_out/s5c/c2_${m}_$rep.json 2>> gpurun_out/s5c/err.txt
done; done
unset CIRS_HIP_LIB
python - <<'P'
import json
for rep in (1,2):
  for m in ("new","old"):
    c=json.loads(open(f"gpurun_out/s5c/c2_{m}_{rep}.json").read().strip().splitlines()[-1])
    print(rep, m, "c2", round(c["ms_per_step"],4), round(c["value"]), c.get("rollout_only_ms_per_collect"))
P
