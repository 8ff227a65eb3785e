mkdir -p gpurun_out/ab3
for rep in 1 2; do
for mode in new old; do
  if [ $mode = old ]; then export CIRS_PERMS_SEPARATE=1 CIRS_READBACK_COPIES=1 CIRS_ROLLOUT_STEPWISE_RESET=1; else unset CIRS_PERMS_SEPARATE CIRS_READBACK_COPIES CIRS_ROLLOUT_STEPWISE_RESET; fi
  python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-probes > gpurun_out/ab3/a_${mode}_$rep.json 2> gpurun_out/ab3/err.txt
  python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 150 --warmup 150 > gpurun_out/ab3/c2_${mode}_$rep.json 2>> gpurun_out/ab3/err.txt
done; done
python - <<'P'
import json
for rep in (1,2):
  for mode in ("new","old"):
    d=json.loads(open(f"gpurun_out/ab3/a_{mode}_{rep}.json").read().strip().splitlines()[-1]); c=json.loads(open(f"gpurun_out/ab3/c2_{mode}_{rep}.json").read().strip().splitlines()[-1])
    print(rep, mode, "c3", round(d["ms_per_step"],4), "c2", round(c["ms_per_step"],4), round(c["value"]))
P
