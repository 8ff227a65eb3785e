mkdir -p gpurun_out/s3a
python -m pytest tests/test_gpu_learn.py tests/test_gpu_tracker_bwd.py tests/test_gpu_rollout.py tests/test_gpu_engine_dp.py tests/test_gpu_dropout.py -q -x 2>&1 | grep -E "passed|failed|rror|FAILED|assert" | tail -6 > gpurun_out/s3a/pytest.txt
cat gpurun_out/s3a/pytest.txt
for rep in 1 2; do
for mode in new old; do
  if [ $mode = old ]; then export CIRS_EMB_SORT_RADIX=1 CIRS_GAE_DIRECT_STORES=1; else unset CIRS_EMB_SORT_RADIX CIRS_GAE_DIRECT_STORES; fi
  python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 150 --warmup 150 > gpurun_out/s3a/c2_${mode}_$rep.json 2>> gpurun_out/s3a/err.txt
done; done
unset CIRS_EMB_SORT_RADIX CIRS_GAE_DIRECT_STORES
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probes > gpurun_out/s3a/a.json 2>> gpurun_out/s3a/err.txt
python - <<'P'
import json
for rep in (1,2):
  for mode in ("new","old"):
    c=json.loads(open(f"gpurun_out/s3a/c2_{mode}_{rep}.json").read().strip().splitlines()[-1])
    print(rep, mode, "c2", round(c["ms_per_step"],4), round(c["value"]))
d=json.loads(open("gpurun_out/s3a/a.json").read().strip().splitlines()[-1]); print("drv", d["ms_per_step"], d["value"])
P
