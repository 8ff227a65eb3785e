mkdir -p gpurun_out/s3c
for rep in 1 2; do
for cap in 256 128 64; do
  export CIRS_DW_MAX_SLABS=$cap
  python bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-probes > gpurun_out/s3c/a_${cap}_$rep.json 2>> gpurun_out/s3c/err.txt
done; done
python - <<'P'
import json
for rep in (1,2):
  for cap in (256,128,64):
    d=json.loads(open(f"gpurun_out/s3c/a_{cap}_{rep}.json").read().strip().splitlines()[-1])
    print(rep, cap, round(d["ms_per_step"],4), round(d["update_only_ms"],4))
P
