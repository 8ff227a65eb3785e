mkdir -p gpurun_out/s3b
for rep in 1 2; do
for mode in both sortold gaeold; do
  unset CIRS_EMB_SORT_RADIX CIRS_GAE_DIRECT_STORES
  [ $mode = sortold ] && export CIRS_EMB_SORT_RADIX=1
  [ $mode = gaeold ] && export CIRS_GAE_DIRECT_STORES=1
  python bench.py --workload c2 --no-probes --no-cpu-baseline --steps 150 --warmup 150 > gpurun_out/s3b/c2_${mode}_$rep.json 2>> gpurun_out/s3b/err.txt
done; done
unset CIRS_EMB_SORT_RADIX CIRS_GAE_DIRECT_STORES
export TMPDIR=/tmp; root=$(pwd); cd /tmp; rm -rf /tmp/prof_c2
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_c2 -o p -- python $root/bench.py --workload c2 --no-probes --no-cpu-baseline > /dev/null 2>> $root/gpurun_out/s3b/err.txt
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python $root/tools/kstats.py $db $root/gpurun_out/s3b/c2_kernel_stats.csv 60 > /dev/null
cd $root
python - <<'P'
import json,csv
for rep in (1,2):
  for mode in ("both","sortold","gaeold"):
    c=json.loads(open(f"gpurun_out/s3b/c2_{mode}_{rep}.json").read().strip().splitlines()[-1])
    print(rep, mode, "c2", round(c["ms_per_step"],4), round(c["value"]))
for r in csv.reader(open("gpurun_out/s3b/c2_kernel_stats.csv")):
    if any(k in r[0] for k in ("small_sort","gae_kernel","pack_tracker","prepare_tail","emb_subrun","emb_segment")): print(r[0][:50], r[1], r[3])
P
