mkdir -p gpurun_out/s2i
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
rm -rf /tmp/prof_c2
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_c2 -o p -- python $root/bench.py --workload c2 --no-probes --no-cpu-baseline > $root/gpurun_out/s2i/c2_prof.json 2> $root/gpurun_out/s2i/c2_prof.err
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python $root/tools/kstats.py $db $root/gpurun_out/s2i/c2_kernel_stats.csv 60 > $root/gpurun_out/s2i/kstats_c2.txt
python $root/tools/step_timeline.py $db $root/gpurun_out/s2i/c2_step_timeline.md > /dev/null
cd $root
python bench.py --workload c2 --no-probes --no-cpu-baseline > gpurun_out/s2i/c2.json 2>> gpurun_out/s2i/c2_prof.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2i/c2.json").read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], d["value"], d["steps"], d["warmup"])
P
