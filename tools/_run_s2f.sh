mkdir -p gpurun_out/s2f
python -m pytest tests/test_gpu_dropout.py -q -x 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -5 > gpurun_out/s2f/pytest.txt
cat gpurun_out/s2f/pytest.txt
python bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > gpurun_out/s2f/rd.json 2> gpurun_out/s2f/rd.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/s2f/rd.json").read().strip().splitlines()[-1])
print("redraw", d["ms_per_step"], d["value"], d.get("rollout_only_ms_per_collect"), d.get("update_only_ms"))
P
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
rm -rf /tmp/prof_rd
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_rd -o p -- python $root/bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > $root/gpurun_out/s2f/rd_prof.json 2> $root/gpurun_out/s2f/rd_prof.err
db=$(find /tmp/prof_rd -name "*.db" | head -1)
python $root/tools/kstats.py $db $root/gpurun_out/s2f/rd_kernel_stats.csv 40 > $root/gpurun_out/s2f/kstats.txt
