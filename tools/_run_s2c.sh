mkdir -p gpurun_out/s2c
export TMPDIR=/tmp
root=$(pwd)
cd /tmp
rm -rf /tmp/prof_rd
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_rd -o p -- python $root/bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline > $root/gpurun_out/s2c/rd_prof.json 2> $root/gpurun_out/s2c/rd_prof.err
db=$(find /tmp/prof_rd -name "*.db" | head -1)
python $root/tools/kstats.py $db $root/gpurun_out/s2c/rd_kernel_stats.csv 40 > $root/gpurun_out/s2c/kstats.txt
tail -5 $root/gpurun_out/s2c/kstats.txt
