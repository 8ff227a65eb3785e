root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/t2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp
timeout 600 rocprofv3 --kernel-trace -d /tmp/rp -o p -- python $root/bench.py --steps 20 --warmup 5 --dropout-redraw --no-probes --no-cpu-baseline --pretrain-steps 5 > /dev/null 2> $root/gpurun_out/t2/err.txt
db=$(find /tmp/rp -name "*.db" | head -1)
python $root/tools/kstats.py $db $root/gpurun_out/t2/rd_kernel_stats.csv 24
