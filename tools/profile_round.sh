#!/bin/bash
# All measurements of a round in one GPU call:  bash tools/profile_round.sh <tag>     -> gpurun_out/<tag>/
#   bench JSON lines (c3, c2, c3 with --dropout 0.1), rocprofv3 kernel-trace summaries of the timed step (eval-mode and dropout-mode)
#   and of the K1-K2 / sweep workload (tools/pmc_workload.py), the phase timeline, the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs).
tag=${1:-r05}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd $root
# (the default line is the reference's training mode: tracker Dropout 0.1, position-keyed masks; --dropout 0 = eval-mode tracker)
python bench.py > $out/${tag}_bench_c3.json 2> $out/bench_c3.err
python bench.py --workload c2 --no-cpu-baseline --no-probes > $out/${tag}_bench_c2.json 2>> $out/bench_c3.err
python bench.py --dropout 0 --no-cpu-baseline --no-probes > $out/${tag}_bench_c3_dropout_off.json 2>> $out/bench_c3.err
python bench.py --dropout 0.1 --dropout-redraw --steps 20 --warmup 5 --no-cpu-baseline --no-probes > $out/${tag}_bench_c3_dropout_redraw.json 2>> $out/bench_c3.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/${tag}_bench_c3_driver_flags.json 2>> $out/bench_c3.err
cd /tmp
for mode in train eval; do
  flags="--no-probes --no-cpu-baseline --steps 50 --warmup 15"; [ $mode = eval ] && flags="$flags --dropout 0"
  rm -rf /tmp/prof_$mode
  timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_$mode -o p -- python $root/bench.py $flags > $out/${tag}_bench_c3_prof_$mode.json 2> $out/prof_$mode.err
  db=$(find /tmp/prof_$mode -name "*.db" | head -1)
  sfx=""; [ $mode = eval ] && sfx="_dropout_off"
  python $root/tools/kstats.py $db $out/${tag}_bench_c3${sfx}_kernel_stats.csv 40 > $out/kstats_$mode.txt
  [ $mode = train ] && python $root/tools/step_timeline.py $db $out/${tag}_step_timeline.md > /dev/null
done
# C2 (steady state) and the exact-redraw mode: kernel stats + timeline of the same commands as their bench lines
rm -rf /tmp/prof_c2
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_c2 -o p -- python $root/bench.py --workload c2 --no-probes --no-cpu-baseline > $out/${tag}_bench_c2_prof.json 2> $out/prof_c2.err
db=$(find /tmp/prof_c2 -name "*.db" | head -1)
python $root/tools/kstats.py $db $out/${tag}_bench_c2_kernel_stats.csv 40 > $out/kstats_c2.txt
python $root/tools/step_timeline.py $db $out/${tag}_c2_step_timeline.md > /dev/null
rm -rf /tmp/prof_rd
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_rd -o p -- python $root/bench.py --dropout 0.1 --dropout-redraw --steps 20 --warmup 5 --no-probes --no-cpu-baseline > $out/${tag}_bench_c3_dropout_redraw_prof.json 2> $out/prof_rd.err
db=$(find /tmp/prof_rd -name "*.db" | head -1)
python $root/tools/kstats.py $db $out/${tag}_bench_c3_dropout_redraw_kernel_stats.csv 40 > $out/kstats_rd.txt
rm -rf /tmp/prof_k12
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_k12 -o p -- python $root/tools/pmc_workload.py > /dev/null 2> $out/prof_k12.err
db=$(find /tmp/prof_k12 -name "*.db" | head -1)
python $root/tools/kstats.py $db $out/${tag}_k1k2_sweep_kernel_stats.csv 40 > $out/kstats_k12.txt
cd $root
python tools/pmc_traffic.py collect $tag > $out/pmc.txt 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/${tag}_pmc_traffic.md $out/ 2>/dev/null
# stage stamps (probe build: bash tools/probes/build_prof_lib.sh before the GPU call)
if [ -f tools/probes/libcirs_prof.so ]; then
  { python tools/probes/head_prof.py; python tools/probes/step_prof.py; python tools/probes/tbwd_prof.py; python tools/probes/attn_prof.py c3; for r in 1 16 30; do python tools/probes/prefix_prof.py $r | grep -v launches; done; } > $out/${tag}_stage_stamps.txt 2> $out/stamps.err
fi
tail -3 $out/kstats_train.txt; head -c 600 $out/${tag}_bench_c3.json
