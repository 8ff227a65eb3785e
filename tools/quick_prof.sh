#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> gpurun_out/<tag>/kstats.txt + step timeline:  bash tools/quick_prof.sh <tag> [bench flags]
tag=${1:-q}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/qprof
timeout 900 rocprofv3 --kernel-trace -d /tmp/qprof -o p -- python $root/bench.py --no-probes --no-cpu-baseline --steps 50 --warmup 15 "$@" > $out/bench.json 2> $out/prof.err
db=$(find /tmp/qprof -name "*.db" | head -1)
python $root/tools/kstats.py $db $out/kernel_stats.csv 60 > $out/kstats.txt
python $root/tools/step_timeline.py $db $out/step_timeline.md > /dev/null
head -12 $out/step_timeline.md
