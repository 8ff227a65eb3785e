#!/bin/bash
# builds library variants of csrc/ppo.hip with different -D knobs -> tools/probes/ab/<name>.so   usage: build_ab.sh name "-DX=1 -DY=0" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/probes/ab
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 $flags -c cirs-codes_amd/csrc/ppo.hip -o tools/probes/ab/ppo_$name.o
  objs=$(ls cirs-codes_amd/csrc/_obj/*.o | grep -v "/ppo.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/probes/ab/ppo_$name.o -o tools/probes/ab/$name.so
  echo built $name
done
