#!/bin/bash
# the whole library at a git revision -> tools/probes/ab/<name>.so (for same-box A/B runs: tools/ab_rollout.py, tools/ab_headbwd.py)   usage: build_rev.sh <name> <rev>
set -e
cd "$(dirname "$0")/.."
name=$1; rev=${2:-HEAD}
tmp=$(mktemp -d /tmp/cirs_rev.XXXX)
git archive $rev cirs-codes_amd/csrc include | tar -x -C $tmp
mkdir -p tools/probes/ab $tmp/obj
for f in $tmp/cirs-codes_amd/csrc/*.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -c $f -o $tmp/obj/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $tmp/obj/*.o -o tools/probes/ab/$name.so
rm -rf $tmp
echo built tools/probes/ab/$name.so from $rev
