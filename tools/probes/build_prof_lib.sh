#!/bin/bash
# Probe build of the library with stage timestamps inside tracker_step_kernel (-DCIRS_TRK_PROF), head_bwd_fused_kernel (-DCIRS_HEAD_PROF) and the rollout's actor_mass_kernel (-DCIRS_MASS_PROF) -> tools/probes/libcirs_prof.so
set -e
cd "$(dirname "$0")/../.."
OUT=tools/probes/_prof_obj; mkdir -p $OUT
for f in cirs-codes_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" = "tracker" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DCIRS_TRK_PROF ${CIRS_PROF_FLAGS} -c $f -o $OUT/$b.o;
  elif [ "$b" = "ppo" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DCIRS_HEAD_PROF ${CIRS_PROF_FLAGS} -c $f -o $OUT/$b.o;
  elif [ "$b" = "tracker_bwd" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DCIRS_TBWD_PROF ${CIRS_PROF_FLAGS} -c $f -o $OUT/$b.o;
  elif [ "$b" = "rollout" ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -DCIRS_MASS_PROF ${CIRS_PROF_FLAGS} -c $f -o $OUT/$b.o;
  else cp cirs-codes_amd/csrc/_obj/$b.o $OUT/$b.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OUT/*.o -o tools/probes/libcirs_prof.so
echo built tools/probes/libcirs_prof.so
