mkdir -p gpurun_out/r06d
rocprofv3 -L 2>/dev/null | grep -i -o "SQC\?_[A-Z_]*\(ICACHE\|IFETCH\|INST_CACHE\)[A-Z_]*" | sort -u > gpurun_out/r06d/icache_counters.txt
cat gpurun_out/r06d/icache_counters.txt
python tools/pmc_rollout.py r06d_sq SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_SMEM c3 2>&1 | tail -12
python tools/pmc_rollout.py r06d_ic SQ_IFETCH,SQ_WAVE_CYCLES,SQ_INSTS_VMEM_RD,SQ_INSTS_LDS,SQ_INSTS_BRANCH,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_VMEM c3 2>&1 | tail -12
