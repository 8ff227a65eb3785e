mkdir -p gpurun_out/r06e
python -m pytest tests/test_gpu_head_precision.py tests/test_gpu_learn.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r06e/pytest_bf16.txt
CIRS_HIP_LIB=$PWD/tools/probes/ab/f16.so python -m pytest tests/test_gpu_head_precision.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06e/pytest_f16_precision.txt
cp gpurun_out/parity_margins.json gpurun_out/r06e/margins_bf16.json 2>/dev/null
CIRS_HIP_LIB=$PWD/tools/probes/ab/f16.so python -m pytest tests/test_gpu_learn.py tests/test_gpu_tp_learner.py tests/test_gpu_engine_dp.py -m gpu -q 2>&1 | tail -30 > gpurun_out/r06e/pytest_f16_learn.txt
cp gpurun_out/parity_margins.json gpurun_out/r06e/margins_f16.json 2>/dev/null
cat gpurun_out/r06e/pytest_bf16.txt gpurun_out/r06e/pytest_f16_precision.txt gpurun_out/r06e/pytest_f16_learn.txt
python tools/ab_step_libs.py - tools/probes/ab/f16.so > gpurun_out/r06e/ab_step.txt 2>&1
cat gpurun_out/r06e/ab_step.txt
