mkdir -p gpurun_out/r06g
python -m pytest tests/test_gpu_policy.py tests/test_gpu_rollout.py tests/test_gpu_edges.py tests/test_gpu_sharded.py tests/test_gpu_distributed.py tests/test_gpu_dropout.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r06g/pytest.txt
cat gpurun_out/r06g/pytest.txt
python tools/ab_rollout.py c3 --dropout 0.1 --rounds 1 tools/probes/ab/r06b.so - > gpurun_out/r06g/ab_c3_drop.txt 2>&1
python tools/ab_rollout.py c2 --dropout 0.1 --rounds 1 tools/probes/ab/r06b.so - > gpurun_out/r06g/ab_c2.txt 2>&1
cat gpurun_out/r06g/ab_*.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06g/bench_driver_flags.json 2> gpurun_out/r06g/bench.err
python - <<'PY'
import json
z=json.load(open('gpurun_out/r06g/bench_driver_flags.json'))
print(json.dumps(z["summary"], indent=0))
print("roofline", z["roofline"]["frac"], z["roofline"]["seconds_per_launch"], "mbstep", z["minibatch_step"]["seconds"])
PY
