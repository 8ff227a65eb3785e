mkdir -p gpurun_out/r06l
python -m pytest tests/test_gpu_rollout.py tests/test_gpu_tracker.py tests/test_gpu_dropout.py tests/test_gpu_edges.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r06l/pytest.txt
cat gpurun_out/r06l/pytest.txt
for m in 1 0; do echo "COOP=$m"; CIRS_ROLLOUT_COOP=$m python tools/ab_rollout.py c3 --dropout 0.1 --rounds 1 tools/probes/ab/r06d.so - ; CIRS_ROLLOUT_COOP=$m python tools/ab_rollout.py c2 --dropout 0.1 --rounds 1 - ; done > gpurun_out/r06l/ab.txt 2>&1
cat gpurun_out/r06l/ab.txt
python tools/probes/trk_prof.py c3 2>&1 | tail -23 > gpurun_out/r06l/trk_coop.txt; cat gpurun_out/r06l/trk_coop.txt
CIRS_ROLLOUT_COOP=0 python tools/probes/trk_prof.py c3 2>&1 | tail -19 > gpurun_out/r06l/trk_wave.txt; cat gpurun_out/r06l/trk_wave.txt
