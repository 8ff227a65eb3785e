mkdir -p gpurun_out/r06f
python tools/probes/head_prof.py > gpurun_out/r06f/head_prof.txt 2>&1
head -45 gpurun_out/r06f/head_prof.txt
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r06f/pytest_full.txt
cat gpurun_out/r06f/pytest_full.txt
