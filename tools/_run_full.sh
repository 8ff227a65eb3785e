mkdir -p gpurun_out/full
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/full/pytest.txt
cat gpurun_out/full/pytest.txt
