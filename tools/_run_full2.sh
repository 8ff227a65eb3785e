mkdir -p gpurun_out/full2
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -8 > gpurun_out/full2/pytest.txt
cat gpurun_out/full2/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -2
cat gpurun_out/draw_margin_stats.json
