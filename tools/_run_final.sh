mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror|FAILED" | tail -6 > gpurun_out/final/pytest.txt
cat gpurun_out/final/pytest.txt
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/profile_round.sh r06b > gpurun_out/final/profile.log 2>&1
tail -3 gpurun_out/final/profile.log
