timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
bash tools/profile_round.sh r03g "round 3 (g): final code of the round" > gpurun_out/r03g_tail.txt 2>&1
