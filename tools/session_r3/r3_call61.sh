timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
bash tools/profile_round.sh r03f "round 3 (f): final code of the round" > gpurun_out/r03f_tail.txt 2>&1
