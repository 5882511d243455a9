timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed" | tail -2
bash tools/profile_round.sh r03h "round 3 (h): final code of the round" > gpurun_out/r03h_tail.txt 2>&1
python tools/infer_layers.py 32 576 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r03h_infer_layers_b32.txt
