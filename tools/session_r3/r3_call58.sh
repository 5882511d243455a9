bash tools/profile_round.sh r03e "round 3 (e): final code of the round" > gpurun_out/r03e_tail.txt 2>&1
python tools/infer_layers.py 32 576 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r03e_infer_layers_b32.txt
tail -3 gpurun_out/r03e_infer_layers_b32.txt | head -1
