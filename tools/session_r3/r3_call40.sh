bash tools/profile_round.sh r03b "round 3 (b): fused conv1+2, residual block 1, BN finalize operand prefetch" > gpurun_out/r03b_tail.txt 2>&1
python tools/infer_layers.py 32 576 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r03b_infer_layers_b32.txt
tail -22 gpurun_out/r03b_infer_layers_b32.txt
