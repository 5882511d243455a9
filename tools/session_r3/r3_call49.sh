timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
bash tools/profile_round.sh r03d "round 3 (d): fused conv1+2, residual blocks 1-3, mask head; BN finalize operand prefetch" > gpurun_out/r03d_tail.txt 2>&1
python tools/infer_layers.py 32 576 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r03d_infer_layers_b32.txt
tail -20 gpurun_out/r03d_infer_layers_b32.txt
