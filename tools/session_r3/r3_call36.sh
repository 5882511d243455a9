python tools/infer_layers.py 32 576 2>&1 | grep -v "amdgpu.ids" > gpurun_out/infer_layers_b32_after.txt
tail -20 gpurun_out/infer_layers_b32_after.txt
python bench.py --no-secondary --no-cpu-baseline --task infer --batch 32 2>/dev/null | tail -1
python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1
