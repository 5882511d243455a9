mkdir -p gpurun_out/r3
python tools/infer_layers.py 32 576 > gpurun_out/r3/infer_layers_b32.txt 2>&1; cat gpurun_out/r3/infer_layers_b32.txt
