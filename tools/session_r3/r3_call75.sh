python tools/tune_split.py 32 576 2>&1 | grep -v amdgpu.ids | tail -12
