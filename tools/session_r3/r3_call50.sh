COLD=1 python tools/bench_conv.py 32 576 1,8,12,0x20c,16,3,10,0x201,0x208 2>&1 | grep -v amdgpu.ids | grep -E "shape|, 3, 1|, 3, 2" | head -40
