timeout 600 python tools/bench_conv.py 32 576 12,8,0x208,1,0x201,7,0x20c 2>&1 | grep -E ", 3, [12]\)"
timeout 600 python tools/bench_conv.py 8 576 12,8,0x208,1,0x201,7,0x20c 2>&1 | grep -E ", 3, [12]\)"
