timeout 600 python tools/bench_conv.py 8 576 21,6,0x206,0x204,0x202,2 2>&1 | grep -E ", 1, 1\)" | grep -E "^\((288|144)"
timeout 600 python tools/bench_conv.py 32 576 21,6,0x206,0x204,0x202,2 2>&1 | grep -E ", 1, 1\)" | grep -E "^\((288|144)"
