timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -k "stream" 2>&1 | tail -12
python tools/infer_layers.py 32 576 2>&1 | grep -E "^ +(3|6|8|77|79|80|82) |^sum"
python tools/infer_layers.py 8 576 2>&1 | grep -E "^ +(3|6|8|77|79|80|82) |^sum"
