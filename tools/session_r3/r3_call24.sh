timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
for t in force202 force202; do
python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache tools/tune_$t.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new $t', d['value'], d['ms_per_step'])"
done
export DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_prev.so
for t in force202 force202; do
python bench.py --steps 20 --warmup 5 --repeats 5 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache tools/tune_$t.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('prev $t', d['value'], d['ms_per_step'])"
done
