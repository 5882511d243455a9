# A/B of two builds inside ONE gpurun call (boxes differ by >10 %): prev = tools/bin/libdisyolo_prev.so
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events $@"
for i in 1 2; do
  echo -n "prev: "; DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_prev.so python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo -n "new:  "; python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
