run() { env "$@" python bench.py $FLAGS --no-secondary --no-cpu-baseline --no-box --no-kernel-events --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$FLAGS] [$*]', d['value'], d['ms_per_step'])"; }
for q in 8 12 8 12; do
FLAGS="" run GPU_MAX_HW_QUEUES=$q
FLAGS="--force-dp" run GPU_MAX_HW_QUEUES=$q
FLAGS="--task infer --batch 32 --steps 10 --warmup 3" run GPU_MAX_HW_QUEUES=$q
done
