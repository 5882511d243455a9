run() { env "$@" python bench.py $FLAGS --no-secondary --no-cpu-baseline --no-box --no-kernel-events --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$FLAGS] [$*]', d['value'], d['ms_per_step'], d['ms_per_step_min_max'])"; }
for i in 1 2; do
FLAGS="" run A=1
for d in 0 1 2 4; do FLAGS="--force-dp" run DISYOLO_DP_SWEEP_DELAY=$d; done
FLAGS="--force-dp" run DISYOLO_DP_SWEEP_DELAY=0 DISYOLO_OPT_CHUNK_M=4.5
FLAGS="" run DISYOLO_OPT_CHUNK_M=4.5
done
