ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events"
run() { python bench.py $ARGS "$@" 2>gpurun_out/ab_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do
echo -n "prev lib: "; DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_prev.so run
echo -n "rules: "; run
echo -n "rules + autotune: "; run --autotune on; grep autotune gpurun_out/ab_err.txt | cut -c1-1500
done
echo "stage 2"
echo -n "prev lib: "; DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_prev.so run --stage 2
echo -n "rules: "; run --stage 2
echo -n "rules + autotune: "; run --autotune on --stage 2; grep autotune gpurun_out/ab_err.txt | cut -c1-2500
