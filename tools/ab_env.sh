# sweep an environment variable inside ONE gpurun call: bash tools/ab_env.sh VAR v1 v2 ... [-- bench args]
VAR=$1; shift
VALS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do VALS+=("$1"); shift; done; [ "$1" == "--" ] && shift
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-kernel-events $@"
for rep in 1 2; do for v in "${VALS[@]}"; do
  echo -n "$VAR=$v: "; env $VAR=$v python bench.py $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
