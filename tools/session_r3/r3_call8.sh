mkdir -p gpurun_out/r3
B="python bench.py --no-kernel-events --no-secondary --no-cpu-baseline --steps 20 --warmup 5 --repeats 5"
run() { # tag env... -- args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$(env "${envs[@]}" $B "$@" 2>/dev/null | tail -1)
  echo "$tag $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ms_per_step_min_max"], d["config"].get("loss_last"))')"
}
for st in 1 2; do
  $B --stage $st --tune-cache gpurun_out/r3/tune_s$st.json > /dev/null 2>&1
  run s$st-default A=1 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-nowgrad DISYOLO_EXP_SKIP_WGRAD=1 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-onelane DISYOLO_SIDE_LANE=0 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-nooptoverlap DISYOLO_OPT_OVERLAP=0 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-chunk2M DISYOLO_OPT_CHUNK_M=2 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-default2 A=1 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
done
