mkdir -p gpurun_out/r3
B="python bench.py --no-kernel-events --no-secondary --no-cpu-baseline --steps 20 --warmup 5 --repeats 5"
run() { tag=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  out=$(env "${envs[@]}" $B "$@" 2>gpurun_out/r3/err_$tag.txt | tail -1)
  echo "$tag $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["ms_per_step_min_max"], d["config"].get("loss_first"), d["config"].get("loss_last"))')"
}
for st in 1 2; do
  $B --stage $st --tune-cache gpurun_out/r3/tune_s$st.json > /dev/null 2>&1
  run s$st-old DISYOLO_HEADS_SIDE=0 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-new DISYOLO_HEADS_SIDE=1 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-old2 DISYOLO_HEADS_SIDE=0 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
  run s$st-new2 DISYOLO_HEADS_SIDE=1 -- --stage $st --tune-cache gpurun_out/r3/tune_s$st.json
done
timeout 1500 python -m pytest tests/test_gpu_net.py tests/test_gpu_configs.py tests/test_gpu_canary.py tests/test_gpu_dp2.py -x -q 2>&1 | tail -5
