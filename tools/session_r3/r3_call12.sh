B="python bench.py --no-kernel-events --no-secondary --no-cpu-baseline --steps 20 --warmup 5 --repeats 5 --tune-cache none"
for st in 1 2; do
  for i in 1 2; do $B --stage $st 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("stage", d["config"]["stage"][:1], d["value"], d["ms_per_step"], d["config"].get("loss_last"))'; done
done
python bench.py --task infer --batch 32 --steps 10 --warmup 3 --repeats 5 --tune-cache none 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("infer", d["value"], d["ms_per_step"])'
python bench.py --task infer --batch 32 --steps 10 --warmup 3 --repeats 5 --tune-cache none 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("infer", d["value"], d["ms_per_step"])'
