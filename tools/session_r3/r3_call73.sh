SECONDS=0
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -2
echo "smoke took $SECONDS s"
SECONDS=0
python bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['n_gpus'], d['steps'], d['warmup'], d['dtype'], d['scaling'], d['vs_baseline']); print(sorted(d.keys()))"
echo "default bench took $SECONDS s"
