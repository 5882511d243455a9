#!/bin/bash
# A/B of whole training steps under different env settings, same box, same tile cache.
# usage: tools/ab_step.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
args=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out
python bench.py $args --no-cpu-baseline --no-kernel-events --tune-cache $O/ab_tune.json > /dev/null 2>&1   # creates the cache
for cfg in "$@"; do
  for rep in 1 2; do
    r=$(env $cfg python bench.py $args --no-cpu-baseline --no-kernel-events --tune-cache $O/ab_tune.json 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "[$cfg] $r"
  done
done
