#!/bin/bash
# kernel trace of the recorded training step + per-lane timeline (GPU box). usage: tools/step_timeline.sh <tag> [bench args]
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_$tag -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $O/ab_tune.json "$@" > $O/tl_${tag}_bench.json 2>/dev/null
f=$(find /tmp/tl_$tag -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/timeline.py $f $O/tl_${tag}_step.txt > $O/tl_${tag}.txt 2>&1
cp $(find /tmp/tl_$tag -name '*kernel_stats.csv' | head -1) $O/tl_${tag}_kernel_stats.csv
cat $O/tl_${tag}.txt
