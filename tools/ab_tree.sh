# same-box A/B of two whole trees: the working tree against tools/tmp/<name> (e.g. the previous round's commit exported by
# `git worktree` + make); usage (GPU box): bash tools/ab_tree.sh r04 "<bench args>"
P=$GRAFT_REPO_ROOT/tools/tmp/$1; shift
for i in 1 2; do
  for t in prev new; do
    if [ $t = prev ]; then cd $P; else cd $GRAFT_REPO_ROOT; fi
    python bench.py $@ 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['config']['workload'], d['value'], d['ms_per_step'])"
  done
done
