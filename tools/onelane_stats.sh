#!/bin/bash
# stand-alone kernel times: the recorded step on ONE lane under rocprofv3 (no two kernels overlap). usage: tools/onelane_stats.sh <tag> [bench args]
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/ol_$tag
python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $O/ol_tune_$tag.json "$@" > /dev/null 2>&1
DISYOLO_SIDE_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ol_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 4 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary --tune-cache $O/ol_tune_$tag.json "$@" > $O/ol_${tag}_bench.json 2>/dev/null
cp $(find /tmp/ol_$tag -name '*kernel_stats.csv' | head -1) $O/ol_${tag}_kernel_stats.csv
python3 - <<PY
import csv,re,json
rows=list(csv.DictReader(open("$O/ol_${tag}_kernel_stats.csv")))
steps=24
fam={}
for r in rows:
    n=re.sub(r"\(anonymous namespace\)::|void |HIP_vector_type<[^>]*>|\(.*$","",r['Name'])
    k=('conv3x3/1x1 fwd+dgrad (igemm)' if 'conv_igemm' in n else 'conv3x3 fwd+dgrad (halo)' if 'conv_halo' in n else 'wgrad 3x3 tap-fused' if 'wgrad3x3' in n else 'wgrad im2col' if 'conv_wgrad' in n else 'slab_reduce' if 'slab_reduce' in n else 'BN family' if ('bn_' in n or 'colreduce_kernel<1>' in n) else 'adam+pack' if ('adam' in n or 'pack_all' in n) else 'first layer' if 'conv_first' in n else 'detect/loss' if any(x in n for x in ('nms','decode','yolo_loss','mask_rois','psroi','shuffle')) else 'other')
    fam[k]=fam.get(k,0)+float(r['TotalDurationNs'])/steps/1e3
print(json.load(open("$O/ol_${tag}_bench.json"))['ms_per_step'], "ms/step one lane under rocprof")
for k,v in sorted(fam.items(), key=lambda kv:-kv[1]): print(f"{k:36s} {v:8.1f} us/step")
print("sum", sum(fam.values()))
PY
