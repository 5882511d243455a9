timeout 900 python -m pytest tests/test_gpu_conv.py -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -1
for f in 0 1 0 1; do
  DISYOLO_XCD_N=$f python bench.py --no-secondary --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); h=k.get('conv_halo_kernel<8,3,2>',{}); g=k.get('conv_halo_kernel<8,3,4>',{})
print('XCD_N=$f stage1', d['value'], d['ms_per_step'], 'halo<8,3,2>', h.get('avg_us'), h.get('launches_per_step'), 'halo<8,3,4>', g.get('avg_us'))"
done
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for f in 0 1; do
DISYOLO_XCD_N=$f rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pfx$f -- python3 $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-kernel-events --no-secondary > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pfx$f/*/*counter_collection.csv')[0]
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='FETCH_SIZE':
        n=r['Kernel_Name']
        if 'conv_halo_kernel' in n:
            k=n.split('(')[0][-40:]
            acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
for k,(c,v) in acc.items(): print('XCD_N=$f', k, 'launches', c, 'FETCH_SIZE KB per launch %.0f'%(v/c))
PY
done
