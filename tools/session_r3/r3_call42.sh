for b in 8 16; do
python tools/infer_layers.py $b 576 2>&1 | grep -v "amdgpu.ids" > gpurun_out/infer_layers_b$b.txt
python - <<PY
import re
tot=0; n=0
for line in open('gpurun_out/infer_layers_b$b.txt'):
    m=re.match(r'\s*(\d+) \S+ \S+ @\d+\s+([\d.]+)', line)
    if m and int(m.group(1))<=52: tot+=float(m.group(2)); n+=1
print("B=$b: layers 1-52 sum %.1f us (%d layers) -> %.1f us per image"%(tot,n,tot/$b))
PY
tail -18 gpurun_out/infer_layers_b$b.txt | head -2
done
