export DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_probe.so
for shape in "36 256 512 3 1 16" "72 128 256 3 1 16" "144 64 128 3 1 16" "18 512 1024 3 1 17"; do
  for pr in 0 0x4000 0x10000 0x14000 0x80000 0x94000; do
    echo -n "$shape :: "; PROBE=$pr python tools/one_conv.py $shape 8 0 2>&1 | grep probe
  done
done
