export DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_probe.so
for shape in "36 512 256 1 1 6" "72 256 128 1 1 0x202" "18 1024 512 1 1 6" "72 128 256 1 1 3"; do
  for pr in 0 0x100000 0x94000 0x14000 0x80000 0x10000; do
    echo -n "$shape :: "; PROBE=$pr python tools/one_conv.py $shape 8 0 2>&1 | grep probe
  done
done
