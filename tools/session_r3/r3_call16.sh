export DISYOLO_LIB=$GRAFT_REPO_ROOT/tools/bin/libdisyolo_probe.so
for B in 8 32; do
for tile in 20; do
  for pr in 0 0x10000 0x80000 0x90000; do
    echo -n "B=$B tile=$tile 288 32->64 :: "; PROBE=$pr python tools/one_conv.py 288 32 64 3 1 $tile $B 0 2>&1 | grep probe
  done
done
done
