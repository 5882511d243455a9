mkdir -p gpurun_out/r3
for cfgs in "A:1:1" "B:0:1" "C:1:0" "D:0:0"; do
  IFS=: read tag bn sc <<< "$cfgs"
  DISYOLO_BN_FUSE=$bn DISYOLO_SHORTCUT_FUSE=$sc timeout 300 python tools/nan_bisect.py --stage 2 --steps 70 --tag "$tag(bn=$bn,sc=$sc)" > gpurun_out/r3/bisect_$tag.log 2>&1
  tail -2 gpurun_out/r3/bisect_$tag.log
done
DISYOLO_BN_FUSE=1 timeout 300 python tools/nan_bisect.py --stage 2 --steps 70 --tune-cache profiles/r02c_tune_cache.json --tag "E(cache)" > gpurun_out/r3/bisect_E.log 2>&1; tail -2 gpurun_out/r3/bisect_E.log
DISYOLO_BN_FUSE=1 timeout 300 python tools/nan_bisect.py --stage 2 --steps 70 --autotune off --tag "F(notune)" > gpurun_out/r3/bisect_F.log 2>&1; tail -2 gpurun_out/r3/bisect_F.log
