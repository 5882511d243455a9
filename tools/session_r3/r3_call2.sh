mkdir -p gpurun_out/r3
timeout 900 python tools/tile_check.py --stage 2 > gpurun_out/r3/tile_check_s2.log 2>&1; tail -5 gpurun_out/r3/tile_check_s2.log
for i in 1 2 3; do
  timeout 300 python tools/nan_bisect.py --stage 2 --steps 4 --tag "G$i" > gpurun_out/r3/bisect_G$i.log 2>&1; tail -1 gpurun_out/r3/bisect_G$i.log
done
