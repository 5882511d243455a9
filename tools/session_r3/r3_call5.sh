mkdir -p gpurun_out/r3
bash tools/onelane_stats.sh r3a_s1 --stage 1 > gpurun_out/r3/onelane_s1.txt 2>&1
bash tools/onelane_stats.sh r3a_s2 --stage 2 > gpurun_out/r3/onelane_s2.txt 2>&1
cat gpurun_out/r3/onelane_s1.txt gpurun_out/r3/onelane_s2.txt
bash tools/infer_profile.sh r3a > gpurun_out/r3/infer_prof.txt 2>&1; tail -40 gpurun_out/r3/infer_prof.txt
