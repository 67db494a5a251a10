# SQ / LDS counters of the config-4 pass with k_exec_c (exec_variant 5) and k_exec_b (2), kernels alone (--no-split)
bash tools/profile_counters.sh r4xc 4 --exec-variant 5 --no-split > gpurun_out/r4xc_counters.txt 2>&1
cp gpurun_out/r4xc_cfg4_sq_counters.csv gpurun_out/r4xc_sq.csv
bash tools/profile_counters.sh r4xb 4 --exec-variant 2 --no-split > gpurun_out/r4xb_counters.txt 2>&1
cp gpurun_out/r4xb_cfg4_sq_counters.csv gpurun_out/r4xb_sq.csv
