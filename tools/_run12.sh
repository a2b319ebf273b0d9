bash tools/profile_round.sh r03_b > gpurun_out/prof_r03_b.log 2>&1; grep "^{" gpurun_out/prof_r03_b.log | head -1 | cut -c1-200
grep "weight_layout3\|wgrad_reduce_multi\|conv_wgrad_kernel\|kernel time" gpurun_out/prof_r03_b/r03_b_step_sequence.txt
bash tools/profile_round.sh r03_nerv nerv > gpurun_out/prof_r03_nerv.log 2>&1; grep "^{" gpurun_out/prof_r03_nerv.log | head -1 | cut -c1-200
tail -3 gpurun_out/prof_r03_nerv/r03_nerv_step_sequence.txt
