#!/bin/bash
# round-5 baseline: per-kernel split of the strict gausslq fit at boxes 3, 7, 9 + today's MLE residual distances
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
python3 tools/diag_mle_residuals.py 2>&1 | grep -v amdgpu.ids > $OUT/mle_residuals.txt
for b in 7 3 9; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lq$b -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $b > /tmp/prof_lq$b.log 2>&1)
  python3 tools/rocprof_summary.py /tmp/prof_lq$b > $OUT/lq_box${b}_kernel_stats.txt
  tail -5 /tmp/prof_lq$b.log > $OUT/lq_box${b}.log
done
ls -la $OUT
