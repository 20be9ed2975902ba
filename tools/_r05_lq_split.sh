#!/bin/bash
# per-kernel split of the strict gausslq fit at the boxes given (default 3 9)
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp
for b in ${BOXES:-3 9}; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lq$b -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $b > /tmp/prof_lq$b.log 2>&1)
  python3 tools/rocprof_summary.py /tmp/prof_lq$b > $OUT/lq_box${b}_kernel_stats.txt
  tail -5 /tmp/prof_lq$b.log | grep -v amdgpu > $OUT/lq_box${b}.log
  cat $OUT/lq_box${b}.log; head -14 $OUT/lq_box${b}_kernel_stats.txt | cut -c1-70,100-160
done
