#!/bin/bash
# per-kernel split of the strict gausslq fit at the boxes given (default 3 9)
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp; PD=$(mktemp -d /tmp/prof_XXXXXX)    # (a box may be one an earlier call left its /tmp on)
for b in ${BOXES:-3 9}; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $PD -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $b > $PD.log 2>&1)
  python3 tools/rocprof_summary.py $PD > $OUT/lq_box${b}_kernel_stats.txt
  tail -5 $PD.log | grep -v amdgpu > $OUT/lq_box${b}.log
  cat $OUT/lq_box${b}.log; head -14 $OUT/lq_box${b}_kernel_stats.txt | cut -c1-70,100-160
done
