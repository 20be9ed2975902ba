#!/bin/bash
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp
b=${BOX:-3}
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lq$b -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $b > /tmp/prof_lq$b.log 2>&1)
grep -E "^N=|mean nfev|second pass" /tmp/prof_lq$b.log
python3 tools/rocprof_calls.py /tmp/prof_lq$b > $OUT/lq_box${b}_calls.txt 2>&1
wc -l $OUT/lq_box${b}_calls.txt; tail -110 $OUT/lq_box${b}_calls.txt | cut -c1-110
