#!/bin/bash
# per-dispatch timeline of the strict gausslq fit at the boxes given (default 7)
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp
for b in ${BOXES:-7}; do
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lq$b -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $b > /tmp/prof_lq$b.log 2>&1)
grep -E "^N=|mean nfev|second pass" /tmp/prof_lq$b.log
python3 tools/rocprof_calls.py /tmp/prof_lq$b > $OUT/lq_box${b}_calls.txt 2>&1
n=$(grep -n "lq_init_kernel" $OUT/lq_box${b}_calls.txt | tail -2 | head -1 | cut -d: -f1)
tail -n +$n $OUT/lq_box${b}_calls.txt | cut -c1-100 | head -60
done
