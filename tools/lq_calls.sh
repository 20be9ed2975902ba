#!/bin/bash
# per-dispatch timeline of the strict gausslq fit at the boxes given (default 7)
OUT=${OUT:-gpurun_out/r06c}; mkdir -p $OUT; export TMPDIR=/tmp; PD=$(mktemp -d /tmp/prof_XXXXXX)    # (a box may be one an earlier call left its /tmp on)
for b in ${BOXES:-7}; do
(cd /tmp && rocprofv3 --kernel-trace --stats -d $PD -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $b > $PD.log 2>&1)
grep -E "^N=|mean nfev|second pass" $PD.log
python3 tools/rocprof_calls.py $PD > $OUT/lq_box${b}_calls.txt 2>&1
n=$(grep -n "lq_init_kernel" $OUT/lq_box${b}_calls.txt | tail -2 | head -1 | cut -d: -f1)
tail -n +$n $OUT/lq_box${b}_calls.txt | cut -c1-100 | head -60
done
