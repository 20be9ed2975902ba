#!/bin/bash
# per-dispatch timeline of one config-5 step (MLE route): launch order, start offset, duration of every kernel
OUT=${OUT:-gpurun_out/c5t}; mkdir -p $OUT; export TMPDIR=/tmp; PD=$(mktemp -d /tmp/prof_XXXXXX)
(cd /tmp && rocprofv3 --kernel-trace -d $PD -- python3 $OLDPWD/tools/bench_configs.py --only 5 --no-lq3d --cpu-seconds 0 --steps 2 > $PD.log 2>&1)
tail -2 $PD.log | cut -c1-400
python3 tools/rocprof_calls.py $PD > $OUT/config5_calls_all.txt 2>&1
# the last step: from the last but one range_rows / scan launch on
n=$(grep -n "identify_scan_u16_fast_kernel" $OUT/config5_calls_all.txt | tail -2 | head -1 | cut -d: -f1)
tail -n +$n $OUT/config5_calls_all.txt | grep -v "at::\|rocclr" | cut -c1-140 > $OUT/config5_calls.txt
wc -l $OUT/config5_calls.txt
