#!/bin/bash
# per-dispatch timeline of one benchmark step (one range in flight unless RANGES=2)
OUT=gpurun_out/r05s; mkdir -p $OUT; export TMPDIR=/tmp; PD=$(mktemp -d /tmp/prof_XXXXXX)    # (a box may be one an earlier call left its /tmp on)
(cd /tmp && rocprofv3 --kernel-trace --stats -d $PD -- python3 $OLDPWD/bench.py --cpu-seconds 0 --steps 4 --warmup 1 --ranges ${RANGES:-1} > $PD.log 2>&1)
python3 tools/rocprof_calls.py $PD pmi > $OUT/bench_calls.txt 2>&1
n=$(grep -n "identify_scan" $OUT/bench_calls.txt | tail -2 | head -1 | cut -d: -f1)
tail -n +$n $OUT/bench_calls.txt | cut -c1-120 | head -${LINES_N:-40}
