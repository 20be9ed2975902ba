#!/bin/bash
# strict gausslq: bit-identity tests + timings at boxes 3/5/7 (new column-per-lane kernel) on one box
OUT=gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "gausslq" 2>&1 | tail -15 > $OUT/pytest_lq.txt
for b in 7 5 3; do python3 tools/time_gausslq.py 1048576 $b 2>&1 | grep -v amdgpu.ids | tail -3 | sed "s/^/[box $b] /"; done > $OUT/times.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lq7 -- python3 $OLDPWD/tools/time_gausslq.py 1048576 7 > /tmp/prof_lq7.log 2>&1)
python3 tools/rocprof_summary.py /tmp/prof_lq7 > $OUT/lq_box7_kernel_stats.txt
cat $OUT/pytest_lq.txt $OUT/times.txt; head -8 $OUT/lq_box7_kernel_stats.txt
