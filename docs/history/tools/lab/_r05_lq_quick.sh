#!/bin/bash
# quick A/B of the strict gausslq fit: bit-identity tests of the strict mode, timings, per-kernel stats at 7x7
OUT=gpurun_out/r05q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "gausslq_strict or gausslq_all_boxes" 2>&1 | tail -3
for b in ${BOXES:-7 5 3}; do python3 tools/time_gausslq.py 1048576 $b 2>&1 | grep -v amdgpu.ids | tail -4 | head -2 | sed "s/^/[box $b] /"; done
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_lq7 -- python3 $OLDPWD/tools/time_gausslq.py 1048576 ${PBOX:-7} > /tmp/prof_lq7.log 2>&1)
python3 tools/rocprof_summary.py /tmp/prof_lq7 | head -8 | cut -c1-60,100-150
