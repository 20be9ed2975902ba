#!/bin/bash
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
for b in 7 5 3; do python3 tools/time_gausslq.py 1048576 $b 2>&1 | grep -v amdgpu.ids | tail -4 | sed "s/^/[box $b] /"; done > $OUT/times.txt 2>&1
cat $OUT/times.txt
(time timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -25) > $OUT/pytest_gpu.txt 2>&1
cat $OUT/pytest_gpu.txt
