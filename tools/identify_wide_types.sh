#!/bin/bash
# float32 / 32-bit integer movies through identify at boxes 7 ... 17 and frames up to 2048^2 (the shapes round 5 did not time)
# usage: bash tools/identify_wide_types.sh <out file>
OUT=${1:-gpurun_out/identify_wide.txt}; export TMPDIR=/tmp; mkdir -p $(dirname $OUT)
(for b in 7 9 11 13 15 17; do python3 tools/time_identify_shapes.py $b 512 512 "float32 x1.37" 2>&1 | grep -v amdgpu.ids | sed "s/^/[box $b] /"; done
 for b in 7 13; do for hw in 1024 2048; do python3 tools/time_identify_shapes.py $b $hw $hw "float32 x1.37" 2>&1 | grep -v amdgpu.ids | sed "s/^/[box $b] /"; done; done
 for b in 7 13; do python3 tools/time_identify_shapes.py $b 512 512 "float32" 2>&1 | grep -v amdgpu.ids | sed "s/^/[box $b] /"; done
 for b in 7 13; do for dt in "int32" "int32 x70000"; do python3 tools/time_identify_shapes.py $b 512 512 "$dt" 2>&1 | grep -v amdgpu.ids | sed "s/^/[box $b] /"; done; done) | tee $OUT
