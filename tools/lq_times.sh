#!/bin/bash
# per-box times of the least-squares fit (strict), 2^20 resident spots: bash tools/lq_times.sh <out file> [boxes...]
OUT=$1; shift; BOXES=${@:-3 5 7 9 11 13 15 21}
export TMPDIR=/tmp
for b in $BOXES; do python3 tools/time_gausslq.py 1048576 $b 2>&1 | grep "^N=" | tail -1 | sed "s/^/[strict] /"; done | tee $OUT
