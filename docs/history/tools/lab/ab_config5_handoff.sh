#!/bin/bash
# config 5 (13x13 MLE on 50 000 frames) with and without the pixel hand-off, alternating: bash tools/lab/ab_config5_handoff.sh <out>
OUT=${1:-gpurun_out/r06c/config5_handoff.txt}; export TMPDIR=/tmp; mkdir -p $(dirname $OUT); : > $OUT
for h in 0 1 0 1; do
  python3 tools/bench_configs.py --only 5 --no-lq3d --cpu-seconds 0 --handoff $h 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('handoff $h', 'ms_per_step', round(d['ms_per_step'], 3), json.dumps(d.get('roofline', {}).get('kernels', d.get('kernels', {}))))" >> $OUT
done
cat $OUT
