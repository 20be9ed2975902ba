#!/bin/bash
# first-launch duration of lq_jacobian_w_kernel under each ablation build in tools/lab/_abl (a measurement aid: the ablated
# builds compute garbage).  usage: bash tools/lab/ablate_run.sh <box> <out>
BOX=${1:-13}; OUT=${2:-gpurun_out/r06b/ablate_$BOX.txt}; export TMPDIR=/tmp; mkdir -p $(dirname $OUT); : > $OUT
for lib in tools/lab/_abl/*.so; do
  PD=$(mktemp -d /tmp/prof_XXXXXX)
  (cd /tmp && PICASSO_AMD_LIB=$OLDPWD/$lib rocprofv3 --kernel-trace -d $PD -- python3 $OLDPWD/tools/time_gausslq.py 1048576 $BOX > $PD.log 2>&1)
  echo "== $lib" >> $OUT
  grep -E "^N=" $PD.log | tail -1 >> $OUT
  python3 tools/rocprof_calls.py $PD lq_jacobian_w | awk '{print $3}' | head -40 | tr '\n' ' ' >> $OUT; echo >> $OUT
done
cat $OUT
