#!/bin/bash
# config 5: kernel stats of the strict re-fit with and without a tuning variable.  usage: tools/c5_strict_stats.sh VAR VALUE
export TMPDIR=/tmp; export PICASSO_AMD_LIB=$PWD/picasso_amd/libpicasso_hip_tuning.so
for v in off on; do
  if [ $v = on ]; then export $1=$2; else unset $1; fi
  PD=$(mktemp -d /tmp/prof_XXXXXX)
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $PD -- python3 $OLDPWD/tools/bench_configs.py --only 5 --cpu-seconds 0 --steps 3 > $PD.log 2>&1)
  echo "== $1 $v"; python3 tools/rocprof_summary.py $PD | grep -E "mle_strict|g8_iterate|identify_scan" | cut -c1-60,100-150
done
