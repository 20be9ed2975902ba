#!/bin/bash
# same-box A/B of a tuning variable on the tuning build (picasso_amd/libpicasso_hip_tuning.so: make TUNING=1 objects): usage
# tools/ab_env.sh VAR VALUE
export PICASSO_AMD_LIB=$PWD/picasso_amd/libpicasso_hip_tuning.so
for i in 1 2; do
for v in off on; do
  if [ $v = on ]; then export $1=$2; else unset $1; fi
  echo "== $1 $v"
  python3 bench.py --allow-env --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], 'strict', d['ms_per_step_strict'])"
  python3 tools/time_mle_eps.py 2>&1 | grep '"eps": 0.0001' | head -1 | cut -c1-60
  python3 tools/bench_configs.py --only 5 --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5', d['ms_per_step'])"
done; done
