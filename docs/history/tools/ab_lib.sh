#!/bin/bash
# same-box A/B of two builds: picasso_amd/libpicasso_hip_old.so against the tree's library
for i in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export PICASSO_AMD_LIB=$PWD/picasso_amd/libpicasso_hip_old.so; else unset PICASSO_AMD_LIB; fi
  echo "== $lib"
  python3 bench.py --allow-env --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], 'strict', d['ms_per_step_strict'])"
  python3 tools/time_mle_eps.py 2>&1 | grep '"eps": 0.0001' | head -1 | cut -c1-60
  python3 tools/bench_configs.py --only 5 --cpu-seconds 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5', d['ms_per_step'])"
done; done
