#!/bin/bash
# A/B on one box: erf / exp of the reference-arithmetic MLE kernel as glibc computes them (default) or the device library's
OUT=${OUT:-gpurun_out/libm}; mkdir -p $OUT
for rep in 1 2; do for lib in device glibc; do
  PMI_MLE_LIBM=$lib python3 bench.py --cpu-seconds 0 --allow-env 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('$lib', 'step %.4f' % d['ms_per_step'], 'strict %.4f' % d['ms_per_step_strict'], 'fit', d['roofline']['kernels'])" | cut -c1-260
done; done | tee $OUT/bench_ab.txt
for lib in device glibc; do echo "== $lib"; PMI_MLE_LIBM=$lib python3 tools/time_mle_eps.py 2>&1 | grep -v amdgpu.ids | head -3 | cut -c1-120; done | tee $OUT/eps_ab.txt
for lib in device glibc; do echo "== $lib"; PMI_MLE_LIBM=$lib python3 tools/bench_configs.py --only 5 --no-lq3d --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['stages_ms'], d['mle']['refit_spots'])"; done | tee $OUT/config5_ab.txt
