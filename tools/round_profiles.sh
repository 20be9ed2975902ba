#!/bin/bash
# The measurement artefacts of a round on ONE box: bench line, rocprofv3 kernel stats of the same command (default and
# --ranges 1), PMC of the scan as the benchmark launches it (fused path, exact stage deferred) and alone, instruction-mix
# PMC of the Newton loop and of the least-squares kernels, the config-3 / config-5 lines.
# usage (on the GPU box, from the repo root): bash tools/round_profiles.sh <outdir> [round tag]
OUT=${1:-gpurun_out/prof}; TAG=${2:-r06}
rm -rf /tmp/pmc_scan_fused /tmp/pmc_scan_alone /tmp/pmc_g8 /tmp/pmc_g8i /tmp/pmc_lqj /tmp/pmc_lqs /tmp/prof_bench /tmp/prof_bench1 /tmp/prof_c3 /tmp/prof_c5      # (a box may carry an earlier call's /tmp)
mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_scan.sh /tmp/pmc_scan_fused 10000 7 fused > $OUT/${TAG}_scan_pmc.txt 2>&1; cp /tmp/pmc_scan_fused/traffic.json $OUT/${TAG}_identify_pmc.json
cp /tmp/pmc_scan_fused/traffic.json profiles/${TAG}_identify_pmc.json      # bench.py quotes roofline.traffic from this file and checks the kernel
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python3 $OLDPWD/bench.py --cpu-seconds 0 > /tmp/prof_bench.log 2>&1)
python3 tools/rocprof_summary.py /tmp/prof_bench > $OUT/${TAG}_bench_kernel_stats.txt
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_bench1 -- python3 $OLDPWD/bench.py --cpu-seconds 0 --ranges 1 > /tmp/prof_bench1.log 2>&1)
python3 tools/rocprof_summary.py /tmp/prof_bench1 > $OUT/${TAG}_bench_ranges1_kernel_stats.txt
bash tools/pmc_scan.sh /tmp/pmc_scan_alone 10000 7 identify > $OUT/${TAG}_scan_alone_pmc.txt 2>&1; cp /tmp/pmc_scan_alone/traffic.json $OUT/${TAG}_identify_alone_pmc.json
bash tools/pmc_fit.sh /tmp/pmc_g8 "g8_iterate" python3 tools/ab_defer.py 10000 7 1 1 > $OUT/${TAG}_g8_iterate_pmc.txt 2>&1
bash tools/pmc_fit.sh /tmp/pmc_g8i "g8_init" python3 tools/ab_defer.py 10000 7 1 1 > $OUT/${TAG}_g8_init_pmc.txt 2>&1
bash tools/pmc_first.sh /tmp/pmc_lqj "lq_jacobian_w" python3 tools/time_gausslq.py 1048576 7 > $OUT/${TAG}_lq_jacobian_pmc.txt 2>&1
bash tools/pmc_first.sh /tmp/pmc_lqs "lq_step_kernel" python3 tools/time_gausslq.py 1048576 7 > $OUT/${TAG}_lq_step_pmc.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -- python3 $OLDPWD/tools/bench_configs.py --only 3 --cpu-seconds 0 --steps 3 > /tmp/prof_c3.log 2>&1)
python3 tools/rocprof_summary.py /tmp/prof_c3 > $OUT/${TAG}_config3_kernel_stats.txt
python3 tools/bench_configs.py --only 3 > $OUT/${TAG}_config3.jsonl 2> $OUT/config3.err || tail -3 $OUT/config3.err
python3 tools/bench_configs.py --only 5 > $OUT/${TAG}_config5.jsonl 2> $OUT/config5.err || tail -3 $OUT/config5.err
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -- python3 $OLDPWD/tools/bench_configs.py --only 5 --cpu-seconds 0 --steps 3 > /tmp/prof_c5.log 2>&1)
python3 tools/rocprof_summary.py /tmp/prof_c5 > $OUT/${TAG}_config5_kernel_stats.txt
python3 tools/time_mle_eps.py 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_mle_eps.jsonl
PMI_MLE_MODE=strict bash tools/pmc_first.sh /tmp/pmc_strict_$$ "mle_strict_kernel" python3 tools/ab_defer.py 10000 7 1 1 > $OUT/${TAG}_mle_strict_pmc.txt 2>&1     # every spot in the reference's arithmetic
PMI_MLE_LIBM=device PMI_MLE_MODE=strict bash tools/pmc_first.sh /tmp/pmc_strict_dev_$$ "mle_strict_kernel" python3 tools/ab_defer.py 10000 7 1 1 > $OUT/${TAG}_mle_strict_device_libm_pmc.txt 2>&1     # ... with the device library's erf / exp
python3 tools/time_identify_shapes.py 7 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_identify_shapes.txt
bash tools/identify_wide_types.sh $OUT/${TAG}_identify_wide_types_now.txt > /dev/null 2>&1
(for m in refit strict; do for b in 3 5 7 9 11 13 15 21; do PMI_LQ_MODE=$m python3 tools/time_gausslq.py 1048576 $b 2>&1 | grep "^N=" | tail -1 | sed "s/^/[$m] /"; done; done; python3 tools/time_lq_ranges.py 2>&1 | grep -v amdgpu.ids) > $OUT/${TAG}_gausslq_times.txt 2>&1
ls -la $OUT
