#!/bin/bash
# PMC passes for the identify scan kernel on config 2 (each pass its own rocprofv3 run, bounded by timeout).
# usage: tools/pmc_scan.sh <outdir> [frames] [box] [identify|fused]
#   identify: pmi_identify_dev alone (the scan with its own exact stage, tools/time_identify.py)
#   fused:    pmi_localize_mle_dev, one frame range, the exact stage deferred to the fit (the benchmark's launches, tools/ab_defer.py)
OUT=${1:-gpurun_out/pmc_scan}; F=${2:-10000}; BOX=${3:-7}; MODE=${4:-identify}
mkdir -p $OUT
if [ "$MODE" = fused ]; then PROG="python3 tools/ab_defer.py $F $BOX 1 1"; export PMC_DEFER=1; else PROG="python3 tools/time_identify.py $F $BOX 1"; export PMC_DEFER=0; fi
export PMC_COMMAND="$PROG"
timeout 300 bash tools/pmc_quick.sh $OUT/rd identify_scan "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" $PROG > $OUT/rd.txt 2>&1
timeout 300 bash tools/pmc_quick.sh $OUT/wr identify_scan "WRITE_SIZE" $PROG > $OUT/wr.txt 2>&1
timeout 300 bash tools/pmc_quick.sh $OUT/sq identify_scan "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" $PROG > $OUT/sq.txt 2>&1
timeout 300 bash tools/pmc_quick.sh $OUT/sq2 identify_scan "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_LDS GRBM_GUI_ACTIVE" $PROG > $OUT/sq2.txt 2>&1
cat $OUT/rd.txt $OUT/wr.txt $OUT/sq.txt $OUT/sq2.txt
python3 tools/pmc_traffic.py $OUT/traffic.json $F 512 512 $BOX $OUT/rd $OUT/wr | tail -12
