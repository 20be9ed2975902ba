#!/bin/bash
# PMC passes (instruction mix) for one fit kernel regex on resident spots.  usage: tools/pmc_fit.sh <outdir> <regex> <program> [args...]
OUT=$1; RX=$2; shift 2
mkdir -p $OUT
timeout 300 bash tools/pmc_quick.sh $OUT/sq "$RX" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "$@" > $OUT/sq.txt 2>&1
timeout 300 bash tools/pmc_quick.sh $OUT/sq2 "$RX" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "$@" > $OUT/sq2.txt 2>&1
timeout 300 bash tools/pmc_quick.sh $OUT/sq3 "$RX" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH" "$@" > $OUT/sq3.txt 2>&1
cat $OUT/sq.txt $OUT/sq2.txt $OUT/sq3.txt
