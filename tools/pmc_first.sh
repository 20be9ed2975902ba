#!/bin/bash
# PMC passes for the FIRST (largest) dispatch matching a kernel regex.  usage: tools/pmc_first.sh <outdir> <regex> <program> [args...]
set -u
OUT=$(realpath -m "$1"); RX=$2; shift 2
mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_IFETCH"; do
  i=$((i+1))
  (timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --kernel-include-regex "$RX" --output-format csv -d "$OUT/p$i" -- "$@" > "$OUT/log$i.txt" 2>&1)
done
python3 - "$OUT" "$RX" <<'PY'
import csv, glob, re, sys, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float); first = None
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            d = int(r["Dispatch_Id"])
            first = d if first is None else min(first, d)
            per[(r["Counter_Name"], d)] += float(r["Counter_Value"])
    for (c, d), v in sorted(per.items()):
        if d == first: print(f"{c:32s} {v:18.1f}  dispatch {d}")
PY
