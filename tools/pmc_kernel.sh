#!/bin/bash
# PMC counters for ONE kernel (regex) in separate passes; kernel-trace only.
# Usage: tools/pmc_kernel.sh <outdir> <kernel-regex> <program> [args...]   (program = python3 / ELF, no wrappers)
set -u
OUT=$(realpath -m "$1"); KREGEX=$2; shift 2
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
 "SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT/pass$i" -- "$@" > "$OUT/pass$i.log" 2>&1
  i=$((i+1))
done
python3 - "$OUT" "$KREGEX" <<'PY'
import csv, glob, re, sys, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if rx.search(r["Kernel_Name"]):
                per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    byc = collections.defaultdict(list)
    for (c, d), v in per.items():
        byc[c].append(v)
    for c, v in byc.items():
        agg[c] = (sum(v) / len(v), len(v), min(v), max(v))
with open(out + "/summary.txt", "w") as fh:
    fh.write(f"# PMC per dispatch (mean over dispatches) for kernels matching /{sys.argv[2]}/\n")
    for c, (v, n, lo, hi) in agg.items():
        fh.write(f"{c:32s} {v:18.1f}   (n={n}, min {lo:.1f}, max {hi:.1f})\n")
print(open(out + "/summary.txt").read())
PY
