#!/bin/bash
# Collect PMC counters for a command in separate passes (kernel-trace only, no other trace
# domains).  Usage: tools/pmc_run.sh <outdir> <kernel-regex> -- <program> [args...]
# (the program must be python3/an ELF itself: no env/bash wrappers under rocprofv3)
set -u
OUT=$1; KREGEX=$2; shift 3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
 "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
 "FETCH_SIZE"
 "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"
 "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PERF_SEL_TOTAL_READ"
)
i=0
for P in "${PASSES[@]}"; do
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT/pass$i" -- "$@" > "$OUT/pass$i.log" 2>&1
  i=$((i+1))
done
python3 - "$OUT" "$KREGEX" <<'PY'
import csv, glob, re, sys, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
agg = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if rx.search(r["Kernel_Name"]):
                per[(r["Counter_Name"], r["Dispatch_Id"])].append(float(r["Counter_Value"]))
    byc = collections.defaultdict(list)
    for (c, d), v in per.items():
        byc[c].append(sum(v))
    for c, v in byc.items():
        agg[c] = (sum(v) / len(v), len(v))
with open(out + "/summary.txt", "w") as fh:
    fh.write(f"# PMC per dispatch (mean over dispatches) for kernels matching /{sys.argv[2]}/\n")
    for c, (v, n) in agg.items():
        fh.write(f"{c:40s} {v:20.1f}   (n={n})\n")
print(open(out + "/summary.txt").read())
PY
