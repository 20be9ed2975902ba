#!/bin/bash
# One PMC pass for one kernel regex (kernel-trace only).  Keep the sets small and each pass under `timeout`: passes with
# FETCH_SIZE together with other TCC counters, or with TCC_HIT / TCC_MISS / TCC_REQ, hung on this pool (the whole timeout is charged).  Usage: tools/pmc_quick.sh <outdir> <regex> "<counters>" <program> [args...]
set -u
OUT=$(realpath -m "$1"); KREGEX=$2; CTRS=$3; shift 3
mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $CTRS --kernel-include-regex "$KREGEX" --output-format csv -d "$OUT" -- "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" "$KREGEX" <<'PY'
import csv, glob, re, sys, collections
out, rx = sys.argv[1], re.compile(sys.argv[2])
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    byc = collections.defaultdict(list)
    for (c, d), v in per.items(): byc[c].append(v)
    for c, v in sorted(byc.items()): print(f"{c:32s} {sum(v)/len(v):18.1f}  n={len(v)}")
PY
