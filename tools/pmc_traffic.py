#!/usr/bin/env python3
"""Turn the counter_collection.csv files of tools/pmc_quick.sh passes into profiles/<name>.json
(HBM-side bytes per launch of the identify scan kernel, gfx950 corrections of MI355X_MICROARCH.md).
usage: tools/pmc_traffic.py <out.json> <frames> <height> <width> <box> <pass dir> [<pass dir> ...]"""
import collections, csv, glob, json, os, re, sys

out, F, H, W, box = sys.argv[1], *map(int, sys.argv[2:6])
rx = re.compile("identify_scan")
ctr, kname = {}, None
for d in sys.argv[6:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if rx.search(r["Kernel_Name"]):
                kname = r["Kernel_Name"]
                per[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        byc = collections.defaultdict(list)
        for (c, _), v in per.items():
            byc[c].append(v)
        for c, v in byc.items():
            ctr[c] = sum(v) / len(v)
rd = ctr.get("TCC_EA0_RDREQ_sum")
rd32 = ctr.get("TCC_EA0_RDREQ_32B_sum", 0.0)
fetch_kb = ctr.get("FETCH_SIZE")
read_bytes = None
if rd is not None:
    read_bytes = (rd - rd32) * 128 + rd32 * 32          # requests are 128 B unless counted as 32 B
elif fetch_kb is not None:
    read_bytes = 2 * fetch_kb * 1024                     # FETCH_SIZE counts 64 B per 128-B request on gfx950
write_bytes = ctr.get("WRITE_SIZE", 0.0) * 1024
alg = F * H * W * 2
prog = os.environ.get("PMC_COMMAND", "python3 tools/time_identify.py %d %d 1" % (F, box))
rec = {"kernel": kname, "frames": F, "height": H, "width": W, "box": box,
       "defer": int(os.environ.get("PMC_DEFER", "0")),      # 1: the launches of the fused path, exact stage left to the fit's start-value kernel
       "command": "rocprofv3 --kernel-trace --pmc <set> --kernel-include-regex identify_scan -- %s   "
                  "(one --pmc pass per counter set, tools/pmc_quick.sh)" % prog,
       "counters_per_launch": ctr,
       "hbm_read_bytes_per_launch": int(read_bytes) if read_bytes is not None else None,
       "hbm_write_bytes_per_launch": int(write_bytes),
       "algorithmic_bytes_per_launch": alg,
       "traffic_over_algorithmic": (read_bytes + write_bytes) / alg if read_bytes is not None else None,
       "note": "FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md): bytes = 2 x FETCH_SIZE x 1024 "
               "= RDREQ x 128.  Counted at the L2-fabric interface: requests served by the Infinity Cache are included."}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
