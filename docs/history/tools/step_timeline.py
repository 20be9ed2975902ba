#!/usr/bin/env python3
"""Timeline of ONE pass of pmi_localize_mle_dev from a rocprofv3 --kernel-trace csv: every launch with its duration and the
idle gap before it.  usage: python tools/step_timeline.py <dir with *_kernel_trace.csv> [pass index from the end, default 1]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# passes are delimited by the scan kernel
starts = [i for i, r in enumerate(rows) if "identify_scan" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = starts[-k - 1] if len(starts) > k else 0
# go back to the first launch of that pass: the memsets in front of the scan
while lo > 0 and rows[lo][0] - rows[lo - 1][1] < 20000 and "locs_from_fits" not in rows[lo - 1][2]:
    lo -= 1
hi = starts[-k] if k > 0 else len(rows)
while hi > lo and rows[hi - 1][0] - rows[hi - 2][1] < 20000 and "locs_from_fits" not in rows[hi - 1][2]:
    hi -= 1
seg = rows[lo:hi]
t0 = seg[0][0]
busy = 0
gap_total = 0
prev_end = t0
for s, e, name in seg:
    gap = max(0, s - prev_end)
    gap_total += gap
    busy += e - max(s, prev_end) if e > prev_end else 0
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {gap / 1e3:6.1f}  {name[:90]}")
    prev_end = max(prev_end, e)
print(f"pass: {(prev_end - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us busy, {gap_total / 1e3:.1f} us idle between launches, {len(seg)} launches")
