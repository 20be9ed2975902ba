#!/usr/bin/env python3
"""Two-stream timeline of ONE default pass of pmi_localize_mle_dev (two frame ranges in flight) from a rocprofv3 --kernel-trace csv:
every launch with its queue, start, duration, and what runs beside it.  usage: python tools/step_timeline2.py <dir> [pass from the end]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "?"))))
rows.sort()
tabs = [i for i, r in enumerate(rows) if "locs_from_fits" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# a pass ends with its two locs_from_fits launches
end = tabs[-1 - 2 * (k - 1)]
begin = tabs[-1 - 2 * k] + 1
seg = rows[begin:end + 1]
t0 = seg[0][0]
queues = sorted({r[3] for r in seg})
for s, e, n, q in seg:
    lane = queues.index(q)
    name = n.replace("void pmi::", "").replace("pmi::", "").split("(")[0][:48]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f}  {'    ' * lane}[q{lane}] {name}")
print(f"pass: {(seg[-1][1] - t0) / 1e3:.1f} us, {len(seg)} launches, queues {queues}")
