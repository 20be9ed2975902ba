"""TOOL: per-launch durations of the least-squares kernels of one pmi_gausslq_dev call, from a rocprofv3 kernel trace.
usage: cd /tmp; rocprofv3 --kernel-trace --output-format csv -d /tmp/lqtrace -- python3 /root/repo/tools/time_gausslq.py
       python3 tools/trace_lq_rounds.py /tmp/lqtrace"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last call: everything after the last lq_init launch
starts = [i for i, r in enumerate(rows) if "lq_init" in r[2]]
# a call has one lq_init per pass; take the launches from the third-last init on (pass 0 and pass 1 of the last call)
first = starts[-2] if len(starts) >= 2 else starts[-1]
t0 = rows[first][0]
tot = {}
for s, e, n in rows[first:]:
    short = n.split("(")[0].replace("void pmi::lq::", "")[:60]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {short}")
    tot[short] = tot.get(short, 0) + (e - s) / 1e3
print("totals:", {k: round(v, 1) for k, v in tot.items()}, "span", (rows[-1][1] - t0) / 1e3)
