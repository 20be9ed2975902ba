#!/usr/bin/env python3
"""Summarise a rocprofv3 run (rocpd .db or *_kernel_stats.csv) as a text table.

    python tools/rocprof_summary.py gpurun_out/prof1 > profiles/r01_xxx_kernel_stats.txt
"""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
    return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in rows]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main():
    root = sys.argv[1]
    rows = []
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    csvs = glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)
    if csvs:
        rows = from_csv(csvs[0])
        src = csvs[0]
    elif dbs:
        rows = from_db(dbs[0])
        src = dbs[0]
    else:
        sys.exit("no rocprofv3 output found under " + root)
    print(f"# rocprofv3 --kernel-trace --stats summary ({os.path.basename(src)}); durations in microseconds")
    print(f"{'kernel':100s} {'calls':>7s} {'total_us':>14s} {'avg_us':>12s} {'pct':>7s}")
    for name, calls, total, avg, pct in sorted(rows, key=lambda r: -r[2])[:25]:
        print(f"{name[:100]:100s} {calls:7d} {total:14.1f} {avg:12.1f} {pct:7.2f}")


if __name__ == "__main__":
    main()
