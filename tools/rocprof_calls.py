#!/usr/bin/env python3
"""Every kernel dispatch of a rocprofv3 --kernel-trace run in launch order: start offset, duration, grid, name.

    python tools/rocprof_calls.py /tmp/prof_dir [name filter] > calls.txt
"""
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
    if not dbs:
        sys.exit("no rocpd database under " + root)
    cur = sqlite3.connect(dbs[0]).cursor()
    views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('view','table')")]
    src = "kernels" if "kernels" in views else None
    if src is None:
        sys.exit("no `kernels` view; have: " + ", ".join(views))
    cols = [r[1] for r in cur.execute(f"pragma table_info({src})")]
    grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = f"select name, start, end{', ' + grid if grid else ''} from {src} order by start"
    rows = cur.execute(q).fetchall()
    if not rows:
        sys.exit("no dispatches")
    t0 = rows[0][1]
    for r in rows:
        if flt and flt not in r[0]:
            continue
        g = f"{r[3]:>9}" if grid else ""
        print(f"{(r[1] - t0) / 1e3:12.1f} us  {(r[2] - r[1]) / 1e3:10.1f} us {g}  {r[0][:90]}")


if __name__ == "__main__":
    main()
