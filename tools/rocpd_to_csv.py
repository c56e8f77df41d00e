#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (the default output format of ROCm 7.2) into the two CSV
summaries kept under profiles/: the --stats kernel table and the --pmc counter collection.

usage: rocpd_to_csv.py stats   <results.db> <kernel_stats.csv>
       rocpd_to_csv.py counter <results.db> <counter_collection.csv>"""
import csv
import math
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    per = {}
    for name, dur in c.execute("select name, duration from kernels"):
        per.setdefault(name, []).append(dur)
    total = sum(sum(v) for v in per.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            sd = math.sqrt(sum((x - avg) ** 2 for x in v) / (len(v) - 1)) if len(v) > 1 else 0.0
            w.writerow([name, len(v), sum(v), round(avg, 3), round(100.0 * sum(v) / total, 4), min(v), max(v), round(sd, 3)])


def counter(db, out):
    c = sqlite3.connect(db)
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"])
        for row in c.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value "
                             "from counters_collection order by dispatch_id"):
            w.writerow(row)


if __name__ == "__main__":
    {"stats": stats, "counter": counter}[sys.argv[1]](sys.argv[2], sys.argv[3])
