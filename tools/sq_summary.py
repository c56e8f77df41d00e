#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc rocpd database (SQ_* counters make the
database too large to carry around, so this runs next to it and prints one line per
kernel and counter).  usage: sq_summary.py <results.db> [kernel substring]"""
import collections
import sqlite3
import sys

db, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
c = sqlite3.connect(db)
agg = collections.defaultdict(lambda: [0, 0.0])
for name, counter, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
    if needle not in name:
        continue
    short = name.split("(")[0].split("::")[-1]
    a = agg[(short, counter)]
    a[0] += 1
    a[1] += value
for (k, counter), (n, v) in sorted(agg.items()):
    print(f"{k:32s} {counter:24s} dispatches {n:5d}  sum {v:.6g}  per dispatch {v / n:.6g}")
