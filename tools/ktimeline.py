#!/usr/bin/env python3
"""Developer helper: print the kernel timeline around the last-but-one dispatch of a kernel whose
name contains <substr>, from a rocprofv3 kernel trace CSV (tools/kstats.sh writes one)."""
import sys as _sys
if "--help" in _sys.argv[1:] or "-h" in _sys.argv[1:]:  # every tool answers --help without touching the GPU (tests/test_tools.py)
    print(__doc__)
    _sys.exit(0)
import csv
import sys

trace, substr = sys.argv[1], sys.argv[2]
before, after = int(sys.argv[3]) if len(sys.argv) > 3 else 6, int(sys.argv[4]) if len(sys.argv) > 4 else 8
rows = list(csv.DictReader(open(trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if substr in r["Kernel_Name"]]
i = idx[-2] if len(idx) > 1 else idx[-1]
t0 = None
for r in rows[max(0, i - before):i + after]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 is None:
        t0 = s
    print("%9.1f %9.1f dur %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:80]))
