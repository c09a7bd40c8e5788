#!/usr/bin/env python3
"""Shorten the kernel names of a rocprofv3 *_kernel_stats.csv so the summary is readable (profiles/)."""
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(rows[0])
for r in rows[1:]:
    name = r[0]
    if len(name) > 110:
        name = name[:107] + "..."
    w.writerow([name] + r[1:])
