#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats CSV: ms per bench step per kernel."""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms per step %.3f" % (tot / steps / 1e6))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-72s %5s calls %8.1f us avg %6.3f ms/step %6.2f%%" % (
        r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3,
        float(r["TotalDurationNs"]) / steps / 1e6, float(r["Percentage"])))
