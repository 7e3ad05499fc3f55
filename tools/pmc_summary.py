#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters (counter_collection.csv)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0].replace("msmd::", "")
    agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name][r["Counter_Name"]] += 1
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for name in sorted(agg):
    if filt and filt not in name:
        continue
    print(name[:80], {k: round(v / cnt[name][k], 1) for k, v in agg[name].items()},
          "launches", max(cnt[name].values()))
