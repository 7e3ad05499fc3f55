#!/usr/bin/env python
"""Merge the counter_collection.csv files of tools/pmc_collect.sh into one
per-kernel JSON (means per launch) with the derived figures bench.py reads:
hbm_bytes_per_launch (gfx950 FETCH_SIZE correction) and mfma_pipe_busy_frac."""
import collections
import csv
import glob
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "transfusion_l"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for path in glob.glob(os.path.join(src, "g*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name).replace("msmd::", "")
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
out = {}
for name in sorted(agg):
    if not (name.startswith("spconv") or name.startswith("bn_") or name.startswith("wgrad")
            or name.startswith("subm") or name.startswith("conv_") or name.startswith("vox")
            or name.startswith("dense") or name.startswith("pack") or name.startswith("permute")
            or name.startswith("row_mask") or name.startswith("scan") or name.startswith("fps")
            or name.startswith("nn_") or name.startswith("ball") or name.startswith("split_")
            or name.startswith("mark") or name.startswith("add_rows")):
        continue
    e = {k: round(v / cnt[name][k], 1) for k, v in agg[name].items()}
    e["launches_sampled"] = max(cnt[name].values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    if e.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in e:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs
        e["mfma_pipe_busy_frac"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"]
                                         / (e["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if e.get("TCC_REQ_sum"):
        e["l2_hit_rate"] = round(e.get("TCC_HIT_sum", 0.0) / e["TCC_REQ_sum"], 4)
    out[name] = e
note = ("rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only) over the default, "
        "pipelined schedule: `python bench.py --workload %s --no-also --steps 3 --warmup 2 --no-cpu-baseline "
        "--no-profile`; values are means per launch.  " % workload +
        "FETCH_SIZE/WRITE_SIZE are in KiB as rocprofv3 reports them; hbm_bytes_per_launch applies "
        "the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B for "
        "wide coalesced reads: x2) -- an upper bound here because part of the reads are narrow row "
        "gathers.  mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).")
json.dump({"note": note, "workload": workload, "kernels": out}, open(dst, "w"), indent=1, sort_keys=True)
print("wrote", dst, len(out), "kernels")
