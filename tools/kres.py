#!/usr/bin/env python
"""Per-kernel register / LDS / occupancy table of one csrc/*.hip file (hipcc
-Rpass-analysis=kernel-resource-usage, cross-compiled: no GPU needed).

    python tools/kres.py msmdfusion_amd/csrc/spconv_split.hip [filter]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
       "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in err.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|"
                  r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v],
                                      capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    else:
        cur[k.split(" [")[0]] = v
for r in rows:
    name = re.sub(r"msmd::\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name:
        continue
    print("%-52s V%-4s A%-4s spill %-3s scratch %-4s lds %-6s occ %s" % (
        name[:52], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize"),
        r.get("LDS Size"), r.get("Occupancy")))
