#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc CSVs under a directory per kernel (largest kernel first): counter totals per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
launches = defaultdict(lambda: defaultdict(set))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add(r["Dispatch_Id"])
rows = sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0.0))
for k, c in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 2]:
    print(k[:140])
    per = {n: v / max(len(launches[k][n]), 1) for n, v in c.items()}
    for n in sorted(per):
        print(f"   {n:28s} {per[n]:.4g}   ({len(launches[k][n])} launches)")
    if "SQ_ACTIVE_INST_VALU" in per and "GRBM_GUI_ACTIVE" in per:
        # SQ_ACTIVE_INST_VALU counts in quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        busy = per["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * per["GRBM_GUI_ACTIVE"] / 8)
        print(f"   VALU busy fraction = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8) = {busy:.3f}")
