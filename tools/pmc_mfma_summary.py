#!/usr/bin/env python3
"""MFMA busy fraction and effective clock per kernel from `rocprofv3 -i tools/pmc_mfma.txt --output-format csv -d DIR -- <cmd>`:
   busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles), cycles = GRBM_GUI_ACTIVE / 8 XCDs; clock = cycles / duration
   (MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
match = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if match and match not in k:
            continue
        a = acc[k]
        a[row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            a["n"] += 1
for k, a in sorted(acc.items(), key=lambda kv: -kv[1].get("ns", 0)):
    if not a.get("GRBM_GUI_ACTIVE"):
        continue
    cyc = a["GRBM_GUI_ACTIVE"] / 8.0
    busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4 * 256 * cyc)
    print(f"{k[:110]:110s} n={int(a['n']):4d} avg={a['ns']/a['n']/1e3:9.1f} us  clock={cyc/a['ns']:.3f} GHz  mfma_busy={busy:.3f}")
