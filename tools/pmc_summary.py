#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs for the render kernel."""
import csv
import sys
from collections import defaultdict

d = defaultdict(dict)
meta = {}
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r["Kernel_Name"]:
            continue
        d["k"][r["Counter_Name"]] = d["k"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        meta = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size", "Grid_Size")}
        meta["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
v = d["k"]
print(meta)
for k in sorted(v):
    print(f"  {k:28s} {v[k]:.4g}")
if "SQ_THREAD_CYCLES_VALU" in v:
    print("  active-lane fraction        %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"] / 64))
if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v:
    print("  wait fraction of wave cycles %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
