#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSVs per kernel (counters summed over all dispatches of a kernel)."""
import csv
import re
import sys
from collections import defaultdict

vals = defaultdict(lambda: defaultdict(float))
meta = {}
calls = defaultdict(set)
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m:
            continue
        k = m.group(1)
        vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add((f, r["Dispatch_Id"]))
        meta[k] = {x: r[x] for x in ("VGPR_Count", "SGPR_Count", "Scratch_Size", "LDS_Block_Size")}
for k, v in vals.items():
    print(k, meta[k], "dispatches/pass:", len(calls[k]) // max(1, len(sys.argv) - 1))
    for c in sorted(v):
        print(f"  {c:28s} {v[c]:.4g}")
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        print("  active-lane fraction        %.3f" % (v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"] / 64))
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        print("  wait fraction of wave cycles %.3f" % (v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]))
    if v.get("SQ_BUSY_CU_CYCLES") and "SQ_WAVE_CYCLES" in v:
        # SQ_WAVE_CYCLES counts quad-cycles per resident wave, SQ_BUSY_CU_CYCLES cycles per busy CU (4 SIMDs):
        # their ratio is the average number of waves resident per SIMD while the CU is busy
        print("  occupancy (waves per SIMD)  %.2f" % (v["SQ_WAVE_CYCLES"] / v["SQ_BUSY_CU_CYCLES"]))
    if v.get("SQ_LDS_IDX_ACTIVE"):
        print("  LDS bank-conflict fraction  %.3f  (conflict cycles / LDS-array cycles)" % (v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]))
    if v.get("SQ_INSTS_VALU") and v.get("SQ_BUSY_CU_CYCLES"):
        print("  VALU issue utilisation      %.3f  (4 cycles per wave64 instruction and SIMD)" % (v["SQ_INSTS_VALU"] * 4.0 / (v["SQ_BUSY_CU_CYCLES"] * 4.0)))
