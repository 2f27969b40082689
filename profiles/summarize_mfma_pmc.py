#!/usr/bin/env python
"""MFMA utilisation of the matrix-core kernels from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
SQ_INSTS_VALU_MFMA_MOPS_F32, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_ACTIVE_INST_VALU; --output-format csv).
util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs): the MfmaUtil expression of rocprofv3 -L written out (the shipped
derived-metric tables have no gfx950 section); the CSV reports GRBM_GUI_ACTIVE summed over the 8 XCDs where the expression
takes the max, hence the /8 (cross-check: GUI_ACTIVE/8 / duration = the ~2.06 GHz clock under MFMA load).
flops = MOPS_F32 * 512.   usage: summarize_mfma_pmc.py <csv> [out]"""
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[(k, r["Dispatch_Id"])] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
lines = ["# MFMA counters per dispatch (mean), %s" % sys.argv[1]]
for k in rows:
    if not any(s in k for s in ("cov_mfma", "wpe_herk", "wpe_lagprod", "wpe_solve")):
        continue
    c = {a: sum(v) / len(v) for a, v in rows[k].items()}
    d = [v for (kk, _), v in dur.items() if kk == k]
    t_us = sum(d) / len(d)
    flops = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
    clk = c["GRBM_GUI_ACTIVE"] / 8 / t_us / 1e3
    lines.append("%s: %.1f us (profiled), MFMA flops %.3e -> %.1f TFLOP/s, MFMA busy %.1f %% of SIMD-cycles at %.2f GHz "
                 "(fp32 MFMA peak at that clock: %.0f TFLOP/s)"
                 % (k, t_us, flops, flops / t_us / 1e6, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), clk,
                    256 * 4 * 64 * clk / 1e3))
    for a, v in sorted(c.items()):
        lines.append("    %-32s %16.0f" % (a, v))
txt = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
print(txt)
