#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel, per counter, mean value per dispatch
(and duration from the matching kernel trace).  Usage: summarize_pmc.py <dir> [out.txt]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    rows = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:60]
            rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    lines = ["# rocprofv3 --pmc summary of %s (mean per dispatch; FETCH_SIZE/WRITE_SIZE in KiB as reported, uncorrected)" % d]
    for k in rows:
        if not any(s in k for s in ("analysis", "bf_apply", "synthesis", "nlms", "cov_", "zelinski", "mvdr")):
            continue
        lines.append(k)
        if dur[k]:
            lines.append("    %-28s %14.1f us (profiled)" % ("duration", sum(dur[k]) / len(dur[k])))
        for c, v in sorted(rows[k].items()):
            lines.append("    %-28s %18.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)

main()
