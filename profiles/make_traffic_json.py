#!/usr/bin/env python
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, --output-format csv) -> per-kernel HBM traffic per
launch in bytes, corrected as /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]" prescribes for gfx950:
FETCH_SIZE (KiB) is doubled for wide coalesced streaming reads (128-B requests tallied at 64 B); WRITE_SIZE (KiB) is
used as reported (calibrated here: analysis512_kernel writes exactly its 8*K*N*S*T snapshot bytes).
Usage: make_traffic_json.py <pmc_dir> <out.json> S T"""
import csv, glob, hashlib, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_util import kernel_source_sha

d, out, S, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
# bench.py only quotes these numbers while the kernel sources are the ones they were measured on
res = {"S": S, "T": T, "N": 64, "M": 512, "kernel_source_sha256": kernel_source_sha(), "correction": "read = 2 * FETCH_SIZE KiB * 1024; write = WRITE_SIZE KiB * 1024", "kernels": {},
       "kernels_i16": {}}      # the instances that read 16-bit PCM (btk_fb_analysis_bf_i16) are kept apart: bench.py stages.fused_i16
for k, c in acc.items():
    if not any(s in k for s in ("analysis", "bf_apply", "synthesis")):
        continue
    e = {}
    if "FETCH_SIZE" in c:
        e["read_bytes"] = 2.0 * 1024.0 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    if "WRITE_SIZE" in c:
        e["write_bytes"] = 1024.0 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    if "read_bytes" in e and "write_bytes" in e:
        e["traffic_bytes"] = e["read_bytes"] + e["write_bytes"]
    res["kernels_i16" if ", short>" in k else "kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
