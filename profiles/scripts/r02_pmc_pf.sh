#!/bin/bash
# matrix-core statistics kernel of the McCowan / Lefkimmiatis post-filters (profiles/pf_one.py): kernel-trace durations and
# SQ / MFMA counters, one --pmc pass per group; PF_LEF / PF_PAD select the form and the row layout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_pf_l${PF_LEF:-0}_p${PF_PAD:-0}; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python $R/profiles/pf_one.py > $O/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(list); dur = []
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stats2" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
print("kernel duration under the profiler: %.1f us (n=%d)" % (sum(dur) / max(1, len(dur)), len(dur)))
for c in sorted(rows):
    print("    %-32s %16.0f  (n=%d)" % (c, sum(rows[c]) / len(rows[c]), len(rows[c])))
if "SQ_VALU_MFMA_BUSY_CYCLES" in rows:
    g = sum(rows["GRBM_GUI_ACTIVE"]) / len(rows["GRBM_GUI_ACTIVE"]) / 8
    print("    MFMA busy %.1f %% of SIMD-cycles" % (100 * sum(rows["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(rows["SQ_VALU_MFMA_BUSY_CYCLES"]) / (g * 1024)))
PY
