#!/bin/bash
# SQ counters of the NLMS canceller kernel (profiles/nlms_one.py launch; NLMS_S streams, BTK_NLMS_ALT form), one --pmc pass per group
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_nlms_s${NLMS_S:-16}_a${BTK_NLMS_ALT:-0}; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python $R/profiles/nlms_one.py > $O/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nlms_bin2_kernel" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("$O/summary.txt", "w")
for c in sorted(rows):
    line = "    %-28s %16.0f  (n=%d)" % (c, sum(rows[c]) / len(rows[c]), len(rows[c]))
    print(line); out.write(line + "\n")
PY
