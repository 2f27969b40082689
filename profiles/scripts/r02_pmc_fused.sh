#!/bin/bash
# SQ counters of the fused kernel (profiles/fused_ab.py launch), one --pmc pass per counter group; BTK_FUSED_VAR selects the form.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; V=${BTK_FUSED_VAR:-16}; O=$R/gpurun_out/pmc_fused_v$V; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_WAVE_DEP_WAIT SQ_IFETCH SQ_INSTS_BRANCH"; do
  i=$((i+1))
  BTK_FUSED_VAR=$V rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o p -- python $R/profiles/fused_ab.py > $O/g$i.log 2>&1
done
python $R/profiles/summarize_pmc.py_all.py $O 2>/dev/null
python - <<PY
import csv, glob, collections
rows = collections.defaultdict(list)
for f in glob.glob("$O/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bfz_kernel" in r["Kernel_Name"] or "bfp_kernel" in r["Kernel_Name"]:
            rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = open("$O/summary.txt", "w")
for c in sorted(rows):
    line = "    %-28s %16.0f  (n=%d)" % (c, sum(rows[c]) / len(rows[c]), len(rows[c]))
    print(line); out.write(line + "\n")
PY
