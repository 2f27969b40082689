#!/bin/bash
# SQ counters of analysis_bfz_big_kernel (profiles/fused_big_one.py), one --pmc pass per counter group, for the LDS form of the 1a -> 1b
# hand-over (BTK_FUSED_VAR=3) and the row-swap form (7), at M = 2048 (256 x 1 x 4096) and M = 1024 (64 x 4 x 8192).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_big; rm -rf $O; mkdir -p $O
for shape in 2048,256,1,4096 1024,64,4,8192; do
 for V in 3 7; do
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    BIG_SHAPE=$shape BTK_FUSED_VAR=$V rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/s${shape%%,*}_v${V}_g$i -o p -- python $R/profiles/fused_big_one.py > $O/s${shape%%,*}_v${V}_g$i.log 2>&1
  done
 done
done
python - <<PY
import csv, glob, collections
out = open("$O/summary.txt", "w")
for shape in ("2048", "1024"):
    tab = {}
    for V in ("3", "7"):
        rows = collections.defaultdict(list)
        for f in glob.glob("$O/s%s_v%s_g*/**/*counter_collection.csv" % (shape, V), recursive=True):
            for r in csv.DictReader(open(f)):
                if "bfz_big_kernel" in r["Kernel_Name"]:
                    rows[r["Counter_Name"]].append(float(r["Counter_Value"]))
        tab[V] = {c: sum(v) / len(v) for c, v in rows.items()}
    line = "M = %s   %-24s %16s %16s" % (shape, "counter (per launch)", "LDS form (VAR 3)", "row swaps (VAR 7)")
    print(line); out.write(line + "\n")
    for c in sorted(set(tab["3"]) | set(tab["7"])):
        line = "           %-24s %16.0f %16.0f" % (c, tab["3"].get(c, float("nan")), tab["7"].get(c, float("nan")))
        print(line); out.write(line + "\n")
PY
