#!/bin/bash
# kernel trace of one WPE estimation at the reference configuration (profiles/wpe_one.py, WPE_S streams)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/wpe_prof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o wpe -- python $R/profiles/wpe_one.py > $O/run.log 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB $O/kernel_stats.txt > /dev/null 2>&1
cat $O/run.log; head -12 $O/kernel_stats.txt | cut -c1-200
