cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/meas_r1e; rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py > $O/bench_profiled.json 2> $O/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/profiles/pmc_workload.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/profiles/pmc_workload.py > $O/pmc_write.log 2>&1
ls $O $O/prof | head -30
tail -c 600 $O/bench_default.json
