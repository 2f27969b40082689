# refresh every committed measurement of the round: bench default, profiled bench, PMC traffic, stage / config benches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/meas_final; rm -rf $O; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench_stages.py > $O/bench_stages.json 2> $O/bench_stages.err
python bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py > $O/bench_profiled.json 2> $O/prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/profiles/pmc_workload.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/profiles/pmc_workload.py > $O/pmc_write.log 2>&1
tail -1 $O/smoke.log; ls $O
