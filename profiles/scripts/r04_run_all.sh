#!/bin/bash
# Round-4 measurement set (one gpurun call; order: final sources -> PMC traffic -> bench): smoke, PMC traffic passes of the headline
# launch (FETCH_SIZE / WRITE_SIZE separately), bench default, profiled bench (rocprofv3 --kernel-trace --stats), SQ counters + phase
# timing + clock probe of the fused kernel, fused A/B (79 = round-3 butterflies, 207 = folded constants, 463 = + scalar bin-256 weight, 33231 = + polyphase stage at wave priority 1: the default), stage / config / bin-shard
# benches, large-geometry fused A/B, WPE profile, pseudo-inverse bench.  Everything lands under gpurun_out/r04/;
# profiles/scripts/r04_collect.sh copies the summaries that are committed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; rm -rf $O; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
cd /tmp
PMC_S=32 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/profiles/pmc_workload.py > $O/pmc_fetch.log 2>&1
PMC_S=32 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/profiles/pmc_workload.py > $O/pmc_write.log 2>&1
mkdir -p $O/pmc_rw && cp -r $O/pmc_fetch $O/pmc_rw/ && cp -r $O/pmc_write $O/pmc_rw/
python $R/profiles/make_traffic_json.py $O/pmc_rw $O/pmc_traffic.json 32 4096 > /dev/null 2>&1
cp $O/pmc_traffic.json $R/profiles/r04_pmc_traffic.json          # bench.py below quotes it (same sources, same launch)
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --no-cpu > $O/bench_profiled.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB $O/bench_kernel_stats.txt > /dev/null 2>&1
cd $R
python bench_stages.py > $O/bench_stages.json 2> $O/bench_stages.err
python bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err
python bench_bin_sharded.py > $O/bench_bin_sharded.json 2> $O/bench_bin_sharded.err
BTK_FUSED_VAR=33231 bash profiles/scripts/r02_pmc_fused.sh > $O/pmc_fused_sq.txt 2>&1
for v in 33743; do BTK_FUSED_VAR=$v python profiles/fused_ab.py 2>&1 | grep -E "phases|ms"; done > $O/fused_phase_timing.txt 2>&1
VARS="33231 463 207 79" ROUNDS=4 bash profiles/scripts/r03_fused_ab.sh > $O/fused_ab.txt 2>&1
cp gpurun_out/r03_fused_ab.log $O/fused_ab_raw.log
for v in 33231; do BTK_FUSED_VAR=$v PROBE_SECONDS=6 python profiles/clock_probe.py 2>/dev/null | tail -1; done > $O/clock_probe.txt
python profiles/fused_big_ab.py > $O/fused_big_ab.txt 2>/dev/null
WPE_S=2 bash profiles/scripts/r02_wpe_profile.sh > $O/wpe_profile.txt 2>&1
(echo "# solver A/B in separate processes (BTK_WPE_TIMING=1 prints the phase shares of wave 0, shader cycles per system)"; for v in REG PANEL; do echo "BTK_WPE_SOLVE_$v=1"; env BTK_WPE_SOLVE_$v=1 BTK_WPE_TIMING=1 WPE_S=2 python profiles/wpe_one.py 2>&1 | grep -E "wpe_solve phases|wpe_estimate" | tail -2; done) > $O/wpe_solver_ab.txt 2>&1
python profiles/fb_ab.py 2>/dev/null | tail -4 > $O/fb_ab.txt
python profiles/pinv_bench.py > $O/pinv_bench.txt 2>/dev/null
python profiles/mvdr256_time.py 2>/dev/null | tail -1 > $O/mvdr_solve.json
python profiles/nlms_ab.py > $O/nlms_ab.txt 2>/dev/null
tail -1 $O/smoke.log; ls $O
