#!/bin/bash
# Round-5 measurement set (one gpurun call; order: final sources -> PMC traffic -> bench): smoke, PMC traffic passes of the headline
# launch (FETCH_SIZE / WRITE_SIZE separately), bench default, profiled bench (rocprofv3 --kernel-trace --stats), SQ counters + phase
# timing + clock probe of the fused kernel, stage / config / bin-shard benches, the adaptive-chain overlap A/B, the linpack-rule
# timing, WPE profile.  Everything lands under gpurun_out/r05/; profiles/scripts/r05_collect.sh copies the summaries that are committed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; rm -rf $O; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
cd /tmp
PMC_S=32 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/profiles/pmc_workload.py > $O/pmc_fetch.log 2>&1
PMC_S=32 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/profiles/pmc_workload.py > $O/pmc_write.log 2>&1
mkdir -p $O/pmc_rw && cp -r $O/pmc_fetch $O/pmc_rw/ && cp -r $O/pmc_write $O/pmc_rw/
python $R/profiles/make_traffic_json.py $O/pmc_rw $O/pmc_traffic.json 32 4096 > /dev/null 2>&1
cp $O/pmc_traffic.json $R/profiles/r05_pmc_traffic.json          # bench.py below quotes it (same sources, same launch)
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --no-cpu > $O/bench_profiled.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB $O/bench_kernel_stats.txt > /dev/null 2>&1
cd $R
python bench_stages.py > $O/bench_stages.json 2> $O/bench_stages.err
python bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err
python bench_bin_sharded.py > $O/bench_bin_sharded.json 2> $O/bench_bin_sharded.err                                   # (default: frame-range partition of the fused operator)
python bench_bin_sharded.py --analysis-input replicated > $O/bench_bin_sharded_replicated.json 2> $O/bench_bin_sharded_replicated.err
BTK_FUSED_VAR=33231 bash profiles/scripts/r02_pmc_fused.sh > $O/pmc_fused_sq.txt 2>&1
for v in 33743; do BTK_FUSED_VAR=$v python profiles/fused_ab.py 2>&1 | grep -E "phases|ms"; done > $O/fused_phase_timing.txt 2>&1
for v in 33231; do BTK_FUSED_VAR=$v PROBE_SECONDS=6 python profiles/clock_probe.py 2>/dev/null | tail -1; done > $O/clock_probe.txt
python profiles/adaptive_overlap_ab.py 2>/dev/null | grep -v amdgpu.ids > $O/adaptive_overlap_ab.txt
python profiles/linpack_rule_time.py 2>/dev/null | tail -1 > $O/linpack_rule_time.json
python profiles/fused_big_ab.py > $O/fused_big_ab.txt 2>/dev/null
WPE_S=2 bash profiles/scripts/r02_wpe_profile.sh > $O/wpe_profile.txt 2>&1
python profiles/nlms_ab.py > $O/nlms_ab.txt 2>/dev/null
tail -1 $O/smoke.log; ls $O
