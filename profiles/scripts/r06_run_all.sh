#!/bin/bash
# Round-6 measurement set (one gpurun call; order: final sources -> PMC traffic -> bench): smoke, PMC traffic passes of the headline
# launch incl. the int16 entry (FETCH_SIZE / WRITE_SIZE separately), bench default (with stages.node_api / fused_i16), profiled bench
# (rocprofv3 --kernel-trace --stats), HIP-API traces of the node-API bench at two stream lengths (allocations per block), stage /
# config / bin-shard benches, WPE profile + non-stationary probe, partition probe.  Everything lands under gpurun_out/r06/;
# profiles/scripts/r06_collect.sh copies the summaries that are committed.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
cd /tmp
PMC_S=32 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python $R/profiles/pmc_workload.py > $O/pmc_fetch.log 2>&1
PMC_S=32 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python $R/profiles/pmc_workload.py > $O/pmc_write.log 2>&1
mkdir -p $O/pmc_rw && cp -r $O/pmc_fetch $O/pmc_rw/ && cp -r $O/pmc_write $O/pmc_rw/
python $R/profiles/make_traffic_json.py $O/pmc_rw $O/pmc_traffic.json 32 4096 > /dev/null 2>&1
cp $O/pmc_traffic.json $R/profiles/r06_pmc_traffic.json          # bench.py below quotes it (same sources, same launch)
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err      # the driver's flags
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --no-cpu > $O/bench_profiled.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1)
[ -n "$DB" ] && python $R/profiles/summarize_rocpd.py $DB $O/bench_kernel_stats.txt > /dev/null 2>&1
# the node API: HIP-API trace at two stream lengths (8 and 32 blocks of 1024 frames per pass): what is called per block
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from bench_util import design_prototype
np.concatenate([design_prototype(512, 4), design_prototype(512, 4, "g")]).astype(np.float64).tofile("/tmp/c512.f64")
PY
B=$R/distant_speech_recognition_amd/host/examples/node_api_bench
for F in 8192 32768; do
  rocprofv3 --hip-trace --stats --output-format csv -d $O/hip_$F -o t -- $B /tmp/c512.f64 512 4 1 64 $F 1 1024 0 > $O/hip_$F.json 2> $O/hip_$F.err
done
for t in 1 2 4 8; do BTK_NODE_I16=0 BTK_NODE_THREADS=$t $B /tmp/c512.f64 512 4 1 64 8192 1 8192 0; done > $O/node_api_threads.txt 2>&1
# the node API on 16-bit streams against the float path, with and without the next block's upload under the current one
( for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
    echo "BTK_NODE_I16=$1 BTK_NODE_PREFETCH=$2: one graph, 32768 frames in blocks of 8192 | 8 graphs in a pool | 8 graphs one by one (8192 frames, blocks of 2048)"
    BTK_NODE_I16=$1 BTK_NODE_PREFETCH=$2 $B /tmp/c512.f64 512 4 1 64 32768 1 8192 0
    BTK_NODE_I16=$1 BTK_NODE_PREFETCH=$2 $B /tmp/c512.f64 512 4 1 64 8192 8 2048 1
    BTK_NODE_I16=$1 BTK_NODE_PREFETCH=$2 $B /tmp/c512.f64 512 4 1 64 8192 8 2048 0
  done ) > $O/node_api_i16.txt 2>&1
# the rows of a block by one gather kernel (btk_gather_rows) against one copy per row
( for g in 1 0; do
    echo "BTK_NODE_GATHER=$g: one graph 32768 frames blocks of 8192 | 32 graphs x 2048 pool blocks 1024 | 8 graphs x 8192 pool blocks 2048 | 8 graphs one by one"
    BTK_NODE_GATHER=$g $B /tmp/c512.f64 512 4 1 64 32768 1 8192 0
    BTK_NODE_GATHER=$g $B /tmp/c512.f64 512 4 1 64 2048 32 1024 1
    BTK_NODE_GATHER=$g $B /tmp/c512.f64 512 4 1 64 8192 8 2048 1
    BTK_NODE_GATHER=$g $B /tmp/c512.f64 512 4 1 64 8192 8 2048 0
  done ) > $O/node_api_gather.txt 2>&1
cd $R
python bench_stages.py > $O/bench_stages.json 2> $O/bench_stages.err
python bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err
python bench_bin_sharded.py > $O/bench_bin_sharded.json 2> $O/bench_bin_sharded.err
python profiles/partition_probe.py > $O/partition_probe.txt 2>/dev/null
python profiles/i16_banks_ab.py > $O/i16_banks_ab.json 2>/dev/null
python profiles/node_api_python.py > $O/node_api_python.txt 2>/dev/null
python profiles/wpe_envelope_probe.py 2>/dev/null | tail -1 > $O/wpe_envelope.txt
BTK_WPE_LAGPROD_F32=1 python profiles/wpe_envelope_probe.py 2>/dev/null | tail -1 >> $O/wpe_envelope.txt
WPE_S=2 bash profiles/scripts/r02_wpe_profile.sh > $O/wpe_profile.txt 2>&1
python profiles/fused_big_ab.py > $O/fused_big_ab.txt 2>/dev/null
python profiles/linpack_rule_time.py > $O/linpack_rule_time.json 2>/dev/null
python profiles/csvdc_split.py > $O/csvdc_split.txt 2>/dev/null
python profiles/nlms_residency_probe.py > $O/nlms_residency_probe.txt 2>/dev/null
python profiles/nlms_stagger_probe.py > $O/nlms_stagger_probe.txt 2>/dev/null
python profiles/adaptive_groups_ab.py > $O/adaptive_groups_ab.txt 2>/dev/null
tail -1 $O/smoke.log; ls $O
