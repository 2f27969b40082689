#!/bin/bash
# copy the summaries of gpurun_out/r06 (profiles/scripts/r06_run_all.sh) that are committed into profiles/r06_*
R=gpurun_out/r06
cp $R/bench_default.json profiles/r06_bench_c0.json; cp $R/bench_driver_flags.json profiles/r06_bench_c0_driver_flags.json
cp $R/bench_profiled.json profiles/r06_bench_c0_profiled_run.json
cp $R/bench_kernel_stats.txt profiles/r06_bench_c0_kernel_stats.txt; cp $R/pmc_traffic.json profiles/r06_pmc_traffic.json
cp $R/bench_stages.json profiles/r06_bench_stages.json; cp $R/bench_configs.json profiles/r06_bench_configs.json
cp $R/bench_bin_sharded.json profiles/r06_bench_bin_sharded.json
cp $R/fused_big_ab.txt profiles/r06_fused_big_ab.txt
cp $R/linpack_rule_time.json profiles/r06_linpack_rule_time.json
(echo "# profiles/partition_probe.py: the fused kernels are partition-exact in (t0, tcount) and in the sample window they are given"; cat $R/partition_probe.txt) > profiles/r06_partition_probe.txt
(echo "# host/examples/node_api_bench (C0 graph: 64 SampleFeature -> 64 banks -> SubbandGSC -> synthesis, pulled with next()), second pass timed."; echo "# BTK_NODE_I16=1: utterances of 16-bit PCM go up as int16 from the sources' pinned copies (round 6); =0: float blocks through SampleFeature::next_blocks (round 5)."; echo "# BTK_NODE_PREFETCH=1: the next block's upload runs under the current block's kernels, download and serving."; cat $R/node_api_i16.txt) > profiles/r06_node_api_i16.txt
(echo "# the float path of the node API against the number of helper threads that pull the SampleFeature sources (BTK_NODE_THREADS)"; cat $R/node_api_threads.txt) > profiles/r06_node_api_threads.txt
(echo "# WPE estimate, reference configuration (8 ch x lags 0..32, 2 iterations, 1000 frames), 2 streams per call: rocprofv3 --kernel-trace --stats"; cat $R/wpe_profile.txt) > profiles/r06_wpe_kernel_stats.txt
(echo "# final sources: node_api_bench with the block's rows going up by ONE gather kernel (BTK_NODE_GATHER=1, default) against one hipMemcpyAsync per row (=0); sweep of the kernel's launch: profiles/r06_node_api_gather.txt"; cat $R/node_api_gather.txt) > profiles/r06_node_api_gather_final.txt
python - <<'PY'
import csv, collections, json
# HIP-API traces of the node-API bench at two stream lengths: calls per API and what a block costs in calls
out = ["# rocprofv3 --hip-trace --stats of host/examples/node_api_bench (C0 graph, blocks of 1024 frames), 8192 and 32768 frames per pass (2 passes):",
       "# the allocation calls do not grow with the number of blocks (the SampleFeature sources' pinned int16 copies are made at load time: 64 per pass)"]
for F in (8192, 32768):
    rows = list(csv.DictReader(open("gpurun_out/r06/hip_%d/t_hip_api_stats.csv" % F)))
    out.append("frames per pass %d:" % F)
    for r in rows:
        if any(k in r["Name"] for k in ("Malloc", "Free", "Memcpy", "LaunchKernel", "StreamSynchronize", "Memset")):
            out.append("  %-28s calls %7s  total %10.1f us" % (r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e3))
open("profiles/r06_node_api_hip_trace.txt", "w").write("\n".join(out) + "\n")
o = []
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/r06/%s/p_counter_collection.csv" % name)):
        if r["Counter_Name"] == ctr: acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "at::" not in k and "rocclr" not in k: o.append("%-11s avg %14.1f KiB over %3d dispatches  %s" % (ctr, sum(v) / len(v), len(v), k))
open("profiles/r06_pmc_c0_traffic_raw.txt", "w").write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of profiles/pmc_workload.py, PMC_S=32, T=4096, C0 (float and int16 entries)\n# gfx950: HBM read bytes = 2 * FETCH_SIZE KiB * 1024 (MI355X guide); write bytes = WRITE_SIZE KiB * 1024\n" + "\n".join(o) + "\n")
PY
