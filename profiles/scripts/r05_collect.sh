#!/bin/bash
# copy the summaries of gpurun_out/r05 (profiles/scripts/r05_run_all.sh) that are committed into profiles/r05_*
R=gpurun_out/r05
cp $R/bench_default.json profiles/r05_bench_c0.json; cp $R/bench_profiled.json profiles/r05_bench_c0_profiled_run.json
cp $R/bench_kernel_stats.txt profiles/r05_bench_c0_kernel_stats.txt; cp $R/pmc_traffic.json profiles/r05_pmc_traffic.json
cp $R/pmc_fused_sq.txt profiles/r05_pmc_fused_sq.txt
cp $R/bench_stages.json profiles/r05_bench_stages.json; cp $R/bench_configs.json profiles/r05_bench_configs.json
cp $R/bench_bin_sharded.json profiles/r05_bench_bin_sharded.json
[ -f $R/bench_bin_sharded_replicated.json ] && cp $R/bench_bin_sharded_replicated.json profiles/r05_bench_bin_sharded_replicated.json
(head -3 $R/fused_phase_timing.txt; echo "..."; tail -4 $R/fused_phase_timing.txt) > profiles/r05_fused_phase_timing.txt
cp $R/fused_big_ab.txt profiles/r05_fused_big_ab.txt
cp $R/linpack_rule_time.json profiles/r05_linpack_rule_time.json
(echo "# profiles/adaptive_overlap_ab.py on MI355X (round 5): the adaptive chain at C0 (32 streams x 4096 frames), one launch per kernel against"; echo "# forms that let the analysis bank run BESIDE the canceller.  Concurrent HIP streams alone change nothing (the two kernels of a group queue up behind"; echo "# the other group's analysis); staggering does: stream groups (analysis of group g+1 beside the canceller of group g) and frame chunks (the bank runs"; echo "# ahead, the canceller follows with its full occupancy) both gain ~8 %, bit-identical.  The chain moves 43 GB per launch"; echo "# (8.6 PCM + 17.2 snapshots written + 17.2 read): 8.1 ms at the 5.3 TB/s the staged bank reaches alone is the floor of ANY overlap through HBM."; cat $R/adaptive_overlap_ab.txt) > profiles/r05_adaptive_overlap_ab.txt
(echo "# WPE estimate, reference configuration (8 ch x lags 0..32, 2 iterations, 1000 frames), 2 streams per call: profiles/wpe_one.py under rocprofv3 --kernel-trace --stats"; cat $R/wpe_profile.txt) > profiles/r05_wpe_kernel_stats.txt
(echo "# profiles/nlms_ab.py (NLMS canceller at C0 channel count, round 5: streaming energy kernel)"; cat $R/nlms_ab.txt) > profiles/r05_nlms_ab.txt
python - <<'PY'
import json, csv, collections
out = []
for l in open("gpurun_out/r05/clock_probe.txt"):
    d = json.loads(l); sc = sorted(d["sclk_MHz_samples"]); pw = sorted(d["power_W_samples"])
    out.append("%-6s var=%s  ms first/min/last %.3f / %.3f / %.3f   sclk MHz median %d (min %d max %d, %d samples)   package power W median %.0f (max %.0f)"
               % (d["work"], d["var"], d["ms_first"], d["ms_min"], d["ms_last"], sc[len(sc) // 2], sc[0], sc[-1], len(sc), pw[len(pw) // 2], pw[-1]))
open("profiles/r05_clock_probe.txt", "w").write("# profiles/clock_probe.py: kernel back to back for 6 s, rocm-smi sampled from the same process (power cap 1400 W, nominal sclk 2400 MHz)\n" + "\n".join(out) + "\n")
o = []
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/r05/%s/p_counter_collection.csv" % name)):
        if r["Counter_Name"] == ctr: acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "at::" not in k and "rocclr" not in k: o.append("%-11s avg %14.1f KiB over %3d dispatches  %s" % (ctr, sum(v) / len(v), len(v), k))
open("profiles/r05_pmc_c0_traffic_raw.txt", "w").write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of profiles/pmc_workload.py, PMC_S=32, T=4096, C0\n# gfx950: HBM read bytes = 2 * FETCH_SIZE KiB * 1024 (MI355X guide); write bytes = WRITE_SIZE KiB * 1024\n" + "\n".join(o) + "\n")
PY
