#!/bin/bash
# copy the summaries of gpurun_out/r04 (profiles/scripts/r04_run_all.sh) that are committed into profiles/r04_*
R=gpurun_out/r04
cp $R/bench_default.json profiles/r04_bench_c0.json; cp $R/bench_profiled.json profiles/r04_bench_c0_profiled_run.json
cp $R/bench_kernel_stats.txt profiles/r04_bench_c0_kernel_stats.txt; cp $R/pmc_traffic.json profiles/r04_pmc_traffic.json
cp $R/pmc_fused_sq.txt profiles/r04_pmc_fused_sq.txt
cp $R/bench_stages.json profiles/r04_bench_stages.json; cp $R/bench_configs.json profiles/r04_bench_configs.json
cp $R/bench_bin_sharded.json profiles/r04_bench_bin_sharded.json
(head -3 $R/fused_phase_timing.txt; echo "..."; tail -4 $R/fused_phase_timing.txt) > profiles/r04_fused_phase_timing.txt
cp $R/mvdr_solve.json profiles/r04_mvdr_solve.json; cp $R/fused_big_ab.txt profiles/r04_fused_big_ab_final.txt; cp $R/fused_ab.txt profiles/r04_fused_ab_final.txt
(echo "# WPE estimate, reference configuration (8 ch x lags 0..32, 2 iterations, 1000 frames), 2 streams per call: profiles/wpe_one.py under rocprofv3 --kernel-trace --stats"; cat $R/wpe_profile.txt) > profiles/r04_wpe_kernel_stats.txt
(cat $R/wpe_solver_ab.txt) >> profiles/r04_wpe_kernel_stats.txt
(echo "# profiles/fb_ab.py (staged filter banks, 8 streams x 64 channels, round 4)"; cat $R/fb_ab.txt) > profiles/r04_fb_ab.txt
(echo "# profiles/nlms_ab.py (NLMS canceller at C0 channel count, round 4)"; cat $R/nlms_ab.txt) > profiles/r04_nlms_ab.txt
python - <<'PY'
import json, csv, collections
out = []
for l in open("gpurun_out/r04/clock_probe.txt"):
    d = json.loads(l); sc = sorted(d["sclk_MHz_samples"]); pw = sorted(d["power_W_samples"])
    out.append("%-6s var=%s  ms first/min/last %.3f / %.3f / %.3f   sclk MHz median %d (min %d max %d, %d samples)   package power W median %.0f (max %.0f)"
               % (d["work"], d["var"], d["ms_first"], d["ms_min"], d["ms_last"], sc[len(sc) // 2], sc[0], sc[-1], len(sc), pw[len(pw) // 2], pw[-1]))
open("profiles/r04_clock_probe.txt", "w").write("# profiles/clock_probe.py: kernel back to back for 6 s, rocm-smi sampled from the same process (power cap 1400 W, nominal sclk 2400 MHz)\n" + "\n".join(out) + "\n")
o = []
for name, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/r04/%s/p_counter_collection.csv" % name)):
        if r["Counter_Name"] == ctr: acc[r["Kernel_Name"][:90]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "at::" not in k and "rocclr" not in k: o.append("%-11s avg %14.1f KiB over %3d dispatches  %s" % (ctr, sum(v) / len(v), len(v), k))
open("profiles/r04_pmc_c0_traffic_raw.txt", "w").write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of profiles/pmc_workload.py, PMC_S=32, T=4096, C0\n# gfx950: HBM read bytes = 2 * FETCH_SIZE KiB * 1024 (MI355X guide); write bytes = WRITE_SIZE KiB * 1024\n" + "\n".join(o) + "\n")
PY
