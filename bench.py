#!/usr/bin/env python
"""bench.py -- headline benchmark: beamformed subband frames/s, 64-mic 512-bin SubbandGSC.

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM:
    S streams x 64 channels of PCM -> oversampled-DFT analysis (M=512, m=4, r=1)
    -> SubbandGSC apply (wq - B wa, 257 bins) -> synthesis -> S output signals.
`value` counts beamformed frames (S*T per step per GPU) per second, aggregated over all ranks.
Multi-GPU: streams are sharded over ranks (independent utterances, no data-path collective);
one process per GPU, launched by torch.distributed.run; scaling is weak.

Prints ONE JSON line on rank 0 (see the driver contract in the task description), including
  roofline     -- the dominant kernel (analysis bank) against the HBM roofline,
  cpu_baseline -- the oracle (loop-faithful port of the reference's CPU path) on 1 host core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
VECTOR_PEAK = 157.3e12     # flop/s: 256 CUs x 4 SIMDs x 64 float32 flops per clock x 2.4 GHz (MI355X_MICROARCH.md)
SCLK_PEAK = 2.4e9          # Hz, nominal peak engine clock; under the 1 400 W cap the fused kernel runs at ~2.1 GHz (profiles/*_clock_probe.txt)
# Interior loop of analysis512_bfz_kernel<2,33231,16>, per wavefront (= 4 frames) and channel, counted in the ISA of the sources
# with this sha256 (tools/isa_loop_count.py on `hipcc -S --cuda-device-only fb_analysis512.hip`, the loop that holds
# the 15 window loads of a channel): packed float32 instructions; a wave64 packed instruction occupies its SIMD for 4 cycles.
FUSED_ISA = {"kernel_source_sha256": "98cd814b2e72da2cb6b34596f639a17dff228c65bb3486bede0db4fc3363d9f9",
             "v_pk_fma_f32": 184, "v_pk_add_f32": 89, "v_pk_mul_f32": 23, "frames_per_wave": 4}
TRAFFIC_JSON = "r06_pmc_traffic.json"      # profiles/: PMC traffic of the kernels below, with the launch size and kernel-source hash it holds for
FS = 16000.0


def synth_pcm_device(torch, dev, S, N, L, delays, seed):
    """int16-scale synthetic PCM [S][N][L] generated on the device (SURVEY 8(d) distribution:
    iid N(0,1000^2) noise + a common N(0,3000^2) target delayed per channel, rounded, clipped)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    pcm = torch.empty((S, N, L), dtype=torch.float32, device=dev)
    for s in range(S):
        tgt = torch.randn(L + 64, generator=g, device=dev) * 3000.0
        noise = torch.randn((N, L), generator=g, device=dev) * 1000.0
        for c in range(N):
            sh = int(round(delays[c] * FS))
            noise[c] += tgt[32 + sh: 32 + sh + L]
        pcm[s] = noise.round_().clamp_(-32767, 32767)
    return pcm


def ula_positions(N, pitch_mm=20.0):
    """uniform linear array, centred, 20 mm pitch; positions in mm"""
    x = (np.arange(N) - (N - 1) / 2.0) * pitch_mm
    return np.stack([x, np.zeros(N), np.zeros(N)], axis=1)


def synth_pcm_host(N, L, delays, seed):
    """the same distribution as synth_pcm_device for one stream, on the host (CPU baseline input)"""
    rng = np.random.default_rng(seed)
    tgt = rng.normal(0.0, 3000.0, L + 64)
    out = np.empty((N, L), np.float32)
    for c in range(N):
        sh = int(round(delays[c] * FS))
        out[c] = np.clip(np.rint(rng.normal(0.0, 1000.0, L) + tgt[32 + sh: 32 + sh + L]), -32767, 32767)
    return out


def _cpu_stream(a):
    """one utterance stream through the oracle's frame-by-frame pull graph; returns (frames, seconds, t_start, t_end)"""
    N, M, m, r, dct, frames, seed = a
    from oracle import oracle as orc
    from distant_speech_recognition_amd import prototypes
    from distant_speech_recognition_amd.pybeamformer import calc_la_delays
    D = M >> r
    h, g = prototypes.load(M, m, r)
    delays = calc_la_delays(ula_positions(N), -1.306379)
    pcm = synth_pcm_host(N, frames * D, delays, seed)
    wq = orc.calc_mainlobe(M, N, FS, delays)
    wl = np.zeros((M, N), np.complex128)
    w0 = time.time()                                      # wall clock: comparable across the pool's processes
    t0 = time.perf_counter()
    _, nbf = orc.pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl)
    dt = time.perf_counter() - t0
    return nbf, dt, w0, w0 + dt


def usable_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (v2 cpu.max, v1 cfs_quota)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(N, M, m, r, dct, frames, all_cores=True):
    """Time the oracle's frame-by-frame pull graph (N analysis banks -> SubbandGSC -> synthesis) on a bounded sample of the
    same workload: ONE host core (the reference is single-threaded: the faithful figure, `value`), and -- BASELINE.md
    section 2 item 2 -- one independent process per utterance stream on all cores the box gives this process."""
    D = M >> r
    nbf, dt, _, _ = _cpu_stream((N, M, m, r, dct, frames, 20260927))
    res = {"value": nbf / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "%d-mic %d-bin SubbandGSC chain, %d frames, 1 stream, oracle/btk_oracle.c -O3" % (N, M, nbf),
           "xRT": nbf / dt / (FS / D)}
    if all_cores:
        import multiprocessing as mp
        cores, quota = usable_cores()
        per = max(frames // 4, 256)                       # ~1/4 of the single-core sample per process
        with mp.get_context("fork").Pool(cores) as pool:
            outs = pool.map(_cpu_stream, [(N, M, m, r, dct, per, 20260927 + 1000 * i) for i in range(cores)], chunksize=1)
        tot = sum(o[0] for o in outs)
        wall = max(o[3] for o in outs) - min(o[2] for o in outs)        # first pull-graph start .. last end (input synthesis excluded)
        res["all_cores"] = {"value": tot / wall, "unit": "frames/s", "cores": cores, "kind": "port",
                            "sample": "%d independent streams (one process each) x %d frames" % (cores, outs[0][0]),
                            "xRT": tot / wall / (FS / D), "cgroup_cpu_quota": quota,
                            "per_process_frames_per_s": [min(o[0] / o[1] for o in outs), max(o[0] / o[1] for o in outs)]}
    return res


def node_api_stage(h, g, N, M, m, r, engine_frames_per_s):
    """What a drop-in caller of the reference's node API gets: host/examples/node_api_bench builds the graph of
    src/beamformerDS.cc:144-223 (SampleFeature x N -> OverSampledDFTAnalysisBank x N -> SubbandGSC -> OverSampledDFTSynthesisBank)
    on in-memory utterances and pulls it block by block through next(): one graph of 32768 frames (four blocks: the upload of a
    block runs under the computing and serving of the one before), 32 graphs of 2048 frames one after the other
    and the same 32 graphs in a SubbandGraphPool (one S = 32 launch per round).  The binary reports where the host time goes; the
    device work per frame is the engine's, the rest is the reference's own interface: every sample lives in a SampleFeature in
    host memory and crosses PCIe.  Round 6: utterances of 16-bit PCM (what a WAV read delivers) go up AS int16, from where the
    source keeps them (SampleFeature::pcm16: no host copies, 2 B per sample, the same bits out); `float_path` repeats the runs with
    BTK_NODE_I16=0 -- the samples through SampleFeature's float blocks, as in round 5."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "node_api_bench")
    if not os.path.exists(exe):
        return {"error": "host/examples/node_api_bench is not built (make -C distant_speech_recognition_amd/host)"}
    out = {"what": "C++ node graph of src/beamformerDS.cc at C0 (%d banks -> SubbandGSC -> synthesis), pulled with next() until "
                   "jiterator_error; wall clock of the pull loop, inputs in host memory (SampleFeature), second pass timed" % N}
    with tempfile.NamedTemporaryFile(suffix=".f64") as f:
        np.concatenate([h, g]).astype(np.float64).tofile(f.name)
        for key, frames, graphs, block, pool in (("one_graph", 32768, 1, 8192, 0), ("graphs_32_one_by_one", 2048, 32, 1024, 0),
                                                 ("graph_pool_32", 2048, 32, 1024, 1)):
            for i16 in (1, 0):
                dst = out if i16 else out.setdefault("float_path", {"what": "the same runs with BTK_NODE_I16=0: float samples through "
                                                                            "SampleFeature::next_blocks and over PCIe (4 B per sample)"})
                try:
                    res = subprocess.run([exe, f.name, str(M), str(m), str(r), str(N), str(frames), str(graphs), str(block), str(pool)],
                                         capture_output=True, text=True, timeout=600, env=dict(os.environ, BTK_NODE_I16=str(i16)))
                    if res.returncode != 0:
                        dst[key] = {"error": res.stderr.strip()[-300:]}
                        continue
                    j = json.loads(res.stdout.strip().splitlines()[-1])
                    j["xRT"] = j["frames_per_s"] / (FS / (M >> r))
                    j["fraction_of_engine_rate"] = j["frames_per_s"] / engine_frames_per_s
                    j["pcm_over_pcie"] = "int16" if i16 else "float32"
                    dst[key] = j
                except (OSError, ValueError, subprocess.TimeoutExpired) as e:
                    dst[key] = {"error": repr(e)}
    return out


def c5_frame_sharded_stage(torch, dist, dev, rank, world):
    """First-contact kit for the multi-GPU node (every rank calls this; DESIGN.md section 6 predicts both curves): BASELINE's C5
    shape -- 256 mics, 2048 bins, ONE stream, static weights -- partitioned by frame range over the ranks (sharding.
    pipeline_frame_sharded: the fused kernel over this rank's frames, one all-gather of Y along the frame axis, synthesis on rank
    0).  Times, max over ranks, for 512- and 4096-frame blocks: the whole block with the all-gather inside the timed region, and the
    all-gather alone.  Strong scaling: the block is the same at every world size."""
    from distant_speech_recognition_amd import engine as eng, prototypes, sharding
    N, M, m, r, dct = 256, 2048, 4, 1, 2
    D, K = M >> r, M // 2 + 1
    h, g = prototypes.load(M, m, r)
    afb = eng.FilterBank(h, M, m, r, dct)
    sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
    gen = torch.Generator(device=dev).manual_seed(4)                 # the same PCM and weights on every rank (replicated input)
    out = {"what": "C5 (256 mics x 2048 bins, 1 stream, static weights) by frame range: fused kernel on T / world frames per rank + ONE "
                   "all-gather of Y (8 K T bytes) + synthesis on rank 0; ms = max over ranks", "world": world}
    for T in (512, 4096):
        L = (T - afb.processing_delay + afb.lookahead) * D
        pcm = (torch.randn((1, N, L), device=dev, generator=gen) * 1000.0).round_()
        W = (torch.randn((K, N), device=dev, generator=gen) + 1j * torch.randn((K, N), device=dev, generator=gen)).to(torch.complex64) / N
        t0r, t1r = sharding.frame_range_for_rank(T, rank, world, 16)
        Yl = torch.zeros((1, K, t1r - t0r), dtype=torch.complex64, device=dev)

        def timed(fn, n=5):
            for _ in range(2):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            return sharding.max_over_ranks((time.perf_counter() - t0) / n, dev)
        t_block = timed(lambda: sharding.pipeline_frame_sharded(afb, sfb, pcm, W, rank, world, synth_rank=0))
        t_gather = timed(lambda: sharding.allgather_frames(Yl, T))
        t_one = timed(lambda: afb.analysis_beamform(pcm, W, t0=t0r, tcount=t1r - t0r)) if t1r > t0r else 0.0
        out["frames_%d" % T] = {"block_ms": t_block * 1e3, "allgather_alone_ms": t_gather * 1e3, "rank_kernel_ms": t_one * 1e3,
                                "frames_per_s": T / t_block, "allgather_bytes": 8 * K * T, "frames_per_rank": t1r - t0r}
        del pcm, W, Yl
    return out


def guarded_c5_stage(torch, dist, dev, rank, world, res, limit_s=300.0):
    """c5_frame_sharded_stage under a timer.  The stage is made of collectives: a rank that fails alone leaves the others waiting
    inside one, and the line would never be printed.  When the timer fires rank 0 prints the line it already holds (`res`) with the
    stage marked as timed out, and every rank ends its process."""
    import threading
    done = threading.Event()

    def fire():
        if done.is_set():
            return
        if rank == 0:
            res["stages"]["c5_frame_sharded"] = {"error": "no result within %.0f s (a rank failed or a collective hung): stage dropped" % limit_s}
            print(json.dumps(res), flush=True)
        os._exit(0)
    timer = threading.Timer(limit_s, fire)
    timer.daemon = True
    timer.start()
    try:
        out = c5_frame_sharded_stage(torch, dist, dev, rank, world)
    except Exception as e:
        out = {"error": repr(e)}
    done.set()
    timer.cancel()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)       # ~1.8 s of timed GPU work at C0
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps run before the warmup until this much wall time has passed: the host-side weight design "
                         "leaves the GPU idle for seconds and its clocks need tens of ms of load to come back up")
    ap.add_argument("--mics", type=int, default=64)
    ap.add_argument("--bins", type=int, default=512)
    ap.add_argument("--streams", type=int, default=32, help="utterance streams per GPU (BASELINE.md section 3: C0 = 32 on 1 GPU)")
    ap.add_argument("--frames", type=int, default=4096, help="frames per stream per step")
    ap.add_argument("--cpu-frames", type=int, default=12000, help="frames of the single-core CPU-baseline sample (~9 s of one core)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--staged", action="store_true",
                    help="time the staged chain (analysis -> HBM snapshots -> apply) instead of the fused kernel")
    args = ap.parse_args()

    # --gpus N is the contract: one rank per GPU.  Launched bare (no torch.distributed.run environment) with N > 1 the
    # script starts its own N ranks; launched by the driver (RANK / WORLD_SIZE set) the two must agree.
    if args.gpus > 1 and "RANK" not in os.environ:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%s -- launch with torch.distributed.run --nproc-per-node %d (or without it: "
                 "the script spawns its own ranks)" % (args.gpus, os.environ.get("WORLD_SIZE", "1"), args.gpus))

    import torch
    from distant_speech_recognition_amd import engine as eng, prototypes
    from distant_speech_recognition_amd.pybeamformer import calc_la_delays

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # one rank per GPU; BTK_DIST_BACKEND=gloo lets a test run several ranks on one device (RCCL refuses that)
    backend = os.environ.get("BTK_DIST_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    # (BTK_BENCH_FORCE_DIST=1: a world of ONE rank still goes through torch.distributed / RCCL -- how a 1-GPU box runs the multi-GPU
    #  code path of this script, the C5 stage included: tests/test_gpu_sharded_2rank.py)
    if world > 1 or os.environ.get("BTK_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    N, M, m, r, dct = args.mics, args.bins, 4, 1, 2
    D, K = M >> r, M // 2 + 1
    S, T = args.streams, args.frames
    h, g = prototypes.load(M, m, r)                           # Nyquist(M) prototypes from the reference's designer (fixtures)
    afb = eng.FilterBank(h, M, m, r, dct)
    sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D          # so that num_frames(L) == T
    assert afb.num_frames(L) == T
    delays = calc_la_delays(ula_positions(N), -1.306379)
    pcm = synth_pcm_device(torch, dev, S, N, L, delays, seed=20260927 + 1000 * rank)

    # SubbandGSC weights: quiescent + blocking matrix + a fixed (non-zero) active weight vector
    wq = eng.weights_mainlobe(M, N, FS, delays)
    rng = np.random.default_rng(0)
    wl = np.zeros((M, N), np.complex128)
    for k in range(1, K):
        B = eng.weights_blocking_matrix(wq[k], 1)
        wl[k] = eng.weights_sidelobe(B, (rng.normal(size=N - 1) + 1j * rng.normal(size=N - 1)) * 0.01)
    W = torch.from_numpy(eng.weights_gsc_effective(wq, wl, M)).to(dev)

    # snapshots and beamformed block with padded rows (engine.padded_rows: 48 frames wider whenever a contiguous row would be a
    # multiple of 4 KiB -- power-of-two row pitches put the 257 bin rows of a tile on the same HBM channels; the C-ABI takes any
    # T_stride >= T).  The staged stages' Y shares the snapshots' row stride (one T_stride in btk_bf_apply / btk_nlms_process).
    X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
    Y = eng.padded_rows((S, K, T), torch.complex64, dev)
    Yc = eng.rows_like(X, (S, K, T))
    if args.staged:
        Y = Yc
    nblk = sfb.num_blocks(T)
    out = torch.empty((S, nblk * D), dtype=torch.float32, device=dev)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    fused = not args.staged

    def step(e=None):
        if e: e[0].record()
        if fused:
            afb.analysis_beamform(pcm, W, out=Y)          # analysis + SubbandGSC apply, snapshots stay on chip
            if e: e[1].record(); e[2].record()
        else:
            afb.analysis(pcm, out=X)
            if e: e[1].record()
            eng.bf_apply(W, X, out=Y)
            if e: e[2].record()
        sfb.synthesize(Y, out=out)
        if e: e[3].record()

    prewarm_steps = 0
    t_pw = time.perf_counter()
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        for _ in range(8):                       # back to back: a host synchronisation after every step lets the clock sag
            step()
        torch.cuda.synchronize()
        prewarm_steps += 8
    from bench_util import ClockPowerSampler
    # shader clock and package power while the timed steps run (sysfs, 2 ms period).  The sampler is set up and its thread started
    # BEFORE the warm-up: between the synchronisation that ends the warm-up and the first timed launch the GPU is idle, and its
    # clock sags within a millisecond or two of idling
    with ClockPowerSampler(torch, dev) as clk:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        if dist: dist.barrier()
        torch.cuda.synchronize()
        clk.begin()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(ev[i])
        torch.cuda.synchronize()
        if dist: dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    my_elapsed = elapsed
    per_rank = [S * T * args.steps / my_elapsed]
    rccl_world = None
    if dist:
        from distant_speech_recognition_amd import sharding
        elapsed = sharding.max_over_ranks(elapsed, dev)
        # every rank's own rate (frames/s between the two barriers) and the world size the communicator itself reports
        mine = torch.tensor([per_rank[0]], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(v.item()) for v in allr]
        rccl_world = dist.get_world_size()

    t_a = np.mean([e[0].elapsed_time(e[1]) for e in ev]) * 1e-3          # fused: analysis+apply kernel
    t_b = np.mean([e[1].elapsed_time(e[2]) for e in ev]) * 1e-3
    t_syn = np.mean([e[2].elapsed_time(e[3]) for e in ev]) * 1e-3

    # per-stage reference measurements of the staged kernels (outside the timed region)
    def _time(fn, n=3):
        from bench_util import gpu_time
        return gpu_time(torch, fn, n=n)[0]
    if fused:
        t_ana = _time(lambda: afb.analysis(pcm, out=X))
        t_bf = _time(lambda: eng.bf_apply(W, X, out=Yc))
    else:
        t_ana, t_bf = t_a, t_b
    # the same fused operator on the samples as they are stored -- 16-bit PCM, widened inside the kernel (btk_fb_analysis_bf_i16):
    # reported BESIDE the float32 headline, with its own algorithmic bytes 2 D N + 8 K per frame; bit-identical output
    pcm16 = pcm.to(torch.int16)
    clk16 = ClockPowerSampler(torch, dev)
    with clk16:
        t_i16 = _time(lambda: afb.analysis_beamform(pcm16, W, out=Y)) if fused and afb.fused_i16() else None
    i16_same = None
    if t_i16 is not None:
        Yf = afb.analysis_beamform(pcm, W)
        Yi = afb.analysis_beamform(pcm16, W)
        i16_same = bool(torch.equal(Yf[..., :T].contiguous().view(torch.float32).view(torch.int32), Yi[..., :T].contiguous().view(torch.float32).view(torch.int32)))
        del Yf, Yi
    # the adaptive variant of the same beamformer (SubbandGSCLMSBeamformer: NLMS canceller on the snapshots), reported
    # next to the static-weight chain; the recursion is sequential in t, so its rate depends on the number of streams
    vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (FS / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
    nst = eng.NLMSState(S, M, N, dev)
    t_nlms = _time(lambda: eng.nlms_process(vs, X, nst, out=Yc))
    t_nlms_one = _time(lambda: eng.nlms_process(vs, X, nst, out=Yc, interleave=(1, T)))

    # the ADAPTIVE chain end to end (north_star's GSC with the NLMS canceller: analysis -> snapshots in HBM -> canceller ->
    # synthesis), at the headline launch and with the same number of frames laid out as four times as many, shorter streams (the
    # recursion is sequential in t: more streams per GPU is what fills the chip, and 288 GB hold them)
    def adaptive(pcm_, X_, Yc_, out_, nst_):
        def chain():
            afb.analysis(pcm_, out=X_)
            eng.nlms_process(vs, X_, nst_, out=Yc_)
            sfb.synthesize(Yc_, out=out_)
        return _time(chain)
    t_chain = adaptive(pcm, X, Yc, out, nst)
    # the same chain on the samples as they are stored (16-bit PCM): the staged bank widens them on its way into LDS
    # (btk_fb_analysis_i16, the same bits) -- reported BESIDE the float32 chain, with the bytes it moves
    t_ana_i16 = t_chain_i16 = None
    chain_i16_same = None
    if afb.analysis_i16():
        t_ana_i16 = _time(lambda: afb.analysis(pcm16, out=X))
        nst_i = eng.NLMSState(S, M, N, dev)
        t_chain_i16 = adaptive(pcm16, X, Yc, out, nst_i)
        Xa = afb.analysis(pcm[:1]); Xb = afb.analysis(pcm16[:1])
        chain_i16_same = bool(torch.equal(Xa.view(torch.float32).view(torch.int32), Xb.view(torch.float32).view(torch.int32)))
        del Xa, Xb, nst_i
    del pcm16
    # the same chain with the analysis bank running ahead of the canceller on a second HIP stream (engine.AdaptiveGSCChain):
    # identical output, the two kernels share the chip instead of taking turns
    pipe = eng.AdaptiveGSCChain(afb, sfb, chunk_frames=512)
    nst_p = eng.NLMSState(S, M, N, dev)
    t_chain_pipe = _time(lambda: pipe(pcm, vs, nst_p, X, Yc, out))
    adaptive_wide = None
    if T % 4 == 0 and not args.no_cpu and world == 1:                       # (skipped with --no-cpu: the profiling runs want the headline launch only)
        S2, T2 = 4 * S, T // 4
        L2 = (T2 - afb.processing_delay + afb.lookahead) * D
        q = (L - L2) // 3
        pcm2 = torch.cat([pcm[:, :, i * q:i * q + L2] for i in range(4)]).contiguous()
        X2 = eng.padded_rows((S2, K, N, T2), torch.complex64, dev)
        Yc2 = eng.rows_like(X2, (S2, K, T2))
        out2 = torch.empty((S2, sfb.num_blocks(T2) * D), dtype=torch.float32, device=dev)
        t2 = adaptive(pcm2, X2, Yc2, out2, eng.NLMSState(S2, M, N, dev))
        adaptive_wide = {"streams": S2, "frames_per_stream": T2, "ms": t2 * 1e3, "frames_per_s": S2 * T2 / t2, "xRT": S2 * T2 / t2 / (FS / D)}
        del pcm2, X2, Yc2, out2

    res = None
    if rank == 0:
        frames_per_step = S * T * world
        value = frames_per_step * args.steps / elapsed
        b_ana = (4 * D + 8 * K) * N * S * T            # algorithmic bytes: analysis, per launch
        b_bf = 8 * K * (N + 1) * S * T                 # beamformer apply
        b_syn = (8 * K + 4 * D) * S * T
        b_fused_hbm = (4 * D * N + 8 * K) * S * T      # what the fused kernel actually has to move
        from bench_util import kernel_source_sha
        traffic_note = {}
        def pmc_traffic(kernel_substr):
            """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/make_traffic_json.py) -- refused (null,
            with the reason in traffic_source) unless they were taken at this launch size AND on the kernel sources of this
            checkout (sha256 of fb_analysis512.hip + fft_packed.h stored with the counters)"""
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", TRAFFIC_JSON)
            try:
                j = json.load(open(path))
            except (OSError, ValueError):
                traffic_note["why"] = "no PMC passes committed (profiles/%s missing)" % TRAFFIC_JSON
                return None
            if (j.get("S"), j.get("T"), j.get("N"), j.get("M")) != (S, T, N, M):
                traffic_note["why"] = "profiles/%s was taken at another launch size: refused" % TRAFFIC_JSON
                return None
            if j.get("kernel_source_sha256") != kernel_source_sha():
                traffic_note["why"] = "profiles/%s was taken on other kernel sources (sha256 differs): refused as stale" % TRAFFIC_JSON
                return None
            for kname, e in j.get("kernels", {}).items():
                if kernel_substr in kname and "traffic_bytes" in e:
                    traffic_note["why"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of profiles/pmc_workload.py at this launch size and "
                                           "on these kernel sources (profiles/%s, kernel %s; gfx950 correction 2 x FETCH_SIZE)" % (TRAFFIC_JSON, kname))
                    return e["traffic_bytes"]
            traffic_note["why"] = "profiles/%s holds no kernel named *%s*: refused" % (TRAFFIC_JSON, kernel_substr)
            return None
        def pmc_traffic_i16():
            """PMC traffic of the int16 launch (same file, key kernels_i16), under the same staleness rules"""
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", TRAFFIC_JSON)
            try:
                j = json.load(open(path))
            except (OSError, ValueError):
                return None
            if (j.get("S"), j.get("T"), j.get("N"), j.get("M")) != (S, T, N, M) or j.get("kernel_source_sha256") != kernel_source_sha():
                return None
            for kname, e in j.get("kernels_i16", {}).items():
                if "analysis512_bfz_kernel" in kname and "traffic_bytes" in e:
                    return e["traffic_bytes"]
            return None
        def compute_side(t_kernel):
            """What the dominant kernel does to the vector ALU next to what it does to HBM: packed instructions issued (ISA count x
            launches) against the issue slots of the launch's duration, and the flops they carry against the vector peak."""
            if FUSED_ISA["kernel_source_sha256"] != kernel_source_sha():
                return {"valu_slot_frac": None, "executed_TFLOPs": None,
                        "why": "the kernel sources changed since the ISA count in bench.py (FUSED_ISA) was taken: refused as stale"}
            n_pk = FUSED_ISA["v_pk_fma_f32"] + FUSED_ISA["v_pk_add_f32"] + FUSED_ISA["v_pk_mul_f32"]
            waves_ch = N * S * T / FUSED_ISA["frames_per_wave"]                 # (wavefront, channel) passes per launch
            flop = waves_ch * 64 * (4 * FUSED_ISA["v_pk_fma_f32"] + 2 * (FUSED_ISA["v_pk_add_f32"] + FUSED_ISA["v_pk_mul_f32"]))
            cyc_per_simd = waves_ch * n_pk * 4 / 1024.0                         # 256 CUs x 4 SIMDs
            return {"packed_instr_per_wave_and_channel": n_pk, "flop_per_launch": flop,
                    "executed_TFLOPs": flop / t_kernel / 1e12, "vector_peak_TFLOPs": VECTOR_PEAK / 1e12,
                    "flop_frac": flop / t_kernel / VECTOR_PEAK,
                    "valu_slot_frac": cyc_per_simd / (t_kernel * SCLK_PEAK),
                    "valu_slot_frac_at_2p1GHz": cyc_per_simd / (t_kernel * 2.1e9),
                    "note": "issue cycles of the packed float32 instructions (4 per wave64 instruction) / cycles of the launch at the "
                            "2.4 GHz peak clock; the part sits at its 1 400 W cap and clocks ~2.1 GHz under this kernel "
                            "(profiles/*_clock_probe.txt), second figure.  Neither roof is reached: two wavefronts per SIMD at 227 "
                            "VGPRs hide LDS and issue latency only partly (DESIGN.md 3.1b)"}
        if fused:
            # algorithmic bytes of the FUSED operator: every PCM sample in once, every beamformed bin out once (4 D N + 8 K
            # per frame); the N x K snapshots that SURVEY 8(d) prices for the staged pair never exist in HBM
            roof = {"bound": "power / issue", "roof": "hbm",
                    "kernel": "analysis512_bfz_kernel (fused analysis bank + SubbandGSC apply)",
                    "achieved": b_fused_hbm / t_a / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                    "frac": b_fused_hbm / t_a / HBM_PEAK, "frac_survey_8d": (b_ana + b_bf) / t_a / HBM_PEAK,
                    "traffic": pmc_traffic("analysis512_bfz_kernel"), "traffic_source": None,
                    "bytes_per_launch": b_fused_hbm, "avg_launch_ms": t_a * 1e3,
                    "compute": compute_side(t_a),
                    "staged_equivalent": {"bytes_per_launch": b_ana + b_bf, "GBps": (b_ana + b_bf) / t_a / 1e9,
                                          "frac": (b_ana + b_bf) / t_a / HBM_PEAK},
                    "clock_and_power_in_timed_region": clk.summary(),
                    "note": "achieved / peak / frac are against the HBM roof (`roof`), bytes_per_launch = 4DN+8K per frame (PCM in, Y out): "
                            "the fused kernel keeps the N x K snapshots on chip.  `bound` says what the counters say limits it: HBM "
                            "traffic is at its floor (1.05 x algorithmic), the vector ALU issues ~40 % of its slots, and the package sits "
                            "at its power cap with the shader clock pulled down (clock_and_power_in_timed_region; DESIGN.md 3.1b).  "
                            "staged_equivalent prices the same launch with SURVEY 8(d)'s staged figures N(4D+8K)+8K(N+1) per frame "
                            "(what analysis + apply through HBM would have to move in that time)"}
        else:
            roof = {"bound": "hbm", "kernel": "analysis512_kernel", "achieved": b_ana / t_ana / 1e9, "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": b_ana / t_ana / HBM_PEAK, "traffic": pmc_traffic("analysis512_kernel"),
                    "traffic_source": None, "bytes_per_launch": b_ana, "avg_launch_ms": t_ana * 1e3}
        roof["traffic_source"] = traffic_note.get("why")
        res = {
            "metric": "beamformed subband frames/sec, 64-mic 512-bin SubbandGSC",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "prewarm_steps": prewarm_steps,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "element": "complex64", "data": "synthetic",
            "xRT": value / (FS / D),
            "config": {"workload": "C0: %d-mic %d-bin SubbandGSC, analysis->GSC apply->synthesis (%s), m=4 r=1 (D=%d), "
                                   "%d streams/GPU x %d frames/step" % (N, M, "fused analysis+apply" if fused else "staged", D, S, T),
                       "streams_per_gpu": S, "frames_per_stream": T, "parallelism": "stream-sharded x%d" % world,
                       "rccl_world": rccl_world, "dist_backend": (backend if dist else None),
                       "per_rank_frames_per_s": per_rank,
                       "y_row_stride_frames": int(Y.stride(1))},
            "roofline": roof,
            "stages": {
                "fused_analysis_apply": ({"ms": t_a * 1e3, "frames_per_s": S * T / t_a,
                                          "hbm_GBps_actual": b_fused_hbm / t_a / 1e9} if fused else None),
                "fused_i16": ({"ms": t_i16 * 1e3, "frames_per_s": S * T / t_i16, "bytes_per_launch": (2 * D * N + 8 * K) * S * T,
                               "GBps": (2 * D * N + 8 * K) * S * T / t_i16 / 1e9, "frac": (2 * D * N + 8 * K) * S * T / t_i16 / HBM_PEAK,
                               "speedup_vs_f32_entry": t_a / t_i16, "bit_identical_to_f32_entry": i16_same,
                               "traffic": pmc_traffic_i16(),
                               "clock_and_power": clk16.summary(),
                               "what": "btk_fb_analysis_bf_i16: the same launch reading the PCM as int16 (as stored in a WAV and as it crosses "
                                       "PCIe) and widening it in registers; algorithmic bytes 2DN+8K per frame; reported beside the float32 "
                                       "headline, never instead of it"} if t_i16 is not None else None),
                "analysis": {"ms": t_ana * 1e3, "GBps": b_ana / t_ana / 1e9, "frac": b_ana / t_ana / HBM_PEAK},
                "gsc_apply": {"ms": t_bf * 1e3, "GBps": b_bf / t_bf / 1e9, "frac": b_bf / t_bf / HBM_PEAK,
                              "frames_per_s": S * T / t_bf},
                "synthesis": {"ms": t_syn * 1e3, "GBps": b_syn / t_syn / 1e9, "frac": b_syn / t_syn / HBM_PEAK},
                "adaptive_nlms_canceller": {"ms": t_nlms * 1e3, "GBps": b_bf / t_nlms / 1e9, "frac": b_bf / t_nlms / HBM_PEAK,
                                            "frames_per_s": S * T / t_nlms,
                                            "one_launch_ms": t_nlms_one * 1e3,
                                            "note": "sequential recursion per (stream, bin): %d streams x %d bin groups = %d single-wavefront workgroups, "
                                                    "2048 resident (249 VGPRs); engine.nlms_process runs a launch beyond that as two staggered groups of "
                                                    "streams on two HIP streams in 512-frame chunks (bit-identical); one_launch_ms = the same work as ONE "
                                                    "launch (rounds 2-5)" % (S, (K + 3) // 4, S * ((K + 3) // 4))},
                "adaptive_chain": {"ms": t_chain * 1e3, "frames_per_s": S * T / t_chain, "xRT": S * T / t_chain / (FS / D),
                                   "what": "analysis -> snapshots [S][K][N][T] in HBM -> NLMS sidelobe canceller -> synthesis, end to end, "
                                           "%d streams x %d frames" % (S, T),
                                   "same_frames_as_more_streams": adaptive_wide},
                "adaptive_chain_i16": (None if t_chain_i16 is None else
                                       {"ms": t_chain_i16 * 1e3, "frames_per_s": S * T / t_chain_i16, "xRT": S * T / t_chain_i16 / (FS / D),
                                        "analysis_ms": t_ana_i16 * 1e3, "analysis_bytes_per_launch": (2 * D + 8 * K) * N * S * T,
                                        "analysis_GBps": (2 * D + 8 * K) * N * S * T / t_ana_i16 / 1e9,
                                        "analysis_frac": (2 * D + 8 * K) * N * S * T / t_ana_i16 / HBM_PEAK,
                                        "snapshots_bit_identical_to_f32_entry": chain_i16_same,
                                        "what": "the adaptive chain from int16 PCM (as stored in a WAV and as it crosses PCIe): btk_fb_analysis_i16 "
                                                "-> snapshots -> NLMS canceller -> synthesis; the bank reads 2 D instead of 4 D bytes per frame and "
                                                "channel (algorithmic bytes 2D+8K per frame and channel); beside the float32 chain, never instead of it"}),
                "adaptive_chain_two_hip_streams": {"ms": t_chain_pipe * 1e3, "frames_per_s": S * T / t_chain_pipe, "xRT": S * T / t_chain_pipe / (FS / D),
                                                   "what": "the same chain, the analysis bank running 512-frame chunks ahead of the canceller on a second "
                                                           "HIP stream (engine.AdaptiveGSCChain); bit-identical output"},
            },
        }
        if not args.no_cpu and world == 1:
            res["stages"]["node_api"] = node_api_stage(h, g, N, M, m, r, value)
            res["cpu_baseline"] = cpu_baseline(N, M, m, r, dct, args.cpu_frames)
        else:
            res["cpu_baseline"] = None
    # multi-GPU runs also carry the strong-scaling partition of C5 (all ranks take part; RCCL only).  It runs LAST, with the line
    # already complete on rank 0 and a timer armed: the extra stage must never cost the driver its headline line
    if dist and backend == "nccl":
        c5_sharded = guarded_c5_stage(torch, dist, dev, rank, world, res)
        if rank == 0:
            res["stages"]["c5_frame_sharded"] = c5_sharded
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist:
        if backend == "nccl" and isinstance(c5_sharded, dict) and "error" in c5_sharded:
            os._exit(0)                                         # (other ranks may still sit in a collective: no orderly shutdown to wait for)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
