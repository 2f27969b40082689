#!/usr/bin/env python
"""bench_bin_sharded.py -- BASELINE config C5 (256-mic super-directive array, 2048 bins) sharded by frequency-bin range:
`python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 bench_bin_sharded.py --gpus G`.
Every rank designs the MVDR weights of ITS bins (the O(K N^3) part), analyses the replicated PCM, beamforms its bin range;
one RCCL all-gather of Y (8 K T S bytes) precedes the synthesis on rank 0.  Strong scaling: the total work is fixed.
Prints one JSON line on rank 0.  (Not the driver's bench: bench.py measures the stream-sharded headline.)"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mics", type=int, default=256)
    ap.add_argument("--bins", type=int, default=2048)
    ap.add_argument("--frames", type=int, default=512)
    # "replicated" = option (i), every rank transforms all channels of the same PCM and stores its bins (no exchange
    # before the beamformer); "channels" = option (ii), every rank holds and transforms N / world channels and ONE all-to-all
    # regroups the snapshots by bin (predicted 2.7 x at 8 GPUs against 1.6 x, DESIGN.md section 6 -- to be decided on the node)
    # "frames": static weights only -- no bin shards at all: every rank runs the FUSED analysis -> beamformer kernel over its range of
    # frames and the one all-gather runs along the frame axis (sharding.pipeline_frame_sharded)
    # round 5: the default -- this bench's weights (diffuse-noise MVDR) ARE static; "replicated" / "channels" are what adaptive
    # statistics per bin need
    ap.add_argument("--analysis-input", choices=["replicated", "channels", "frames"], default="frames")
    args = ap.parse_args()
    import torch
    from distant_speech_recognition_amd import engine as eng, sharding
    from bench_util import design_prototype, ula_positions, la_delays
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    N, M, T, S = args.mics, args.bins, args.frames, 1
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    g = torch.Generator(device=dev).manual_seed(4)                     # the same PCM on every rank (replicated input)
    pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
    mpos = ula_positions(N, 10.0)
    k0, k1 = sharding.bin_range_for_rank(K, rank, world)
    wq = torch.from_numpy(eng.weights_mainlobe(M, N, 16000.0, la_delays(mpos, 0.8))[:K].astype(np.complex64)).to(dev)

    def design():
        Rd = eng.mvdr_diffuse_model(mpos, M, 16000.0, device=dev)[k0:k1].contiguous()
        eng.mvdr_diagonal_loading(Rd, 0.01)
        Wl, _ = eng.mvdr_weights(Rd, wq[k0:k1].contiguous(), first_bin=k0)
        return Wl
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    W_local = design()
    torch.cuda.synchronize()
    t_design = time.perf_counter() - t0

    # what the shard saves on ONE GPU (options (i) vs full): the analysis bank storing K/8 of the bins vs all
    shard_probe = None
    if world == 1:
        def _t(fn, n=3):
            fn(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / n
        a8, b8 = sharding.bin_range_for_rank(K, 3, 8)
        Xf = torch.empty((S, K, N, T), dtype=torch.complex64, device=dev)
        Xs = torch.empty((S, b8 - a8, N, T), dtype=torch.complex64, device=dev)
        t_full = _t(lambda: afb.analysis(pcm, out=Xf))
        t_shard = _t(lambda: afb.analysis(pcm, out=Xs, bins=(a8, b8)))
        t_slice = _t(lambda: Xf[:, a8:b8].contiguous())
        Wf = torch.randn((K, N), dtype=torch.complex64, device=dev)
        t_bf_full = _t(lambda: eng.bf_apply(Wf, Xf))
        t_bf_shard = _t(lambda: eng.bf_apply(Wf[a8:b8].contiguous(), Xs))
        # option (iii) of SURVEY 8(e): the rank's K/8 bins as a pruned DFT = GEMM (K/8 x M) . (M x N T) on the polyphase outputs.
        # Priced here with the vendor fp32 GEMM (torch.matmul -> hipBLASLt; a measurement proxy, not a product path): the cos and
        # sin planes of the rank's rows against the real polyphase block, 4 (K/8) M N T flop
        Pm = torch.randn((M, N * T), dtype=torch.float32, device=dev)
        Cm = torch.randn((2 * (b8 - a8), M), dtype=torch.float32, device=dev)
        t_gemm = _t(lambda: torch.matmul(Cm, Pm))
        gemm_flop = 2.0 * 2 * (b8 - a8) * M * N * T
        t_poly = _t(lambda: afb.analysis_polyphase(pcm))      # the polyphase stage that would still precede the GEMM
        # option (ii): every rank transforms N/8 channels completely, then an all-to-all hands each rank its bins of all channels:
        # volume per block 8 K N T bytes in total, 7/8 of a rank's 1/8 leaves it over its 7 xGMI links (153 GB/s each, MI355X guide)
        X8 = torch.empty((S, K, N // 8, T), dtype=torch.complex64, device=dev)
        t_ana8 = _t(lambda: afb.analysis(pcm[:, : N // 8].contiguous(), out=X8))
        a2a_bytes_per_link = 8.0 * K * N * T * S / 8 / 8
        del Pm, Cm, X8
        shard_probe = {"option_iii_pruned_dft_gemm_ms": t_gemm, "option_iii_gemm_TFLOPs": gemm_flop / t_gemm / 1e9,
                       "option_iii_polyphase_stage_ms": t_poly,
                       "option_ii_analysis_of_N_over_8_channels_ms": t_ana8, "option_ii_all_to_all_bytes_per_link": a2a_bytes_per_link,
                       "option_ii_all_to_all_ms_at_153GBps_per_link": a2a_bytes_per_link / 153e9 * 1e3,
                       "analysis_all_bins_ms": t_full, "analysis_one_of_8_bin_shards_ms": t_shard,
                       "slice_copy_the_round1_path_needed_ms": t_slice, "bf_apply_all_bins_ms": t_bf_full,
                       "bf_apply_one_shard_ms": t_bf_shard,
                       "note": "rank-side cost of option (i) (replicated PCM + FFT, only the rank's bins stored) on one GPU"}
        del Xf, Xs

    pcm_in = pcm
    if args.analysis_input == "channels" and world > 1:
        c0, c1 = sharding.bin_range_for_rank(N, rank, world)
        pcm_in = pcm[:, c0:c1].contiguous()                 # this rank's microphones only

    W_all = None
    if args.analysis_input == "frames":
        if world > 1:                                       # every rank needs the whole weight set: gather the designed shards once
            parts = [torch.zeros((-(-K // world), N), dtype=torch.complex64, device=dev) for _ in range(world)]
            mine = torch.zeros_like(parts[0]); mine[: k1 - k0] = W_local
            dist.all_gather(parts, mine)
            W_all = torch.cat(parts)[:K].contiguous()
        else:
            W_all = W_local

    def step():
        if args.analysis_input == "frames":
            return sharding.pipeline_frame_sharded(afb, sfb, pcm, W_all, rank, world, synth_rank=0)
        return sharding.pipeline_bin_sharded(afb, sfb, pcm_in, W_local, K, rank, world, synth_rank=0, analysis_input=args.analysis_input)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, Y = step()
    torch.cuda.synchronize()
    if dist: dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if dist:
        el = sharding.max_over_ranks(el, dev)
        td = sharding.max_over_ranks(t_design, dev)
    else:
        td = t_design
    if rank == 0:
        print(json.dumps({"metric": "beamformed subband frames/sec, %d-mic %d-bin super-directive, bin-sharded" % (N, M),
                          "value": S * T * args.steps / el, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
                          "scaling": "strong", "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "C5: %d mics, %d bins, %d frames, bins [%d,%d) on rank 0 of %d" % (N, M, T, k0, k1, world),
                                     "analysis_input": args.analysis_input,
                                     "parallelism": "bin-sharded x%d, 1 all-gather of %d bytes per step%s" % (
                                         world, 8 * K * T * S, ", 1 all-to-all of %d bytes before the beamformer" % (8 * K * N * T * S) if args.analysis_input == "channels" and world > 1 else "")},
                          "weight_design_ms": td * 1e3, "pcm_checksum": float(out.double().abs().sum()),
                          "shard_probe": shard_probe}))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
