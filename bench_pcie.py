#!/usr/bin/env python
"""bench_pcie.py -- the headline chain (C0: 64-mic 512-bin SubbandGSC, analysis -> apply -> synthesis) measured HOST TO HOST:
utterances start and end in pinned host memory.  Reports, for the same utterances,
  resident      the chain alone with the PCM already in HBM (bench.py's number),
  f32_serial    float32 samples uploaded, transformed and downloaded one batch after the other on one stream,
  f32_pipelined float32 samples, three streams (upload / compute / download) and three buffer sets,
  i16_pipelined int16 samples over the link, widened on the device (what SampleFeature's 16-bit WAVs are), int16 output,
  i16_interleaved_pipelined  the same with the samples as a multi-channel WAV stores them ([L][N] frames), de-interleaved on the device,
and the raw pinned host-to-device copy rate, which is what bounds the last two.  Not the driver's bench (bench.py is)."""
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
    ap.add_argument("--mics", type=int, default=64)
    ap.add_argument("--streams", type=int, default=64, help="utterances in host memory")
    ap.add_argument("--batch", type=int, default=8, help="utterances per launch of the chain")
    ap.add_argument("--frames", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from distant_speech_recognition_amd import engine as eng
    from distant_speech_recognition_amd.serving import BatchBeamformerPipeline
    from bench_util import design_prototype, ula_positions, la_delays
    dev = torch.device("cuda:0")
    N, M, m, r, S, B, T = args.mics, 512, 4, 1, args.streams, args.batch, args.frames
    D = M >> r
    afb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    sfb = eng.FilterBank(design_prototype(M, m, "g"), M, m, r, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    rng = np.random.default_rng(3)
    host16 = torch.from_numpy(rng.integers(-3000, 3000, size=(S, N, L), dtype=np.int16)).pin_memory()
    host32 = host16.to(torch.float32).pin_memory()
    delays = la_delays(ula_positions(N), -1.306379)
    wq = eng.weights_mainlobe(M, N, 16000.0, delays)
    W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
    frames = S * T
    res = {"config": {"workload": "C0 host to host: %d utterances x %d mics x %d frames, %d per launch" % (S, N, T, B)}}

    def wall(fn, reps=args.reps):
        fn(); torch.cuda.synchronize()
        best = 1e30
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best

    # raw link rate
    dbuf = torch.empty((B, N, L), dtype=torch.float32, device=dev)
    t = wall(lambda: [dbuf.copy_(host32[i: i + B], non_blocking=True) for i in range(0, S, B)])
    res["h2d_pinned_GBps"] = host32.numel() * 4 / t / 1e9
    # resident chain
    pcm = host32[:B].to(dev)
    Y = eng.padded_rows((B, afb.K, T), torch.complex64, dev)
    out = torch.empty((B, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)
    def resident():
        for _ in range(S // B):
            afb.analysis_beamform(pcm, W, out=Y); sfb.synthesize(Y, out=out)
    t = wall(resident)
    res["resident"] = {"frames_per_s": frames / t, "ms": t * 1e3}
    # serial: one stream, float32
    hout = torch.empty((B, out.shape[1]), dtype=torch.float32).pin_memory()
    def serial():
        for i in range(0, S, B):
            pcm.copy_(host32[i: i + B], non_blocking=True)
            afb.analysis_beamform(pcm, W, out=Y); sfb.synthesize(Y, out=out)
            hout.copy_(out, non_blocking=True)
    t = wall(serial)
    res["f32_serial"] = {"frames_per_s": frames / t, "ms": t * 1e3}
    host16il = host16.permute(0, 2, 1).contiguous().pin_memory()            # [S][L][N]: the frames of a multi-channel WAV as stored
    t0 = time.perf_counter(); _ = np.ascontiguousarray(np.transpose(host16il.numpy()[:4], (0, 2, 1))); t_host_tr = (time.perf_counter() - t0) / 4
    res["host_side_deinterleave_ms_per_utterance"] = t_host_tr * 1e3  # what numpy needs to do the same on one core
    for name, kw, src in (("f32_pipelined", dict(int16_in=False, int16_out=False), host32),
                          ("i16_pipelined", dict(int16_in=True, int16_out=True), host16),
                          ("i16_interleaved_pipelined", dict(int16_in=True, int16_out=True, interleaved=True), host16il)):
        pipe = BatchBeamformerPipeline(afb, sfb, W, N, L, streams_per_batch=B, depth=3, **kw)
        o = np.empty((S, pipe.out_len), np.int16 if kw["int16_out"] else np.float32)
        t = wall(lambda: pipe.run(src, out=o))
        bytes_up = src.numel() * src.element_size()
        res[name] = {"frames_per_s": frames / t, "ms": t * 1e3, "link_GBps": bytes_up / t / 1e9,
                     "bytes_per_frame_over_the_link": bytes_up / frames}
        del pipe
    res["xRT_i16_pipelined"] = res["i16_pipelined"]["frames_per_s"] / (16000.0 / D)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
