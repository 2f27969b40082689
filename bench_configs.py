#!/usr/bin/env python
"""bench_configs.py -- every configuration of BASELINE.json at its full channel/bin count on one MI355X, stage by
stage (HIP events), with the algorithmic-byte roofline of SURVEY 8(d) per stage.  Not the driver's bench (bench.py
measures the headline C0); results are copied into profiles/ and DESIGN.md.  Synthetic int16-scale input.

  C1 (BASELINE configs[1])  8-mic SubbandGSC, 512 bins, single stream: analysis -> GSC apply -> synthesis
                            (static weights, fused kernel) and the adaptive NLMS / RLS cancellers on the same snapshots
  C2 (configs[2])           64-mic SubbandMVDR, 1024 bins: analysis -> covariance (MFMA HERK) -> diagonal loading ->
                            MVDR solve -> apply -> synthesis
  C3 (configs[3])           8-mic WPE -> SubbandGSC + Zelinski -> synthesis, 16 streams per GPU (128 over 8 GPUs)
  C4 (configs[4])           256-mic super-directive array, 2048 bins: diffuse-noise MVDR design + apply + synthesis
                            (one bin shard per rank with torchrun; world_size 1 here)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM = 8.0e12
FS = 16000.0


def timed(torch, fn, n=3, warm=1, prewarm_ms=250.0):
    """(seconds per call, last result) at settled clocks (bench_util.gpu_time)"""
    from bench_util import gpu_time
    return gpu_time(torch, fn, n=n, prewarm_ms=prewarm_ms)


def pcm_for(torch, dev, afb, S, N, T, D, seed):
    L = (T - afb.processing_delay + afb.lookahead) * D
    g = torch.Generator(device=dev).manual_seed(seed)
    return (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_(), L


def stage(ms, frames, bytes_=None, extra=None):
    d = {"ms": ms * 1e3, "frames_per_s": frames / ms}
    if bytes_:
        d["GBps"] = bytes_ / ms / 1e9
        d["hbm_frac"] = bytes_ / ms / HBM
    if extra:
        d.update(extra)
    return d


def main():
    import torch
    from distant_speech_recognition_amd import engine as eng
    from bench_util import design_prototype, ula_positions, la_delays
    dev = torch.device("cuda:0")
    out = {}

    # ------------------------------------------------------------------ C1
    N, M, S, T = 8, 512, 1, 4096
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    pcm, _ = pcm_for(torch, dev, afb, S, N, T, D, 1)
    delays = la_delays(ula_positions(N), -1.306379)
    wq = eng.weights_mainlobe(M, N, FS, delays)
    W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    t_f, _ = timed(torch, lambda: afb.analysis_beamform(pcm, W, out=Y), n=10, warm=3)
    t_s, _ = timed(torch, lambda: sfb.synthesize(Y), n=10, warm=3)
    X = afb.analysis(pcm)
    vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (FS / M) * delays) / N for k in range(K)])).to(dev)
    nst = eng.NLMSState(S, M, N, dev)
    t_n, _ = timed(torch, lambda: eng.nlms_process(vs.to(torch.complex64), X, nst, out=Y))
    rst = eng.RLSState(1, S, M, N, vs, min_frames=0)
    t_r, _ = timed(torch, lambda: eng.rls_process(X, rst, out=Y))
    b = (N * (4 * D + 8 * K) + 8 * K * (N + 1)) * S * T
    out["C1_8mic_gsc_512bins_1stream"] = {
        "frames": S * T,
        "fused_analysis_gsc_apply": stage(t_f, S * T, b),
        "synthesis": stage(t_s, S * T, (8 * K + 4 * D) * S * T),
        "chain_static": {"ms": (t_f + t_s) * 1e3, "frames_per_s": S * T / (t_f + t_s), "xRT": S * T / (t_f + t_s) / (FS / D)},
        "nlms_canceller": stage(t_n, S * T, 8 * K * (N + 1) * S * T),
        "rls_canceller_f64": stage(t_r, S * T),
        "note": "one stream = 257 sequential recursions for the adaptive cancellers: latency-bound by construction",
    }
    del X, pcm, Y

    # ------------------------------------------------------------------ C2
    N, M, S, T = 64, 1024, 4, 2048
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    pcm, _ = pcm_for(torch, dev, afb, S, N, T, D, 2)
    X = eng.padded_rows((S, K, N, T), torch.complex64, dev)          # row-padded snapshots (engine.padded_rows); Y shares the row stride
    t_a, _ = timed(torch, lambda: afb.analysis(pcm, out=X))
    R = torch.zeros((S, K, N, N), dtype=torch.complex64, device=dev)

    def cov():
        R.zero_()
        return eng.cov_accumulate(X, R=R)
    t_c, _ = timed(torch, cov)
    cnt = torch.full((S,), float(T), dtype=torch.float32, device=dev)
    eng.cov_finalize(R, cnt)
    eng.mvdr_diagonal_loading(R, 100.0)
    delays = la_delays(ula_positions(N), -1.306379)
    wqd = torch.from_numpy(eng.weights_mainlobe(M, N, FS, delays)[:K].astype(np.complex64)).to(dev)
    t_m, (Wm, nfb) = timed(torch, lambda: eng.mvdr_weights(R[0], wqd))      # default svd_rule "linpack": + the reference's csvdc decision per bin
    t_m_exact, _ = timed(torch, lambda: eng.mvdr_weights(R[0], wqd, svd_rule="exact"))
    wqs = wqd.unsqueeze(0).expand(S, K, N).contiguous()
    t_ms, (Wms, nfbs) = timed(torch, lambda: eng.mvdr_weights(R, wqs))      # the S streams' designs in one launch (btk_mvdr_weights_streams)
    t_ms_exact, _ = timed(torch, lambda: eng.mvdr_weights(R, wqs, svd_rule="exact"))
    Y = eng.rows_like(X, (S, K, T))
    t_b, _ = timed(torch, lambda: eng.bf_apply(Wms, X, out=Y))               # per-stream weights
    t_s, _ = timed(torch, lambda: sfb.synthesize(Y))
    tot = t_a + t_c + t_ms + t_b + t_s
    out["C2_64mic_mvdr_1024bins"] = {
        "frames": S * T, "streams": S,
        "analysis": stage(t_a, S * T, (4 * D + 8 * K) * N * S * T),
        "covariance_mfma": stage(t_c, S * T, None, {"TFLOPs": 8.0 * K * N * N * S * T / t_c / 1e12}),
        "mvdr_solve_per_stream": {"ms": t_m * 1e3, "ms_svd_rule_exact": t_m_exact * 1e3, "identity_fallbacks": nfb,
                                  "GFLOPs_of_the_solve": (32.0 / 3) * K * N ** 3 / t_m_exact / 1e9,
                                  "note": "ms = default svd_rule 'linpack' (solve + the reference's float32 csvdc decision per bin, DESIGN 3.7)"},
        "mvdr_solve_all_streams_one_launch": {"ms": t_ms * 1e3, "ms_svd_rule_exact": t_ms_exact * 1e3, "identity_fallbacks": nfbs,
                                              "GFLOPs_of_the_solve": (32.0 / 3) * S * K * N ** 3 / t_ms_exact / 1e9},
        "apply": stage(t_b, S * T, 8 * K * (N + 1) * S * T),
        "synthesis": stage(t_s, S * T, (8 * K + 4 * D) * S * T),
        "chain": {"ms": tot * 1e3, "frames_per_s": S * T / tot, "xRT": S * T / tot / (FS / D)},
    }
    del X, R, pcm, Y

    # ------------------------------------------------------------------ C3
    N, M, S, T = 8, 512, 16, 1000
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    pcm, _ = pcm_for(torch, dev, afb, S, N, T, D, 3)
    t_a, X = timed(torch, lambda: afb.analysis(pcm))
    # unit_test/confs/wpe.json: lower_num 0, upper_num 32 -> 33 lags, P = 8 x 33 = 264 taps per channel and bin
    t_we, G = timed(torch, lambda: eng.wpe_estimate(X, M, 0, 32, 2, -18.0, 0.0, 1e-4), n=2, warm=1)
    t_wa, Xd = timed(torch, lambda: eng.wpe_apply(X, G, M, 0, 32))
    delays = la_delays(ula_positions(N), -1.306379)
    wq = eng.weights_mainlobe(M, N, FS, delays)
    W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
    Dv = torch.from_numpy(wq[:K].astype(np.complex64)).to(dev)
    zs = eng.ZelinskiState(S, K, dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    t_z, _ = timed(torch, lambda: eng.bf_apply_zelinski(W, Dv, Xd, zs, alpha=0.7, out=Y))
    t_s, _ = timed(torch, lambda: sfb.synthesize(Y))
    tot = t_a + t_we + t_wa + t_z + t_s
    out["C3_8mic_wpe_gsc_zelinski_16streams_per_gpu"] = {
        "frames": S * T, "streams": S,
        "analysis": stage(t_a, S * T, (4 * D + 8 * K) * N * S * T),
        "wpe_estimate_2it_lags0to32": stage(t_we, S * T),
        "wpe_apply": stage(t_wa, S * T),
        "gsc_apply_zelinski": stage(t_z, S * T, 8 * K * (N + 1) * S * T),
        "synthesis": stage(t_s, S * T, (8 * K + 4 * D) * S * T),
        "chain": {"ms": tot * 1e3, "frames_per_s": S * T / tot, "xRT_aggregate": S * T / tot / (FS / D),
                  "xRT_per_stream": T / tot / (FS / D)},
    }
    del X, Xd, pcm, G, Y

    # ------------------------------------------------------------------ C4
    N, M, S, T = 256, 2048, 1, 512
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    pcm, _ = pcm_for(torch, dev, afb, S, N, T, D, 4)
    X = eng.padded_rows((S, K, N, T), torch.complex64, dev)          # row-padded snapshots (engine.padded_rows); Y shares the row stride
    t_a, _ = timed(torch, lambda: afb.analysis(pcm, out=X))
    mpos = ula_positions(N, 10.0)
    delays = la_delays(mpos, 0.8)
    wqd = torch.from_numpy(eng.weights_mainlobe(M, N, FS, delays)[:K].astype(np.complex64)).to(dev)

    def design():
        Rd = eng.mvdr_diffuse_model(mpos, M, FS, device=dev)
        eng.mvdr_diagonal_loading(Rd, 0.01)
        return eng.mvdr_weights(Rd, wqd)
    t_m, (Wm, nfb) = timed(torch, design, n=2, warm=1)

    def design_exact():
        Rd = eng.mvdr_diffuse_model(mpos, M, FS, device=dev)
        eng.mvdr_diagonal_loading(Rd, 0.01)
        return eng.mvdr_weights(Rd, wqd, svd_rule="exact")
    t_m_exact4, _ = timed(torch, design_exact, n=2, warm=1)
    Y = eng.rows_like(X, (S, K, T))
    t_b, _ = timed(torch, lambda: eng.bf_apply(Wm, X, out=Y))
    t_s, _ = timed(torch, lambda: sfb.synthesize(Y))
    tot = t_a + t_b + t_s
    # round 4: the superdirective weights are static, so the block runs through the fused analysis -> apply kernel of the large
    # geometries (fb_fused_big.hip): the 256 x 1025 snapshot block never reaches HBM
    Yf = eng.padded_rows((S, K, T), torch.complex64, dev)
    t_f, _ = timed(torch, lambda: afb.analysis_beamform(pcm, Wm, out=Yf))
    t_sf, _ = timed(torch, lambda: sfb.synthesize(Yf))
    out["C4_256mic_superdirective_2048bins"] = {
        "frames": S * T,
        "fused_analysis_apply": stage(t_f, S * T, (4 * D * N + 8 * K) * S * T, {"staged_pair_ms": (t_a + t_b) * 1e3,
                                      "rel_diff_vs_staged": float((Yf - Y).abs().max() / Y.abs().max())}),
        "chain_fused_without_design": {"ms": (t_f + t_sf) * 1e3, "frames_per_s": S * T / (t_f + t_sf), "xRT": S * T / (t_f + t_sf) / (FS / D)},
        "analysis": stage(t_a, S * T, (4 * D + 8 * K) * N * S * T),
        "superdirective_design": {"ms": t_m * 1e3, "ms_svd_rule_exact": t_m_exact4 * 1e3, "identity_fallbacks": nfb,
                                  "GFLOPs_of_the_solve": (32.0 / 3) * K * N ** 3 / t_m_exact4 / 1e9,
                                  "note": "ms = default svd_rule 'linpack': the reference's csvdc returns INFO != 0 on identity_fallbacks bins of this model "
                                          "and the design is delay-and-sum there (DESIGN 3.7)"},
        "apply": stage(t_b, S * T, 8 * K * (N + 1) * S * T),
        "synthesis": stage(t_s, S * T, (8 * K + 4 * D) * S * T),
        "chain_without_design": {"ms": tot * 1e3, "frames_per_s": S * T / tot, "xRT": S * T / tot / (FS / D)},
        "allgather_bytes_if_bin_sharded": 8 * K * T * S,
    }
    del X, pcm, Y

    # ------------------------------------------------------------------ pseudo-inverse fall-back (a12)
    # SMI-MVDR covariances from fewer frames than microphones (unit_test/confs/smimvdr.json): every bin fails the Cholesky solve
    # and takes the reference's float32-SVD pseudo-inverse rule -- the batched GPU Jacobi solve (pinv_kernels.hip)
    def pinv_case(N, K, T, seed):
        g = torch.Generator(device=dev).manual_seed(seed)
        Xs = (torch.randn((K, N, T), device=dev, generator=g) + 1j * torch.randn((K, N, T), device=dev, generator=g)).to(torch.complex64) * 1000.0
        Rs = torch.einsum("knt,kmt->knm", Xs, Xs.conj()) / T
        Rs = Rs + 1.0e-4 * torch.diag_embed(torch.diagonal(Rs, dim1=1, dim2=2).real.mean(dim=1, keepdim=True).expand(K, N)).to(torch.complex64)
        dq = torch.polar(torch.full((K, N), 1.0 / N, device=dev), torch.rand((K, N), device=dev, generator=g) * 6.2831853).to(torch.complex64)
        eng.mvdr_weights(Rs, dq)                                   # warm-up
        t, (Wp, nid) = timed(torch, lambda: eng.mvdr_weights(Rs, dq), n=2, warm=1, prewarm_ms=0.0)
        return {"N": N, "bins": K, "frames_in_covariance": T, "ms_cholesky_attempt_plus_pinv_all_bins": t * 1e3, "identity_fallbacks": nid,
                "us_per_bin": t / K * 1e6}
    out["pinv_fallback_all_bins"] = {"N64_513bins": pinv_case(64, 513, 40, 11), "N256_1025bins_C4_size": pinv_case(256, 1025, 128, 12),
                                     "note": "round 2 (host Jacobi, one thread): 15 ms per 64 x 64 bin, 1.74 s per 256 x 256 bin"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
