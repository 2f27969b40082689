#!/usr/bin/env python
"""Batched GPU pseudo-inverse MVDR solve (pinv_kernels.hip) against the round-2 host loop: every bin of a covariance estimated
from fewer frames than microphones fails the Cholesky solve.  PINV_HOST=1 also times the host loop on a sample of bins."""
import os, sys, json, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng, _lib

dev = torch.device("cuda:0")
L = _lib.lib()
res = []
for N, K, T in ((64, 513, 40), (128, 513, 64), (256, 1025, 128)):
    g = torch.Generator(device=dev).manual_seed(N)
    X = (torch.randn((K, N, T), device=dev, generator=g) + 1j * torch.randn((K, N, T), device=dev, generator=g)).to(torch.complex64) * 1000.0
    R = torch.einsum("knt,kmt->knm", X, X.conj()) / T
    R = R + 1.0e-4 * torch.diag_embed(torch.diagonal(R, dim1=1, dim2=2).real.mean(dim=1, keepdim=True).expand(K, N)).to(torch.complex64)
    d = torch.polar(torch.full((K, N), 1.0 / N, device=dev), torch.rand((K, N), device=dev, generator=g) * 6.2831853).to(torch.complex64)
    flags = torch.ones(K, dtype=torch.int32, device=dev)
    W = torch.zeros((K, N), dtype=torch.complex64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    sb = L.btk_mvdr_pinv_scratch_bytes(K, N)
    scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=dev)
    args = (R.data_ptr(), d.data_ptr(), W.data_ptr(), K, N, 1, 1e-8, flags.data_ptr(), cnt.data_ptr(), scratch.data_ptr() if sb else None, None)
    _lib.check(L.btk_mvdr_pinv_fallback_async(*args)); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); _lib.check(L.btk_mvdr_pinv_fallback_async(*args)); e1.record(); torch.cuda.synchronize()
    r = {"N": N, "bins": K, "gpu_ms": e0.elapsed_time(e1), "gpu_us_per_bin": e0.elapsed_time(e1) / K * 1e3, "scratch_MB": sb / 1e6}
    if os.environ.get("PINV_HOST"):
        nb = 4 if N >= 128 else 16
        Wh = torch.zeros((nb, N), dtype=torch.complex64, device=dev)
        ni = C.c_int(0)
        t0 = time.time()
        _lib.check(L.btk_mvdr_pinv_fallback_host(R.data_ptr(), d.data_ptr(), Wh.data_ptr(), nb, N, 1, 1e-8, flags.data_ptr(), C.byref(ni), None))
        r["host_ms_per_bin"] = (time.time() - t0) / nb * 1e3
        r["max_rel_diff_vs_host"] = float(((W[:nb] - Wh).abs().amax(dim=1) / Wh.abs().amax(dim=1)).max())
    res.append(r)
    print(json.dumps(r), flush=True)
