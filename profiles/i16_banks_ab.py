#!/usr/bin/env python
"""The banks on int16 PCM against the same banks on its float copies (bit-identical snapshots / outputs): staged analysis at
M = 256 ... 2048 (btk_fb_analysis_i16: four samples per typed buffer load) and the fused M = 256 kernel (btk_fb_analysis_bf_i16)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, gpu_time

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
res = {}
for M, S, N, T in ((256, 16, 64, 8192), (512, 32, 64, 4096), (1024, 8, 64, 2048), (2048, 4, 64, 2048)):
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    L = (T - afb.processing_delay + afb.lookahead) * D
    pf = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_().clamp_(-32767, 32767)
    pi = pf.to(torch.int16)
    X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
    e = {}
    for rep in range(2):
        e["staged_f32_ms_%d" % rep] = round(gpu_time(torch, lambda: afb.analysis(pf, out=X))[0] * 1e3, 4)
        e["staged_i16_ms_%d" % rep] = round(gpu_time(torch, lambda: afb.analysis(pi, out=X))[0] * 1e3, 4)
    e["staged_same_bits"] = bool(torch.equal(afb.analysis(pf[:1]).view(torch.float32).view(torch.int32), afb.analysis(pi[:1]).view(torch.float32).view(torch.int32)))
    del X
    if M == 256:
        W = (torch.randn((K, N), device=dev, generator=g) + 1j * torch.randn((K, N), device=dev, generator=g)).to(torch.complex64) / N
        Y = eng.padded_rows((S, K, T), torch.complex64, dev)
        for rep in range(2):
            e["fused_f32_ms_%d" % rep] = round(gpu_time(torch, lambda: afb.analysis_beamform(pf, W, out=Y))[0] * 1e3, 4)
            e["fused_i16_ms_%d" % rep] = round(gpu_time(torch, lambda: afb.analysis_beamform(pi, W, out=Y))[0] * 1e3, 4)
        # what a 16-bit stream at this geometry cost before: the widening pass in front of the float kernel
        buf = torch.empty_like(pf)
        from distant_speech_recognition_amd import _lib
        e["widen_pass_ms"] = round(gpu_time(torch, lambda: _lib.check(_lib.lib().btk_pcm_i16_to_f32(pi.data_ptr(), buf.data_ptr(), pi.numel(), None)))[0] * 1e3, 4)
        del buf, Y
    res["M%d_%dx%dx%d" % (M, S, N, T)] = e
    del pf, pi
print(json.dumps(res))
