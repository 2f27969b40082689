#!/usr/bin/env python
"""A/B of the M = 512 synthesis kernels (BTK_SYN_NARROW=1: the round-2 kernel; unset: the wide-access form) at the bench launch:
time per launch by HIP events and a hash of the output (the two forms must agree bit for bit)."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, gpu_time

dev = torch.device("cuda:0")
res = {}
for r, S, T in ((1, 32, 4096), (1, 1, 4096), (0, 16, 2048)):
    M = 512
    D = M >> r
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, r, 2, synthesis=True)
    g = torch.Generator(device=dev).manual_seed(3)
    Y = eng.padded_rows((S, 257, T), torch.complex64, dev)
    Y.copy_(torch.randn((S, 257, T), device=dev, generator=g) + 1j * torch.randn((S, 257, T), device=dev, generator=g))
    t, o = gpu_time(torch, lambda: sfb.synthesize(Y), n=10)
    odd = sfb.synthesize(Y[:, :, 3:].contiguous())                     # an unaligned view must take the narrow path and still be right
    nb = S * sfb.num_blocks(T)
    res["r%d_S%d" % (r, S)] = {"ms": t * 1e3, "hbm_frac": (8 * 257 + 4 * D) * nb / t / 8e12,
                               "sha": hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16],
                               "sha_odd": hashlib.sha256(odd.cpu().numpy().tobytes()).hexdigest()[:16]}
print(json.dumps({"narrow": bool(os.environ.get("BTK_SYN_NARROW")), "cases": res}))
if os.environ.get("SYN_DUMP"):
    import numpy as np
    np.save(os.environ["SYN_DUMP"], o[:2].cpu().numpy())
