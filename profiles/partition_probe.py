"""Which kernels give the same BITS for a frame / block whatever launch partition computes it?  (The node layer's block-size
invariance rests on it.)  Fused analysis -> apply: whole launch against pieces with other t0 / tcount / truncated sample windows
(the window a bounded block uploads ends right behind the block's last frame: tiles that are interior in the whole launch are
edge tiles there).  Synthesis: whole launch against pieces taken from windows with other row strides / b0 parities."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from tests.util import design_prototype, synthetic_pcm

dev = torch.device("cuda:0")
for M in (256, 512, 1024, 2048):
    m, r, dct, N = 4, 1, 2, 8
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    afb = eng.FilterBank(h, M, m, r, dct)
    sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
    T = 200
    pcm, delays = synthetic_pcm(1, N, (T + 8) * D, seed=9)
    pd = torch.from_numpy(pcm).to(dev)
    rng = np.random.default_rng(1)
    W = torch.from_numpy((rng.normal(size=(afb.K, N)) + 1j * rng.normal(size=(afb.K, N))).astype(np.complex64) / N).to(dev)
    nfr = afb.num_frames(pd.shape[-1])
    Y = afb.analysis_beamform(pd, W)[..., :nfr].contiguous()
    laN = 3
    worst = 0
    for (t0, tc) in ((0, 16), (16, 16), (16, 24), (40, 37), (77, 50), (5, 11), (127, nfr - 127)):
        # (a) same samples, other (t0, tcount)
        Ya = afb.analysis_beamform(pd, W, t0=t0, tcount=tc)[..., :tc]
        da = int((Ya.contiguous().view(torch.float32).view(torch.int32) != Y[..., t0:t0 + tc].contiguous().view(torch.float32).view(torch.int32)).sum())
        # (b) the window a bounded block uploads: blocks b0 .. t0 + tc + laN of the input
        b0 = max(0, t0 + laN + 1 - m * 2)
        b1 = min(pd.shape[-1] // D, t0 + tc + laN)
        win = pd[..., b0 * D:b1 * D].contiguous()
        if t0 + tc <= nfr - 8:
            Yb = afb.analysis_beamform(win, W, t0=t0 - b0, tcount=tc)[..., :tc]
            db = int((Yb.contiguous().view(torch.float32).view(torch.int32) != Y[..., t0:t0 + tc].contiguous().view(torch.float32).view(torch.int32)).sum())
        else:
            db = -1
        worst = max(worst, da, db)
        print("M=%d fused  t0=%3d tcount=%3d: differing words same-window %d, block-window %d" % (M, t0, tc, da, db))
    out = sfb.synthesize(Y)
    nb = out.shape[-1] // D
    pds = 4
    H = 10
    for (b0, bc, hist, pad) in ((0, 29, 0, 0), (29, 32, 10, 0), (61, 31, 10, 0), (61, 31, 10, 1), (92, 50, 9, 0), (92, 50, 10, 3), (142, nb - 142, 10, 0)):
        # the round window of the node layer: frames [b0 + pd - hist - (frames of the round)...]: build [hist | frames base .. base + Tn)
        base = b0 + pds if b0 > 0 else 0
        Tn = bc if b0 > 0 else bc + pds
        if base + Tn > nfr:
            Tn = nfr - base
        w0 = base - hist
        if w0 < 0:
            continue
        Lw = hist + Tn
        buf = torch.zeros((1, afb.K, Lw + pad), dtype=torch.complex64, device=dev)
        buf[..., :Lw] = Y[..., w0:w0 + Lw]
        view = buf[..., :Lw]
        o = torch.empty((1, bc * D), dtype=torch.float32, device=dev)
        eng.check(eng._lib.lib().btk_fb_synthesis(sfb._h, view.data_ptr(), Lw, Lw + pad, 1, o.data_ptr(), bc * D, b0 - w0, bc, 0))
        torch.cuda.synchronize()
        d = int((o.view(torch.int32) != out[:, b0 * D:(b0 + bc) * D].contiguous().view(torch.int32)).sum())
        print("M=%d synth  b0=%3d bcount=%3d hist=%2d stride=%d: differing words %d (max |diff| %.3g)" % (
            M, b0, bc, hist, Lw + pad, d, float((o - out[:, b0 * D:(b0 + bc) * D]).abs().max())))
