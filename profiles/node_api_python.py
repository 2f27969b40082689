#!/usr/bin/env python
"""What a PYTHON caller of the node API gets at C0 (64 SampleFeature -> 64 banks -> SubbandGSC -> synthesis; the reference's
unit_test/test_online_beamforming.py pattern `for b in synthesis_bank:`): the per-block loop against next_blocks() (engine
extension: a round per call).  Second pass timed, 16-bit streams."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr, OverSampledDFTSynthesisBankPtr)
from bench_util import design_prototype, ula_positions, la_delays

M, m, r, N = 512, 4, 1, 64
D = M >> r
T = int(os.environ.get("T", "32768"))
BF = int(os.environ.get("BLOCK_FRAMES", "8192"))
h, g = design_prototype(M, m), design_prototype(M, m, "g")
rng = np.random.default_rng(3)
delays = la_delays(ula_positions(N), -1.306379)
srcs, keep = [], []
bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
for c in range(N):
    sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
    x = rng.integers(-3000, 3000, size=T * D).astype(np.float32)
    sf.set_samples(x)
    a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2)
    a.set_block_frames(BF)
    bf.set_channel(a)
    srcs.append((sf, x)); keep.append(a)
bf.calc_gsc_weights(16000, delays)
sfb = OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)


def reload():
    for sf, x in srcs:
        sf.set_samples(x)


res = {"frames": T, "block_frames": BF}
for mode in ("per_block_loop", "next_blocks"):
    for rep in range(2):
        reload()
        sfb.reset()
        t0 = time.perf_counter()
        n, acc = 0, 0.0
        if mode == "per_block_loop":
            for b in sfb:
                n += 1
                acc += float(np.asarray(b)[0])
        else:
            while True:
                a = sfb.next_blocks()
                if a.shape[0] == 0:
                    break
                n += a.shape[0]
                acc += float(a[:, 0].sum())
        dt = time.perf_counter() - t0
    res[mode] = {"blocks": n, "wall_s": round(dt, 5), "frames_per_s": round(n / dt, 1), "checksum": round(acc, 3)}
print(json.dumps(res))
