"""The fused analysis -> apply kernel at the bench launch (32 streams x 64 mics x 4096 frames, M = 512) on float32 PCM against the
same samples as int16 (btk_fb_analysis_bf_i16), alternating in one process, each timed over blocks of 20 back-to-back launches."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays, ClockPowerSampler

dev = torch.device("cuda:0")
N, M, S, T = 64, 512, int(os.environ.get("AB_S", "32")), 4096
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
L = (T - afb.processing_delay + afb.lookahead) * D
g = torch.Generator(device=dev).manual_seed(1)
pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
pcm16 = pcm.to(torch.int16)
wq = eng.weights_mainlobe(M, N, 16000.0, la_delays(ula_positions(N), -1.306379))
W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
Y = eng.padded_rows((S, K, T), torch.complex64, dev)
for _ in range(100):
    afb.analysis_beamform(pcm, W, out=Y)
torch.cuda.synchronize()
res = {"f32": [], "i16": []}
clk = {}
for rep in range(8):
    for name, p in (("f32", pcm), ("i16", pcm16)):
        with ClockPowerSampler(torch, dev) as c:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(10):
                afb.analysis_beamform(p, W, out=Y)
            e0.record()
            for _ in range(60):
                afb.analysis_beamform(p, W, out=Y)
            e1.record()
            torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1) / 60)
        clk[name] = c.summary()
print(json.dumps({"ms_f32": [round(v, 4) for v in res["f32"]], "ms_i16": [round(v, 4) for v in res["i16"]],
                  "median_f32": float(np.median(res["f32"])), "median_i16": float(np.median(res["i16"])),
                  "clock_f32": clk["f32"].get("sclk_MHz"), "power_f32": clk["f32"].get("package_power_W"),
                  "clock_i16": clk["i16"].get("sclk_MHz"), "power_i16": clk["i16"].get("package_power_W")}))
