"""C1 chain (8 mics, 512 bins, one stream of 4096 frames) eager vs captured in a HIP graph (torch.cuda.CUDAGraph around the
C-ABI launches): the chain is two kernels of ~20-40 us, i.e. launch-bound when utterances are processed one by one."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays
dev = torch.device("cuda:0")
N, M, S, T = 8, 512, 1, 4096
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * D
pcm = (torch.randn((S, N, L), device=dev) * 1000).round_()
wq = eng.weights_mainlobe(M, N, 16000.0, la_delays(ula_positions(N), -1.3))
W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
out = torch.empty((S, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)

def chain():
    afb.analysis_beamform(pcm, W, out=Y)
    sfb.synthesize(Y, out=out)

for _ in range(5): chain()
torch.cuda.synchronize()
ref = out.clone()
n = 200
t0 = time.perf_counter()
for _ in range(n): chain()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / n
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): chain()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    chain()
out.zero_()
g.replay(); torch.cuda.synchronize()
assert torch.equal(out, ref), "graph replay differs"
t0 = time.perf_counter()
for _ in range(n): g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / n
print("C1 chain per utterance: eager %.1f us, hipGraph replay %.1f us (%.2fx), %.1f M frames/s" % (eager * 1e6, graph * 1e6, eager / graph, T / graph / 1e6))
