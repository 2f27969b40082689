import sys, json
sys.path.insert(0, "/root/repo")
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda:0")
out = {}
for N, K in ((8, 257), (20, 257), (64, 257), (100, 257)):
    g = torch.Generator(device=dev).manual_seed(N)
    def spd(extra):
        A = torch.randn((K, N, N + extra), device=dev, generator=g) + 1j * torch.randn((K, N, N + extra), device=dev, generator=g)
        return (A @ A.conj().transpose(1, 2) / (N + extra) + 0.01 * torch.eye(N, device=dev)).to(torch.complex64).contiguous()
    Rt, Rn = spd(2), spd(8)
    t = gpu_time(torch, lambda: eng.gev_weights(Rt, Rn), n=3)[0]
    out["N%d" % N] = round(t * 1e3, 3)
print(json.dumps(out))
