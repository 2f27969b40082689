"""csvdc rule at 64 channels: time against the number of bins in the batch (one residency round or several?), random Hermitian
matrices (QR sweeps) and diagonal ones (reduction only)."""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda:0")
for N in (64, 32):
    for K in (256, 513, 1024, 1025, 2052, 4104):
        g = torch.Generator(device=dev).manual_seed(N)
        A = torch.randn((K, N, N + 8), device=dev, generator=g) + 1j * torch.randn((K, N, N + 8), device=dev, generator=g)
        R = (A @ A.conj().transpose(1, 2) / (N + 8)).to(torch.complex64).contiguous()
        Dg = torch.diag_embed(torch.rand((K, N), device=dev, generator=g) + 0.5).to(torch.complex64).contiguous()
        t_r = gpu_time(torch, lambda: eng.csvdc_values(R), n=3)[0]
        t_d = gpu_time(torch, lambda: eng.csvdc_values(Dg), n=3)[0]
        print("N=%d K=%d: random Hermitian %.2f ms, diagonal (no QR sweeps) %.2f ms" % (N, K, t_r * 1e3, t_d * 1e3), flush=True)
