"""csvdc: time of the two launches against the batch shape, for BTK_CSVDC_QRW bins per QR workgroup (set in the environment)."""
import sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda:0")
out = []
for N, K in ((64, 1024), (64, 2048), (128, 513), (256, 256), (256, 1024)):
    g = torch.Generator(device=dev).manual_seed(N)
    A = torch.randn((K, N, N + 8), device=dev, generator=g) + 1j * torch.randn((K, N, N + 8), device=dev, generator=g)
    R = (A @ A.conj().transpose(1, 2) / (N + 8)).to(torch.complex64).contiguous()
    t_r = gpu_time(torch, lambda: eng.csvdc_values(R), n=2)[0]
    out.append("N=%d K=%d %.2f" % (N, K, t_r * 1e3))
print("QRW=%s SPLIT=%s: " % (os.environ.get("BTK_CSVDC_QRW", "4"), os.environ.get("BTK_CSVDC_SPLIT", "1")) + " | ".join(out), flush=True)
