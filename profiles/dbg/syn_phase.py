import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from distant_speech_recognition_amd import engine as eng, _lib
from bench_util import design_prototype
dev = torch.device("cuda:0")
M, S, T = 512, 32, 4096
sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
Y = eng.padded_rows((S, 257, T), torch.complex64, dev); Y.normal_()
for _ in range(3): o = sfb.synthesize(Y)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * 16)()
L.btk_debug_syn_phases(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): o = sfb.synthesize(Y)
e1.record(); torch.cuda.synchronize()
L.btk_debug_syn_phases(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["loop", "A prepass", "barrier1", "prefetch issue", "B fft", "barrier2", "C ola+store", "barrier3"]
tot = v[:8].sum()
print("ms per launch %.4f; WGs %d; cycles per WG %.0f" % (e0.elapsed_time(e1) / 10, v[8] / 10, tot / v[8]))
for n, x in zip(names, v[:8]): print("  %-16s %5.1f%%  %.0f cycles per chunk" % (n, 100 * x / tot, x / v[8] / 17))
