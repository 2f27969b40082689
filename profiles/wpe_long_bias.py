"""Long utterances through the float16-split lag-product kernel: do the filters keep their accuracy when the accumulators run over
tens of thousands of frames?  8 channels, lags 0..7, M = 16, T frames of a synthetic reverberant mixture; G of wpe_estimate against
the float64 oracle (oracle/liboracle.so, the checker), for the default kernel and, with BTK_WPE_LAGPROD_F32=1, the float32 one."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from oracle import oracle as orc
dev = torch.device("cuda", 0)
M, C = 16, 8
K = M // 2 + 1
for T in [int(t) for t in os.environ.get("WPE_T", "1000,20000,100000").split(",")]:
    g = torch.Generator(device=dev).manual_seed(3)
    src = (torch.randn((K, T + 16), device=dev, generator=g) + 1j * torch.randn((K, T + 16), device=dev, generator=g)) * 500
    X = torch.zeros((1, K, C, T), dtype=torch.complex64, device=dev)
    for c in range(C):
        for dd in range(6):
            X[0, :, c] += (0.6 ** dd) * np.exp(1j * (c + dd)) * src[:, 16 - dd: 16 - dd + T]
    X += 5 * (torch.randn(X.shape, device=dev, generator=g) + 1j * torch.randn(X.shape, device=dev, generator=g))
    G = eng.wpe_estimate(X, M, lower_num=0, upper_num=7, iterations_num=2, load_db=-18.0, diagonal_bias=1e-4).cpu().numpy()[0]
    Y = np.zeros((T, C, M), np.complex128)
    Xh = X[0].cpu().numpy()                                       # [K][C][T]
    Y[:, :, :K] = Xh.transpose(2, 1, 0)
    Y[:, :, K:] = np.conj(Y[:, :, M // 2 - 1:0:-1])
    Go = orc.wpe_estimate(Y, 0, 7, 2, -18.0, diagonal_bias=1e-4)  # [C][M][C*L]
    Gd = G if G.shape == Go[:, :K].shape else None
    if Gd is None:
        print("shapes", G.shape, Go.shape); break
    ref = Go[:, :K]
    err = np.abs(Gd - ref).max() / np.abs(ref).max()
    print("T=%6d  max |G - G64| / max |G64| = %.3e   (%s)" % (T, err,
          "float32 instruction" if os.environ.get("BTK_WPE_LAGPROD_F32") else "float16 split"), flush=True)
