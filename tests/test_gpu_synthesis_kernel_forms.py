"""GPU: the wide synthesis kernels of round 3 (synthesis512w_kernel, fast_synthesis_w_kernel: 16-byte loads and stores, history in
registers at M >= 1024) against the ring kernels they replace (BTK_SYN_NARROW=1; read once per process, hence two child processes) on
launches around their limits: fewer frames than one chunk, ragged last chunks, several streams, block sub-ranges (even b0 keeps the wide
path, odd b0 falls back), row-padded inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = [  # (M, r, dct, S, T, b0, bcount or None)
    (512, 1, 2, 3, 5, 0, None), (512, 1, 2, 2, 257, 0, None), (512, 0, 0, 2, 100, 0, None), (512, 1, 2, 1, 600, 32, 200), (512, 1, 2, 1, 600, 33, 100),
    (1024, 1, 2, 3, 7, 0, None), (1024, 1, 2, 2, 161, 0, None), (1024, 0, 2, 2, 90, 0, None), (1024, 1, 0, 1, 400, 48, 130),
    (2048, 1, 2, 3, 3, 0, None), (2048, 1, 2, 2, 83, 0, None), (2048, 0, 0, 2, 50, 0, None), (2048, 1, 2, 1, 300, 16, 100), (2048, 1, 2, 1, 300, 17, 64),
]

# edge launches that are ALSO checked against the oracle (block sub-ranges with even and odd b0, several row-padded streams, a
# launch shorter than one chunk) -- the kernel-against-kernel comparison alone would not notice an error both forms share
ORACLE_CASES = (3, 4, 6, 8, 11, 12, 13)

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from distant_speech_recognition_amd import engine as eng, prototypes
dev = torch.device("cuda:0")
out = {}
for i, (M, r, dct, S, T, b0, bc) in enumerate(%r):
    K = M // 2 + 1
    h, g = prototypes.load(M, 4, 1)
    sfb = eng.FilterBank(g, M, 4, r, dct, synthesis=True)
    gen = torch.Generator(device=dev).manual_seed(7 + i)
    Y = eng.padded_rows((S, K, T), torch.complex64, dev)
    Y.copy_((torch.randn((S, K, T), device=dev, generator=gen) + 1j * torch.randn((S, K, T), device=dev, generator=gen)) * 500)
    o = sfb.synthesize(Y, b0=b0, bcount=bc)
    out["o%%d" %% i] = o.cpu().numpy()
    if i in %r:
        out["y%%d" %% i] = Y.cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def _run(tmp_path, name, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    path = str(tmp_path / (name + ".npz"))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, CASES, ORACLE_CASES), path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


def test_wide_synthesis_kernels_match_the_ring_kernels(dev, tmp_path):
    wide = _run(tmp_path, "wide", {})
    ring = _run(tmp_path, "ring", {"BTK_SYN_NARROW": "1"})
    for i, case in enumerate(CASES):
        a, b = ring["o%d" % i], wide["o%d" % i]
        assert a.shape == b.shape and np.all(np.isfinite(b)), case
        if a.size == 0:                                       # fewer frames than the synthesis delay: no block yet
            continue
        # same arithmetic per output sample up to the contraction order of the multiply-adds
        assert np.max(np.abs(a - b)) <= 2e-6 * max(np.max(np.abs(a)), 1.0), (case, np.max(np.abs(a - b)), np.max(np.abs(a)))


def test_wide_synthesis_edge_launches_match_the_oracle(orc, dev, tmp_path):
    from distant_speech_recognition_amd import prototypes
    wide = _run(tmp_path, "wide_o", {})
    for i in ORACLE_CASES:
        M, r, dct, S, T, b0, bc = CASES[i]
        K, D = M // 2 + 1, M >> r
        g = prototypes.load(M, 4, 1)[1]
        Y, got = wide["y%d" % i], wide["o%d" % i]
        for s_ in range(S):
            Yc = Y[s_].astype(np.complex128)
            full = np.zeros((T, M), np.complex128)
            full[:, :K] = Yc.T
            full[:, K:] = np.conj(Yc.T[:, M // 2 - 1:0:-1])
            ref = orc.synthesis(g, M, 4, r, dct, full).reshape(-1, D)
            ref = ref[b0:] if bc is None else ref[b0:b0 + bc]
            assert got[s_].shape == (ref.size,), (CASES[i], got[s_].shape, ref.shape)
            if ref.size:
                assert np.max(np.abs(got[s_] - ref.reshape(-1))) <= 2e-6 * np.max(np.abs(ref)) * np.sqrt(M), CASES[i]
