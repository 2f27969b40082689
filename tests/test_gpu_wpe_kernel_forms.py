"""GPU: the round-3 WPE kernels (normal equations as lag products, prediction on the matrix cores) against the round-2 kernels they
replace (block HERK, vector prediction; selected with BTK_WPE_HERK_BLOCKS / BTK_WPE_PREDICT_VALU, which the library reads once per
process -- hence two child processes) on shapes around their limits: fewer frames than one tile, one frame, the smallest and largest
lag counts each kernel accepts, delayed prediction, 4 / 8 / 16 channels, several streams."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [  # (S, C, M, T, lower, upper, iterations)
    (1, 8, 16, 1, 0, 7, 1), (2, 8, 16, 37, 0, 32, 2), (1, 8, 16, 64, 1, 4, 1), (1, 8, 16, 65, 2, 9, 2), (1, 8, 8, 700, 0, 64, 1),
    (3, 4, 16, 129, 0, 7, 2), (1, 4, 16, 300, 3, 20, 1), (1, 16, 8, 200, 0, 3, 1), (2, 6, 16, 90, 1, 5, 2), (1, 8, 16, 1000, 0, 32, 2),
]

CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from distant_speech_recognition_amd import engine as eng
dev = torch.device("cuda:0")
out = {}
for i, (S, C, M, T, lo, up, it) in enumerate(%r):
    K = M // 2 + 1
    g = torch.Generator(device=dev).manual_seed(100 + i)
    src = torch.randn((S, K, 1, T + 12), device=dev, generator=g) + 1j * torch.randn((S, K, 1, T + 12), device=dev, generator=g)
    taps = (torch.randn((S, K, C, 12), device=dev, generator=g) + 1j * torch.randn((S, K, C, 12), device=dev, generator=g)) * (0.7 ** torch.arange(12, device=dev))
    X = torch.zeros((S, K, C, T), dtype=torch.complex64, device=dev)
    for d in range(12):
        X += (taps[..., d:d + 1] * src[..., 12 - d: 12 - d + T]).to(torch.complex64) * 300
    G = eng.wpe_estimate(X, M, lower_num=lo, upper_num=up, iterations_num=it, load_db=-18.0, diagonal_bias=1e-4)
    Y = eng.wpe_apply(X, G, M, lower_num=lo, upper_num=up)
    out["G%%d" %% i] = G.cpu().numpy(); out["Y%%d" %% i] = Y.cpu().numpy()
np.savez(sys.argv[1], **out)
'''


def _run(tmp_path, name, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    path = str(tmp_path / (name + ".npz"))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, SHAPES), path], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(path)


def test_round3_wpe_kernels_match_the_kernels_they_replace(dev, tmp_path):
    new = _run(tmp_path, "new", {})
    old = _run(tmp_path, "old", {"BTK_WPE_HERK_BLOCKS": "1", "BTK_WPE_PREDICT_VALU": "1"})
    for i, shape in enumerate(SHAPES):
        G0, G1, Y0, Y1 = old["G%d" % i], new["G%d" % i], old["Y%d" % i], new["Y%d" % i]
        assert np.all(np.isfinite(G1)) and np.all(np.isfinite(Y1)), shape
        gs, ys = max(np.max(np.abs(G0)), 1e-6), np.max(np.abs(Y0))
        # both are float32 normal equations solved by the same Cholesky: they differ by the summation order of the accumulations only
        assert np.max(np.abs(G1 - G0)) <= 2e-3 * gs, (shape, np.max(np.abs(G1 - G0)) / gs)
        assert np.max(np.abs(Y1 - Y0)) <= 1e-3 * ys, (shape, np.max(np.abs(Y1 - Y0)) / ys)


def test_register_resident_solver_matches_the_panel_solver(dev, tmp_path):
    """round 4: chol_reg.h (matrix in the accumulator registers, default for 112 <= P <= 271) forced for every P <= 271 against the panel
    solver of chol_blocked.h forced for every P, same normal equations: P = 30 ... 264, one shape above the register solver's limit."""
    reg = _run(tmp_path, "reg", {"BTK_WPE_SOLVE_REG": "1"})
    pan = _run(tmp_path, "pan", {"BTK_WPE_SOLVE_PANEL": "1"})
    for i, shape in enumerate(SHAPES):
        G0, G1, Y0, Y1 = pan["G%d" % i], reg["G%d" % i], pan["Y%d" % i], reg["Y%d" % i]
        assert np.all(np.isfinite(G1)) and np.all(np.isfinite(Y1)), shape
        gs, ys = max(np.max(np.abs(G0)), 1e-6), np.max(np.abs(Y0))
        assert np.max(np.abs(G1 - G0)) <= 2e-3 * gs, (shape, np.max(np.abs(G1 - G0)) / gs)
        assert np.max(np.abs(Y1 - Y0)) <= 1e-3 * ys, (shape, np.max(np.abs(Y1 - Y0)) / ys)


@pytest.mark.parametrize("switch", ["BTK_WPE_SOLVE_REG", "BTK_WPE_SOLVE_PANEL"])
def test_both_solvers_against_the_oracle(dev, switch):
    """the oracle tests of tests/test_gpu_wpe.py once with each solver forced (the default picks by size: small systems would never
    reach the register solver, the reference configuration never the panel solver)"""
    env = dict(os.environ)
    env[switch] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_wpe.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])
    assert " passed" in r.stdout
