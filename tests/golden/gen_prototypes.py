#!/usr/bin/env python
"""Dev-container-only generator of the Nyquist(M) filter-bank prototypes (reference tools/filterbank/
design_nyquist_filter.py:87-278; Kumatani et al., ICASSP 2018) for the geometries the reference does not ship:
M = 512, 1024, 2048 with m = 4, r = 1 (the BASELINE configs).  Output: distant_speech_recognition_amd/prototypes/
nyquist_m4_r1.npz (h_M, g_M float64 [m M]) -- data only; nothing of the reference travels.

How the reference is used (it is imported from /root/reference, never copied):
  * analysis prototype: the reference function design_Nyquist_analyasis_filter_prototype itself (numba is absent:
    an identity `jit` and the numpy aliases it expects are stubbed in);
  * synthesis prototype: the reference function design_Nyquist_synthesis_filter_prototype itself, with ONE helper
    replaced: create_E_f_P is an O(L_g^2 L_max) pure-Python triple loop without numba (hours at M = 256, years at 2048).
    fast_create_E_f_P below computes the same three arrays from correlations; it is checked against the reference loop
    at M = 8..64 (<= 1e-12) and the whole pipeline against the prototypes the reference ships for M = 256
    (unit_test/prototype.ny/{h,g}-M256-m4-r1.pickle, <= 1e-10).
Run: python tests/golden/gen_prototypes.py [--check-only]"""
import importlib.util
import os
import pickle
import sys
import time
import types

import numpy as np

sys.dont_write_bytecode = True          # the reference tree is read-only: importing its designer must not leave a __pycache__ there

REF = "/root/reference/btk20_src"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), "distant_speech_recognition_amd", "prototypes", "nyquist_m4_r1.npz")


def load_reference_designer():
    numba = types.ModuleType("numba")

    def jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    numba.jit = jit
    sys.modules["numba"] = numba
    if not hasattr(np, "float_"):
        np.float_ = np.float64
    if not hasattr(np, "float"):
        np.float = float
    spec = importlib.util.spec_from_file_location("ref_design_nyquist", os.path.join(REF, "tools/filterbank/design_nyquist_filter.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def fast_create_E_f_P(L_g, L_h, M, m, D, tau_t, h):
    """Same arrays as the reference's create_E_f_P (design_nyquist_filter.py:170-196):
         E[i][j] = M^2 sum_{k=0}^{2m} h[kM-i] h[kM-j]            (indices inside [0, L_h))
         P[i][j] = (M / D^2) factor(i-j) sum_l h[l+i] h[l+j]     = factor x autocorrelation of h at lag i-j
         f[i]    = (M / (pi D)) h[tau_t - i]"""
    hv = np.asarray(h, np.float64).reshape(-1)
    E = np.zeros((L_g, L_g))
    idx = np.arange(L_g)
    for k in range(0, 2 * m + 1):
        src = k * M - idx
        ok = (src >= 0) & (src < L_h)
        v = np.where(ok, hv[np.clip(src, 0, L_h - 1)], 0.0)
        E += np.outer(v, v)
    full = np.correlate(hv, hv, mode="full")               # lag d at index d + L_h - 1
    lag = idx[:, None] - idx[None, :]
    r = np.where(np.abs(lag) < L_h, full[np.clip(np.abs(lag), 0, L_h - 1) + L_h - 1], 0.0)
    factor = np.where(lag % D == 0, D - 1.0, -1.0)
    P = r * factor
    f = np.zeros((L_g, 1))
    src = tau_t - idx
    ok = (src >= 0) & (src < L_h)
    f[ok, 0] = hv[src[ok]]
    E = ((M * M) / float(D / D)) * E
    f = (M / (np.pi * D)) * f
    P = (M / float(D * D)) * P
    return E, f, P


def design(ref, M, m, r):
    D = M // (2 ** r)
    t = time.time()
    h, beta = ref.design_Nyquist_analyasis_filter_prototype(M, m, D)
    t1 = time.time()
    g, eps = ref.design_Nyquist_synthesis_filter_prototype(h, M, m, D)
    print("M=%d m=%d r=%d: analysis %.1f s (inband aliasing %.1f dB), synthesis %.1f s (residual aliasing %.1f dB)"
          % (M, m, r, t1 - t, 10 * np.log10(float(beta)), time.time() - t1, 10 * np.log10(abs(float(eps)))), flush=True)
    return np.asarray(h, np.float64).reshape(-1), np.asarray(g, np.float64).reshape(-1)


def main():
    ref = load_reference_designer()
    slow = ref.create_E_f_P
    # 1. the replacement helper == the reference loop (small sizes the pure-Python loop finishes)
    for (M, m, r) in ((8, 2, 1), (16, 4, 1), (32, 4, 2), (64, 2, 0)):
        D = M // (2 ** r)
        h, _ = ref.design_Nyquist_analyasis_filter_prototype(M, m, D)
        L = M * m
        tau_t = int(L / 2 + L / 2)
        a = slow(L, L, M, m, D, tau_t, h)
        b = fast_create_E_f_P(L, L, M, m, D, tau_t, h)
        err = max(float(np.max(np.abs(x - y))) / max(float(np.max(np.abs(x))), 1e-300) for x, y in zip(a, b))
        assert err < 1e-12, (M, m, r, err)
        print("create_E_f_P restatement vs reference loop M=%d m=%d r=%d: %.1e" % (M, m, r, err))
    ref.create_E_f_P = fast_create_E_f_P
    # 2. the whole pipeline == the prototypes the reference ships
    h256, g256 = design(ref, 256, 4, 1)
    hs = pickle.load(open(os.path.join(REF, "unit_test/prototype.ny/h-M256-m4-r1.pickle"), "rb"), encoding="latin1")
    gs = pickle.load(open(os.path.join(REF, "unit_test/prototype.ny/g-M256-m4-r1.pickle"), "rb"), encoding="latin1")
    eh, eg = float(np.max(np.abs(h256 - hs))), float(np.max(np.abs(g256 - gs)))
    print("M=256 vs shipped pickles: h %.1e, g %.1e (max |g| %.3g)" % (eh, eg, float(np.max(np.abs(gs)))))
    assert eh < 1e-10 and eg < 1e-10 * max(1.0, float(np.max(np.abs(gs))))
    if "--check-only" in sys.argv:
        return
    out = {}
    for M in (512, 1024, 2048):
        h, g = design(ref, M, 4, 1)
        out["h_%d" % M], out["g_%d" % M] = h, g
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
