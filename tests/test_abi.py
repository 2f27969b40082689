"""CPU: the C-ABI library loads and exports every symbol include/btkhip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "btkhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(btk_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_symbols():
    syms = _declared_symbols()
    assert "btk_fb_analysis" in syms and "btk_bf_apply" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol():
    from distant_speech_recognition_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    # the ctypes table mirrors the header one to one
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_host_side_entry_points_without_gpu(orc):
    """Weight design is host code (no device needed): compare with the oracle."""
    import numpy as np
    from distant_speech_recognition_amd import engine
    from tests.util import ula_positions, la_delays
    M, N = 64, 8
    delays = la_delays(ula_positions(N), -1.306379)
    wq = engine.weights_mainlobe(M, N, 16000, delays)
    assert np.max(np.abs(wq - orc.calc_mainlobe(M, N, 16000, delays))) < 1e-15
    wh = engine.weights_mainlobe(M, N, 16000, delays, half_band_shift=True)          # beamformer.cc:515-527
    assert np.max(np.abs(wh - orc.calc_mainlobe_halfband(M, N, 16000, delays))) < 1e-15
    assert np.max(np.abs(wh[M - 1] - np.conj(wh[0]))) == 0.0 and abs(np.angle(wh[0][0] * N) + np.pi * 16000 * delays[0] / M) < 1e-12
    for k in (0, 3, 32, 40):
        B = engine.weights_blocking_matrix(wq[k], 1)
        assert np.max(np.abs(B - orc.blocking_matrix(wq[k], 1))) < 1e-12
        wa = np.random.default_rng(k).normal(size=N - 1) + 1j * np.random.default_rng(k + 1).normal(size=N - 1)
        assert np.max(np.abs(engine.weights_sidelobe(B, wa) - orc.sidelobe_canceller(B, wa))) < 1e-12
    with pytest.raises(Exception):
        engine.weights_blocking_matrix(wq[1][:1], 1)       # N - NC <= 0 -> dimension error


def test_error_reporting_no_gpu():
    from distant_speech_recognition_amd import _lib
    L = _lib.lib()
    h = ctypes.c_void_p()
    import numpy as np
    proto = np.zeros(12, np.float64)
    rc = L.btk_fb_create(ctypes.byref(h), 100, 4, 1, 0, 0, proto.ctypes.data_as(ctypes.c_void_p))
    assert rc == _lib.BTK_ERR_PARAMETER
    assert b"power of two" in L.btk_last_error()


def test_lcmv_weights_host(orc):
    """calcMainlobe2 (LCMV, NC=2): product vs oracle, distortionless + null known answers."""
    import numpy as np
    from distant_speech_recognition_amd import engine
    from tests.util import ula_positions, la_delays
    M, N = 128, 8
    dt, di = la_delays(ula_positions(N), 0.4), la_delays(ula_positions(N), 1.9)
    wq = engine.weights_mainlobe_2(M, N, 16000, dt, di)
    assert np.max(np.abs(wq - orc.calc_mainlobe_2(M, N, 16000, dt, di))) < 1e-14
    for k in (1, 9, 63):
        vt = np.exp(-2j * np.pi * k * dt * 16000 / M)
        vj = np.exp(-2j * np.pi * k * di * 16000 / M)
        assert abs(np.vdot(wq[k], vt) - 1.0) < 1e-12 and abs(np.vdot(wq[k], vj)) < 1e-12
    B = engine.weights_blocking_matrix(wq[9], 2)
    assert B.shape == (N, N - 2) and np.max(np.abs(wq[9] @ B)) < 1e-12


def test_lcmv_weights_n_constraints_host(orc):
    """calcMainlobeN with NC = 3, 4 (beamformer.cc:600-721): product (btk_pinv of the float32-rounded Gram matrix) vs the oracle
    (the reference's float32 csvdc pseudoinverse, compiled into oracle/_ref when /root/reference is present, numpy
    float32 SVD otherwise) + distortionless / null known answers; NC = 2 falls through to calcMainlobe2."""
    import numpy as np
    from distant_speech_recognition_amd import engine, _lib
    from tests.util import ula_positions, la_delays
    M, N = 64, 8
    mp = ula_positions(N, 40.0)
    dt = la_delays(mp, 0.3)
    nulls = np.stack([la_delays(mp, a) for a in (1.2, 2.0, 2.6)])
    for NC in (3, 4):
        wq = engine.weights_mainlobe_n(M, N, 16000, dt, nulls[: NC - 1], NC)
        ref = orc.calc_mainlobe_n(M, N, 16000, dt, nulls[: NC - 1], NC)
        K = M // 2 + 1
        # the reference inverts the Gram matrix with a float32 SVD: error ~ 1e-6 x cond(C^H C) per bin
        # (bins above M/2 keep calcMainlobe's mirror in both)
        for k in range(1, M // 2):
            Cm = np.stack([np.exp(-2j * np.pi * k * d * 16000 / M) for d in [dt] + list(nulls[: NC - 1])], axis=1)
            cond = np.linalg.cond(np.conj(Cm.T) @ Cm)
            assert np.max(np.abs(wq[k] - ref[k])) <= 2e-6 * cond * np.max(np.abs(ref[k])), (NC, k, cond)
        assert np.array_equal(wq[0], ref[0]) and np.array_equal(wq[K:], ref[K:])
        for k in (1, 9, 31):
            # known answers hold to the float32 precision the reference inverts the Gram matrix in (x its condition number)
            Cm = np.stack([np.exp(-2j * np.pi * k * d * 16000 / M) for d in [dt] + list(nulls[: NC - 1])], axis=1)
            tol = 1e-6 * np.linalg.cond(np.conj(Cm.T) @ Cm)
            vt = np.exp(-2j * np.pi * k * dt * 16000 / M)
            assert abs(np.vdot(wq[k], vt) - 1.0) < tol
            for n in range(NC - 1):
                assert abs(np.vdot(wq[k], np.exp(-2j * np.pi * k * nulls[n] * 16000 / M))) < tol
        B = engine.weights_blocking_matrix(wq[9], NC)
        assert B.shape == (N, N - NC) and np.max(np.abs(wq[9] @ B)) < 1e-12
    assert np.array_equal(engine.weights_mainlobe_n(M, N, 16000, dt, nulls[:1], 2), engine.weights_mainlobe_2(M, N, 16000, dt, nulls[0]))
    with pytest.raises(_lib.BtkError):
        engine.weights_mainlobe_n(M, N, 16000, dt, nulls[:1], 9)


def test_pinv_matches_reference_csvdc(orc):
    """btk_pinv (one-sided Jacobi SVD of the float32-rounded matrix) == pseudoinverse() on the reference's own compiled
    LINPACK csvdc (oracle/_ref), incl. the zeroed-singular-value rule and its return value (beamformer.cc:232-289)."""
    import numpy as np
    from distant_speech_recognition_amd import engine
    rng = np.random.default_rng(3)
    for N in (2, 3, 8, 64):
        X = rng.normal(size=(N, 3 * N)) + 1j * rng.normal(size=(N, 3 * N))
        A = X @ X.conj().T / (3 * N) + 0.01 * np.eye(N)
        a, ok = engine.pinv(A)
        b, okb = orc.pseudoinverse(A)
        assert ok and okb and np.max(np.abs(a - b)) <= 3e-6 * np.max(np.abs(b)), N
    H = rng.normal(size=(6, 6)) + 1j * rng.normal(size=(6, 6))
    H = (H + H.conj().T) / 2                                       # indefinite Hermitian
    a, ok = engine.pinv(H); b, okb = orc.pseudoinverse(H)
    assert ok and okb and np.max(np.abs(a - b)) <= 1e-5 * np.max(np.abs(b))
    Z = np.diag([1.0, 2.0, 0.0, 3.0]).astype(complex)              # exact zero singular value -> zeroed, "false"
    a, ok = engine.pinv(Z); b, okb = orc.pseudoinverse(Z)
    assert (not ok) and (not okb) and np.max(np.abs(a - b)) < 1e-7
    T = rng.normal(size=(5, 3)) + 1j * rng.normal(size=(5, 3))    # tall: invA A = I
    a, ok = engine.pinv(T); b, okb = orc.pseudoinverse(T)
    assert a.shape == (3, 5) and np.max(np.abs(a - b)) <= 3e-6 * np.max(np.abs(b)) and np.max(np.abs(a @ T - np.eye(3))) < 1e-6


def test_missing_extension_fails_loudly(monkeypatch):
    """no CPU fallback: without libbtkhip.so every entry of the product raises instead of computing somewhere else"""
    from distant_speech_recognition_amd import _lib, engine
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(ROOT, "does", "not", "exist", "libbtkhip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()
    import numpy as np
    with pytest.raises(ImportError):
        engine.weights_mainlobe(64, 4, 16000, np.zeros(4))
    # and the product never imports the oracle
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import distant_speech_recognition_amd.engine, distant_speech_recognition_amd.btk20, "
            "distant_speech_recognition_amd.pybeamformer, distant_speech_recognition_amd.sharding; "
            "assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'product imports the oracle'" % ROOT)
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).returncode == 0
    src = ""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "distant_speech_recognition_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc")):
                src += open(os.path.join(dirpath, f), errors="ignore").read()
    assert "import oracle" not in src and "from oracle" not in src and "liborc" not in src
