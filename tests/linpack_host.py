"""Test helper: the kernel body of csrc/svd_linpack.hip (csrc/linpack_f32.h) built by g++ as serial host code
(tests/cpp/linpack_host.cc), for bit-for-bit comparisons without a GPU and as the expected value of the GPU tests."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        out = os.path.join(tempfile.mkdtemp(prefix="lpk_"), "liblpkhost.so")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                               "-I" + os.path.join(ROOT, "distant_speech_recognition_amd", "csrc"),
                               os.path.join(ROOT, "tests", "cpp", "linpack_host.cc"), "-o", out])
        _lib = C.CDLL(out)
        _lib.lpk_host_csvdc_values.restype = C.c_int
        _lib.lpk_host_csvdc_values.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def csvdc_values(A):
    """A complex [n][p] -> (s float32 [m], e float32 [m], info) with m = min(n + 1, p)."""
    A = np.ascontiguousarray(A, np.complex64)
    n, p = A.shape
    m = min(n + 1, p)
    s = np.zeros(m, np.float32)
    e = np.zeros(m, np.float32)
    info = lib().lpk_host_csvdc_values(A.ctypes.data_as(C.c_void_p), n, p, s.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p))
    return s, e, int(info)


def ref_csvdc(orc, A, job=11):
    """The reference's compiled csvdc (oracle/_ref) on A complex [n][p]: (s float32 [m], e float32 [m], info)."""
    ref = orc.ref_lib()
    A = np.asarray(A).astype(np.complex64)
    n, p = A.shape
    a = np.asfortranarray(A).copy(order="F")
    s = np.zeros(n + p, np.complex64)
    e = np.zeros(n + p, np.complex64)
    u = np.zeros((n, n), np.complex64, order="F")
    v = np.zeros((p, p), np.complex64, order="F")
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    info = ref.ref_csvdc(P(a), n, n, p, P(s), P(e), P(u), n, P(v), p, job)
    m = min(n + 1, p)
    return np.ascontiguousarray(s[:m].real), np.ascontiguousarray(e[:m].real), int(info)


def test_matrices(seed=0):
    """A fixed family of small matrices: full rank, rank deficient, zero columns, tall, wide, 1 x 1, Hermitian + loading."""
    rng = np.random.default_rng(seed)
    out = []
    for (n, p) in [(1, 1), (2, 2), (3, 3), (4, 4), (5, 3), (3, 5), (8, 8), (16, 16), (7, 7), (33, 33), (64, 64), (40, 20), (20, 40), (100, 100),
                   (137, 137), (150, 150)]:
        for trial in range(4):
            A = rng.normal(size=(n, p)) + 1j * rng.normal(size=(n, p))
            if trial == 1 and p <= n:
                B = rng.normal(size=(n, 2)) + 1j * rng.normal(size=(n, 2))
                A = (B @ B.conj().T)[:, :p]
            if trial == 2:
                A[:, 0] = 0
            if trial == 3 and n == p:
                B = rng.normal(size=(n, 3)) + 1j * rng.normal(size=(n, 3))
                A = B @ B.conj().T + 0.01 * np.eye(n)
            out.append(A.astype(np.complex64))
    return out
