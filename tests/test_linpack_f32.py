"""CPU checks of the float32 LINPACK restatement behind svd_rule = "linpack" (csrc/linpack_f32.h):
the kernel body, compiled by g++ as serial code, against the reference's own compiled csvdc (oracle/_ref) and against
tests/golden/c5_csvdc_info.npz -- bit for bit (s, e and INFO).  The GPU build of the same body is checked in
tests/test_gpu_linpack_rule.py."""
import os
import zlib

import numpy as np
import pytest

from tests import linpack_host as lh
from tests.util import ula_positions

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c5_csvdc_info.npz")


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def test_host_build_matches_reference_csvdc_bit_for_bit(orc):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    for A in lh.test_matrices():
        s, e, info = lh.csvdc_values(A)
        for job in (11, 0):                                   # the vectors never feed back into s, e, INFO
            sr, er, ir = lh.ref_csvdc(orc, A, job)
            assert info == ir, (A.shape, info, ir)
            assert np.array_equal(_bits(s), _bits(sr)) and np.array_equal(_bits(e), _bits(er)), A.shape


def test_c5_fixture_every_16th_bin(orc):
    """The C5 model (256 microphones, 2048 sub-bands, loading 1e-2): INFO and the singular values of every 16th bin at both
    array pitches equal what the reference's compiled csvdc produced (fixture), incl. the bins where it does not converge."""
    z = np.load(GOLDEN)
    N, M = 256, 2048
    for g, pitch in enumerate(z["pitch_mm"]):
        Rs = orc.diffuse_noise_model(ula_positions(N, float(pitch)), M, 16000)[::16]
        R = orc.diagonal_loading(Rs, 2 * (len(Rs) - 1), 0.01)            # (the oracle loads M / 2 + 1 matrices)
        nbad = 0
        for j in range(1, R.shape[0]):
            k = 16 * j
            s, e, info = lh.csvdc_values(R[j])
            assert info == int(z["info"][g, k]), (pitch, k, info, int(z["info"][g, k]))
            assert zlib.crc32(s[:N].tobytes()) == int(z["s_crc"][g, k])
            assert np.array_equal(_bits(s[:N]), _bits(z["s_sub"][g, j]))
            nbad += info != 0
        assert nbad > 10                                        # the branch is exercised: not converging is the rule, not the exception


def test_fixture_statistics():
    """What the fixture says about the reference on C5: INFO != 0 on 505 of 1024 bins at 20 mm, 752 at 10 mm."""
    z = np.load(GOLDEN)
    assert z["info"].shape == (2, 1025) and np.all(z["info"][:, 0] == -1)
    assert [int(np.sum(z["info"][g, 1:] != 0)) for g in range(2)] == [505, 752]


def test_oracle_pseudoinverse_reports_the_fixture_info(orc):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    z = np.load(GOLDEN)
    N, M = 256, 2048
    Rs = orc.diffuse_noise_model(ula_positions(N, 20.0), M, 16000)[::128]
    R = orc.diagonal_loading(Rs, 2 * (len(Rs) - 1), 0.01)
    for j in (1, 2, 5):
        inv, ok, info = orc.pseudoinverse(R[j], return_info=True)
        assert info == int(z["info"][0, 128 * j]) and ok == (info == 0)
