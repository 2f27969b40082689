"""Golden data for the reference's `csvdc INFO != 0 -> identity` branch (beamformer/beamformer.cc:253-260, :2379-2384) on
BASELINE config C5: the 256-microphone diffuse-noise MVDR model, 2048 sub-bands, diagonal loading 1e-2 (confs/sd.json).

For every bin 1..1024 the float32-rounded matrix R_k goes through the REFERENCE's OWN compiled LINPACK csvdc (oracle/_ref,
job = 11 exactly as pseudoinverse() calls it) in the dev container.  Stored -> tests/golden/c5_csvdc_info.npz:
  pitch_mm [2]            the two linear-array geometries swept: 20 mm and 10 mm (tests/test_gpu_configs.py)
  info     int32 [2][1025]  csvdc's INFO per bin (bin 0 is never solved by calc_mvdr_weights: -1)
  s_crc    uint32 [2][1025] zlib.crc32 of the 256 float32 singular values csvdc left in s (converged or not)
  s_sub    float32 [2][65][256] those values themselves for every 16th bin
The model matrix is the oracle's float64 restatement of set_diffuse_noise_model / set_all_diagonal_loading
(beamformer.cc:2442-2523); only numbers are stored.

Run:  python tests/golden/gen_c5_csvdc_info.py   (about 4 minutes on one core)"""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    from oracle import oracle as orc
    from tests.util import ula_positions
    orc.build(ref=True)
    ref = orc.ref_lib()
    assert ref is not None, "oracle/_ref (the reference's compiled LINPACK) is needed"
    N, M = 256, 2048
    K = M // 2 + 1
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    pitches = np.array([20.0, 10.0])
    info = np.full((2, K), -1, np.int32)
    crc = np.zeros((2, K), np.uint32)
    s_sub = np.zeros((2, (K + 15) // 16, N), np.float32)
    for g, pitch in enumerate(pitches):
        R = orc.diagonal_loading(orc.diffuse_noise_model(ula_positions(N, pitch), M, 16000), M, 0.01)
        for k in range(1, K):
            a = np.asfortranarray(R[k].astype(np.complex64))
            s = np.zeros(2 * N, np.complex64)
            e = np.zeros(2 * N, np.complex64)
            u = np.zeros((N, N), np.complex64, order="F")
            v = np.zeros((N, N), np.complex64, order="F")
            info[g, k] = ref.ref_csvdc(P(a), N, N, N, P(s), P(e), P(u), N, P(v), N, 11)
            sv = np.ascontiguousarray(s[:N].real.astype(np.float32))
            assert not np.any(s[:N].imag)
            crc[g, k] = zlib.crc32(sv.tobytes())
            if k % 16 == 0:
                s_sub[g, k // 16] = sv
        print("pitch %g mm: INFO != 0 on %d of %d bins" % (pitch, int(np.sum(info[g, 1:] != 0)), K - 1), flush=True)
    np.savez_compressed(os.path.join(HERE, "c5_csvdc_info.npz"), pitch_mm=pitches, info=info, s_crc=crc, s_sub=s_sub)


if __name__ == "__main__":
    main()
