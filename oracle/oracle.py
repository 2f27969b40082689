"""ctypes front-end of the CPU ORACLE (oracle/btk_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package never imports this module.

All heavy loops live in the C restatement; this file only marshals numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(ref=True):
    """Compile liborc.so (and oracle/_ref when /root/reference exists)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liborc.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        dp, fp, vp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_void_p
        L.orc_ring_index.restype = C.c_uint
        L.orc_ring_index.argtypes = [C.c_uint] * 3
        L.orc_fb_delays.argtypes = [C.c_uint, C.c_uint, C.c_int, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.orc_analysis_run.restype = C.c_long
        L.orc_analysis_run.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, C.c_long, vp, vp, C.c_long]
        L.orc_synthesis_run.restype = C.c_long
        L.orc_synthesis_run.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, vp, C.c_long, vp, C.c_long]
        L.orc_calc_mainlobe.argtypes = [C.c_uint, C.c_uint, C.c_float, vp, vp]
        L.orc_calc_mainlobe_2.argtypes = [C.c_uint, C.c_uint, C.c_float, vp, vp, vp]
        L.orc_blocking_matrix.restype = C.c_int
        L.orc_blocking_matrix.argtypes = [vp, C.c_uint, C.c_uint, vp]
        L.orc_sidelobe_canceller.argtypes = [vp, vp, C.c_uint, C.c_uint, vp]
        L.orc_snapshot_update.argtypes = [vp, C.c_uint, C.c_uint, vp]
        L.orc_gsc_frame.argtypes = [vp, vp, vp, C.c_uint, C.c_uint, C.c_int, vp]
        L.orc_zelinski_frame.argtypes = [vp, vp, C.c_uint, C.c_uint, vp, vp, C.c_double, C.c_int, C.c_int, C.c_int, vp]
        L.orc_mccowan_frame.argtypes = [vp, vp, vp, C.c_uint, C.c_uint, vp, vp, C.c_double, C.c_int, C.c_int, C.c_double,
                                        C.c_int, vp]
        L.orc_lefkimmiatis_frame.argtypes = [vp, vp, vp, vp, C.c_uint, C.c_uint, vp, vp, C.c_double, C.c_int, C.c_int,
                                             C.c_double, C.c_uint, C.c_int, vp]
        L.orc_nlms_new.restype = vp
        L.orc_nlms_new.argtypes = [C.c_uint] * 3 + [C.c_double] * 7 + [C.c_int, C.c_int]
        L.orc_nlms_free.argtypes = [vp]
        L.orc_nlms_wa.restype = vp
        L.orc_nlms_wa.argtypes = [vp]
        L.orc_nlms_subband_energy.restype = vp
        L.orc_nlms_subband_energy.argtypes = [vp]
        L.orc_nlms_energy.restype = C.c_double
        L.orc_nlms_energy.argtypes = [vp]
        L.orc_nlms_frame.argtypes = [vp] * 6
        L.orc_rls_py_frame.argtypes = [C.c_uint] * 3 + [vp] * 9
        L.orc_rls_cc_frame.argtypes = [C.c_uint] * 3 + [C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int] + [vp] * 7
        L.orc_cov_accumulate_frame.argtypes = [vp, C.c_uint, C.c_uint, vp, vp]
        L.orc_frame_energy.restype = C.c_double
        L.orc_frame_energy.argtypes = [vp, C.c_uint]
        L.orc_diffuse_noise_model.argtypes = [vp, C.c_uint, C.c_uint, C.c_float, C.c_float, vp]
        L.orc_diagonal_loading.argtypes = [vp, C.c_uint, C.c_uint, C.c_float]
        L.orc_mvdr_weights_from_inverse.argtypes = [vp, vp, C.c_uint, C.c_uint, vp]
        L.orc_cholesky_solve.restype = C.c_int
        L.orc_cholesky_solve.argtypes = [vp, vp, C.c_uint, vp]
        L.orc_wpe_estimate.restype = C.c_int
        L.orc_wpe_estimate.argtypes = [vp, C.c_long, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_double,
                                       C.c_uint, C.c_uint, C.c_double, vp]
        L.orc_wpe_apply.argtypes = [vp, C.c_long, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]
        L.orc_pipeline_gsc.restype = C.c_long
        L.orc_pipeline_gsc.argtypes = [vp, vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, C.c_uint, C.c_long,
                                       vp, vp, vp, C.c_long, C.POINTER(C.c_long)]
        _LIB = L
    return _LIB


def ref_lib():
    """The REFERENCE's own LINPACK (csvdc) compiled into oracle/_ref, or None."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libbtkref_linpack.so")
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_csvdc.restype = C.c_int
        R.ref_csvdc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int]
        _REF = R
    return _REF


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


# --------------------------------------------------------------------------- filter banks
def fb_delays(m, r, synthesis, dct):
    pd, la = C.c_uint(0), C.c_uint(0)
    lib().orc_fb_delays(m, r, int(bool(synthesis)), dct, C.byref(pd), C.byref(la))
    return pd.value, la.value


def analysis_num_frames(nsamples, M, m, r, dct):
    """ceil(len/D) - laN + pd  (modulated.cc:419-469 bookkeeping); 0 when the source ends inside the look-ahead skip
    (is_end_ is set before the first frame, :423-438)."""
    D = M >> r
    pd, la = fb_delays(m, r, False, dct)
    nblk = -(-nsamples // D)
    return 0 if nblk < la else nblk - la + pd


def analysis(proto, M, m, r, dct, pcm, want_polyphase=False):
    """Run OverSampledDFTAnalysisBank over a whole utterance. pcm float32[len].
    Returns complex128 [T][M] (and float64 [T][M] polyphase sums if asked)."""
    proto = np.ascontiguousarray(proto, np.float64)
    pcm = np.ascontiguousarray(pcm, np.float32)
    maxf = analysis_num_frames(len(pcm), M, m, r, dct) + 8
    out = np.zeros((maxf, M), np.complex128)
    pp = np.zeros((maxf, M), np.float64) if want_polyphase else None
    n = lib().orc_analysis_run(_p(proto), M, m, r, dct, _p(pcm), len(pcm), _p(out),
                               _p(pp) if want_polyphase else None, maxf)
    return (out[:n], pp[:n]) if want_polyphase else out[:n]


def synthesis(proto, M, m, r, dct, frames, gain_factor=1):
    """Run OverSampledDFTSynthesisBank over frames complex128 [T][M]. Returns float32 [B*D]."""
    proto = np.ascontiguousarray(proto, np.float64)
    frames = _c128(frames)
    T = frames.shape[0]
    D = M >> r
    out = np.zeros((T + 1) * D, np.float32)
    nb = lib().orc_synthesis_run(_p(proto), M, m, r, dct, gain_factor, _p(frames), T, _p(out), T + 1)
    return out[:nb * D]


# --------------------------------------------------------------------------- weights
def calc_mainlobe(M, N, samplerate, delays):
    delays = np.ascontiguousarray(delays, np.float64)
    wq = np.zeros((M, N), np.complex128)
    lib().orc_calc_mainlobe(M, N, float(samplerate), _p(delays), _p(wq))
    return wq


def calc_mainlobe_halfband(M, N, samplerate, delays):
    """BeamformerWeights::calcMainlobe with halfBandShift_ == true (beamformer.cc:515-527): bins at (k + 0.5) fs / M, the
    conjugate partner of bin k is bin M - 1 - k."""
    delays = np.asarray(delays, np.float64)
    wq = np.zeros((M, N), np.complex128)
    fshift = np.float32(0.5)
    for k in range(M // 2):
        for c in range(N):
            val = -2.0 * np.pi * float(fshift + k) * float(np.float32(samplerate)) * delays[c] / M
            wq[k, c] = np.exp(1j * val) / N
            wq[M - 1 - k, c] = np.exp(-1j * val) / N
    return wq


def gsc_frames_halfband(X, wq, wl, normalize=False):
    """SubbandDS::next / SubbandGSC::next with halfBandShift_ == true (beamformer.cc:1113-1128, 1276-1285): every one of the
    M bins is computed from its own snapshot and weights, nothing is mirrored.  X [T][N][M] (full spectra), wq, wl [M][N]."""
    T, N, M = X.shape
    Y = np.zeros((T, M), np.complex128)
    for k in range(M):
        w = wq[k] - wl[k]                                            # calc_gsc_output (:1208-1243)
        if normalize:
            w = w / (np.linalg.norm(w) * N)
        Y[:, k] = X[:, :, k] @ np.conj(w)
    return Y


def calc_mainlobe_2(M, N, samplerate, delays_t, delays_i):
    """calcMainlobe2 / calcMainlobeN with NC = 2 (beamformer.cc:572-721): LCMV quiescent weights."""
    dt = np.ascontiguousarray(delays_t, np.float64)
    di = np.ascontiguousarray(delays_i, np.float64)
    wq = np.zeros((M, N), np.complex128)
    lib().orc_calc_mainlobe_2(M, N, float(samplerate), _p(dt), _p(di), _p(wq))
    return wq


def calc_mainlobe_n(M, N, samplerate, delays_t, delays_is, NC):
    """calcMainlobeN (beamformer.cc:600-721) for NC >= 2 with calc_null_beamformer_ (:299-363): the NC x NC Gram matrix
    is inverted by calc_inverse_22mat_ for NC = 2 and by pseudoinverse() (float32 csvdc, oracle/_ref) otherwise."""
    dt = np.ascontiguousarray(delays_t, np.float64)
    dis = np.ascontiguousarray(delays_is, np.float64).reshape(NC - 1, N)
    if NC == 2:
        return calc_mainlobe_2(M, N, samplerate, dt, dis[0])
    wq = calc_mainlobe(M, N, samplerate, dt)                       # :638
    M2 = M // 2
    g = np.zeros(NC, np.complex128)
    g[0] = 1.0

    def null_beamformer(wt, pWj):
        Cm = np.stack([wt] + list(pWj), axis=1)                      # [N][NC]
        G = np.conj(Cm.T) @ Cm
        inv, _ = pseudoinverse(G)                                    # :352-355 (return value ignored there)
        return Cm @ (inv @ g)

    wq[0] = 1.0 / N
    pWj = [np.zeros(N, np.complex128) for _ in range(NC - 1)]
    for k in range(1, M2):                                           # :670-689
        vec = wq[k] * N
        for n in range(NC - 1):
            pWj[n] = np.exp(1j * (-2.0 * np.pi * k * float(np.float32(samplerate)) * dis[n] / M))
        wq[k] = null_beamformer(vec, pWj)
    vec = wq[M2].copy()                                              # :692-703, reproduced literally
    for c in range(N):
        vec[c] = vec[c] * N
        for n in range(NC - 1):
            vec[c] = np.exp(1j * (-np.pi * float(np.float32(samplerate)) * dis[n][c])) / N
        vec = null_beamformer(vec, pWj)
    wq[M2] = vec
    return wq


def blocking_matrix(a, NC=1):
    a = _c128(a)
    N = a.shape[0]
    B = np.zeros((N, N - NC), np.complex128)
    rc = lib().orc_blocking_matrix(_p(a), N, NC, _p(B))
    if rc:
        raise ValueError("blocking matrix failed")
    return B


def sidelobe_canceller(B, wa):
    B, wa = _c128(B), _c128(wa)
    N, bs = B.shape
    wl = np.zeros(N, np.complex128)
    lib().orc_sidelobe_canceller(_p(B), _p(wa), N, N - bs, _p(wl))
    return wl


def gsc_weights(M, N, samplerate, delays, wa=None, NC=1):
    """calc_gsc_weights + optional set_active_weights for bins 0..M/2: returns wq [M][N], B [M][N][N-NC], wl [M][N]."""
    wq = calc_mainlobe(M, N, samplerate, delays)
    B = np.stack([blocking_matrix(wq[k], NC) for k in range(M)])
    wl = np.zeros((M, N), np.complex128)
    if wa is not None:
        for k in range(M // 2 + 1):
            wl[k] = sidelobe_canceller(B[k], wa[k])
    return wq, B, wl


# --------------------------------------------------------------------------- beamformers
def snapshot_update(samples):
    samples = _c128(samples)
    N, M = samples.shape
    snaps = np.zeros((M, N), np.complex128)
    lib().orc_snapshot_update(_p(samples), M, N, _p(snaps))
    return snaps


def gsc_frames(X, wq, wl=None, normalize=False):
    """X complex [T][N][M] analysis outputs -> Y [T][M]  (SubbandGSC::next / SubbandDS::next)."""
    X = _c128(X)
    T, N, M = X.shape
    wq = _c128(wq)
    wl = None if wl is None else _c128(wl)
    out = np.zeros((T, M), np.complex128)
    snaps = np.zeros((M, N), np.complex128)
    L = lib()
    for t in range(T):
        L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
        L.orc_gsc_frame(_p(snaps), _p(wq), None if wl is None else _p(wl), M, N, int(normalize), _p(out[t]))
    return out


def zelinski_frames(X, Y, d, alpha, type_=2, min_frames=0, return_csd=False):
    """ZelinskiPostFilter over a whole utterance. X [T][N][M], Y [T][M] beamformer output,
    d [M][N] (wq or ta_).  Returns filtered Y [T][M] and weights [T][M] (and, with return_csd, BeamformerWeights::CSDs()
    after the last frame: [M][N*N], entry i N + j for i <= j, postfilter.cc:77-116)."""
    X, Y, d = _c128(X), _c128(Y).copy(), _c128(d)
    T, N, M = X.shape
    csd = np.zeros((M, N * N), np.complex128)
    wp1 = np.zeros(M, np.complex128)
    W = np.zeros((T, M), np.complex128)
    snaps = np.zeros((M, N), np.complex128)
    L = lib()
    for t in range(T):
        L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
        L.orc_zelinski_frame(_p(d), _p(snaps), M, N, _p(csd), _p(wp1), float(alpha), int(type_),
                             int(min_frames), t - 1, _p(Y[t]))
        W[t] = wp1
    return (Y, W, csd) if return_csd else (Y, W)


def mccowan_frames(X, Y, d, R, alpha=0.6, type_=2, min_frames=0, threshold=0.99):
    """McCowanPostFilter over a whole utterance (postfilter.cc:843-935). X [T][N][M], Y [T][M] beamformer output,
    d [K..][N] (wq if type & 8 else the array manifold), R [K][N][N] noise coherence.  Returns filtered Y and weights."""
    X, Y, d, R = _c128(X), _c128(Y).copy(), _c128(d), _c128(R)
    T, N, M = X.shape
    csd = np.zeros((M, N * N), np.complex128)
    wp = np.zeros(M, np.complex128)
    W = np.zeros((T, M), np.complex128)
    snaps = np.zeros((M, N), np.complex128)
    L = lib()
    thr = float(np.float32(threshold))                     # float member threshold_of_Rij_ (postfilter.h:157)
    for t in range(T):
        L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
        L.orc_mccowan_frame(_p(d), _p(snaps), _p(R), M, N, _p(csd), _p(wp), float(alpha), int(type_), int(min_frames),
                            thr, t - 1, _p(Y[t]))
        W[t] = wp
    return Y, W


def lefkimmiatis_frames(X, Y, d, R, min_sv=1.0e-8, fbin_x1=0, alpha=0.6, type_=2, min_frames=0, threshold=0.99):
    """LefkimmiatisPostFilter over a whole utterance (postfilter.cc:967-1190)."""
    X, Y, d, R = _c128(X), _c128(Y).copy(), _c128(d), _c128(R)
    T, N, M = X.shape
    K = M // 2 + 1
    invR = np.zeros((K, N, N), np.complex128)
    for k in range(K):                                      # calc_inverse_noise_spatial_spectral_matrix :967-980
        inv, ok = pseudoinverse(R[k], min_sv)
        invR[k] = inv if ok else np.eye(N)
    csd = np.zeros((M, N * N), np.complex128)
    wp = np.zeros(M, np.complex128)
    W = np.zeros((T, M), np.complex128)
    snaps = np.zeros((M, N), np.complex128)
    L = lib()
    thr = float(np.float32(threshold))
    for t in range(T):
        L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
        L.orc_lefkimmiatis_frame(_p(d), _p(snaps), _p(R), _p(invR), M, N, _p(csd), _p(wp), float(alpha), int(type_),
                                 int(min_frames), thr, int(fbin_x1), t - 1, _p(Y[t]))
        W[t] = wp
    return Y, W


class NLMS:
    """SubbandGSCLMSBeamformer restatement (lib/pybeamformer.py:588-762)."""

    def __init__(self, M, N, Nc=1, beta=0.97, gamma=0.01, init_diagonal_load=1.0e6,
                 regularization_param=1.0e-4, energy_floor=90, sil_thresh=1.0e8,
                 max_wa_l2norm=100.0, min_frames=128, slowdown_after=4096):
        self.M, self.N, self.Nc = M, N, Nc
        self.K = M // 2 + 1
        self._h = lib().orc_nlms_new(M, N, Nc, beta, gamma, init_diagonal_load, regularization_param,
                                     energy_floor, sil_thresh, max_wa_l2norm, min_frames, slowdown_after)
        self.BmH = None
        self.wqH = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_nlms_free(self._h)
            self._h = None

    def calc_beamformer_weights(self, samplerate, delays):
        """pybeamformer.py:736-743 with calc_array_manifold_f (:284-306)."""
        delays = np.asarray(delays, np.float64)
        K, N = self.K, self.N
        self.BmH = np.zeros((K, N - self.Nc, N), np.complex128)
        self.wqH = np.zeros((K, N), np.complex128)
        delta_f = samplerate / float(self.M)
        for k in range(K):
            vs = np.exp(-1j * 2.0 * np.pi * k * delta_f * delays) / N
            self.BmH[k] = blocking_matrix(vs, self.Nc).T
            self.wqH[k] = np.conjugate(vs)

    def wa(self):
        ptr = lib().orc_nlms_wa(self._h)
        n = self.K * (self.N - self.Nc)
        buf = (C.c_double * (2 * n)).from_address(ptr)
        return np.frombuffer(buf, np.complex128).reshape(self.K, self.N - self.Nc).copy()

    def run(self, X):
        """X complex [T][N][M] -> Y [T][M]"""
        X = _c128(X)
        T, N, M = X.shape
        out = np.zeros((T, M), np.complex128)
        snaps = np.zeros((M, N), np.complex128)
        L = lib()
        for t in range(T):
            L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
            L.orc_nlms_frame(self._h, _p(X[t]), _p(snaps), _p(self.BmH), _p(self.wqH), _p(out[t]))
        return out


class RLSPy:
    """SubbandGSCRLSBeamformer restatement (lib/pybeamformer.py:765-928)."""

    def __init__(self, M, N, Nc=1, beta=0.97, gamma=0.04, mu=0.97, init_diagonal_load=1.0e6,
                 regularization_param=1.0e-2, sil_thresh=1.0e8, constraint_option=3, alpha2=10.0,
                 max_wa_l2norm=100.0, min_frames=128):
        self.M, self.N, self.Nc, self.K = M, N, Nc, M // 2 + 1
        self.par = np.array([beta, gamma, mu, init_diagonal_load, regularization_param, sil_thresh,
                             constraint_option, alpha2, max_wa_l2norm, min_frames], np.float64)
        bs = N - Nc
        self.scal = np.array([init_diagonal_load, 0.0, 0.0], np.float64)        # reset_stats :913-925
        self.Pz = np.tile(np.identity(bs, np.complex128) / init_diagonal_load, (self.K, 1, 1))
        self.waH = np.zeros((self.K, bs), np.complex128)
        self.BmH = None
        self.wqH = None

    calc_beamformer_weights = NLMS.calc_beamformer_weights                      # :900-908 is the same code

    def run(self, X):
        X = _c128(X)
        T, N, M = X.shape
        out = np.zeros((T, M), np.complex128)
        snaps = np.zeros((M, N), np.complex128)
        L = lib()
        for t in range(T):
            L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
            L.orc_rls_py_frame(M, N, self.Nc, _p(self.par), _p(self.scal), _p(X[t, 0]), _p(snaps),
                               _p(self.BmH), _p(self.wqH), _p(self.Pz), _p(self.waH), _p(out[t]))
        return out


class RLSCc:
    """SubbandGSCRLS restatement (beamformer/beamformer.cc:1447-1645)."""

    def __init__(self, M, N, delays, samplerate, mu=0.9, sigma2=0.0, Nc=1):
        self.M, self.N, self.Nc = M, N, Nc
        self.mu, self.diag_w = float(np.float32(mu)), float(np.float32(sigma2))  # float members, beamformer.h:253-254
        self.qctype, self.alpha, self.normalize, self.update = 0, -1.0, 0, 1
        self.wq = calc_mainlobe(M, N, samplerate, delays)                        # [M][N]
        bs = N - Nc
        self.B = np.zeros((M, N, bs), np.complex128)
        for k in range(M):
            self.B[k] = blocking_matrix(self.wq[k], Nc)
        self.wl = np.zeros((M, N), np.complex128)
        self.wa = np.zeros((M, bs), np.complex128)
        self.Pz = None

    def init_precision_matrix(self, sigma2=0.01):
        """beamformer.cc:1482-1494; 1/sigma2 is a float division there"""
        bs = self.N - self.Nc
        v = float(np.float32(1) / np.float32(sigma2))
        self.Pz = np.tile(np.identity(bs, np.complex128) * v, (self.M, 1, 1))

    def set_quadratic_constraint(self, alpha, qctype=1):
        self.alpha, self.qctype = float(np.float32(alpha)), int(qctype)

    def run(self, X):
        X = _c128(X)
        T, N, M = X.shape
        out = np.zeros((T, M), np.complex128)
        snaps = np.zeros((M, N), np.complex128)
        L = lib()
        for t in range(T):
            L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
            L.orc_rls_cc_frame(M, N, self.Nc, self.mu, self.diag_w, self.qctype, self.alpha, self.normalize,
                               self.update, _p(snaps), _p(self.B), _p(self.wq), _p(self.wl), _p(self.Pz),
                               _p(self.wa), _p(out[t]))
        return out


def frame_energy(X_ch0):
    X_ch0 = _c128(X_ch0)
    return lib().orc_frame_energy(_p(X_ch0), X_ch0.shape[0])


def cov_accumulate(X, frame_weights=None, masks=None):
    """R_k += w x x^H over frames.  X [T][N][M]; frame_weights bool/float [T] (label gating) or
    masks float [T][K] (TF-mask accumulation).  Returns R [K][N][N]."""
    X = _c128(X)
    T, N, M = X.shape
    K = M // 2 + 1
    R = np.zeros((K, N, N), np.complex128)
    snaps = np.zeros((M, N), np.complex128)
    L = lib()
    for t in range(T):
        if frame_weights is not None and not frame_weights[t]:
            continue
        L.orc_snapshot_update(_p(X[t]), M, N, _p(snaps))
        mk = None
        if masks is not None:
            mk = np.ascontiguousarray(masks[t], np.float64)
        L.orc_cov_accumulate_frame(_p(snaps), M, N, None if mk is None else _p(mk), _p(R))
    return R


# --------------------------------------------------------------------------- MVDR
def improve_matrix_condition(x, gamma):
    """lib/pybeamformer.py:1200-1207"""
    x = _c128(x)
    scale = gamma * np.trace(x) / x.shape[-1]
    return (x + np.eye(x.shape[-1]) * scale) / (1 + gamma)


def sos_finalize(cov_t, cov_j, cnt_t, cnt_j, gamma=1e-6, gev=False):
    """SubbandBlindMVDRBeamformer.finalize_stats (:1249-1263) / SubbandGEVBeamformer.finalize_stats (:1305-1328).
    cov_* [K][N][N] raw sums, cnt_* [K].  Returns (target, noise)."""
    ct, cj = _c128(cov_t).copy(), _c128(cov_j).copy()
    K, N = ct.shape[0], ct.shape[1]
    for m in range(K):
        if not gev:
            ct[m] /= cnt_t[m]
        cj[m] /= cnt_j[m]
        if gamma > 0:
            cj[m] = improve_matrix_condition(cj[m], gamma)
        if gev:
            cj[m] /= (np.trace(cj[m]) / N)
    return ct, cj


def blind_mvdr_weights(cov_t, cov_j, ref_micx=0, offset=0.0):
    """SubbandBlindMVDRBeamformer.calc_beamformer_weights (:1225-1247): wqH [K][N]"""
    ct, cj = _c128(cov_t), _c128(cov_j)
    K, N = ct.shape[0], ct.shape[1]
    u = np.zeros(N)
    u[ref_micx] = 1.0
    wqH = np.zeros((K, N), np.complex128)
    for m in range(K):
        no = np.dot(np.linalg.inv(cj[m]), ct[m])
        wqH[m] = np.conjugate(np.dot(no, u) / (offset + np.trace(no)))
    return wqH


def gev_weights(cov_t, cov_j):
    """SubbandGEVBeamformer.calc_beamformer_weights (:1280-1303): scipy.linalg.eigh(target, noise), principal
    eigenvector, phase aligned to the previous bin, conjugated.  wqH [K][N]"""
    import scipy.linalg
    ct, cj = _c128(cov_t), _c128(cov_j)
    K, N = ct.shape[0], ct.shape[1]
    wqH = np.zeros((K, N), np.complex128)
    for m in range(K):
        _, vecs = scipy.linalg.eigh(ct[m], cj[m])
        wqH[m] = vecs[:, -1]
        if m > 0:
            wqH[m] *= np.exp(-1j * np.angle(np.inner(wqH[m], np.conjugate(wqH[m - 1]))))
    return np.conjugate(wqH)


def sos_frames(X, wqH):
    """SubbandSOSBatchBeamformer.__iter__ (:1171-1186): X [T][N][M], wqH [K][N] -> [T][M]"""
    X, wqH = _c128(X), _c128(wqH)
    T, N, M = X.shape
    K = M // 2 + 1
    out = np.zeros((T, M), np.complex128)
    out[:, :K] = np.einsum("kn,tnk->tk", wqH, X[:, :, :K])
    out[:, K:] = np.conj(out[:, M // 2 - 1:0:-1])
    return out


def diffuse_noise_model(mpos, M, samplerate, sspeed=343740.0):
    mpos = np.ascontiguousarray(mpos, np.float64)
    N = mpos.shape[0]
    R = np.zeros((M // 2 + 1, N, N), np.complex128)
    lib().orc_diffuse_noise_model(_p(mpos), N, M, float(samplerate), float(sspeed), _p(R))
    return R


def diagonal_loading(R, M, w):
    R = _c128(R).copy()
    lib().orc_diagonal_loading(_p(R), M, R.shape[1], float(w))
    return R


def pseudoinverse(A, threshold=1.0e-8, return_info=False):
    """pseudoinverse(): beamformer.cc:232-289.  float32 LINPACK csvdc (job=11) through the compiled
    reference (oracle/_ref) when present, numpy float32 SVD otherwise.
    Returns (invA complex128, ok) -- with return_info also csvdc's INFO (non-zero: its QR iteration did not converge
    within 30 sweeps for INFO singular values; the reference then reports failure like for a thresholded value)."""
    A = _c128(A)
    Mr, Nc_ = A.shape
    R = ref_lib()
    ok = True
    info = 0
    if R is not None:
        a = np.asfortranarray(A.astype(np.complex64))
        s = np.zeros(Mr + Nc_, np.complex64)
        e = np.zeros(Mr + Nc_, np.complex64)
        u = np.zeros((Mr, Mr), np.complex64, order="F")
        v = np.zeros((Nc_, Nc_), np.complex64, order="F")
        info = R.ref_csvdc(_p(a), Mr, Mr, Nc_, _p(s), _p(e), _p(u), Mr, _p(v), Nc_, 11)
        if info != 0:
            ok = False
        sv = s[:Nc_].copy()
        U, V = u, v
    else:
        U, sr, Vh = np.linalg.svd(A.astype(np.complex64))
        sv = sr.astype(np.complex64)
        V = Vh.conj().T
    sinv = np.zeros(Nc_, np.complex64)
    for k in range(Nc_):
        if abs(sv[k]) < threshold:
            ok = False
        else:
            sinv[k] = np.complex64(1.0) / sv[k]
    inv = np.zeros((Nc_, Mr), np.complex64)
    for i in range(Mr):
        for j in range(Nc_):
            inv[j, i] = np.sum(V[j, :Nc_] * sinv * np.conj(U[i, :Nc_]))
    if return_info:
        return inv.astype(np.complex128), ok, int(info)
    return inv.astype(np.complex128), ok


def mvdr_weights(R, wq, M, threshold=1.0e-8):
    """calc_mvdr_weights: beamformer.cc:2350-2402.  R [K][N][N], wq [M][N] -> w [K][N]"""
    R, wq = _c128(R), _c128(wq)
    K, N = R.shape[0], R.shape[1]
    invR = np.zeros((K, N, N), np.complex128)
    for k in range(1, K):
        inv, ok = pseudoinverse(R[k], threshold)
        invR[k] = inv if ok else np.eye(N)
    w = np.zeros((K, N), np.complex128)
    lib().orc_mvdr_weights_from_inverse(_p(invR), _p(wq), M, N, _p(w))
    return w


def mvdr_frames(X, w):
    """SubbandMVDR::next: y_k = w_k^H x_k, mirror (beamformer.cc:2537-2587). X [T][N][M], w [K][N]"""
    X, w = _c128(X), _c128(w)
    T, N, M = X.shape
    wfull = np.zeros((M, N), np.complex128)
    wfull[: M // 2 + 1] = w
    return gsc_frames(X, wfull, None)


# --------------------------------------------------------------------------- WPE
def wpe_band(M, band_width, samplerate):
    lower = M // 2 if band_width == 0.0 else int((band_width / (samplerate / 2.0)) * (M // 2))
    return lower, M - lower


def wpe_estimate(Y, lower_num, upper_num, iterations, load_db, band_width=0.0, diagonal_bias=1e-4,
                 samplerate=16000.0):
    """Y complex [T][C][M] -> G [C][M][C*L]"""
    Y = _c128(Y)
    T, Cn, M = Y.shape
    L = upper_num - lower_num + 1
    G = np.zeros((Cn, M, Cn * L), np.complex128)
    lo, up = wpe_band(M, band_width, samplerate)
    rc = lib().orc_wpe_estimate(_p(Y), T, Cn, M, lower_num, upper_num, iterations, float(load_db), lo, up,
                                float(diagonal_bias), _p(G))
    if rc:
        raise ArithmeticError("Cholesky failed")
    return G


def wpe_apply(Y, G, lower_num, upper_num, band_width=0.0, samplerate=16000.0):
    Y, G = _c128(Y), _c128(G)
    T, Cn, M = Y.shape
    out = np.zeros_like(Y)
    lo, up = wpe_band(M, band_width, samplerate)
    lib().orc_wpe_apply(_p(Y), T, Cn, M, lower_num, upper_num, lo, up, _p(G), _p(out))
    return out


# --------------------------------------------------------------------------- whole graph (cpu baseline)
def pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl):
    """pcm float32 [N][len] -> (float32 output, frames_beamformed)."""
    h = np.ascontiguousarray(h, np.float64)
    g = np.ascontiguousarray(g, np.float64)
    pcm = np.ascontiguousarray(pcm, np.float32)
    N, ln = pcm.shape
    D = M >> r
    maxb = analysis_num_frames(ln, M, m, r, dct) + 8
    out = np.zeros(maxb * D, np.float32)
    nbf = C.c_long(0)
    nb = lib().orc_pipeline_gsc(_p(h), _p(g), M, m, r, dct, _p(pcm), N, ln, _p(_c128(wq)), _p(_c128(wl)),
                                _p(out), maxb, C.byref(nbf))
    return out[: nb * D], nbf.value
