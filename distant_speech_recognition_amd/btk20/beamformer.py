"""SnapShotArrayPtr, SubbandDSPtr, SubbandGSCPtr, SubbandMVDRPtr, SubbandMVDRGSCPtr
(beamformer/beamformer.h:28-437, beamformer/beamformer.i): the reference's node API, computed on
the MI355X.  A beamformer node gathers the PCM behind its analysis-bank channels, runs ONE batched
analysis + beamform pass on the device and serves frames; weights are designed host-side in float64
(one-off per look direction) exactly as the reference does."""
import numpy as np

from .. import _lib, engine
from .common import j_error, jallocation_error, jdimension_error, raise_from_code
from .modulated import OverSampledDFTAnalysisBankPtr, _mirror, _pull_all
from .stream import VectorComplexFeatureStream, _BlockServedStream, device

__all__ = ["SSPEED", "SnapShotArrayPtr", "SpectralMatrixArrayPtr", "SubbandDSPtr", "SubbandGSCPtr", "SubbandGSCRLSPtr", "SubbandMVDRPtr",
           "SubbandMVDRGSCPtr", "SubbandDS", "SubbandGSC", "SubbandGSCRLS", "SubbandMVDR", "SubbandMVDRGSC",
           "calc_all_delays"]

SSPEED = 343740.0     # beamformer/beamformer.h:26


class SnapShotArrayPtr(object):
    """SnapShotArray (beamformer/spectralinfoarray.h:6-36, beamformer.cc:18-93)."""

    def __init__(self, fftlen, chan_num):
        self._fftlen, self._chan_num = int(fftlen), int(chan_num)
        self._samples = np.zeros((self._chan_num, self._fftlen), np.complex128)
        self._snapshots = np.zeros((self._fftlen, self._chan_num), np.complex128)

    def fftlen(self):
        return self._fftlen

    def chan_num(self):
        return self._chan_num

    fftLen, nChan = fftlen, chan_num

    def set_samples(self, samp, chan_no):
        self._samples[chan_no] = samp

    def update(self):
        self._snapshots = self._samples.T.copy()

    def snapshot(self, fbin_no):
        return self._snapshots[fbin_no]

    def zero(self):
        self._samples[:] = 0
        self._snapshots[:] = 0


class SpectralMatrixArrayPtr(SnapShotArrayPtr):
    """SpectralMatrixArray (beamformer/spectralinfoarray.h:43-64, beamformer.cc:97-143): a SnapShotArray that also keeps, per
    bin, R_k <- mu R_k + (1 - mu) x_k x_k^T -- the outer product WITHOUT conjugation, as the reference writes it (:131-139).
    Host container like SnapShotArray (the Hermitian covariance the beamformers use is btk_cov_accumulate on the device).
    FBSpectralMatrixArray (:147-173) indexes the per-channel sample vectors with a bin number and is not mirrored."""

    def __init__(self, fftLn, nChn, forgetFact=0.95):
        SnapShotArrayPtr.__init__(self, fftLn, nChn)
        self._mu = float(np.float32(forgetFact))
        self._matrices = np.zeros((self._fftlen, self._chan_num, self._chan_num), np.complex128)

    def matrix_f(self, idx):
        return self._matrices[idx]

    getSpecMatrix = matrix_f

    def update(self):
        SnapShotArrayPtr.update(self)
        x = self._snapshots
        self._matrices = self._mu * self._matrices + (1.0 - self._mu) * (x[:, :, None] * x[:, None, :])

    def zero(self):
        SnapShotArrayPtr.zero(self)
        self._matrices[:] = 0


class _SubbandBeamformer(_BlockServedStream, VectorComplexFeatureStream):
    def __init__(self, fftlen, half_band_shift=False, nm="SubbandBeamformer"):
        _BlockServedStream.__init__(self, fftlen, nm)
        # half_band_shift == true exists for SubbandDS and SubbandGSC with one constraint (beamformer.cc:1113-1128, 1276-1285);
        # SubbandMVDR(GSC) refuse it in their constructors (:2283-2285) and SubbandGSCRLS in next() (:1528-1530)
        self._half_band_shift = bool(half_band_shift)
        self._Yfull = None        # half_band_shift: device output of all M bins [1][M][T]
        self._Xfull = None        # half_band_shift over pulled (non-analysis-bank) sources: all M snapshot bins [1][M][N][T]
        self._fftlen = int(fftlen)
        self._K = self._fftlen // 2 + 1
        self._channels = []
        self._X = None            # device snapshots [1][K][N][T]
        self._Xhost = None
        self._Y = None            # device output [1][K][T]
        self._snapshot_array = None
        self._wversion = 0        # bumped whenever the weights (hence the future output) change

    # ---- wiring (beamformer.h:89-125)
    def set_channel(self, chan):
        self._channels.append(chan)

    def clear_channel(self):
        self._channels = []
        self._snapshot_array = None
        self._X = self._Y = self._Yfull = self._Xhost = self._Xfull = None

    def chan_num(self):
        return len(self._channels)

    def fftlen(self):
        return self._fftlen

    def dim(self):
        return self._fftlen

    def is_end(self):
        return self._is_end

    setChannel, clearChannel, chanN, fftLen = set_channel, clear_channel, chan_num, fftlen

    def snapshot_array(self):
        """Live host mirror of the current frame's snapshots (read by post-filters, postfilter.cc:439-442)."""
        if self._snapshot_array is None:
            self._snapshot_array = SnapShotArrayPtr(self._fftlen, self.chan_num())
        if self._X is not None and self._frame_no >= 0:
            if self._Xhost is None:
                self._Xhost = self._X[0].cpu().numpy()
            xk = self._Xhost[:, :, self._frame_no]                      # [K][N]
            full = np.zeros((self._fftlen, self.chan_num()), np.complex128)
            full[: self._K] = xk
            full[self._K:] = np.conj(xk[self._fftlen // 2 - 1:0:-1])
            self._snapshot_array._snapshots = full
            self._snapshot_array._samples = full.T.copy()
        return self._snapshot_array

    def snapshot_array_f(self, fbin_no):
        return self.snapshot_array().snapshot(fbin_no)

    # ---- device block
    def device_snapshots(self):
        """X complex64 [1][K][N][T] on the device: one batched analysis over all channels."""
        if self._X is None:
            import torch
            chans = self._channels
            if not chans:
                raise j_error("set channels first\n")
            if all(isinstance(c, OverSampledDFTAnalysisBankPtr) for c in chans) and \
                    len(set(c.plan_key()[:4] for c in chans)) == 1:
                pcms = [c.pcm() for c in chans]
                L = min(len(p) for p in pcms)            # is_end_ as soon as any channel ends (beamformer.cc:1269)
                pcm = np.stack([p[:L] for p in pcms])[None]
                plan = chans[0]._plan
                self._X = plan.analysis(torch.from_numpy(np.ascontiguousarray(pcm)).to(device()))
            elif all(hasattr(c, "wpe_source") for c in chans) and len(set(id(c.wpe_source()) for c in chans)) == 1 \
                    and [c.channel_no() for c in chans] == list(range(chans[0].wpe_source()._C)):
                # config C4: the channels are the WPE outputs of one estimator -> stay on the device
                self._X = chans[0].wpe_source().device_output()
            else:
                frames = [_pull_all(c) for c in chans]
                T = min(len(f) for f in frames)
                Xa = np.stack([np.stack(f[:T]) for f in frames])                         # [N][T][M]
                if self._half_band_shift:
                    # halfBandShift: the reference dots every one of the M snapshots as supplied (beamformer.cc:1113-1128); a
                    # generic source owes no conjugate symmetry between its bins, so all M bins go to the device
                    self._Xfull = torch.from_numpy(np.ascontiguousarray(np.transpose(Xa, (2, 0, 1))[None]).astype(np.complex64)).to(device())
                    self._X = self._Xfull[:, : self._K].contiguous()
                else:
                    Xh = Xa[:, :, : self._K]                                             # [N][T][K]
                    self._X = torch.from_numpy(np.ascontiguousarray(np.transpose(Xh, (2, 0, 1))[None]).astype(np.complex64)).to(device())
            self._Xhost = None
        return self._X

    def device_block(self):
        """Y complex64 [1][K][T] on the device (used by downstream GPU nodes without a host round trip)."""
        if self._Y is None:
            self._compute_block()
        return self._Y

    def effective_weights(self):
        raise NotImplementedError

    def _compute_block(self):
        import torch
        X = self.device_snapshots()
        try:
            if self._half_band_shift:
                Wf = torch.from_numpy(self.effective_weights_all_bins().astype(np.complex64)).to(device())
                # analysis-bank channels: bins above M/2 are the conjugate mirrors; pulled sources: every bin as supplied
                self._Yfull = engine.bf_apply_all_bins(Wf, X) if self._Xfull is None else engine.bf_apply(Wf, self._Xfull)
                self._Y = self._Yfull[:, : self._K]          # what a downstream synthesis bank reads (bins 0..M/2)
            else:
                W = torch.from_numpy(self.effective_weights()).to(device())
                self._Y = engine.bf_apply(W, X)
        except _lib.BtkError as e:
            raise_from_code(e)

    def _prepare(self):
        Y = self.device_block()
        if self._half_band_shift:
            self._frames = self._Yfull[0].cpu().numpy().T.astype(np.complex128)      # every bin has its own output
        else:
            self._frames = _mirror(Y[0].cpu().numpy(), self._fftlen)

    def _output_version(self):
        return self._wversion

    def _invalidate_output(self):
        """Weights changed: frames not yet served are recomputed with the new weights (host mirror and device block)."""
        self._wversion += 1
        if self._Y is not None:
            done = self._frame_no + 1
            old, oldY = self._frames, self._Y
            self._Y = self._Yfull = None
            self._frames = None
            if done > 0:
                self._compute_block()
                if self._Y is not None and oldY.shape == self._Y.shape:
                    self._Y[..., :done] = oldY[..., :done]
            if old is not None:
                self._prepare()
                self._frames[:done] = old[:done]

    def reset(self):
        for c in self._channels:
            c.reset()
        if self._snapshot_array is not None:
            self._snapshot_array.zero()
        self._X = self._Y = self._Yfull = self._Xhost = self._Xfull = None
        _BlockServedStream.reset(self)


class _BeamformerWeights(object):
    """BeamformerWeights (beamformer.h:28-82, beamformer.cc:485-965): wq, B, wa, wl, ta host-side in float64."""

    def __init__(self, fftlen, chan_num, NC=1, half_band_shift=False):
        self.fftlen, self.chan_num, self.NC = fftlen, chan_num, NC
        self.half_band_shift = bool(half_band_shift)
        self.wq = np.zeros((fftlen, chan_num), np.complex128)
        self.wl = np.zeros((fftlen, chan_num), np.complex128)
        self.ta = np.zeros((fftlen, chan_num), np.complex128)
        self.B = None if chan_num == 1 or chan_num == NC else np.zeros((fftlen, chan_num, chan_num - NC), np.complex128)
        self.wa = None if self.B is None else np.zeros((fftlen, chan_num - NC), np.complex128)

    def calc_mainlobe(self, samplerate, delays, is_gsc):
        delays = np.asarray(delays, np.float64)
        if delays.size != self.chan_num:
            raise jdimension_error("Number of delays does not match number of channels (%d vs. %d).\n" % (delays.size, self.chan_num))
        if is_gsc and self.chan_num <= 1:
            raise jdimension_error("The number of channels must be > 1 but it is %d\n" % self.chan_num)
        self.wq = engine.weights_mainlobe(self.fftlen, self.chan_num, samplerate, delays, self.half_band_shift)
        self.ta = self.wq.copy()                                           # setTimeAlignment
        if is_gsc:
            for k in range(self.fftlen):                                   # all M bins (beamformer.cc:557-563)
                self.B[k] = engine.weights_blocking_matrix(self.wq[k], 1)

    def calc_mainlobe_2(self, samplerate, delays_t, delays_i, is_gsc):
        """calcMainlobe2 / calcMainlobeN with NC = 2 (beamformer.cc:572-721)."""
        if self.half_band_shift:
            raise j_error("halfBandShift==true with more than one constraint is not supported by this engine\n")
        delays_t, delays_i = np.asarray(delays_t, np.float64), np.asarray(delays_i, np.float64)
        if delays_i.size != self.chan_num:
            raise jdimension_error("The number of delays for an interference signal does not match number of channels (%d vs. %d).\n"
                                   % (delays_i.size, self.chan_num))
        if delays_t.size != self.chan_num:
            raise jdimension_error("The number of delays does not match number of channels (%d vs. %d).\n" % (delays_t.size, self.chan_num))
        try:
            self.wq = engine.weights_mainlobe_2(self.fftlen, self.chan_num, samplerate, delays_t, delays_i)
        except _lib.BtkError as e:
            raise_from_code(e)
        # calcMainlobeN calls calcMainlobe(.., false) first, which sets ta_ to the plain D&S weights (:638, :555)
        self.ta = engine.weights_mainlobe(self.fftlen, self.chan_num, samplerate, delays_t)
        if is_gsc:
            for k in range(self.fftlen):
                self.B[k] = engine.weights_blocking_matrix(self.wq[k], self.NC)

    def calc_mainlobe_n(self, samplerate, delays_t, delays_is, NC, is_gsc):
        """calcMainlobeN (beamformer.cc:600-721), NC >= 2."""
        if NC < 2 or NC > self.chan_num:
            raise jdimension_error("1 < the number of constraints %d <= the number of sensors %d.\n" % (NC, self.chan_num))
        if self.half_band_shift:
            raise j_error("halfBandShift==true with more than one constraint is not supported by this engine\n")
        delays_t = np.asarray(delays_t, np.float64)
        delays_is = np.asarray(delays_is, np.float64).reshape(-1, self.chan_num)
        if delays_t.size != self.chan_num:
            raise jdimension_error("The number of delays does not match number of channels (%d vs. %d).\n" % (delays_t.size, self.chan_num))
        try:
            self.wq = engine.weights_mainlobe_n(self.fftlen, self.chan_num, samplerate, delays_t, delays_is, NC)
        except _lib.BtkError as e:
            raise_from_code(e)
        self.ta = engine.weights_mainlobe(self.fftlen, self.chan_num, samplerate, delays_t)
        if is_gsc:
            for k in range(self.fftlen):
                self.B[k] = engine.weights_blocking_matrix(self.wq[k], self.NC)

    def calc_sidelobe_canceller_f(self, fbin, wa):
        self.wa[fbin] = wa
        self.wl[fbin] = engine.weights_sidelobe(self.B[fbin], wa)


class SubbandDSPtr(_SubbandBeamformer):
    """SubbandDS (beamformer.h:130-165, beamformer.cc:1023-1170)."""

    def __init__(self, fftlen, half_band_shift=False, nm="SubbandDS"):
        _SubbandBeamformer.__init__(self, fftlen, half_band_shift, nm)
        self._bfw = []

    def clear_channel(self):
        _SubbandBeamformer.clear_channel(self)
        self._bfw = []

    def _alloc_bfweight(self, NC):
        # re-creates the BeamformerWeights object -> resets active weights and post-filter state (beamformer.cc:1082-1092)
        self._bfw = [_BeamformerWeights(self._fftlen, self.chan_num(), NC, self._half_band_shift)]
        self._weights_version = getattr(self, "_weights_version", 0) + 1

    def calc_array_manifold_vectors(self, samplerate, delays):
        self._alloc_bfweight(1)
        self._bfw[0].calc_mainlobe(samplerate, delays, False)
        self._invalidate_output()

    def calc_array_manifold_vectors_2(self, samplerate, delays_t, delays_j):
        self._alloc_bfweight(2)
        self._bfw[0].calc_mainlobe_2(samplerate, delays_t, delays_j, False)
        self._invalidate_output()

    def calc_array_manifold_vectors_n(self, samplerate, delays_t, delays_js, NC=2):
        self._alloc_bfweight(NC)
        self._bfw[0].calc_mainlobe_n(samplerate, delays_t, delays_js, NC, False)
        self._invalidate_output()

    def get_weights(self, fbin_no):
        return self._bfw[0].wq[fbin_no]

    def beamformer_weight_object(self, srcX=0):
        return self._bfw[srcX]

    calcArrayManifoldVectors, getWeights = calc_array_manifold_vectors, get_weights

    def _check_weights(self):
        if not self._bfw:
            raise j_error("call calc_array_manifold_vectorsX() once\n")

    def effective_weights(self):
        self._check_weights()
        return engine.weights_gsc_effective(self._bfw[0].wq, None, self._fftlen)

    def effective_weights_all_bins(self):
        """half_band_shift: the weight vector of every one of the M bins, complex128 [M][N] (beamformer.cc:1113-1118)"""
        self._check_weights()
        return self._bfw[0].wq

    def alignment_vector(self, use_wq):
        self._check_weights()
        src = self._bfw[0].wq if use_wq else self._bfw[0].ta
        return src[: self._K].astype(np.complex64)

    def next(self, frame_no=-5):
        if not (frame_no == self._frame_no and self._vector is not None):
            self._check_weights()
        return _SubbandBeamformer.next(self, frame_no)


class SubbandGSCPtr(SubbandDSPtr):
    """SubbandGSC (beamformer.h:169-204, beamformer.cc:1245-1445)."""

    def __init__(self, fftlen, half_band_shift=False, nm="SubbandGSC"):
        SubbandDSPtr.__init__(self, fftlen, half_band_shift, nm)
        self._normalize_weight = False

    def normalize_weight(self, flag):
        self._normalize_weight = bool(flag)
        self._invalidate_output()

    def _check_weights(self):
        if not self._bfw:
            raise j_error("call calc_gsc_weights_X() once\n")

    def calc_gsc_weights(self, samplerate, delays_t):
        self._alloc_bfweight(1)
        self._bfw[0].calc_mainlobe(samplerate, delays_t, True)
        self._invalidate_output()

    def calc_gsc_weights_2(self, samplerate, delays_t, delays_i):
        self._alloc_bfweight(2)
        self._bfw[0].calc_mainlobe_2(samplerate, delays_t, delays_i, True)
        self._invalidate_output()

    def calc_gsc_weights_n(self, samplerate, delays_t, delays_is, NC=2):
        self._alloc_bfweight(NC)
        self._bfw[0].calc_mainlobe_n(samplerate, delays_t, delays_is, NC, True)
        self._invalidate_output()

    def set_quiescent_weights_f(self, fbin_no, src_wq):
        self._alloc_bfweight(1)
        self._bfw[0].wq[fbin_no] = src_wq
        self._bfw[0].B[fbin_no] = engine.weights_blocking_matrix(self._bfw[0].wq[fbin_no], 1)
        self._invalidate_output()

    def set_active_weights_f(self, fbin_no, packed_weight):
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        bw = self._bfw[0]
        packed_weight = np.asarray(packed_weight, np.float64)
        if packed_weight.size != 2 * (bw.chan_num - bw.NC):
            raise jdimension_error("the size of an active weight vector must be %d but it is %d\n"
                                   % (2 * (bw.chan_num - bw.NC), packed_weight.size))
        if fbin_no >= self._fftlen:
            raise jdimension_error("Must be a frequency bin %d < the length of FFT %d\n" % (fbin_no, self._fftlen))
        bw.calc_sidelobe_canceller_f(fbin_no, packed_weight[0::2] + 1j * packed_weight[1::2])
        self._dirty = True

    def zero_active_weights(self):
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        self._bfw[0].wa[:] = 0
        self._bfw[0].wl[:] = 0
        self._invalidate_output()

    def blocking_matrix(self, srcX, fbin_no=None):
        """blocking_matrix(srcX, fbinX) (beamformer.h:186); the one-argument form of earlier rounds means source 0."""
        if fbin_no is None:
            srcX, fbin_no = 0, srcX
        return self._bfw[srcX].B[fbin_no]

    calcGSCWeights, setActiveWeights_f, getBlockingMatrix = calc_gsc_weights, set_active_weights_f, blocking_matrix

    def effective_weights(self):
        self._check_weights()
        self._dirty = False
        return engine.weights_gsc_effective(self._bfw[0].wq, self._bfw[0].wl, self._fftlen, self._normalize_weight)

    def effective_weights_all_bins(self):
        """half_band_shift: wq - wl of every bin, normalised like calc_gsc_output (beamformer.cc:1208-1243, 1276-1285)"""
        self._check_weights()
        self._dirty = False
        w = self._bfw[0].wq - self._bfw[0].wl
        if self._normalize_weight:
            w = w / (np.linalg.norm(w, axis=1, keepdims=True) * self.chan_num())
        return w

    def next(self, frame_no=-5):
        if getattr(self, "_dirty", False) and not (frame_no == self._frame_no and self._vector is not None):
            self._invalidate_output()
        return SubbandDSPtr.next(self, frame_no)


class SubbandGSCRLSPtr(SubbandGSCPtr):
    """SubbandGSCRLS (beamformer.h:207-263, beamformer.cc:1447-1699): GSC whose active weights are adapted by a
    recursive-least-squares recursion per frame (Van Trees pp. 766-767).  The whole utterance runs in one
    btk_rls_process launch (mode 0); the active weights after the last served block are exported to the weight
    object like calcSidelobeCancellerU_f does (:1643)."""

    def __init__(self, fftlen=512, half_band_shift=False, mu=0.9, sigma2=0.0, nm="SubbandGSCRLS"):
        SubbandGSCPtr.__init__(self, fftlen, half_band_shift, nm)
        self._mu = float(np.float32(mu))                       # float members (beamformer.h:253-256)
        self._diagonal_weight = float(np.float32(sigma2))
        self._alpha = -1.0
        self._qctype = 0
        self._is_wa_updated = True
        self._p0 = None
        self._Pz_user = {}
        self._rls = None

    def init_precision_matrix(self, sigma2=0.01):
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        self._p0 = float(np.float32(1) / np.float32(sigma2))   # float division, beamformer.cc:1491
        self._Pz_user = {}
        self._rls = None
        self._invalidate_output()

    def set_precision_matrix(self, fbin_no, Pz):
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        if self._p0 is None:
            self._p0 = 0.0
        self._Pz_user[int(fbin_no)] = np.array(Pz, np.complex128)
        self._rls = None
        self._invalidate_output()

    def update_active_weight_vecotrs(self, flag):              # sic: the reference's spelling
        self._is_wa_updated = bool(flag)

    def set_quadratic_constraint(self, alpha, qctype=1):
        self._alpha, self._qctype = float(np.float32(alpha)), int(qctype)

    initPrecisionMatrix, setPrecisionMatrix = init_precision_matrix, set_precision_matrix
    updateActiveWeightVecotrs, setQuadraticConstraint = update_active_weight_vecotrs, set_quadratic_constraint

    def _compute_block(self):
        if self._half_band_shift:
            raise j_error("not yet implemented\n")                      # beamformer.cc:1528-1530
        import torch
        self._check_weights()
        if self._p0 is None:
            raise j_error("set the precision matrix with init_precision_matrix() or set_precision_matrix()\n")
        X = self.device_snapshots()
        bw = self._bfw[0]
        K, N = self._K, self.chan_num()
        try:
            if self._rls is None:
                wq = torch.from_numpy(np.ascontiguousarray(bw.wq[:K]).astype(np.complex128)).to(device())
                self._rls = engine.RLSState(0, 1, self._fftlen, N, wq, Nc=bw.NC)      # NC > 1: after calc_gsc_weights_2 / _n
                self._rls.init_precision_matrix(self._p0)
                if self._Pz_user or np.any(bw.wl[:K] != 0):
                    P = self._rls.P.cpu().numpy()
                    for k, Pz in self._Pz_user.items():
                        if k < K:
                            n = N - bw.NC
                            P[0, k] = bw.B[k] @ Pz[:n, :n] @ np.conj(bw.B[k].T)
                    self._rls.P.copy_(torch.from_numpy(P))
                    self._rls.w.copy_(torch.from_numpy(np.ascontiguousarray(bw.wl[:K])[None]))
            self._rls.p.update(mu=self._mu, diagonal_weight=self._diagonal_weight, qctype=self._qctype,
                               alpha=self._alpha, normalize_weight=self._normalize_weight, update=self._is_wa_updated)
            self._Y = engine.rls_process(X, self._rls)
        except _lib.BtkError as e:
            raise_from_code(e)
        # export wl / wa of the bins that adapt (1..M/2), as calcSidelobeCancellerU_f leaves them (:1643)
        wl = self._rls.w[0].cpu().numpy()
        for k in range(1, K):
            bw.wl[k] = wl[k]
            bw.wa[k] = np.conj(bw.B[k].T) @ wl[k]

    def _invalidate_output(self):
        # the recursion cannot be re-run from the middle of a block: changes apply from the next reset()
        if self._Y is None:
            return
        self._dirty = False

    def reset(self):
        # beamformer.cc:1565-1575: sources and snapshots are reset, Pz_ and wa are KEPT
        SubbandGSCPtr.reset(self)


class SubbandMVDRPtr(SubbandDSPtr):
    """SubbandMVDR (beamformer.h:333-383, beamformer.cc:2280-2599)."""

    def __init__(self, fftlen, half_band_shift=False, nm="SubbandMVDR"):
        if half_band_shift:
            raise jallocation_error("halfBandShift==true is not yet supported\n")      # beamformer.cc:2283-2285
        SubbandDSPtr.__init__(self, fftlen, half_band_shift, nm)
        self._R = None            # device complex64 [K][N][N]
        self._wmvdr = None        # host complex128 [K][N]
        self._fallbacks = 0

    def clear_channel(self):
        SubbandDSPtr.clear_channel(self)
        self._R = None
        self._wmvdr = None

    def _alloc_R(self):
        import torch
        if self._R is None:
            N = self.chan_num()
            self._R = torch.zeros((self._K, N, N), dtype=torch.complex64, device=device())

    def set_noise_spatial_spectral_matrix(self, fbin_no, Rnn):
        import torch
        Rnn = np.asarray(Rnn)
        N = self.chan_num()
        if Rnn.shape[0] != N or Rnn.shape[1] != N:
            print("The number of the rows/columns of the matrix must be %d" % N)
            return False
        self._alloc_R()
        self._R[fbin_no] = torch.from_numpy(Rnn.astype(np.complex64)).to(device())
        return True

    def set_noise_spatial_spectral_matrices(self, R):
        """All bins at once from a device tensor [K][N][N] (no host round trip)."""
        self._R = R.clone()

    def noise_spatial_spectral_matrix(self, fbin_no):
        return self._R[fbin_no].cpu().numpy().astype(np.complex128)

    def set_diffuse_noise_model(self, mic_positions, samplerate, sspeed=SSPEED):
        mp = np.asarray(mic_positions, np.float64)
        if mp.shape[0] != self.chan_num():
            print("The number of microphones must be %d but it is %d" % (self.chan_num(), mp.shape[0]))
            return False
        if mp.shape[1] < 3:
            print("The microphone positions should be described in the three dimensions")
            return False
        self._R = engine.mvdr_diffuse_model(mp, self._fftlen, samplerate, sspeed, device=device())
        return True

    def set_all_diagonal_loading(self, diagonal_weight):
        if self._R is None:
            raise j_error("Construct first a noise covariance matrix\n")
        engine.mvdr_diagonal_loading(self._R, diagonal_weight)

    def set_diagonal_looading(self, fbin_no, diagonal_weight):          # sic: the reference's spelling
        if self._R is None:
            raise j_error("Construct first a noise covariance matrix\n")
        engine.mvdr_diagonal_loading(self._R[fbin_no], diagonal_weight)

    def divide_nondiagonal_elements(self, fbin_no, mu):
        if self._R is None:
            raise j_error("Construct first a noise covariance matrix\n")
        engine.mvdr_divide_nondiagonal(self._R[fbin_no:fbin_no + 1], mu)      # in place on the bin's slice

    def divide_all_nondiagonal_elements(self, mu):
        if self._R is None:
            raise j_error("Construct first a noise covariance matrix\n")
        engine.mvdr_divide_nondiagonal(self._R, mu)                            # bins 0..M/2 (beamformer.h:357-360)

    def calc_mvdr_weights(self, samplerate, dthreshold=1.0e-8, calc_inverse_matrix=True):
        import torch
        if self._R is None:
            raise jallocation_error("Set a spatial spectral matrix before calling calc_mvdr_weights()\n")
        self._check_weights()
        wq = torch.from_numpy(self._bfw[0].wq[: self._K].astype(np.complex64)).to(device())
        try:
            W, self._fallbacks = engine.mvdr_weights(self._R, wq, dthreshold)
        except _lib.BtkError as e:
            raise_from_code(e)
        self._wmvdr = W.cpu().numpy().astype(np.complex128)
        self._invalidate_output()
        return True

    def mvdr_weights(self, fbin_no):
        return self._wmvdr[fbin_no]

    setDiffuseNoiseModel, setAllLevelsOfDiagonalLoading, calcMVDRWeights, getMVDRWeights = \
        set_diffuse_noise_model, set_all_diagonal_loading, calc_mvdr_weights, mvdr_weights

    def effective_weights(self):
        self._check_weights()
        if self._wmvdr is None:
            raise j_error("call calc_mvdr_weights() once\n")
        return self._wmvdr.astype(np.complex64)

    def next(self, frame_no=-5):
        if not (frame_no == self._frame_no and self._vector is not None) and self._wmvdr is None:
            self._check_weights()
            raise j_error("call calc_mvdr_weights() once\n")
        return SubbandDSPtr.next(self, frame_no)


class SubbandMVDRGSCPtr(SubbandMVDRPtr):
    """SubbandMVDRGSC (beamformer.h:385-437, beamformer.cc:2604-2773): MVDR quiescent + GSC lower branch."""

    def __init__(self, fftlen, half_band_shift=False, nm="SubbandMVDRGSC"):
        SubbandMVDRPtr.__init__(self, fftlen, half_band_shift, nm)
        self._normalize_weight = False

    def normalize_weight(self, flag):
        self._normalize_weight = bool(flag)

    def set_active_weights_f(self, fbin_no, packed_weight):
        if not self._bfw:
            raise j_error("set the quiescent vector once\n")
        bw = self._bfw[0]
        packed_weight = np.asarray(packed_weight, np.float64)
        wa = packed_weight[0::2] + 1j * packed_weight[1::2]
        bw.wa[fbin_no] = wa
        # calcMainlobe(..., isGSC=false) never fills B (beamformer.cc:1045-1049): wl = B wa with B == 0
        bw.wl[fbin_no] = engine.weights_sidelobe(bw.B[fbin_no], wa)
        self._dirty = True

    def zero_active_weights(self):
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        self._bfw[0].wa[:] = 0
        self._bfw[0].wl[:] = 0
        self._invalidate_output()

    def calc_blocking_matrix1(self, samplerate, delays_t):
        self._alloc_bfweight(1)
        self._bfw[0].calc_mainlobe(samplerate, delays_t, True)
        return True

    def calc_blocking_matrix2(self):
        if self._wmvdr is None:
            return False
        self._alloc_bfweight(1)
        for k in range(1, self._K):
            self._bfw[0].wq[k] = self._wmvdr[k]
            self._bfw[0].B[k] = engine.weights_blocking_matrix(self._wmvdr[k], 1)
        return True

    def upgrade_blocking_matrix(self):
        """B <- blocking matrix of the entire vector wq - wl, bins 1..M-1 (beamformer.cc:2674-2691)."""
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        bw = self._bfw[0]
        for k in range(1, self._fftlen):
            bw.B[k] = engine.weights_blocking_matrix(bw.wq[k] - bw.wl[k], bw.NC)

    def blocking_matrix_output(self, out_chan_no=0):
        """b_i^H x of the current frame for bins 0..M/2 (beamformer.cc:2693-2717); like the reference it overwrites
        those bins of the node's output vector and returns it."""
        if not self._bfw:
            raise j_error("call calc_gsc_weights_x() once\n")
        bw = self._bfw[0]
        snaps = self.snapshot_array()
        if self._vector is None:
            self._vector = np.zeros(self._fftlen, np.complex128)
        for k in range(self._K):
            self._vector[k] = np.vdot(bw.B[k][:, out_chan_no], snaps.snapshot(k))
        return self._vector

    upgradeBlockingMatrix, blockingMatrixOutput = upgrade_blocking_matrix, blocking_matrix_output

    def effective_weights(self):
        self._check_weights()
        if self._wmvdr is None:
            raise j_error("call calc_mvdr_weights() once\n")
        self._dirty = False
        full = np.zeros((self._fftlen, self.chan_num()), np.complex128)
        full[: self._K] = self._wmvdr
        return engine.weights_gsc_effective(full, self._bfw[0].wl, self._fftlen, self._normalize_weight)

    def next(self, frame_no=-5):
        if getattr(self, "_dirty", False) and not (frame_no == self._frame_no and self._vector is not None):
            self._invalidate_output()
        return SubbandMVDRPtr.next(self, frame_no)


def calc_all_delays(x, y, z, mpos):
    """calc_all_delays (beamformer.cc:1172-1189)."""
    mpos = np.asarray(mpos, np.float64)
    d = np.sqrt(np.sum(mpos[:, :3] ** 2, axis=1)) / SSPEED
    return d - d[len(d) // 2]


SubbandDS, SubbandGSC, SubbandMVDR, SubbandMVDRGSC = SubbandDSPtr, SubbandGSCPtr, SubbandMVDRPtr, SubbandMVDRGSCPtr
SubbandGSCRLS = SubbandGSCRLSPtr
