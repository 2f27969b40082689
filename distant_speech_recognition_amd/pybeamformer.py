"""Host-side mirror of the reference's Python algorithm library for the beamforming hot path
(btk20_src/lib/pybeamformer.py): same function and class names, constructor arguments and
iterator behaviour; the per-frame / per-bin numpy loops are replaced by HIP kernels.

    calc_delays family            lib/pybeamformer.py:41-153   (host, float64, unchanged arithmetic)
    calc_array_manifold_f         :284-306
    calc_blocking_matrix          :309-341                     (host float64 through the C-ABI)
    SubbandGSCBeamformer          :478-537   D&S / LCMV-ready GSC with static active weights
    SubbandMVDRBeamformer         :540-585   super-directive (diffuse-noise MVDR)
    SubbandGSCLMSBeamformer       :588-762   adaptive NLMS canceller   -> btk_nlms_process
    SubbandSMIMVDRBeamformer      :930-1019  sample-matrix-inversion MVDR -> btk_cov_* + btk_mvdr_weights
"""
import numpy as np

from . import engine
from .btk20 import SSPEED, SubbandDSPtr, SubbandGSCPtr, SubbandMVDRGSCPtr      # whichever host layer btk20 resolves to
from ._hostutil import mirror_bins as _mirror, device


# ------------------------------------------------------------------ delays and steering vectors
# Time delays of arrival for the geometries the reference knows (lib/pybeamformer.py:41-153), one array expression per geometry:
# mpos is anything that converts to an [N][>=1..3] array of positions in millimetres, angles are radians, the result is seconds
# relative to the reference microphone (default: the middle one) where the reference's function is relative.
def _positions(mpos, ncol):
    p = np.asarray(mpos, float)
    if p.ndim != 2 or p.shape[1] < ncol:
        raise ValueError("microphone positions must be [N][>=%d], got %s" % (ncol, p.shape))
    return p


def _ref(chan_num, ref_micx):
    return chan_num // 2 if ref_micx is None else ref_micx


def calc_la_delays(mpos, azimuth, sspeed=SSPEED, ref_micx=None):
    """far field, linear array along x"""
    tau = -_positions(mpos, 1)[:, 0] * np.cos(azimuth) / sspeed
    return tau - tau[_ref(len(tau), ref_micx)]


def calc_pa_delays(mpos, azimuth, polar_angle, sspeed=SSPEED, ref_micx=None):
    """far field, planar array in the x-y plane"""
    p = _positions(mpos, 2)
    d = p[:, :2] - p[_ref(len(p), ref_micx), :2]
    sp = np.sin(polar_angle)
    return -(d[:, 0] * np.cos(azimuth) * sp + d[:, 1] * np.sin(azimuth) * sp) / sspeed


def calc_ca_delays(mpos, azimuth, polar_angle, sspeed=SSPEED):
    """far field, any 3-D geometry: projection of the positions on the direction of arrival (absolute, not relative)"""
    p = _positions(mpos, 3)
    sp = np.sin(polar_angle)
    return ((-sp * np.cos(azimuth)) * p[:, 0] + (-sp * np.sin(azimuth)) * p[:, 1] + (-np.cos(polar_angle)) * p[:, 2]) / sspeed


def calc_nf_delays(mpos, x, y, z, sspeed=SSPEED, ref_micx=None):
    """near field: distances to the point (x, y, z)"""
    p = _positions(mpos, 3)
    tau = np.sqrt((x - p[:, 0]) ** 2 + (y - p[:, 1]) ** 2 + (z - p[:, 2]) ** 2) / sspeed
    return tau - tau[_ref(len(tau), ref_micx)]


_DELAYS = {"linear": (calc_la_delays, 1, True), "planar": (calc_pa_delays, 2, True), "circular": (calc_ca_delays, 2, False)}


def calc_delays(array_type, mpos, position, sspeed=SSPEED, ref_micx=None):
    """dispatch on the array type of the reference's JSON configurations; anything else is the near-field model"""
    fn, nargs, relative = _DELAYS.get(array_type, (calc_nf_delays, 3, True))
    kw = {"ref_micx": ref_micx} if relative else {}
    return fn(mpos, *position[:nargs], sspeed=sspeed, **kw)


def calc_array_manifold_f(fbinX, fftlen, samplerate, delays, half_band_shift):
    """steering vector of one bin (lib/pybeamformer.py:284-306): exp(-j 2 pi f tau) / N with the bin's centre frequency f -- bins above
    fftlen / 2 are the mirror images (conjugate; with half_band_shift the half-bin offset continues into negative frequencies)"""
    tau = np.asarray(delays, float)
    if half_band_shift:
        f = (0.5 + fbinX if fbinX < fftlen / 2 else 0.5 - fftlen + fbinX) * (samplerate / float(fftlen))
        vs = np.exp(-2.0j * np.pi * f * tau)
    else:
        vs = np.exp(-2.0j * np.pi * fbinX * (samplerate / float(fftlen)) * tau)
        if fbinX > fftlen / 2:
            vs = np.conjugate(vs)
    return vs / len(tau)


def calc_blocking_matrix(vs, Nc=1):
    """pybeamformer.py:309-341."""
    return engine.weights_blocking_matrix(np.asarray(vs, complex), Nc)


# ------------------------------------------------------------------ beamformer wrappers
class SubbandBeamformer(object):
    """pybeamformer.py:376-475."""

    def __init__(self, spec_sources):
        """spec_sources: one subband-domain source (analysis bank) per channel, all of one geometry"""
        geometry = {(src.size(), src.shiftlen()) for src in spec_sources}
        if len(geometry) != 1:
            raise AssertionError("channels with inconsistent FFT / shift lengths: %s" % sorted(geometry))
        (self._fftlen, self._shiftlen), = geometry
        self._fftlen2 = self._fftlen // 2
        self._spec_sources, self._chan_num = spec_sources, len(spec_sources)
        self._beamformer = self._wqH = self._BmH = self._waH = None
        self._iter_pos = 0

    def beamformer(self):
        return self._beamformer

    def spec_sources(self):
        return self._spec_sources

    def device_block(self):
        return self._beamformer.device_block()

    def _advance_to(self, idx):
        if self._beamformer is not None:
            self._beamformer._advance_to(idx)

    def _output_version(self):
        return 0 if self._beamformer is None else self._beamformer._output_version()

    # blocks of a bounded number of frames (host/include/modulated/modulated.h, BlockSource): the C++ node computes one block of
    # the stream at a time; device_block() is the current one, _block_base() the stream index of its first frame
    def _block_base(self):
        return 0 if self._beamformer is None else self._beamformer._block_base()

    def _next_block(self):
        return False if self._beamformer is None else self._beamformer._next_block()

    def __iter__(self):
        if self._beamformer is None:
            raise NotImplementedError("Undefined beamformer object")
        while True:
            try:
                yield np.array(self._beamformer.next())
            except StopIteration:
                return

    def reset(self):
        if self._beamformer is None:
            raise NotImplementedError("Undefined beamformer object")
        self._beamformer.reset()

    def next_speaker(self):
        pass

    def chan_num(self):
        return self._chan_num

    def size(self):
        return self._fftlen

    def shiftlen(self):
        return self._shiftlen

    def set_active_weights(self):
        if self._beamformer is None:
            raise NotImplementedError("Undefined beamformer object")
        assert self._waH is not None, "The active weight vectors have to be set"
        for fbinX in range(self._fftlen2 + 1):
            packed_wa = np.zeros(2 * (self._chan_num - self._Nc), float)
            packed_wa[0::2] = np.real(self._waH[fbinX])
            packed_wa[1::2] = np.imag(self._waH[fbinX])
            self._beamformer.set_active_weights_f(fbinX, packed_wa)


class SubbandGSCBeamformer(SubbandBeamformer):
    """pybeamformer.py:478-537."""

    def __init__(self, spec_sources, Nc=1):
        SubbandBeamformer.__init__(self, spec_sources)
        self._beamformer = SubbandGSCPtr(fftlen=self._fftlen, half_band_shift=False)
        for source in self._spec_sources:
            self._beamformer.set_channel(source)
        self._Nc = Nc
        self._waH = np.zeros((self._fftlen, self._chan_num - self._Nc), complex)

    def calc_beamformer_weights(self, samplerate, delays, update_active_weights=True):
        self._beamformer.calc_gsc_weights(samplerate, delays)
        if update_active_weights:
            self.set_active_weights()
        self._wq = np.array([self._beamformer.get_weights(m) for m in range(self._fftlen2 + 1)], complex)


    def calc_beamformer_weights_n(self, samplerate, delays_t, delays_js, update_active_weights=True):
        """pybeamformer.py:517-537 (LCMV: look direction + nulls)."""
        assert (self._Nc - 1) == len(delays_js), 'Mismatch between no. constraints and no. jammers'
        self._beamformer.calc_gsc_weights_n(samplerate, delays_t, delays_js, self._Nc)
        if update_active_weights:
            self.set_active_weights()
        self._wqH = np.conjugate(np.array([self._beamformer.get_weights(m) for m in range(self._fftlen2 + 1)], complex))


class SubbandMVDRBeamformer(SubbandBeamformer):
    """pybeamformer.py:540-585 (super-directive beamformer = diffuse-noise MVDR)."""

    def __init__(self, spec_sources, Nc=1):
        SubbandBeamformer.__init__(self, spec_sources)
        self._beamformer = SubbandMVDRGSCPtr(fftlen=self._fftlen, half_band_shift=False)
        for source in self._spec_sources:
            self._beamformer.set_channel(source)
        self._Nc = Nc
        self._waH = np.zeros((self._fftlen, self._chan_num - self._Nc), complex)

    def calc_sd_beamformer_weights(self, samplerate, delays, mpos, sspeed=SSPEED, mu=0.01, update_active_weights=True):
        self._beamformer.calc_array_manifold_vectors(samplerate, delays)
        self._beamformer.set_diffuse_noise_model(mpos, samplerate, sspeed)
        self._beamformer.set_all_diagonal_loading(mu)
        self._beamformer.calc_mvdr_weights(samplerate, dthreshold=1.0E-8, calc_inverse_matrix=True)
        if update_active_weights:
            self.set_active_weights()
        self._wqH = np.conjugate(np.array([self._beamformer.mvdr_weights(m) for m in range(self._fftlen2 + 1)], complex))


class _OwnBlocks(object):
    """Classes that run their own kernels over the snapshots of a front node (self._front): the adaptive cancellers and the
    batch beamformers.  device_block() is computed from the front's current block of snapshots; moving on to the next block
    first makes sure this one went through the recursion, so the state carries over exactly as frame by frame."""

    def _block_base(self):
        return self._front.chunk_base()

    def _next_block(self):
        self.device_block()
        ok = self._front._next_block()
        self._Y = None
        self._frames = None
        return ok

    def _iter_blocks(self):
        while True:
            Y = self.device_block()
            if self._frames is None:
                self._frames = _mirror(Y[0].cpu().numpy(), self._fftlen)
            frames = self._frames
            for t in range(frames.shape[0]):
                yield frames[t]
            if not self._next_block():
                return

    def _snapshot_blocks(self):
        """(stream index of the first frame, X complex64 [1][K][N][T]) for every block of the front node's stream"""
        while True:
            X = self._front.device_snapshots()
            yield self._front.chunk_base(), X
            if not self._front._next_block():
                return


class SubbandGSCLMSBeamformer(_OwnBlocks, SubbandBeamformer):
    """pybeamformer.py:588-762: leaky power-normalised NLMS in GSC configuration.  The recursion runs
    on the GPU (btk_nlms_process); this class keeps the reference's state names as read-only views."""

    def __init__(self, spec_sources, beta=0.97, gamma=0.01, init_diagonal_load=1.0E+6, regularization_param=1.0E-4,
                 energy_floor=90, sil_thresh=1.0E+8, max_wa_l2norm=100.0, min_frames=128, slowdown_after=4096, Nc=1):
        SubbandBeamformer.__init__(self, spec_sources)
        if not (1 <= Nc <= 8 and Nc < len(spec_sources)):
            raise ValueError("Nc = %d: the GPU canceller takes 1..8 constraints, fewer than the %d channels" % (Nc, len(spec_sources)))
        self._Nc = Nc
        self._front = SubbandGSCPtr(fftlen=self._fftlen, half_band_shift=False)   # owns channels + device snapshots
        for source in self._spec_sources:
            self._front.set_channel(source)
        self._front.set_block_quantum(64)      # the step-size control is scanned in 64-frame chunks (csrc/nlms_kernels.hip)
        self._beamformer = self._front
        self._params = dict(beta=beta, gamma=gamma, init_diagonal_load=init_diagonal_load,
                            regularization_param=regularization_param, energy_floor=float(energy_floor),
                            sil_thresh=sil_thresh, max_wa_l2norm=max_wa_l2norm, min_frames=min_frames,
                            slowdown_after=slowdown_after)
        self._state = None
        self._vs = None
        self._Y = None
        self._frames = None

    def calc_beamformer_weights(self, samplerate, delays):
        """pybeamformer.py:736-743."""
        K = self._fftlen2 + 1
        self._vs = np.stack([calc_array_manifold_f(m, self._fftlen, samplerate, delays, False) for m in range(K)])
        self._wqH = np.conjugate(self._vs)
        self._BmH = None                               # formed lazily (only needed to export wa)
        self._Y = None
        self._frames = None
        if self._state is not None:
            self._state.cextra = None                  # the constraint directions follow the new manifold

    def _blocking(self, m):
        return calc_blocking_matrix(self._vs[m], self._Nc)

    def reset_stats(self):
        if self._state is not None:
            self._state.reset_stats()

    def device_block(self):
        import torch
        if self._Y is None:
            assert self._vs is not None, "call calc_beamformer_weights() first"
            X = self._front.device_snapshots()
            if X.shape[-1] % 2:
                # the canceller picks its kernel by the parity of the row pitch (csrc/nlms_kernels.hip: frame pairs need an even
                # one); an even pitch for every block makes the output independent of how the stream is cut into blocks
                Xp = torch.empty(tuple(X.shape[:-1]) + (X.shape[-1] + 1,), dtype=X.dtype, device=X.device)
                Xp[..., :-1] = X
                X = Xp[..., :-1]
            if self._state is None:
                self._state = engine.NLMSState(1, self._fftlen, self._chan_num, device(), Nc=self._Nc, **self._params)
            if self._Nc > 1 and self._state.cextra is None:
                self._state.set_constraints(self._vs)
            self._Y = engine.nlms_process(torch.from_numpy(self._vs.astype(np.complex64)).to(device()), X, self._state)
        return self._Y

    def __iter__(self):
        return self._iter_blocks()

    @property
    def _waH(self):
        """Active weights in the reference's basis, wa^H = u conj(B) per bin."""
        if self._state is None or self._vs is None:
            return np.zeros((self._fftlen2 + 1, self._chan_num - self._Nc), complex)
        u = self._state.u[0].cpu().numpy().astype(np.complex128)
        return np.stack([engine.nlms_u_to_wa(u[m], self._blocking(m)) for m in range(self._fftlen2 + 1)])

    @_waH.setter
    def _waH(self, v):
        pass

    def reset(self):
        self._front.reset()
        self.reset_stats()
        self._Y = None
        self._frames = None


class SubbandGSCRLSBeamformer(_OwnBlocks, SubbandBeamformer):
    """pybeamformer.py:765-928: RLS beamformer in GSC configuration with a regularisation term.  The recursion runs
    on the GPU (btk_rls_process mode 1, float64); _waH / _Pz are exported in the reference's basis on demand."""

    def __init__(self, spec_sources, beta=0.97, gamma=0.04, mu=0.97, init_diagonal_load=1.0E+6,
                 regularization_param=1.0E-2, sil_thresh=1.0E+8, constraint_option=3, alpha2=10.0,
                 max_wa_l2norm=100.0, min_frames=128, slowdown_after=4096, Nc=1):
        SubbandBeamformer.__init__(self, spec_sources)
        self._Nc = int(Nc)
        self._front = SubbandGSCPtr(fftlen=self._fftlen, half_band_shift=False)   # owns channels + device snapshots
        for source in self._spec_sources:
            self._front.set_channel(source)
        self._beamformer = self._front
        # slowdown_after is accepted and, as in the reference (:777-811), never used
        self._params = dict(beta=beta, gamma=gamma, mu=mu, init_diagonal_load=init_diagonal_load,
                            regularization_param=regularization_param, sil_thresh=sil_thresh,
                            constraint_option=constraint_option, alpha2=alpha2, max_wa_l2norm=max_wa_l2norm,
                            min_frames=min_frames)
        self._state = None
        self._vs = None
        self._Y = None
        self._frames = None

    def calc_beamformer_weights(self, samplerate, delays):
        """pybeamformer.py:900-908."""
        K = self._fftlen2 + 1
        self._vs = np.stack([calc_array_manifold_f(m, self._fftlen, samplerate, delays, False) for m in range(K)])
        self._wqH = np.conjugate(self._vs)
        self._state = None
        self._Y = None
        self._frames = None

    def _blocking(self, m):
        return calc_blocking_matrix(self._vs[m], self._Nc)

    def reset_stats(self):
        if self._state is not None:
            self._state.reset_stats()

    def device_block(self):
        import torch
        if self._Y is None:
            assert self._vs is not None, "call calc_beamformer_weights() first"
            X = self._front.device_snapshots()
            if self._state is None:
                self._state = engine.RLSState(1, 1, self._fftlen, self._chan_num,
                                              torch.from_numpy(np.ascontiguousarray(self._vs)).to(device()), Nc=self._Nc, **self._params)
            self._Y = engine.rls_process(X, self._state)
        return self._Y

    def __iter__(self):
        return self._iter_blocks()

    def _export(self):
        K = self._fftlen2 + 1
        P = self._state.P[0].cpu().numpy()
        w = self._state.w[0].cpu().numpy()
        return [engine.rls_state_to_reference(1, P[m], w[m], self._blocking(m)) for m in range(K)]

    @property
    def _waH(self):
        if self._state is None or self._vs is None:
            return np.zeros((self._fftlen2 + 1, self._chan_num - self._Nc), complex)
        return np.stack([e[1] for e in self._export()])

    @_waH.setter
    def _waH(self, v):
        pass

    @property
    def _Pz(self):
        if self._state is None or self._vs is None:
            n = self._chan_num - self._Nc
            return [np.identity(n) / self._params["init_diagonal_load"] for _ in range(self._fftlen2 + 1)]
        return [e[0] for e in self._export()]

    def reset(self):
        self._front.reset()
        self.reset_stats()
        self._Y = None
        self._frames = None


class SubbandSMIMVDRBeamformer(SubbandMVDRBeamformer):
    """pybeamformer.py:930-1019: MVDR by sample matrix inversion."""

    def __init__(self, spec_sources, Nc=1):
        SubbandMVDRBeamformer.__init__(self, spec_sources, Nc)
        self._noise_covariance_matrices = None      # device complex64 [1][K][N][N]
        self._noise_frame_num = None                # device float32 [1]

    def accu_stats_from_label(self, samplerate, target_labs=[(0.1, -1)], energy_threshold=10):
        import torch
        # noise-frame label exactly as the loop of pybeamformer.py:967-985 walks the VAD segments, block after block of the stream
        elapsed_time, time_delta, labx = 0.0, self.shiftlen() / float(samplerate), 0
        while True:
            X = self._beamformer.device_snapshots()
            T = X.shape[-1]
            label = np.zeros(T, np.float32)
            for t in range(T):
                is_target = False
                if labx < len(target_labs):
                    if elapsed_time >= target_labs[labx][0] and (elapsed_time <= target_labs[labx][1] or target_labs[labx][1] < 0):
                        is_target = True
                    elif elapsed_time > target_labs[labx][1]:
                        labx += 1
                label[t] = 0.0 if is_target else 1.0
                elapsed_time += time_delta
            en = engine.frame_energy(X, self._fftlen)
            w, self._noise_frame_num = engine.cov_frame_gate(en, torch.from_numpy(label[None]).to(device()), energy_threshold,
                                                             count=self._noise_frame_num)
            self._noise_covariance_matrices = engine.cov_accumulate(X, R=self._noise_covariance_matrices, frame_weights=w)
            if not self._beamformer._next_block():
                break
        self._beamformer.reset()          # the reference drained its sources here; they are re-read afterwards

    def finalize_stats(self):
        assert self._noise_frame_num is not None and float(self._noise_frame_num.item()) > 0, \
            "No noise stats accumulated; Use self.accu_stats_from_label()"
        engine.cov_finalize(self._noise_covariance_matrices, self._noise_frame_num)

    def calc_beamformer_weights(self, samplerate, delays, mu=1e-4, update_active_weights=True):
        self._beamformer.calc_array_manifold_vectors(samplerate, delays)
        self._beamformer.set_noise_spatial_spectral_matrices(self._noise_covariance_matrices[0])
        self._beamformer.set_all_diagonal_loading(mu)
        self._beamformer.calc_mvdr_weights(samplerate, dthreshold=1.0E-8, calc_inverse_matrix=True)
        if update_active_weights:
            self.set_active_weights()
        self._wqH = np.conjugate(np.array([self._beamformer.mvdr_weights(m) for m in range(self._fftlen2 + 1)], complex))


def _vad_noise_label(T, time_delta, target_labs):
    """Per-frame target indicator exactly as the loops of pybeamformer.py:967-985 / :1071-1080 walk the VAD segments."""
    elapsed_time, labx = 0.0, 0
    is_target = np.zeros(T, np.float32)
    for t in range(T):
        tgt = False
        if labx < len(target_labs):
            if elapsed_time >= target_labs[labx][0] and (elapsed_time <= target_labs[labx][1] or target_labs[labx][1] < 0):
                tgt = True
            elif elapsed_time > target_labs[labx][1]:
                labx += 1
        is_target[t] = 1.0 if tgt else 0.0
        elapsed_time += time_delta
    return is_target


class SubbandSOSBatchBeamformer(_OwnBlocks, SubbandBeamformer):
    """pybeamformer.py:1022-1197: batch beamformer driven by target / noise spatial covariance matrices.
    Statistics are accumulated on the GPU (btk_cov_accumulate); _target/_noise_covariance_matrices are device
    complex64 [1][K][N][N], the frame counters device float32 [1][K]."""

    def __init__(self, spec_sources):
        SubbandBeamformer.__init__(self, spec_sources)
        self._isamp = 0
        self._front = SubbandDSPtr(fftlen=self._fftlen, half_band_shift=False)    # owns channels + device snapshots
        for source in self._spec_sources:
            self._front.set_channel(source)
        self._beamformer = self._front
        self._wqH = np.ones((self._fftlen2 + 1, self._chan_num), complex)
        self._Y = None
        self._frames = None
        self.reset_stats()

    def reset_stats(self):
        self._target_covariance_matrices = None
        self._noise_covariance_matrices = None
        self._target_frame_counts = None
        self._noise_frame_counts = None

    def _accumulate(self, X, tf_t, tf_j, fw_t, fw_j, cnt_t, cnt_j):
        self._target_covariance_matrices = engine.cov_accumulate(X, R=self._target_covariance_matrices,
                                                                 tf_weights=tf_t, frame_weights=fw_t)
        self._noise_covariance_matrices = engine.cov_accumulate(X, R=self._noise_covariance_matrices,
                                                                tf_weights=tf_j, frame_weights=fw_j)
        self._target_frame_counts = cnt_t if self._target_frame_counts is None else self._target_frame_counts + cnt_t
        self._noise_frame_counts = cnt_j if self._noise_frame_counts is None else self._noise_frame_counts + cnt_j

    def accu_stats_from_label(self, samplerate, target_labs=[(0.1, -1)], energy_threshold=10):
        import torch
        K = self._fftlen2 + 1
        for base, X in self._snapshot_blocks():
            T = X.shape[-1]
            tgt = _vad_noise_label(base + T, self.shiftlen() / float(samplerate), target_labs)[base:]
            en = engine.frame_energy(X, self._fftlen)
            fw_t, ct = engine.cov_frame_gate(en, torch.from_numpy(tgt[None]).to(device()), energy_threshold)
            fw_j, cj = engine.cov_frame_gate(en, torch.from_numpy((1.0 - tgt)[None]).to(device()), energy_threshold)
            self._accumulate(X, None, None, fw_t, fw_j, ct[:, None].expand(1, K).contiguous(), cj[:, None].expand(1, K).contiguous())
        self._front.reset()          # the reference drained its sources here; they are re-read afterwards

    def accu_stats_from_tfmask(self, samplerate, mask_t, mask_j, energy_threshold=10):
        import torch
        K = self._fftlen2 + 1
        for base, X in self._snapshot_blocks():
            T = X.shape[-1]
            mt = np.zeros((K, T), np.float32)
            mj = np.zeros((K, T), np.float32)
            n = max(0, min(T, len(mask_t) - base))
            mt[:, :n] = np.asarray(mask_t, np.float32)[base:base + n, :K].T
            mj[:, :n] = np.asarray(mask_j, np.float32)[base:base + n, :K].T
            mt, mj = np.maximum(mt, 0.0), np.maximum(mj, 0.0)           # only mask > 0 contributes (:1137-1146)
            tf_t, tf_j = torch.from_numpy(mt[None]).to(device()), torch.from_numpy(mj[None]).to(device())
            en = engine.frame_energy(X, self._fftlen)
            fw, _ = engine.cov_frame_gate(en, None, energy_threshold)
            self._accumulate(X, tf_t, tf_j, fw, fw, engine.cov_mask_count(tf_t, fw), engine.cov_mask_count(tf_j, fw))
        self._front.reset()          # the reference drained its sources here; they are re-read afterwards

    def finalize_stats(self):
        pass

    def device_block(self):
        import torch
        if self._Y is None:
            X = self._front.device_snapshots()
            W = torch.from_numpy(np.conjugate(np.asarray(self._wqH)).astype(np.complex64)).to(device())
            self._Y = engine.bf_apply(W, X)
        return self._Y

    def __iter__(self):
        for frame in self._iter_blocks():
            self._isamp += 1
            yield frame

    def reset(self):
        self._front.reset()
        self._isamp = 0
        self._Y = None
        self._frames = None


class SubbandBlindMVDRBeamformer(SubbandSOSBatchBeamformer):
    """pybeamformer.py:1210-1263: MVDR without the look direction (MMSE beamforming)."""

    def _check_stats(self):
        if self._target_covariance_matrices is None:
            raise RuntimeError('No target signal SOS')
        if self._noise_covariance_matrices is None:
            raise RuntimeError('No noise signal SOS')

    def calc_beamformer_weights(self, ref_micx=0, offset=0.0):
        self._check_stats()
        assert offset >= 0 and offset <= 1, "The offset value %f is out of [0, 1]" % (offset)
        W, failed = engine.bmvdr_weights(self._target_covariance_matrices[0], self._noise_covariance_matrices[0],
                                         ref_micx, offset)
        if failed:
            raise ArithmeticError('Matrix inversion failed\nAdd a small value to the diagonal component of the covariance matrix')
        self._wqH = W.cpu().numpy().astype(complex)
        self._Y = None
        self._frames = None

    def finalize_stats(self, gamma=1e-6):
        assert self._target_frame_counts is not None and float(self._target_frame_counts.min().item()) > 0, \
            "No target signal stats accumulated; Use self.accu_stats_from_label() or accu_stats_from_tfmask()"
        assert self._noise_frame_counts is not None and float(self._noise_frame_counts.min().item()) > 0, \
            "No noise stats accumulated; Use self.accu_stats_from_label() or accu_stats_from_tfmask()"
        engine.cov_finalize(self._target_covariance_matrices, self._target_frame_counts)
        engine.cov_finalize(self._noise_covariance_matrices, self._noise_frame_counts, gamma=max(gamma, 0.0))


class SubbandGEVBeamformer(SubbandBlindMVDRBeamformer):
    """pybeamformer.py:1266-1328: generalised eigenvector beamformer."""

    def calc_beamformer_weights(self):
        self._check_stats()
        W, failed = engine.gev_weights(self._target_covariance_matrices[0], self._noise_covariance_matrices[0])
        if failed:
            raise ArithmeticError('GEV failed\nAdd a small value to the diagonal component of the covariance matrix')
        self._wqH = W.cpu().numpy().astype(complex)
        self._Y = None
        self._frames = None

    def finalize_stats(self, gamma=1e-6):
        assert self._target_frame_counts is not None and float(self._target_frame_counts.min().item()) > 0, \
            "No target signal stats accumulated; Use self.accu_stats_from_label() or accu_stats_from_tfmask()"
        assert self._noise_frame_counts is not None and float(self._noise_frame_counts.min().item()) > 0, \
            "No noise stats accumulated; Use self.accu_stats_from_label() or accu_stats_from_tfmask()"
        # the target covariance stays un-normalised: no impact on the GEV solution (:1317-1318)
        engine.cov_finalize(self._noise_covariance_matrices, self._noise_frame_counts, gamma=max(gamma, 0.0))
        engine.cov_trace_normalize(self._noise_covariance_matrices)
