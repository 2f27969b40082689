"""ZelinskiPostFilterPtr (postfilter/postfilter.h:74-108, postfilter/postfilter.cc:348-491), McCowanPostFilterPtr
(postfilter.h:123-162, postfilter.cc:496-935) and LefkimmiatisPostFilterPtr (postfilter.h:174-203, postfilter.cc:938-1190)."""
import numpy as np

from .. import _lib, engine
from .common import j_error, jdimension_error, raise_from_code
from .modulated import _mirror
from .stream import VectorComplexFeatureStream, _BlockServedStream, device

__all__ = ["TYPE_ZELINSKI1_REAL", "TYPE_ZELINSKI1_ABS", "TYPE_APAB", "TYPE_ZELINSKI2", "NO_USE_POST_FILTER",
           "ZelinskiPostFilterPtr", "ZelinskiPostFilter", "McCowanPostFilterPtr", "McCowanPostFilter",
           "LefkimmiatisPostFilterPtr", "LefkimmiatisPostFilter"]

TYPE_ZELINSKI1_REAL, TYPE_ZELINSKI1_ABS, TYPE_APAB, TYPE_ZELINSKI2, NO_USE_POST_FILTER = 0x01, 0x02, 0x04, 0x08, 0x00


class ZelinskiPostFilterPtr(_BlockServedStream, VectorComplexFeatureStream):
    def __init__(self, output, fftlen, alpha=0.6, type=2, min_frames=0, nm="ZelinskPostFilter"):
        _BlockServedStream.__init__(self, fftlen, nm)
        if output.size() != fftlen:
            raise jdimension_error("Input block length (%d) != fftLen (%d)\n" % (output.size(), fftlen))
        self._samp = output
        self._alpha, self._type, self._min_frames = float(alpha), int(type), int(min_frames)
        self._bf = None
        self._Y = None
        self._w_last = None

    def set_beamformer(self, bf):
        """bf: the SubbandDS/GSC/MVDR node (test_online_beamforming.py:204 passes beamformer.beamformer())."""
        self._bf = bf

    setBeamformer = set_beamformer

    def postfilter_weights(self):
        if self._w_last is None:
            return None
        K = self._size // 2 + 1
        w = np.zeros(self._size, np.complex128)
        w[:K] = self._w_last
        w[K:] = w[self._size // 2 - 1:0:-1]
        return w

    getPostFilterWeights = postfilter_weights

    def _find_beamformer(self):
        bf = self._bf
        if bf is None:
            src = self._samp
            bf = src.python_object().beamformer() if hasattr(src, "python_object") and hasattr(src.python_object(), "beamformer") else None
        if bf is None:
            raise j_error("set beamformer's weights \n")
        if getattr(bf, "_half_band_shift", False):
            # the reference runs its post-filters over all fftLen bins of a half-band-shifted beamformer (postfilter.cc:170-182);
            # this engine's post-filter kernels work on the M/2+1 bins of a non-shifted bank: refuse instead of filtering wrongly
            raise j_error("post-filters over a beamformer with halfBandShift==true are not supported by this engine\n")
        return bf

    def _bf_or_none(self):
        try:
            return self._find_beamformer()
        except j_error:
            return None

    def _output_version(self):
        bf = self._bf_or_none()
        return 0 if bf is None else bf._output_version()

    def _advance_to(self, idx):
        if idx > self._frame_no:
            self._frame_no = idx
        bf = self._bf_or_none()
        if bf is not None:
            bf._advance_to(idx)

    def device_block(self):
        bf = self._find_beamformer()
        if self._Y is None:
            self._bf_version = bf._output_version()
            self._compute(0)
        elif bf._output_version() != self._bf_version:
            # weights recomputed between two frames: frames already handed over keep their values, the CSD history
            # restarts while the frame counter keeps counting (alloc_bfweight_, beamformer.cc:1082-1092)
            self._bf_version = bf._output_version()
            self._compute(self._frame_no + 1)
            self._frames = None
        return self._Y

    def _run_filter(self, W, D, X, from_frame):
        """returns (Y [S][K][T'], w_last [K]) for the frames from_frame.. of X"""
        S, K, N, T = X.shape
        st = engine.ZelinskiState(S, K, device())
        st.frames_done = from_frame
        Y = engine.bf_apply_zelinski(W, D, X[..., from_frame:].contiguous(), st, alpha=self._alpha, type_=self._type,
                                     min_frames=self._min_frames)
        return Y, st.w_last[0].cpu().numpy()

    def _compute(self, from_frame=0):
        import torch
        bf = self._find_beamformer()
        X = bf.device_snapshots()
        W = torch.from_numpy(bf.effective_weights()).to(device())
        D = torch.from_numpy(bf.alignment_vector(self._use_wq())).to(device())
        T = X.shape[-1]
        from_frame = max(0, min(int(from_frame), T))
        old = self._Y
        try:
            if from_frame < T:
                Ynew, self._w_last = self._run_filter(W, D, X, from_frame)
            else:
                Ynew = None
        except _lib.BtkError as e:
            raise_from_code(e)
        if from_frame == 0 or old is None:
            self._Y = Ynew
        else:
            self._Y = old.clone()
            if Ynew is not None:
                self._Y[..., from_frame:] = Ynew

    def _use_wq(self):
        return bool(self._type & TYPE_ZELINSKI2)

    def _prepare(self):
        Y = self.device_block()
        self._frames = _mirror(Y[0].cpu().numpy(), self._size)

    def next(self, frame_no=-5):
        bf = self._bf_or_none()
        if self._frames is not None and bf is not None and bf._output_version() != getattr(self, "_bf_version", None):
            done = self._frame_no + 1
            old = self._frames
            self._prepare()
            self._frames[:done] = old[:done]
        return _BlockServedStream.next(self, frame_no)

    def reset(self):
        self._samp.reset()
        self._Y = None
        _BlockServedStream.reset(self)


class McCowanPostFilterPtr(ZelinskiPostFilterPtr):
    """McCowan post-filter: Zelinski's estimator with the pair terms weighted by a noise coherence matrix."""

    def __init__(self, output, fftlen, alpha=0.6, type=2, min_frames=0, threshold=0.99, nm="McCowanPostFilterPtr"):
        ZelinskiPostFilterPtr.__init__(self, output, fftlen, alpha, type, min_frames, nm)
        self._threshold = float(np.float32(threshold))
        self._R = None                        # device complex64 [K][N][N]
        self._K = fftlen // 2 + 1
        self._invR_computed = False

    # ---- noise coherence matrix (postfilter.cc:536-660)
    def noise_spatial_spectral_matrix(self, fbin_no):
        return None if self._R is None else self._R[fbin_no].cpu().numpy().astype(np.complex128)

    def set_noise_spatial_spectral_matrix(self, fbin_no, Rnn):
        import torch
        Rnn = np.asarray(Rnn, np.complex128)
        if Rnn.ndim != 2 or Rnn.shape[0] != Rnn.shape[1]:
            print("The noise coherence matrix should be the square matrix")
            return False
        if self._R is None:
            self._R = torch.zeros((self._K,) + Rnn.shape, dtype=torch.complex64, device=device())
        self._R[fbin_no] = torch.from_numpy(Rnn.astype(np.complex64)).to(device())
        self._invR_computed = False
        self._Y = None
        return True

    def set_diffuse_noise_model(self, mic_positions, samplerate, sspeed=343740.0):
        mp = np.asarray(mic_positions, np.float64)
        if mp.shape[1] < 3:
            print("The microphone positions should be described in the three dimensions")
            return False
        try:
            self._R = engine.mvdr_diffuse_model(mp, self._size, samplerate, sspeed, device=device())
        except _lib.BtkError as e:
            raise_from_code(e)
        self._invR_computed = False
        self._Y = None
        return True

    def set_all_diagonal_loading(self, diagonal_weight):
        if self._R is None:
            raise j_error("Construct/set first a noise coherence matrix\n")
        engine.mvdr_diagonal_loading(self._R, float(np.float32(diagonal_weight)))
        self._Y = None

    def set_diagonal_looading(self, fbin_no, diagonal_weight):               # sic: the reference's spelling
        if self._R is None:
            raise j_error("Construct/set first a noise coherence matrix\n")
        engine.mvdr_diagonal_loading(self._R[fbin_no: fbin_no + 1], float(np.float32(diagonal_weight)))
        self._Y = None

    def divide_nondiagonal_elements(self, fbin_no, mu):
        import torch
        N = self._R.shape[-1]
        eye = torch.eye(N, dtype=torch.bool, device=self._R.device)
        # the division happens in double and is stored back (gsl_complex_div, postfilter.cc:650-658)
        Rk = self._R[fbin_no].to(torch.complex128)
        self._R[fbin_no] = torch.where(eye, Rk, Rk / (1.0 + float(np.float32(mu)))).to(torch.complex64)
        self._Y = None

    def divide_all_nondiagonal_elements(self, mu):
        for k in range(self._K):
            self.divide_nondiagonal_elements(k, mu)

    getNoiseSpatialSpectralMatrix, setNoiseSpatialSpectralMatrix = noise_spatial_spectral_matrix, set_noise_spatial_spectral_matrix
    setDiffuseNoiseModel, setAllLevelsOfDiagonalLoading = set_diffuse_noise_model, set_all_diagonal_loading
    setLevelOfDiagonalLoading = set_diagonal_looading
    divideAllNonDiagonalElements, divideNonDiagonalElements = divide_all_nondiagonal_elements, divide_nondiagonal_elements

    _lefkimmiatis = False
    _no_R_message = "McCowanPostFilter:  construct/set a noise coherence matrix\n"

    def _use_wq(self):
        return bool(self._type & TYPE_ZELINSKI2) and not self._lefkimmiatis         # :858-863 vs :1098

    def _run_filter(self, W, D, X, from_frame):
        if self._R is None:
            raise j_error(self._no_R_message)
        S, K, N, T = X.shape
        if self._R.shape[-1] != N:
            raise jdimension_error("The noise coherence matrix is %dx%d but there are %d channels\n"
                                   % (self._R.shape[-1], self._R.shape[-1], N))
        st = engine.CoherencePostFilterState(S, K, N, device(), lefkimmiatis=self._lefkimmiatis)
        st.frames_done = from_frame
        st.set_coherence(self._R, self._threshold)
        Xs = X[..., from_frame:].contiguous()
        if self._lefkimmiatis:
            st.set_lambda(self._R, D, self._min_sv)                               # :967-995
            self._invR_computed = True
            Y = engine.bf_apply_lefkimmiatis(W, D, Xs, st, fbin_x1=self._fbin_no1, alpha=self._alpha, type_=self._type,
                                             min_frames=self._min_frames)
        else:
            Y = engine.bf_apply_mccowan(W, D, Xs, st, alpha=self._alpha, type_=self._type, min_frames=self._min_frames)
        return Y, st.w_last[0].cpu().numpy()


class LefkimmiatisPostFilterPtr(McCowanPostFilterPtr):
    """Lefkimmiatis post-filter: Wiener gain under the diffuse-noise-field assumption."""

    _lefkimmiatis = True
    _no_R_message = "LefkimmiatisPostFilter:  construct/set a noise coherence matrix\n"

    def __init__(self, output, fftlen, min_sv=1.0E-8, fbin_no1=0, alpha=0.6, type=2, min_frames=0, threshold=0.99,
                 nm="LefkimmiatisPostFilterPtr"):
        McCowanPostFilterPtr.__init__(self, output, fftlen, alpha, type, min_frames, threshold, nm)
        self._min_sv, self._fbin_no1 = float(min_sv), int(fbin_no1)

    def calc_inverse_noise_spatial_spectral_matrix(self):
        """The inverse is formed with the snapshots' look direction when the block is computed (calcLambda needs
        only d^H pinv(R) d); kept for API compatibility."""
        if self._R is None:
            raise j_error(self._no_R_message)
        self._Y = None

    calcInverseNoiseSpatialSpectralMatrix = calc_inverse_noise_spatial_spectral_matrix


ZelinskiPostFilter = ZelinskiPostFilterPtr
McCowanPostFilter = McCowanPostFilterPtr
LefkimmiatisPostFilter = LefkimmiatisPostFilterPtr
