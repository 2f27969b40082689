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

    def device_block(self):
        if self._Y is None:
            self._compute()
        return self._Y

    def _compute(self):
        import torch
        bf = self._bf
        if bf is None:
            src = self._samp
            bf = src.python_object().beamformer() if hasattr(src, "python_object") and hasattr(src.python_object(), "beamformer") else None
        if bf is None:
            raise j_error("set beamformer's weights \n")
        X = bf.device_snapshots()
        W = torch.from_numpy(bf.effective_weights()).to(device())
        D = torch.from_numpy(bf.alignment_vector(bool(self._type & TYPE_ZELINSKI2))).to(device())
        S, K, N, T = X.shape
        st = engine.ZelinskiState(S, K, device())
        try:
            self._Y = engine.bf_apply_zelinski(W, D, X, st, alpha=self._alpha, type_=self._type, min_frames=self._min_frames)
        except _lib.BtkError as e:
            raise_from_code(e)
        self._w_last = st.w_last[0].cpu().numpy()

    def _prepare(self):
        Y = self.device_block()
        self._frames = _mirror(Y[0].cpu().numpy(), self._size)

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

    def _beamformer(self):
        bf = self._bf
        if bf is None:
            src = self._samp
            bf = src.python_object().beamformer() if hasattr(src, "python_object") and hasattr(src.python_object(), "beamformer") else None
        if bf is None:
            raise j_error("set beamformer's weights \n")
        return bf

    _lefkimmiatis = False
    _no_R_message = "McCowanPostFilter:  construct/set a noise coherence matrix\n"

    def _compute(self):
        import torch
        bf = self._beamformer()
        if self._R is None:
            raise j_error(self._no_R_message)
        X = bf.device_snapshots()
        W = torch.from_numpy(bf.effective_weights()).to(device())
        use_wq = bool(self._type & TYPE_ZELINSKI2) and not self._lefkimmiatis      # :858-863 vs :1098
        D = torch.from_numpy(bf.alignment_vector(use_wq)).to(device())
        S, K, N, T = X.shape
        if self._R.shape[-1] != N:
            raise jdimension_error("The noise coherence matrix is %dx%d but there are %d channels\n"
                                   % (self._R.shape[-1], self._R.shape[-1], N))
        st = engine.CoherencePostFilterState(S, K, N, device(), lefkimmiatis=self._lefkimmiatis)
        try:
            st.set_coherence(self._R, self._threshold)
            if self._lefkimmiatis:
                st.set_lambda(self._R, D, self._min_sv)                               # :967-995
                self._invR_computed = True
                self._Y = engine.bf_apply_lefkimmiatis(W, D, X, st, fbin_x1=self._fbin_no1, alpha=self._alpha,
                                                       type_=self._type, min_frames=self._min_frames)
            else:
                self._Y = engine.bf_apply_mccowan(W, D, X, st, alpha=self._alpha, type_=self._type,
                                                  min_frames=self._min_frames)
        except _lib.BtkError as e:
            raise_from_code(e)
        self._w_last = st.w_last[0].cpu().numpy()


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
