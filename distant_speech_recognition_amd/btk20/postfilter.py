"""ZelinskiPostFilterPtr (postfilter/postfilter.h:74-108, postfilter/postfilter.cc:348-491)."""
import numpy as np

from .. import _lib, engine
from .common import j_error, jdimension_error, raise_from_code
from .modulated import _mirror
from .stream import VectorComplexFeatureStream, _BlockServedStream, device

__all__ = ["TYPE_ZELINSKI1_REAL", "TYPE_ZELINSKI1_ABS", "TYPE_APAB", "TYPE_ZELINSKI2", "NO_USE_POST_FILTER",
           "ZelinskiPostFilterPtr", "ZelinskiPostFilter"]

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


ZelinskiPostFilter = ZelinskiPostFilterPtr
