"""OverSampledDFTAnalysisBankPtr / OverSampledDFTSynthesisBankPtr (modulated/modulated.h:268-340,
modulated/modulated.i:124-140): same constructors and kwargs, GPU compute through the C-ABI."""
import numpy as np

from .. import _lib, engine
from .common import jconsistency_error, jdimension_error, raise_from_code
from .stream import VectorComplexFeatureStream, VectorFloatFeatureStream, _BlockServedStream, device

__all__ = ["OverSampledDFTAnalysisBankPtr", "OverSampledDFTSynthesisBankPtr",
           "OverSampledDFTAnalysisBank", "OverSampledDFTSynthesisBank"]


def _pull_all(src):
    """Drain a finite upstream node: list of its frames (copies)."""
    out = []
    while True:
        try:
            out.append(np.array(src.next()))
        except StopIteration:
            break
    return out


from .._hostutil import mirror_bins as _mirror  # noqa: E402


class OverSampledDFTAnalysisBankPtr(_BlockServedStream, VectorComplexFeatureStream):
    def __init__(self, samp, prototype, M, m, r, delay_compensation_type=0, nm="OverSampledDFTAnalysisBank"):
        _BlockServedStream.__init__(self, M, nm)
        prototype = np.asarray(prototype, np.float64)
        if prototype.size != M * m:
            raise jconsistency_error("Prototype sizes do not match (%d vs. %d)." % (prototype.size, M * m))
        self._M, self._m, self._r = int(M), int(m), int(r)
        self._D = self._M >> self._r
        if samp.size() != self._D:
            raise jdimension_error("Input block length (%d) != D_ (%d)\n" % (samp.size(), self._D))
        self._samp = samp
        self._dct = int(delay_compensation_type)
        try:
            self._plan = engine.FilterBank(prototype, self._M, self._m, self._r, self._dct)
        except _lib.BtkError as e:
            raise_from_code(e)
        self._pcm = None

    def fftlen(self):
        return self._M

    def shiftlen(self):
        return self._D

    # legacy camelCase aliases (ENABLE_LEGACY_BTK_API)
    fftLen = fftlen
    nBlocks = lambda self: self._m
    subSampRate = lambda self: self._r

    def plan_key(self):
        return (self._M, self._m, self._r, self._dct, self._plan)

    def pcm(self):
        """All samples of the upstream node, zero-padded to whole blocks like SampleFeature(pad_zeros)."""
        if self._pcm is None:
            blocks = _pull_all(self._samp)
            self._pcm = (np.concatenate(blocks) if blocks else np.zeros(0)).astype(np.float32)
        return self._pcm

    def _prepare(self):
        import torch
        pcm = self.pcm()
        T = self._plan.num_frames(len(pcm))
        X = self._plan.analysis(torch.from_numpy(pcm[None, None, :]).to(device()), tcount=T) if len(pcm) else None
        if X is None:
            X = self._plan.analysis(torch.zeros((1, 1, self._D), device=device()), nsamples=0, tcount=T)
        self._frames = _mirror(X[0, :, 0, :].cpu().numpy(), self._M)

    def reset(self):
        self._samp.reset()
        self._pcm = None
        _BlockServedStream.reset(self)


class OverSampledDFTSynthesisBankPtr(_BlockServedStream, VectorFloatFeatureStream):
    def __init__(self, samp, prototype=None, M=None, m=None, r=0, delay_compensation_type=0, gain_factor=1,
                 nm="OverSampledDFTSynthesisBank"):
        prototype = np.asarray(prototype, np.float64)
        self._M, self._m, self._r = int(M), int(m), int(r)
        _BlockServedStream.__init__(self, self._M >> self._r, nm)
        if prototype.size != self._M * self._m:
            raise jconsistency_error("Prototype sizes do not match (%d vs. %d)." % (prototype.size, self._M * self._m))
        self._samp = samp
        self._gain = int(gain_factor)
        try:
            self._plan = engine.FilterBank(prototype, self._M, self._m, self._r, int(delay_compensation_type), synthesis=True)
        except _lib.BtkError as e:
            raise_from_code(e)

    def _prepare(self):
        import torch
        K = self._M // 2 + 1
        src = self._samp
        Yk = src.device_block() if hasattr(src, "device_block") else None
        if Yk is None:
            frames = _pull_all(src)
            if not frames:
                self._frames = np.zeros((0, self._size), np.float32)
                return
            Yk = torch.from_numpy(np.ascontiguousarray(np.stack(frames)[:, :K].T.astype(np.complex64))[None]).to(device())
        nb = self._plan.num_blocks(Yk.shape[-1])
        if nb <= 0:
            self._frames = np.zeros((0, self._size), np.float32)
            return
        out = self._plan.synthesize(Yk).cpu().numpy()[0]
        if self._gain > 0 and self._gain != 1:
            out = out * np.float32(self._gain)
        self._frames = out.reshape(nb, self._size)

    def _prepare_versioned(self):
        self._src_version = self._samp._output_version() if hasattr(self._samp, "_output_version") else 0
        self._prepare()

    def next(self, frame_no=-5):
        if frame_no == self._frame_no and self._vector is not None:
            return self._vector
        src = self._samp
        if self._frames is None:
            self._prepare_versioned()
        elif hasattr(src, "_output_version") and src._output_version() != self._src_version:
            # the source's weights changed between two blocks (moving look direction): blocks already served stay,
            # the rest is re-synthesised from the source's updated frames (frames it had already handed over keep
            # their old values -- see _advance_to below)
            done = self._frame_no + 1
            old = self._frames
            self._prepare_versioned()
            n = min(done, self._frames.shape[0])
            self._frames[:n] = old[:n]
        out = _BlockServedStream.next(self, frame_no)
        # a per-frame pull graph would by now have pulled pd + 1 frames for the first block and one more per block
        # (modulated.cc:574-578): tell the source, so that a later weight change only touches frames after those
        if hasattr(src, "_advance_to"):
            src._advance_to(self._plan.processing_delay + self._frame_no)
        return out

    def reset(self):
        self._samp.reset()
        _BlockServedStream.reset(self)


OverSampledDFTAnalysisBank = OverSampledDFTAnalysisBankPtr
OverSampledDFTSynthesisBank = OverSampledDFTSynthesisBankPtr
