"""MultiChannelWPEDereverberationPtr / MultiChannelWPEDereverberationFeaturePtr
(dereverberation/dereverberation.h:101-190, dereverberation/dereverberation.i:73-180)."""
import numpy as np

from .. import _lib, engine
from .common import jallocation_error, jindex_error, jinitialization_error, jnumeric_error, raise_from_code
from .modulated import OverSampledDFTAnalysisBankPtr, _mirror, _pull_all
from .stream import VectorComplexFeatureStream, _BlockServedStream, device

__all__ = ["MultiChannelWPEDereverberationPtr", "MultiChannelWPEDereverberationFeaturePtr",
           "SingleChannelWPEDereverberationFeaturePtr"]


class MultiChannelWPEDereverberationPtr(object):
    def __init__(self, subbands_num, channels_num, lower_num, upper_num, iterations_num=2, load_db=-20.0,
                 band_width=0.0, diagonal_bias=0.0, samplerate=16000.0):
        self._M, self._C = int(subbands_num), int(channels_num)
        self._lower, self._upper, self._iters = int(lower_num), int(upper_num), int(iterations_num)
        self._load_db, self._band_width, self._bias, self._fs = float(load_db), float(band_width), float(diagonal_bias), float(samplerate)
        self._sources = []
        self._G = None
        self._estimated = False
        self._frames_num = 0
        self._X = None
        self._OUT = None

    def size(self):
        return self._M

    def set_input(self, samples):
        if len(self._sources) == self._C:
            raise jallocation_error("Channel capacity exceeded.")
        self._sources.append(samples)

    setInput = set_input

    def print_objective_func(self, subbandX):
        pass

    def _snapshots(self):
        """X complex64 [1][K][C][T] on the device from the current state of the input nodes."""
        import torch
        K = self._M // 2 + 1
        src = self._sources
        if all(isinstance(c, OverSampledDFTAnalysisBankPtr) for c in src) and len(set(c.plan_key()[:4] for c in src)) == 1:
            pcms = [c.pcm() for c in src]
            Lmin = min(len(p) for p in pcms)
            pcm = np.stack([p[:Lmin] for p in pcms])[None]
            return src[0]._plan.analysis(torch.from_numpy(np.ascontiguousarray(pcm)).to(device()))
        frames = [_pull_all(c) for c in src]
        T = min(len(f) for f in frames)
        Xh = np.stack([np.stack(f[:T])[:, :K] for f in frames])                      # [C][T][K]
        return torch.from_numpy(np.ascontiguousarray(np.transpose(Xh, (2, 0, 1))[None]).astype(np.complex64)).to(device())

    def estimate_filter(self, start_frame_no=0, end_frame_no=-1):
        X = self._snapshots()
        T = X.shape[-1]
        # fill_buffer_ (dereverberation.cc:506-529) counts frX from 0 and pulls one frame per frX in [start, end) from
        # the inputs' CURRENT position: the estimate sees the FIRST end - start frames (all of them when end < 0)
        n = T if end_frame_no < 0 else min(max(int(end_frame_no) - max(int(start_frame_no), 0), 0), T)
        Xe = X[..., :n].contiguous()
        self._frames_num = Xe.shape[-1]
        try:
            self._G = engine.wpe_estimate(Xe, self._M, self._lower, self._upper, self._iters, self._load_db, self._band_width,
                                          self._bias, self._fs, G=self._G)
        except _lib.BtkError as e:
            if e.code == _lib.BTK_ERR_NUMERIC:
                raise jnumeric_error(str(e))
            raise_from_code(e)
        for s in self._sources:                        # estimate_filter resets its inputs (dereverberation.cc:428-431)
            s.reset()
        self._estimated = True
        self._X = self._OUT = None
        return self._frames_num

    def device_output(self):
        """Dereverberated snapshots of every channel, complex64 [1][K][C][T] on the device."""
        if not self._estimated:
            raise jinitialization_error("Call SingleChannelWPEDereverberationFeature::estimate_filter()\n")
        if self._OUT is None:
            self._X = self._snapshots()
            self._OUT = engine.wpe_apply(self._X, self._G, self._M, self._lower, self._upper, self._band_width, self._fs)
        return self._OUT

    def reset(self):
        for s in self._sources:
            s.reset()
        self._X = self._OUT = None

    def reset_filter(self):
        self._estimated = False
        self._frames_num = 0

    def next_speaker(self):
        self.reset()
        self._G = None
        self._estimated = False

    nextSpeaker = next_speaker


class MultiChannelWPEDereverberationFeaturePtr(_BlockServedStream, VectorComplexFeatureStream):
    def __init__(self, source, channel_no, primary_channel_no=0, nm="MultiChannelWPEDereverberationFeature"):
        _BlockServedStream.__init__(self, source.size(), nm)
        if channel_no >= source._C:
            raise jindex_error("Invalid channel index: it exceeds the number of channels: %u >= %u\n" % (channel_no, source._C))
        self._source, self._channel, self._primary = source, int(channel_no), int(primary_channel_no)

    def wpe_source(self):
        return self._source

    def channel_no(self):
        return self._channel

    def _prepare(self):
        out = self._source.device_output()
        self._frames = _mirror(out[0, :, self._channel, :].cpu().numpy(), self._size)

    def reset(self):
        self._source.reset()
        _BlockServedStream.reset(self)


class SingleChannelWPEDereverberationFeaturePtr(_BlockServedStream, VectorComplexFeatureStream):
    """SingleChannelWPEDereverberationFeature (dereverberation/dereverberation.cc:40-307,
    dereverberation.i:73-81): the C = 1 case of the same estimator, without a diagonal bias."""

    def __init__(self, samples, lower_num, upper_num, iterations_num=2, load_db=-20.0, band_width=0.0,
                 samplerate=16000.0, nm="SingleChannelWPEDereverberationFeature"):
        _BlockServedStream.__init__(self, samples.size(), nm)
        self._core = MultiChannelWPEDereverberationPtr(samples.size(), 1, lower_num, upper_num, iterations_num, load_db,
                                                       band_width, 0.0, samplerate)
        self._core.set_input(samples)

    def estimate_filter(self, start_frame_no=0, frame_num=-1):
        # the header calls the second argument frame_num, the implementation uses it as an END index
        # (dereverberation.cc:74-94, 214-215): same counting rule as the multi-channel estimator
        return self._core.estimate_filter(start_frame_no, frame_num)

    def print_objective_func(self, subband_no):
        pass

    def next_speaker(self):
        self._core.next_speaker()
        _BlockServedStream.reset(self)

    nextSpeaker = next_speaker

    def _prepare(self):
        out = self._core.device_output()
        self._frames = _mirror(out[0, :, 0, :].cpu().numpy(), self._size)

    def reset(self):
        self._core.reset()
        _BlockServedStream.reset(self)
