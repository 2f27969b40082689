"""SampleFeature block reader (feature/feature.h:153-206, feature/feature.cc:238-389,605-649):
16-bit PCM WAV, un-normalised float samples (SFC_SET_NORM_FLOAT=false, feature.cc:265-269)."""
import wave

import numpy as np

from .common import jindex_error, jio_error, jiterator_error, jconsistency_error
from .stream import VectorFloatFeatureStream

__all__ = ["SampleFeaturePtr", "SampleFeature"]


class SampleFeaturePtr(VectorFloatFeatureStream):
    def __init__(self, fn="", block_len=320, shift_len=160, pad_zeros=False, nm="Sample"):
        VectorFloatFeatureStream.__init__(self, block_len, nm)
        self._samples = None
        self._ttl = 0
        self._shift = int(shift_len)
        self._cur = 0
        self._pad_zeros = bool(pad_zeros)
        self._vector = np.zeros(block_len, np.float32)
        self._samplerate = 0
        if fn:
            self.read(fn)

    def read(self, fn, format=0, samplerate=16000, chX=1, chN=1, cfrom=0, to=-1, outsamplerate=-1, norm=0.0):
        """The 2nd positional argument is `format` in the reference (tests pass the sample rate there,
        unit_test/test_online_beamforming.py:83); it is ignored for WAV."""
        try:
            w = wave.open(fn, "rb")
        except Exception as e:
            raise jio_error("Could not open file %s: %s" % (fn, e))
        if w.getsampwidth() != 2:
            raise jio_error("Only 16-bit PCM WAV is supported (%s)" % fn)
        nch = w.getnchannels()
        data = np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.float32)
        self._samplerate = w.getframerate()
        w.close()
        if chX > nch or chX < 1:
            raise jconsistency_error("Selected channel out of range of available channels.")
        data = data.reshape(-1, nch)[:, chX - 1]
        if to > 0:
            data = data[cfrom:to + 1]
        elif cfrom > 0:
            data = data[cfrom:]
        if norm not in (0.0, 1.0):
            data = data * np.float32(norm)
        self.set_samples(data)
        return self._ttl

    def set_samples(self, samples):
        """Feed samples from memory (same state as after read())."""
        self._samples = np.ascontiguousarray(samples, np.float32)
        self._ttl = self._samples.shape[0]
        self._cur = 0
        self.reset()
        self._is_end = False

    def samples(self):
        return self._samples

    def samplerate(self):
        return self._samplerate

    def samplesN(self):
        return self._ttl

    def next(self, frame_no=-5):
        if self._is_end:
            raise jiterator_error("end of samples!")
        if frame_no == self._frame_no:
            return self._vector
        if frame_no >= 0 and frame_no - 1 != self._frame_no:
            raise jindex_error("Problem in Feature %s: %d != %d\n" % (self.name(), frame_no - 1, self._frame_no))
        if self._samples is None or self._cur >= self._ttl:
            self._is_end = True
            self._samples = None
            raise jiterator_error("end of samples!")
        n = self._size
        if self._cur + n >= self._ttl:
            if self._pad_zeros:
                self._vector = np.zeros(n, np.float32)
                rem = self._ttl - self._cur
                self._vector[:rem] = self._samples[self._cur:self._cur + rem]
            else:
                self._is_end = True
                self._samples = None
                raise jiterator_error("end of samples!")
        else:
            self._vector = self._samples[self._cur:self._cur + n].copy()
        self._cur += self._shift
        self._frame_no += 1
        return self._vector

    def reset(self):
        self._cur = 0
        VectorFloatFeatureStream.reset(self)


SampleFeature = SampleFeaturePtr
