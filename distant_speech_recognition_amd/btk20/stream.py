"""FeatureStream protocol (stream/stream.h:16-54, stream/stream.i:145-154, stream/pyStream.h:25-168)."""
import numpy as np

from .._lib import BtkError
from .common import jconsistency_error, jiterator_error, raise_from_code

__all__ = ["FeatureStream", "VectorFloatFeatureStream", "VectorComplexFeatureStream",
           "PyVectorFloatFeatureStreamPtr", "PyVectorComplexFeatureStreamPtr", "device"]

from .._hostutil import device  # noqa: E402,F401  (one device per process, shared with the C++ node layer)


class FeatureStream(object):
    """next(frame_no=-5) returns the node-owned buffer of the next frame; asking again for the
    current frame number returns the same buffer; StopIteration (jiterator_error) at the end."""

    def __init__(self, size, name=""):
        self._size = int(size)
        self._name = name
        self._frame_reset_no = -1
        self._frame_no = -1
        self._is_end = False
        self._vector = None

    def name(self):
        return self._name

    def size(self):
        return self._size

    def frame_no(self):
        return self._frame_no

    def is_end(self):
        return self._is_end

    def current(self):
        if self._frame_no < 0:
            raise jconsistency_error("Frame index (%d) < 0." % self._frame_no)
        return self.next(self._frame_no)

    def reset(self):
        self._frame_no = self._frame_reset_no
        self._is_end = False

    def next(self, frame_no=-5):
        raise NotImplementedError

    # ---- block-served graphs: what a per-frame pull graph does implicitly.
    # A downstream node that computed its whole block at once tells its source how far the reference's frame-by-frame
    # pulling would have advanced (_advance_to), and notices when the source's future output changed (_output_version,
    # e.g. new look direction between two frames): frames already consumed keep their values, later ones are recomputed.
    def _advance_to(self, idx):
        if idx > self._frame_no:
            self._frame_no = idx

    def _output_version(self):
        return 0

    # SWIG: __iter__ = reset(); return self  (stream.i:145-154).  Python 3 adds __next__.
    def __iter__(self):
        self.reset()
        return self

    def __next__(self):
        return self.next()


class VectorFloatFeatureStream(FeatureStream):
    pass


class VectorComplexFeatureStream(FeatureStream):
    pass


class _BlockServedStream(FeatureStream):
    """A node whose frames are computed block-wise on the GPU and served from a host array
    self._frames [T][size]."""

    def __init__(self, size, name=""):
        FeatureStream.__init__(self, size, name)
        self._frames = None

    def _prepare(self):
        raise NotImplementedError

    def _num_frames(self):
        if self._frames is None:
            self._prepare()
        return self._frames.shape[0]

    def next(self, frame_no=-5):
        if frame_no == self._frame_no and self._vector is not None:
            return self._vector
        if self._frames is None:
            self._prepare()
        idx = self._frame_no + 1
        if idx >= self._frames.shape[0]:
            self._is_end = True
            raise jiterator_error("end of samples!")
        self._vector = self._frames[idx]
        self._frame_no = idx
        return self._vector

    def reset(self):
        FeatureStream.reset(self)
        self._frames = None
        self._vector = None


class _PyFeatureStream(FeatureStream):
    """PyFeatureStream (stream/pyStream.h:25-168): any Python object with size()/reset()/__iter__()
    whose iterator yields arrays becomes a source node."""

    def __init__(self, obj, dtype, name="PyFeatureStream"):
        FeatureStream.__init__(self, obj.size(), name)
        self._obj = obj
        self._dtype = dtype
        self._iter = None

    def python_object(self):
        return self._obj

    def device_block(self):
        """Device-resident output block of a GPU-backed Python beamformer (None for plain iterators)."""
        f = getattr(self._obj, "device_block", None)
        return f() if f is not None else None

    def _advance_to(self, idx):
        f = getattr(self._obj, "_advance_to", None)
        if f is not None:
            f(idx)

    def _output_version(self):
        f = getattr(self._obj, "_output_version", None)
        return f() if f is not None else 0

    def next(self, frame_no=-5):
        if frame_no == self._frame_no and self._vector is not None:
            return self._vector
        if self._iter is None:
            self._iter = iter(self._obj)
        try:
            v = next(self._iter)
        except StopIteration:
            self._is_end = True
            raise jiterator_error("end of samples!")
        except RuntimeError as e:
            # PEP 479: a StopIteration that escapes inside a generator body surfaces as
            # RuntimeError(__cause__=StopIteration) -- that, and only that, is an end of stream.
            # Anything else (BtkError from the C-ABI, torch/HIP failures) is an error, as in the
            # reference where every exception but StopIteration becomes jpython_error
            # (stream/pyStream.h:89-111).
            if isinstance(e.__cause__, StopIteration):
                self._is_end = True
                raise jiterator_error("end of samples!")
            if isinstance(e, BtkError):
                raise_from_code(e)
            raise
        v = np.asarray(v, self._dtype)
        if v.shape != (self._size,):
            raise jconsistency_error("Feature size mismatch (%d vs. %d)" % (v.size, self._size))
        self._vector = v
        self._frame_no += 1
        return self._vector

    def reset(self):
        self._obj.reset()
        self._iter = None
        FeatureStream.reset(self)


class PyVectorFloatFeatureStreamPtr(_PyFeatureStream, VectorFloatFeatureStream):
    def __init__(self, obj, name="PyVectorFloatFeatureStream"):
        _PyFeatureStream.__init__(self, obj, np.float32, name)


class PyVectorComplexFeatureStreamPtr(_PyFeatureStream, VectorComplexFeatureStream):
    def __init__(self, obj, name="PyVectorComplexFeatureStream"):
        _PyFeatureStream.__init__(self, obj, np.complex128, name)
