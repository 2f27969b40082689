"""btk20.dereverberation (dereverberation/dereverberation.i): the names of that reference module, resolved to the C++ node layer
(distant_speech_recognition_amd.btk20cpp = host/libbtk20hip.so bound with pybind11)."""
from ..btk20cpp import (  # noqa: F401
    MultiChannelWPEDereverberationPtr, MultiChannelWPEDereverberationFeaturePtr,
    SingleChannelWPEDereverberationFeaturePtr,
)

__all__ = ['MultiChannelWPEDereverberationPtr', 'MultiChannelWPEDereverberationFeaturePtr', 'SingleChannelWPEDereverberationFeaturePtr']
