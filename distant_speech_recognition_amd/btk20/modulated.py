"""btk20.modulated (modulated/modulated.i:124-140): the names of that reference module, resolved to the C++ node layer
(distant_speech_recognition_amd.btk20cpp = host/libbtk20hip.so bound with pybind11)."""
from ..btk20cpp import (  # noqa: F401
    OverSampledDFTAnalysisBankPtr, OverSampledDFTSynthesisBankPtr, OverSampledDFTAnalysisBank,
    OverSampledDFTSynthesisBank,
)

__all__ = ['OverSampledDFTAnalysisBankPtr', 'OverSampledDFTSynthesisBankPtr', 'OverSampledDFTAnalysisBank', 'OverSampledDFTSynthesisBank']
