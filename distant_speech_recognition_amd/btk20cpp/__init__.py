"""The C++ node layer (host/libbtk20hip.so) bound to Python with pybind11: the reference's SWIG class names
(`SampleFeaturePtr`, `OverSampledDFTAnalysisBankPtr`, `SubbandGSCPtr`, `ZelinskiPostFilterPtr`, ...) ARE the C++ nodes here --
`next()` returns a numpy view of the node's own vector (no copy), `for frame in node:` resets and iterates, end of stream is
StopIteration, and `PyVectorComplexFeatureStreamPtr(obj)` / `PyVectorFloatFeatureStreamPtr(obj)` turn any Python object with
size() / __iter__ / next() / reset() into a source node that C++ nodes pull from (reference stream/pyStream.h:25-168).

`distant_speech_recognition_amd.btk20` is the pure-Python mirror of the same classes (ctypes over the C-ABI) with the
virtual-pull protocol for moving look directions; both sit on the same kernels and are tested against the same oracle."""
import importlib.util
import os
import sysconfig

import torch  # noqa: F401  -- first, as in engine.py: torch ships its own libamdhip64.so.7 and one process must run ONE HIP runtime

_HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host")
_PATH = os.path.join(_HOST, "_btk20cpp" + sysconfig.get_config_var("EXT_SUFFIX"))
if not os.path.exists(_PATH):
    raise ImportError("%s is not built: run `python -c 'import __graft_entry__ as g; g.build()'` (make -C %s)" % (_PATH, _HOST))
_spec = importlib.util.spec_from_file_location("_btk20cpp", _PATH)
_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

globals().update({k: v for k, v in vars(_mod).items() if not k.startswith("_")})
__all__ = [k for k in vars(_mod) if not k.startswith("_")]
