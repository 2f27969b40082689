"""The C++ node layer (host/libbtk20hip.so) bound to Python with pybind11: the reference's SWIG class names
(`SampleFeaturePtr`, `OverSampledDFTAnalysisBankPtr`, `SubbandGSCPtr`, `ZelinskiPostFilterPtr`, ...) ARE the C++ nodes here --
`next()` returns a numpy view of the node's own vector (no copy), `for frame in node:` resets and iterates, end of stream is
StopIteration, and `PyVectorComplexFeatureStreamPtr(obj)` / `PyVectorFloatFeatureStreamPtr(obj)` turn any Python object with
size() / __iter__ / next() / reset() into a source node that C++ nodes pull from (reference stream/pyStream.h:25-168).

`distant_speech_recognition_amd.btk20` (and the top-level `btk20`) are the reference's import names for THIS module: there is one
host layer, the C++ nodes; the extension must be built (`make -C distant_speech_recognition_amd/host`, done by
`__graft_entry__.build()`), importing it without the built file raises ImportError with that instruction.
End of stream: the binding raises StopIteration itself (jexception.i:20-29 maps jiterator_error there); `jiterator_error` is
exported as that class, so `except jiterator_error` works but `except j_error` does not see the end of a stream."""
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

# ---- the part of the module surface that is not a C++ class
from .._hostutil import device, device_tensor  # noqa: E402

j_error = _mod.j_error
SSPEED = 343740.0                              # beamformer/beamformer.h:26
TYPE_ZELINSKI1_REAL, TYPE_ZELINSKI1_ABS, TYPE_APAB, TYPE_ZELINSKI2, NO_USE_POST_FILTER = 0x01, 0x02, 0x04, 0x08, 0x00   # postfilter.h:18-24


class jarithmetic_error(j_error, ArithmeticError):
    pass


class jinitialization_error(j_error):
    pass


class jkey_error(j_error, KeyError):
    pass


class jparse_error(j_error):
    pass


class jtype_error(j_error, TypeError):
    pass


jiterator_error = StopIteration               # jexception.i:20-29 maps it there; the binding raises StopIteration directly


def calc_all_delays(x, y, z, mpos):
    """calc_all_delays (beamformer.cc:1172-1189)."""
    import numpy as np
    mpos = np.asarray(mpos, np.float64)
    d = np.sqrt(np.sum(mpos[:, :3] ** 2, axis=1)) / SSPEED
    return d - d[len(d) // 2]


def _device_snapshots(self):
    """X complex64 [1][K][N][T] on the device, a torch view of the node's own block (no copy)."""
    ptr, K, N, T = self._device_snapshots_info()
    return device_tensor(ptr, (1, K, N, T), self)


_mod.SubbandBeamformer.device_snapshots = _device_snapshots

def _set_noise_spatial_spectral_matrices(self, R):
    """All bins at once from a [K][N][N] array or device tensor: set_noise_spatial_spectral_matrix per bin (beamformer.h:331)."""
    import numpy as np
    Rh = R.detach().cpu().numpy() if hasattr(R, "detach") else np.asarray(R)
    for k in range(Rh.shape[0]):
        self.set_noise_spatial_spectral_matrix(k, np.ascontiguousarray(Rh[k]).astype(np.complex128))


_mod.SubbandMVDRPtr.set_noise_spatial_spectral_matrices = _set_noise_spatial_spectral_matrices


# ---- keyword names and legacy aliases of the SWIG interface.  _signatures.py is GENERATED from the reference's .i files (the
# names a script written against the SWIG module may use); _signatures_alt.py holds the spellings earlier versions of this
# package used.  A call with keywords is matched against those tables in that order; keywords none of them knows go to the
# pybind11 callable unchanged (its own py::arg names), so nothing that worked positionally or with the binding's names breaks.
def _with_names(f, tables, what):
    def call(self, *args, **kw):
        if not kw:
            return f(self, *args)
        for params in tables:
            names = [p for p, _ in params]
            if not all(k in names for k in kw) or any(k in names[:len(args)] for k in kw):
                continue
            a, left, ok = list(args), dict(kw), True
            for p, dflt in params[len(a):]:
                if p in left:
                    a.append(left.pop(p))
                elif not left:
                    break
                elif isinstance(dflt, str) and dflt == "required" or dflt is Ellipsis:
                    ok = False                      # a gap the table cannot fill: let the binding decide
                    break
                else:
                    a.append(dflt)
            if ok and not left:
                return f(self, *a)
        return f(self, *args, **kw)
    call.__name__ = getattr(f, "__name__", what)
    call.__doc__ = getattr(f, "__doc__", None)
    return call


def _apply_signatures():
    from . import _signatures as S, _signatures_alt as A

    def alt(params):        # the older tables mark "required" with None
        return [(p, "required" if d is None else d) for p, d in params]
    for cname, cls in vars(_mod).items():
        if not isinstance(cls, type):
            continue
        mro = [c.__name__ for c in cls.__mro__]
        for mname in list(vars(cls)):
            if mname.startswith("__") or not callable(vars(cls)[mname]):
                continue
            tables = [S.METHODS[c][mname] for c in mro if mname in S.METHODS.get(c, {})]
            old = A.CLASS_METHOD_KWARGS.get(cname, {}).get(mname) or A.METHOD_KWARGS.get(mname)
            if old:
                tables.append(alt(old))
            if tables:
                setattr(cls, mname, _with_names(vars(cls)[mname], tables, "%s.%s" % (cname, mname)))
        tables = ([S.CTORS[cname]] if cname in S.CTORS else []) + ([alt(A.CTOR_KWARGS[cname])] if cname in A.CTOR_KWARGS else [])
        if tables:
            cls.__init__ = _with_names(cls.__init__, tables, cname)
    for cname, table in A.ALIASES.items():
        cls = getattr(_mod, cname)
        for alias, target in table.items():
            if not hasattr(cls, alias):
                setattr(cls, alias, getattr(cls, target))


_apply_signatures()

# the reference exports every class under both names (X and XPtr, modulated.i:124-140 etc.)
SampleFeature = _mod.SampleFeaturePtr
OverSampledDFTAnalysisBank, OverSampledDFTSynthesisBank = _mod.OverSampledDFTAnalysisBankPtr, _mod.OverSampledDFTSynthesisBankPtr
SubbandDS, SubbandGSC, SubbandGSCRLS = _mod.SubbandDSPtr, _mod.SubbandGSCPtr, _mod.SubbandGSCRLSPtr
SubbandMVDR, SubbandMVDRGSC = _mod.SubbandMVDRPtr, _mod.SubbandMVDRGSCPtr
ZelinskiPostFilter, McCowanPostFilter, LefkimmiatisPostFilter = _mod.ZelinskiPostFilterPtr, _mod.McCowanPostFilterPtr, _mod.LefkimmiatisPostFilterPtr

__all__ += ["device", "SSPEED", "TYPE_ZELINSKI1_REAL", "TYPE_ZELINSKI1_ABS", "TYPE_APAB", "TYPE_ZELINSKI2", "NO_USE_POST_FILTER",
            "jarithmetic_error", "jinitialization_error", "jkey_error", "jparse_error", "jtype_error", "jiterator_error", "calc_all_delays",
            "SampleFeature", "OverSampledDFTAnalysisBank", "OverSampledDFTSynthesisBank", "SubbandDS", "SubbandGSC", "SubbandGSCRLS",
            "SubbandMVDR", "SubbandMVDRGSC", "ZelinskiPostFilter", "McCowanPostFilter", "LefkimmiatisPostFilter"]
