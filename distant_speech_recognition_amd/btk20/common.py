"""btk20.common (Exception family of the reference (common/jexception.h:26-161, include/jexception.i:20-85)): the names of that reference module, resolved to the C++ node layer
(distant_speech_recognition_amd.btk20cpp = host/libbtk20hip.so bound with pybind11)."""
from ..btk20cpp import (  # noqa: F401
    j_error, jallocation_error, jarithmetic_error, jconsistency_error, jdimension_error, jindex_error,
    jinitialization_error, jio_error, jiterator_error, jkey_error, jnumeric_error, jparameter_error, jparse_error,
    jtype_error,
)

__all__ = ['j_error', 'jallocation_error', 'jarithmetic_error', 'jconsistency_error', 'jdimension_error', 'jindex_error', 'jinitialization_error', 'jio_error', 'jiterator_error', 'jkey_error', 'jnumeric_error', 'jparameter_error', 'jparse_error', 'jtype_error']
