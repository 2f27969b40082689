"""Exception family of the reference (common/jexception.h:26-161) and its Python mapping
(include/jexception.i:20-85): jiterator_error surfaces as StopIteration."""

__all__ = ["j_error", "jallocation_error", "jarithmetic_error", "jconsistency_error", "jdimension_error",
           "jindex_error", "jinitialization_error", "jio_error", "jiterator_error", "jkey_error",
           "jnumeric_error", "jparameter_error", "jparse_error", "jtype_error", "raise_from_code"]


class j_error(Exception):
    code = "JERROR"


class jallocation_error(j_error, MemoryError):
    code = "JALLOCATION"


class jarithmetic_error(j_error, ArithmeticError):
    code = "JARITHMETIC"


class jconsistency_error(j_error):
    code = "JCONSISTENCY"


class jdimension_error(j_error, ValueError):
    code = "JDIMENSION"


class jindex_error(j_error, IndexError):
    code = "JINDEX"


class jinitialization_error(j_error):
    code = "JINITIALIZATION"


class jio_error(j_error, IOError):
    code = "JIO"


class jiterator_error(j_error, StopIteration):
    """'end of samples!' -- mapped to StopIteration by the SWIG layer (jexception.i:20-29)."""
    code = "JITERATOR"


class jkey_error(j_error, KeyError):
    code = "JKEY"


class jnumeric_error(j_error, ArithmeticError):
    code = "JNUMERIC"


class jparameter_error(j_error, ValueError):
    code = "JPARAMETER"


class jparse_error(j_error):
    code = "JPARSE"


class jtype_error(j_error, TypeError):
    code = "JTYPE"


def raise_from_code(err):
    """Translate a C-ABI BtkError into the reference's exception types."""
    from .. import _lib
    table = {_lib.BTK_ERR_DIMENSION: jdimension_error, _lib.BTK_ERR_CONSISTENCY: jconsistency_error,
             _lib.BTK_ERR_ALLOCATION: jallocation_error, _lib.BTK_ERR_PARAMETER: jparameter_error,
             _lib.BTK_ERR_NUMERIC: jnumeric_error}
    raise table.get(getattr(err, "code", None), j_error)(str(err))
