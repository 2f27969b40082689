"""Helpers shared by the two host layers (the pybind11 C++ nodes and the ctypes mirror) and pybeamformer."""
import numpy as np

_DEVICE = None


def device():
    """The HIP device every node of this process uses (one process per GPU)."""
    global _DEVICE
    if _DEVICE is None:
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("btk20 nodes compute on an MI355X: no HIP device visible (there is no CPU fallback)")
        _DEVICE = torch.device("cuda", torch.cuda.current_device())
    return _DEVICE


def mirror_bins(Yk, M):
    """[K][T] bins 0..M/2 -> [T][M] complex128 with conjugate mirror bins."""
    K, T = Yk.shape
    full = np.empty((T, M), np.complex128)
    full[:, :K] = Yk.T
    full[:, K:] = np.conj(full[:, M // 2 - 1:0:-1])
    return full


class _DeviceArray(object):
    """A device buffer owned by a C++ node, described through __cuda_array_interface__ so that torch can alias it."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(int(x) for x in shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


def device_tensor(ptr, shape, owner, typestr="<c8"):
    """torch view (no copy) of `shape` elements at device pointer `ptr`; `owner` (the node) is kept alive by the view's base."""
    import torch
    if any(int(x) == 0 for x in shape):
        return torch.zeros(tuple(int(x) for x in shape), dtype=torch.complex64 if typestr == "<c8" else torch.float32, device=device())
    t = torch.as_tensor(_DeviceArray(ptr, shape, typestr, owner), device=device())
    t._btk_owner = owner
    return t
