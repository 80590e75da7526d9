"""Test infrastructure: device buffers for the "_dev" entry points without torch -- hipMalloc / hipMemcpy / hipFree of the
HIP runtime that libmimc3_hip.so already loaded (one runtime per process)."""
import ctypes as C

import numpy as np

from mimc3_amd import api  # noqa: F401  (loads libmimc3_hip.so and with it libamdhip64)

_hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
_hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
_hip.hipFree.argtypes = [C.c_void_p]
_hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
_hip.hipDeviceSynchronize.argtypes = []
H2D, D2H = 1, 2


class DevArray:
    """a device copy of a numpy array (or an uninitialised device array of that shape / dtype)"""

    def __init__(self, shape=None, dtype=None, src=None):
        if src is not None:
            src = np.ascontiguousarray(src)
            shape, dtype = src.shape, src.dtype
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 8)
        p = C.c_void_p()
        rc = _hip.hipMalloc(C.byref(p), self.nbytes)
        if rc != 0:
            raise MemoryError(f"hipMalloc({self.nbytes}) -> {rc}")
        self.ptr = p.value
        if src is not None and src.nbytes:
            assert _hip.hipMemcpy(self.ptr, src.ctypes.data, src.nbytes, H2D) == 0

    def numpy(self):
        _hip.hipDeviceSynchronize()
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            assert _hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, D2H) == 0
        return out

    def free(self):
        if self.ptr:
            _hip.hipFree(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
