"""Minimal device-memory helpers over the HIP runtime the library itself links (ctypes), so GPU
tests of the device-pointer entry points need no other framework in the process."""
import ctypes as C

import numpy as np

_rt = None


def rt():
    global _rt
    if _rt is None:
        L = C.CDLL("libamdhip64.so")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _rt = L
    return _rt


class DeviceArray:
    """A device allocation with the shape/dtype of a numpy array."""

    def __init__(self, shape, dtype, fill_byte=None):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        p = C.c_void_p()
        assert rt().hipMalloc(C.byref(p), max(self.nbytes, 1)) == 0
        self.ptr = p.value
        if fill_byte is not None:
            assert rt().hipMemset(self.ptr, fill_byte, self.nbytes) == 0
            assert rt().hipDeviceSynchronize() == 0

    @classmethod
    def from_host(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        assert rt().hipMemcpy(d.ptr, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        return d

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        assert rt().hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def fill(self, byte):
        """Sets every byte; returns when the device has done it."""
        assert rt().hipMemset(self.ptr, byte, self.nbytes) == 0
        assert rt().hipDeviceSynchronize() == 0

    def free(self):
        if self.ptr:
            rt().hipFree(self.ptr)
            self.ptr = None


def free_device_bytes():
    """Free device memory as the runtime reports it (hipMemGetInfo)."""
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert rt().hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value
