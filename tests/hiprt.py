"""The HIP runtime the product library already loaded, through ctypes: device buffers and streams
for tests that hand device pointers to the C ABI.  (A second copy of the runtime would see no
device, so the one mapped into the process is looked up.)"""
import ctypes as C

import numpy as np

D2H = 2  # hipMemcpyDeviceToHost


def runtime():
    with open("/proc/self/maps") as f:
        paths = sorted({line.split()[-1] for line in f if "libamdhip64.so" in line})
    assert paths, "librtx.so did not load a HIP runtime"
    hip = C.CDLL(paths[0])
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    hip.hipDeviceSynchronize.argtypes = []
    return hip


class Buffer:
    """Device memory (hipMalloc)."""

    def __init__(self, hip, nbytes):
        self.hip, self.nbytes = hip, nbytes
        self.ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(self.ptr), nbytes) == 0

    def fill(self, byte, stream=None):
        if stream is None:
            assert self.hip.hipMemset(self.ptr, byte, self.nbytes) == 0
        else:
            assert self.hip.hipMemsetAsync(self.ptr, byte, self.nbytes, stream) == 0

    def read(self, shape, dtype, stream=None):
        """Copy out; with a stream, asynchronously on it followed by a wait for that stream only."""
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if stream is None:
            assert self.hip.hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, D2H) == 0
        else:
            assert self.hip.hipMemcpyAsync(out.ctypes.data, self.ptr, out.nbytes, D2H, stream) == 0
            assert self.hip.hipStreamSynchronize(stream) == 0
        return out

    def free(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)
            self.ptr = C.c_void_p()


def new_stream(hip, non_blocking=True):
    s = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(s), 1 if non_blocking else 0) == 0
    return s
