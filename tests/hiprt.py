"""minimal ctypes access to the HIP runtime for tests that need buffers resident in HBM (keeps torch out of the
test process: torch bundles its own libamdhip64, and two HIP runtimes in one process do not mix)."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64 not found")
    return _hip


class DeviceBuffer:
    def __init__(self, host: np.ndarray):
        host = np.ascontiguousarray(host)
        p = C.c_void_p()
        assert hip().hipMalloc(C.byref(p), C.c_size_t(host.nbytes)) == 0
        assert hip().hipMemcpy(p, host.ctypes.data_as(C.c_void_p), C.c_size_t(host.nbytes), 1) == 0  # H2D
        assert hip().hipDeviceSynchronize() == 0
        self.ptr = p.value
        self.nbytes = host.nbytes

    def to_host(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        assert hip().hipDeviceSynchronize() == 0
        assert hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), C.c_size_t(out.nbytes), 2) == 0  # D2H
        return out

    def free(self):
        if self.ptr:
            hip().hipFree(C.c_void_p(self.ptr))
            self.ptr = None
