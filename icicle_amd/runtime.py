"""Host-side mirror of icicle-runtime (wrappers/rust/icicle-runtime/src/{runtime,stream,memory}.rs)."""
import ctypes
import numpy as np
from ._lib import lib, check, Device


def set_device(device_id: int = 0, device_type: str = "HIP"):
    d = Device(device_type.encode(), device_id)
    check(lib.icicle_set_device(ctypes.byref(d)), f"set_device({device_type},{device_id})")


def get_active_device():
    d = Device()
    check(lib.icicle_get_active_device(ctypes.byref(d)))
    return d.type.decode(), d.id


def get_device_count() -> int:
    n = ctypes.c_int()
    check(lib.icicle_get_device_count(ctypes.byref(n)))
    return n.value


def get_available_memory():
    total, free = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.icicle_get_available_memory(ctypes.byref(total), ctypes.byref(free)))
    return total.value, free.value


def device_synchronize():
    check(lib.icicle_device_synchronize())


class Stream:
    """IcicleStream (wrappers/rust/icicle-runtime/src/stream.rs:21-46)."""

    def __init__(self):
        h = ctypes.c_void_p()
        check(lib.icicle_create_stream(ctypes.byref(h)))
        self.handle = h.value

    def synchronize(self):
        check(lib.icicle_stream_synchronize(self.handle))

    def destroy(self):
        if self.handle is not None:
            check(lib.icicle_destroy_stream(self.handle))
            self.handle = None


class DeviceVec:
    """DeviceVec<T> (wrappers/rust/icicle-runtime/src/memory.rs): owns an icicle_malloc allocation."""

    def __init__(self, nbytes: int):
        p = ctypes.c_void_p()
        check(lib.icicle_malloc(ctypes.byref(p), nbytes), f"icicle_malloc({nbytes})")
        self.ptr = p.value
        self.nbytes = nbytes

    @classmethod
    def from_host(cls, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        v = cls(arr.nbytes)
        check(lib.icicle_copy_to_device(v.ptr, arr.ctypes.data, arr.nbytes))
        return v

    def to_host(self, dtype=np.uint32, shape=None) -> np.ndarray:
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        check(lib.icicle_copy_to_host(out.ctypes.data, self.ptr, self.nbytes))
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr is not None:
            check(lib.icicle_free(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
