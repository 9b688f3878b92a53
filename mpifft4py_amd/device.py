"""DeviceArray: a minimal HBM-resident n-d array (hipMalloc-backed) so that
fftn/ifftn can run without PCIe copies.  It is deliberately tiny: shape, dtype,
pointer, get()/set().  numpy arrays are accepted everywhere too (copied in/out).
"""
import ctypes

import numpy as np

from . import _lib


class DeviceArray(object):
    """`pitch` (round 6, pitched spectra): elements between consecutive rows of the LAST axis in memory (>= shape[-1]; None =
    compact).  shape / size stay logical -- what numpy sees through get() / set() --, nbytes is what is allocated."""

    def __init__(self, shape, dtype, ptr=None, owner=True, pitch=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if len(self.shape) else 1
        self.pitch = None if (pitch is None or not self.shape or int(pitch) == self.shape[-1]) else int(pitch)
        if self.pitch is not None and self.pitch < self.shape[-1]:
            raise ValueError("pitch %d shorter than the last axis %d" % (self.pitch, self.shape[-1]))
        self.nbytes = (self.size if self.pitch is None else self.size // self.shape[-1] * self.pitch) * self.dtype.itemsize
        self._owner = owner and ptr is None
        if ptr is None:
            p = ctypes.c_void_p()
            _lib.call("mfft_malloc", ctypes.byref(p), max(self.nbytes, 16))
            ptr = p.value
        self.ptr = ptr

    # -- construction -------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype, pitch=None):
        return cls(shape, dtype, pitch=pitch)

    @classmethod
    def zeros(cls, shape, dtype, pitch=None):
        a = cls(shape, dtype, pitch=pitch)
        _lib.call("mfft_memset", a.ptr, 0, a.nbytes)
        return a

    @classmethod
    def from_numpy(cls, arr, pitch=None):
        arr = np.ascontiguousarray(arr)
        a = cls(arr.shape, arr.dtype, pitch=pitch)
        a.set(arr)
        return a

    @classmethod
    def random(cls, shape, dtype, seed=0, pitch=None):
        """U[0,1) synthetic data generated on the device (bench / large tests); a pitched array is filled whole, the
        elements between the rows included."""
        a = cls(shape, dtype, pitch=pitch)
        dt = np.dtype(dtype)
        count = a.nbytes // dt.itemsize * (2 if dt.kind == "c" else 1)
        _lib.call("mfft_fill_uniform", a.ptr, count, _lib.precision_code(dt), seed)
        return a

    # -- transfers ----------------------------------------------------------
    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.shape != self.shape:
            raise ValueError("shape mismatch %s vs %s" % (arr.shape, self.shape))
        if self.pitch is not None:
            w = self.shape[-1] * self.dtype.itemsize
            _lib.call("mfft_memcpy_rows_h2d", self.ptr, self.pitch * self.dtype.itemsize, arr.ctypes.data, w, self.size // self.shape[-1])
            return self
        _lib.call("mfft_memcpy_h2d", self.ptr, arr.ctypes.data, self.nbytes)
        return self

    def _d2h(self, host):
        if self.pitch is not None:
            w = self.shape[-1] * self.dtype.itemsize
            _lib.call("mfft_memcpy_rows_d2h", host.ctypes.data, self.ptr, self.pitch * self.dtype.itemsize, w, self.size // self.shape[-1])
        else:
            _lib.call("mfft_memcpy_d2h", host.ctypes.data, self.ptr, self.nbytes)

    def get(self, out=None):
        if out is None:
            out = np.empty(self.shape, dtype=self.dtype)
        if out.shape != self.shape or out.dtype != self.dtype or not out.flags["C_CONTIGUOUS"]:
            tmp = np.empty(self.shape, dtype=self.dtype)
            self._d2h(tmp)
            out[...] = tmp
            return out
        self._d2h(out)
        return out

    def copy_from(self, other):
        assert other.nbytes == self.nbytes and other.pitch == self.pitch
        _lib.call("mfft_memcpy_d2d", self.ptr, other.ptr, self.nbytes)
        return self

    def view(self, shape, dtype=None):
        """Reinterpret the same memory (no ownership)."""
        if self.pitch is not None:
            raise ValueError("a pitched array has no flat views")
        dtype = self.dtype if dtype is None else np.dtype(dtype)
        v = DeviceArray(shape, dtype, ptr=self.ptr, owner=False)
        if v.nbytes > self.nbytes:
            raise ValueError("view larger than the allocation")
        v._base = self
        return v

    def leading(self, i0, i1):
        """Non-owning view of rows [i0, i1) of the leading axis."""
        row = self.nbytes // self.shape[0]
        v = DeviceArray((i1 - i0,) + self.shape[1:], self.dtype, ptr=self.ptr + i0 * row, owner=False,
                        pitch=self.pitch if len(self.shape) > 1 else None)
        v._base = self
        return v

    def component(self, i):
        """View of self[i] (leading axis dropped)."""
        row = self.nbytes // self.shape[0]
        v = DeviceArray(self.shape[1:], self.dtype, ptr=self.ptr + i * row, owner=False, pitch=self.pitch)
        v._base = self
        return v

    def element(self, flat_index):
        """One element (by flat C-order index) copied to the host."""
        i = int(flat_index)
        if not 0 <= i < self.size:
            raise IndexError(flat_index)
        out = np.empty(1, dtype=self.dtype)
        if self.pitch is not None:
            i = i // self.shape[-1] * self.pitch + i % self.shape[-1]
        _lib.call("mfft_memcpy_d2h", out.ctypes.data, self.ptr + i * self.dtype.itemsize, self.dtype.itemsize)
        return out[0]

    def free(self):
        if self._owner and self.ptr:
            _lib.call("mfft_free", self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, ptr=0x%x%s)" % (self.shape, self.dtype, self.ptr or 0,
                                                               "" if self.pitch is None else ", pitch=%d" % self.pitch)


def is_device_array(x):
    return isinstance(x, DeviceArray)
