"""Per-rank FFTs on the GPU with the signatures of the reference's serial
backends (mpiFFT4py/serialFFT/numpy_fft.py:25-107, pyfftw_fft.py:26-203):

    f(a, b=None, axis|axes, overwrite_input=False, threads=1, planner_effort=...) -> b

`a`/`b` may be numpy arrays (copied to/from HBM) or DeviceArrays of up to three
dimensions.  Multi-axis transforms are chains of the single-axis kernels.
`threads`, `planner_effort` and `overwrite_input` are accepted and ignored
(`a` is never modified).  `dct` (numpy_fft.py:11-22; no file of the reference calls it) is provided
for types 2 and 3 through one complex transform of the same length on the device.
"""
import ctypes

import numpy as np

from .. import _lib
from ..device import DeviceArray, is_device_array

__all__ = ['dct', 'fft', 'ifft', 'fft2', 'ifft2', 'fftn', 'ifftn',
           'rfft', 'irfft', 'rfft2', 'irfft2', 'rfftn', 'irfftn']


def _shape3(shape):
    shape = tuple(int(s) for s in shape)
    if len(shape) > 3:
        raise ValueError("at most 3-D arrays are supported")
    return (1,) * (3 - len(shape)) + shape, 3 - len(shape)


def _to_dev(a, dtype):
    if is_device_array(a):
        if a.dtype != np.dtype(dtype):
            raise TypeError("expected %s device array, got %s" % (np.dtype(dtype), a.dtype))
        return a
    return DeviceArray.from_numpy(np.ascontiguousarray(a, dtype=dtype))


def _finish(res, b):
    if b is None:
        return res.get()
    if is_device_array(b):
        if b is not res:
            b.copy_from(res)
        return b
    b[...] = res.get().reshape(b.shape)
    return b


def _c2c(d_in, shape3, axis3, inverse, out=None):
    out = out if out is not None else DeviceArray(d_in.shape, d_in.dtype)
    s = (ctypes.c_int64 * 3)(*shape3)
    _lib.call("mfft_c2c_axis", d_in.ptr, out.ptr, s, axis3, 1 if inverse else 0, _lib.precision_code(d_in.dtype))
    return out


def _cplx_dtype(a):
    dt = a.dtype if is_device_array(a) else np.asarray(a).dtype
    return np.complex64 if dt in (np.dtype(np.float32), np.dtype(np.complex64)) else np.complex128


def _real_dtype(a):
    return np.float32 if _cplx_dtype(a) == np.complex64 else np.float64


def _c2c_axes(a, b, axes, inverse):
    ct = _cplx_dtype(a)
    d = _to_dev(a, ct)
    shape3, off = _shape3(d.shape)
    ndim = len(d.shape)
    cur = d
    for ax in sorted(set(int(x) % ndim for x in axes), reverse=True):
        dst = DeviceArray(d.shape, ct) if cur is d else cur     # never write into the caller's input
        cur = _c2c(cur, shape3, ax + off, inverse, out=dst)
    return _finish(cur, b)


def fft(a, b=None, axis=0, overwrite_input=False, threads=1, **kw):
    return _c2c_axes(a, b, (axis,), False)


def ifft(a, b=None, axis=0, overwrite_input=False, threads=1, **kw):
    return _c2c_axes(a, b, (axis,), True)


def fft2(a, b=None, axes=(0, 1), overwrite_input=False, threads=1, **kw):
    return _c2c_axes(a, b, axes, False)


def ifft2(a, b=None, axes=(0, 1), overwrite_input=False, threads=1, **kw):
    return _c2c_axes(a, b, axes, True)


def fftn(a, b=None, axes=(0, 1, 2), overwrite_input=False, threads=1, **kw):
    return _c2c_axes(a, b, axes, False)


def ifftn(a, b=None, axes=(0, 1, 2), overwrite_input=False, threads=1, **kw):
    return _c2c_axes(a, b, axes, True)


def _rfft_axes(a, b, axes):
    rt, ct = _real_dtype(a), _cplx_dtype(a)
    d = _to_dev(a, rt)
    ndim = len(d.shape)
    axes = [int(x) % ndim for x in axes]
    if axes[-1] != ndim - 1:
        raise ValueError("the real transform must be along the last axis")
    shape3, off = _shape3(d.shape)
    cshape = d.shape[:-1] + (d.shape[-1] // 2 + 1,)
    out = DeviceArray(cshape, ct)
    _lib.call("mfft_r2c_last", d.ptr, out.ptr, (ctypes.c_int64 * 3)(*shape3), _lib.precision_code(rt))
    cshape3, _ = _shape3(cshape)
    for ax in sorted(set(axes[:-1]), reverse=True):
        _c2c(out, cshape3, ax + off, False, out=out)
    return _finish(out, b)


def _irfft_axes(a, b, axes, n_last=None):
    rt, ct = _real_dtype(a), _cplx_dtype(a)
    d = _to_dev(a, ct)
    ndim = len(d.shape)
    axes = [int(x) % ndim for x in axes]
    if axes[-1] != ndim - 1:
        raise ValueError("the real transform must be along the last axis")
    if n_last is None:
        n_last = b.shape[-1] if b is not None else 2 * (d.shape[-1] - 1)
    if d.shape[-1] != n_last // 2 + 1:
        raise ValueError("last axis of the spectrum must have n//2+1 entries")
    cshape3, off = _shape3(d.shape)
    cur = d
    for ax in sorted(set(axes[:-1]), reverse=True):
        dst = DeviceArray(d.shape, ct) if cur is d else cur
        cur = _c2c(cur, cshape3, ax + off, True, out=dst)
    rshape = d.shape[:-1] + (n_last,)
    rshape3, _ = _shape3(rshape)
    out = DeviceArray(rshape, rt)
    _lib.call("mfft_c2r_last", cur.ptr, out.ptr, (ctypes.c_int64 * 3)(*rshape3), _lib.precision_code(rt))
    return _finish(out, b)


def rfft(a, b=None, axis=-1, overwrite_input=False, threads=1, **kw):
    return _rfft_axes(a, b, (axis,))


def irfft(a, b=None, axis=-1, overwrite_input=False, threads=1, **kw):
    return _irfft_axes(a, b, (axis,))


def rfft2(a, b=None, axes=(0, 1), overwrite_input=False, threads=1, **kw):
    return _rfft_axes(a, b, axes)


def irfft2(a, b=None, axes=(0, 1), overwrite_input=False, threads=1, **kw):
    return _irfft_axes(a, b, axes)


def rfftn(a, b=None, axes=(0, 1, 2), overwrite_input=False, threads=1, **kw):
    return _rfft_axes(a, b, axes)


def irfftn(a, b=None, axes=(0, 1, 2), overwrite_input=False, threads=1, **kw):
    return _irfft_axes(a, b, axes)


def _dct_real(x, type, axis):
    """scipy.fftpack.dct(x, type, axis) (unnormalised) of a real host array, type 2 or 3, through one complex
    transform of the same length on the device (even/odd reordering + a quarter-wave twiddle)."""
    x = np.moveaxis(np.asarray(x), axis, 0)
    n = x.shape[0]
    h = (n + 1) // 2
    k = np.arange(n).reshape((n,) + (1,) * (x.ndim - 1))
    ct = np.complex64 if x.dtype == np.float32 else np.complex128
    if type == 2:
        v = np.empty(x.shape, dtype=ct)
        v[:h] = x[0::2]
        v[h:] = x[1::2][::-1]
        V = fft(v, axis=0)
        y = 2 * np.real(V * np.exp(-1j * np.pi * k / (2 * n)))
    elif type == 3:
        xr = np.zeros_like(x)
        xr[1:] = x[1:][::-1]
        V = ((x - 1j * xr) * np.exp(1j * np.pi * k / (2 * n))).astype(ct)
        v = np.real(ifft(V, axis=0)) * n
        y = np.empty(x.shape, dtype=x.dtype)
        y[0::2] = v[:h]
        y[1::2] = v[h:][::-1]
    else:
        raise NotImplementedError("dct type %r: types 2 and 3 are provided" % (type,))
    return np.moveaxis(y.astype(x.dtype, copy=False), 0, axis)


def dct(a, b=None, type=2, axis=0, **kw):
    """numpy_fft.py:14-22: real and imaginary parts are transformed separately."""
    a = a.get() if is_device_array(a) else np.asarray(a)
    if a.ndim > 3:
        raise ValueError("at most 3-D arrays are supported")
    if np.iscomplexobj(a):
        res = _dct_real(a.real, type, axis) + 1j * _dct_real(a.imag, type, axis)
    else:
        res = _dct_real(a, type, axis)
    if b is None:
        return res
    b[...] = res
    return b
