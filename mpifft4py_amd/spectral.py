"""Element-wise pieces of a pseudo-spectral Navier-Stokes step on DeviceArrays
(mfft_ew_* of the C ABI).  They are enqueued on the FFT object's own stream, so a
whole RK4 step -- 36 transforms plus these kernels -- runs without a host copy
or a host synchronisation.  Vector fields are DeviceArrays of shape (3,) + local
shape; `Wavenumbers` holds the three 1-D scaled wavenumber vectors on the device."""
import ctypes

import numpy as np

from . import _lib
from .device import DeviceArray


class Wavenumbers(object):
    """Scaled local wavenumbers (2*pi/L * k) of an FFT object's complex layout."""

    def __init__(self, FFT):
        K = FFT.get_local_wavenumbermesh(scaled=True)
        self.shape = tuple(int(s) for s in FFT.complex_shape())
        vecs = [np.ascontiguousarray(np.asarray(K[i], dtype=FFT.float).reshape(-1)) for i in range(3)]
        assert tuple(len(v) for v in vecs) == self.shape
        # pitched spectra (FFT.complex_pitch): the element-wise kernels sweep the rows as they lie in memory, the elements
        # between the rows included (wavenumber 0 there; nothing reads what they compute)
        pitch = getattr(FFT, "complex_pitch", None)
        if pitch:
            vecs[2] = np.concatenate([vecs[2], np.zeros(pitch - len(vecs[2]), dtype=vecs[2].dtype)])
        self.dev = [DeviceArray.from_numpy(v) for v in vecs]
        self.cshape = (ctypes.c_int64 * 3)(self.shape[0], self.shape[1], len(vecs[2]))


def _prec(FFT):
    return _lib.precision_code(FFT.precision)


def cross(FFT, a, b, out):
    """out = a x b for real vector fields of shape (3,) + real shape."""
    assert a.pitch is None and b.pitch is None and out.pitch is None
    n = a.size // 3
    _lib.call("mfft_ew_cross", FFT._plan, a.ptr, b.ptr, out.ptr, n, _prec(FFT))
    return out


def curl_hat(FFT, K, U_hat, out):
    """out = i K x U_hat."""
    _lib.call("mfft_ew_curl_hat", FFT._plan, U_hat.ptr, out.ptr, K.dev[0].ptr, K.dev[1].ptr, K.dev[2].ptr,
              K.cshape, _prec(FFT))
    return out


def ns_rhs(FFT, K, dU, U_hat, nu):
    """Pressure projection and viscous term, in place on dU."""
    _lib.call("mfft_ew_ns_rhs", FFT._plan, dU.ptr, U_hat.ptr, K.dev[0].ptr, K.dev[1].ptr, K.dev[2].ptr,
              K.cshape, float(nu), _prec(FFT))
    return dU


def cross_transform(FFT, a_hat, b_hat, out_hat, dealias=None):
    """out_hat = fftn(ifftn(a_hat) x ifftn(b_hat)), the nonlinear term of a pseudo-spectral step as ONE operation of the
    plan (mfft_nonlinear_cross): what the reference demo composes from six `FFT.ifftn(.., dealias)`, a cross product of
    numpy arrays and three `FFT.fftn(.., dealias)` (demo/spectral_dns_solver.py:53-71).  All three are DeviceArrays of
    shape (3,) + FFT.complex_shape(); out_hat may be a_hat or b_hat.  On one rank (slab) the z stages are one fused kernel
    and no real-space work array exists (`FFT.plan_info("nonlinear_fused_3_2")`); elsewhere the plan composes it."""
    from ._base import _DEALIAS
    assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
    shape = (3,) + tuple(int(s) for s in FFT.complex_shape())
    for x in (a_hat, b_hat, out_hat):
        assert x.shape == shape and x.dtype == np.dtype(FFT.complex), (x.shape, x.dtype, shape)
        FFT._check_pitch(x, FFT.complex_pitch)
    code = _DEALIAS[dealias]
    FFT.comm.use_device()
    if code == _lib.DEALIAS_2_3:
        FFT._ensure_mask()
    _lib.call("mfft_nonlinear_cross", FFT._plan, a_hat.ptr, b_hat.ptr, out_hat.ptr, code)
    return out_hat


def ns_rk_stage(FFT, K, N_hat, U_hat, U_hat0, U_hat1, nu, a_dt, b_dt, last):
    """One Runge-Kutta stage in one sweep (mfft_ew_ns_rk_stage): N_hat holds the nonlinear term on entry and the curl of
    the updated U_hat on return; U_hat1 += a_dt dU; U_hat = U_hat0 + b_dt dU, or (last) U_hat = U_hat0 = U_hat1."""
    _lib.call("mfft_ew_ns_rk_stage", FFT._plan, N_hat.ptr, U_hat.ptr, U_hat0.ptr, U_hat1.ptr, K.dev[0].ptr, K.dev[1].ptr,
              K.dev[2].ptr, K.cshape, float(nu), float(a_dt), float(b_dt), 1 if last else 0, _prec(FFT))
    return U_hat


def axpbz(FFT, y, x, z, alpha, beta):
    """y = alpha * x + beta * z (element-wise over the raw real storage; aliasing allowed)."""
    assert x.nbytes == y.nbytes == z.nbytes and x.pitch == y.pitch == z.pitch
    n_real = y.nbytes // np.dtype(FFT.float).itemsize          # the rows as they lie in memory (pitched arrays: all of it)
    _lib.call("mfft_ew_axpbz", FFT._plan, y.ptr, x.ptr, z.ptr, float(alpha), float(beta), n_real, _prec(FFT))
    return y


def sumsq(FFT, x):
    if x.pitch is not None:
        raise ValueError("sumsq of a pitched array would count the elements between its rows")
    n_real = x.size * (2 if x.dtype.kind == "c" else 1)
    r = ctypes.c_double(0.0)
    _lib.call("mfft_ew_sumsq", FFT._plan, x.ptr, n_real, _prec(FFT), ctypes.byref(r))
    return r.value


def dft_bins(FFT, u, bins, start, is_input=True, inverse=False):
    """Partial sums of DFT bins over this rank's block of a field (mfft_ew_dft_bins): `u` a DeviceArray holding the
    block whose first element sits at the global index `start`; bins an (nb, 3) integer array.  Returns a complex
    vector of length nb; the global bin is the sum over the ranks.  Evaluated by definition in double precision --
    an independent check of a transform too large for any host FFT."""
    bins = np.ascontiguousarray(np.asarray(bins, dtype=np.int64).reshape(-1, 3))
    N = FFT.global_shape() if hasattr(FFT, "global_shape") else FFT.global_real_shape()
    out = np.zeros(len(bins), dtype=np.complex128)
    shape = (ctypes.c_int64 * 3)(*[int(x) for x in u.shape])
    st = (ctypes.c_int64 * 3)(*[int(x) for x in start])
    n3 = (ctypes.c_int64 * 3)(*[int(x) for x in N])
    for i in range(0, len(bins), 16):
        chunk = bins[i:i + 16]
        res = (ctypes.c_double * (2 * len(chunk)))()
        _lib.call("mfft_ew_dft_bins", FFT._plan, u.ptr, 1 if u.dtype.kind == "c" else 0, shape, st, n3,
                  1 if inverse else 0, len(chunk), chunk.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), _prec(FFT), res)
        out[i:i + len(chunk)] = np.array(res[:]).view(np.complex128)
    return out
