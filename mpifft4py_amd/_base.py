"""Shared machinery of the slab / pencil classes: plan handle, host<->device
marshalling, dealias handling.  All arithmetic happens in libmpifft4py_amd.so."""
import ctypes
import zlib
from collections import defaultdict

import numpy as np

from . import _lib
from .comm import MAX, as_comm
from .device import DeviceArray, is_device_array
from ._mesh import Block
from .mpibase import datatypes, work_arrays

_DEALIAS = {None: _lib.DEALIAS_NONE, "None": _lib.DEALIAS_NONE,
            "2/3-rule": _lib.DEALIAS_2_3, "3/2-rule": _lib.DEALIAS_3_2}


try:                       # a 64-bit hash at memory speed where the module exists; zlib's adler32 (~3 GB/s) otherwise
    from xxhash import xxh3_64_intdigest as _hash_bytes
except ImportError:        # pragma: no cover
    def _hash_bytes(buf):
        return zlib.adler32(buf)


def default_planner_effort():
    return defaultdict(lambda: "FFTW_MEASURE")


class DistFFTBase(object):
    """Owns one mfft_plan_t.  Subclasses define the shape API of the reference."""

    def _init_common(self, N, L, comm, precision, communication, padsize, threads, planner_effort, ndim=3):
        assert len(L) == ndim
        assert len(N) == ndim
        self.N = np.asarray(N, dtype=int)
        self.comm = as_comm(comm)
        self.float, self.complex, self.mpitype = datatypes(precision)
        self.precision = precision
        self.communication = communication
        self.num_processes = self.comm.Get_size()
        self.rank = self.comm.Get_rank()
        self.padsize = padsize
        self.threads = threads                  # accepted, unused (FFTW knob)
        self.planner_effort = planner_effort    # accepted, unused (FFTW knob)
        self._mask_set = False
        self._mask_fp = None
        self.dealias = np.zeros(0)
        self.work_arrays = work_arrays()
        self._plan = None
        self._stage = {}
        if not hasattr(self, "_comm_cus"):
            self._comm_cus = 0

    # The reference reads `self.dealias` on every '2/3-rule' call (slab.py:237-245, pencil.py:455-462), so a caller may
    # replace the filter, or edit it in place, at any time.  Here the filter lives on the device.  Assigning to `dealias`
    # marks the device copy stale.  In-place edits are found by a fingerprint taken at upload and compared on every
    # '2/3-rule' call (`dealias_check`, on by default).  It has to stay far below the transform it guards -- 50 us at 128^3,
    # 0.2 ms at 256^3 -- so it hashes the whole array only up to 256 KiB (a 64^3 filter: ~10 us) and above that 8192
    # elements at fixed pseudo-random positions (~40 us at 128^3, ~0.15 ms for the 540 MB filter of 1024^3; the first
    # version hashed 4 MB / 65 536 samples: 0.4 - 1 ms per call, several times the transform between 128^3 and 256^3).
    # An edit of a fraction f of a large mask goes unnoticed with probability (1 - f)^8192: planes, bands and blocks
    # are always seen (one plane of 1024: 3e-4), a single changed element is not -- re-assign then.
    # The upload is COLLECTIVE for plans over more than one rank (mfft_plan_set_dealias_mask), so the ranks vote on
    # "somebody's filter changed" with one small host all-reduce per '2/3-rule' call; `F.dealias_check = False` (on
    # every rank) switches fingerprint and vote off, and only assignment re-uploads.
    dealias_check = True
    _FULL_HASH_BYTES = 256 << 10
    _SAMPLES = 1 << 13
    _sample_index = {}

    @property
    def dealias(self):
        return self._dealias

    @dealias.setter
    def dealias(self, value):
        self._dealias = value
        self._mask_set = False

    def _mask_fingerprint(self):
        a = np.asarray(self._dealias)
        flat = a.reshape(-1)
        if flat.nbytes > self._FULL_HASH_BYTES:
            idx = DistFFTBase._sample_index.get(flat.size)
            if idx is None:
                idx = np.sort(np.random.default_rng(0x6d666674).integers(0, flat.size, self._SAMPLES))
                DistFFTBase._sample_index = {flat.size: idx}
            flat = flat[idx]
        return (a.shape, a.dtype.str, _hash_bytes(np.ascontiguousarray(flat).view(np.uint8)))

    def _describe(self, kind, decomp, mesh=None, p1=0, pipeline=0, drop_nyquist=False, line2d=False):
        """Fill the plan descriptor and ask the library -- on the host, no device involved (mfft_layout_query) -- for
        this rank's block: local shapes, global start offsets, process grid.  Raises what plan creation would raise for
        an impossible decomposition or a length without a kernel."""
        d = _lib.PlanDesc()
        for i, n in enumerate(self.N if mesh is None else mesh):
            d.n[i] = int(n)
        d.precision = _lib.precision_code(self.precision)
        d.kind = kind
        d.decomp = decomp
        d.p1 = int(p1 or 0)
        d.padsize = float(self.padsize)
        d.pipeline = int(pipeline)
        d.drop_nyquist = 1 if drop_nyquist else 0
        d.line2d = 1 if line2d else 0
        d.comm_cus = int(getattr(self, "_comm_cus", 0) or 0)
        self._desc = d
        arrs = [(ctypes.c_int64 * 3)() for _ in range(5)]
        grid = (ctypes.c_int64 * 2)()
        sub = (ctypes.c_int64 * 2)()
        _lib.call("mfft_layout_query", ctypes.byref(d), self.num_processes, self.rank,
                  arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], grid, sub)
        self._c_real_shape = tuple(arrs[0])
        self._c_complex_shape = tuple(arrs[1])
        self._c_real_start = tuple(arrs[2])
        self._c_complex_start = tuple(arrs[3])
        self._c_real_shape_padded = tuple(arrs[4])
        self._c_grid = tuple(grid)
        self._c_sub = tuple(sub)

    def _block(self, half_axis=2, drop_axes=0):
        """This rank's block for the mesh helpers (`drop_axes`: leading axes of the plan's mesh the class does not
        show, 1 for the 2-D class)."""
        a = drop_axes
        return Block(self.N, self.L,
                     list(zip(self._c_real_start[a:], self._c_real_shape[a:])),
                     list(zip(self._c_complex_start[a:], self._c_complex_shape[a:])),
                     None if half_axis is None else half_axis - a)

    def _create_plan(self):
        """The device side.  A class built on a LayoutComm (no device) stays without a plan: its shape, slice and mesh
        helpers work, its transforms raise."""
        if getattr(self.comm, "_handle", None) is None:
            self._plan = None
            return
        self.comm.use_device()
        h = ctypes.c_void_p()
        _lib.call("mfft_plan_create", self.comm._handle, ctypes.byref(self._desc), ctypes.byref(h))
        self._plan = h.value

    # -- marshalling ----------------------------------------------------------
    def _staging(self, tag, shape, dtype):
        key = (tag, tuple(shape), np.dtype(dtype).str)
        buf = self._stage.get(key)
        if buf is None:
            buf = DeviceArray(shape, dtype)
            self._stage[key] = buf
        return buf

    def _dev_in(self, tag, arr, shape, dtype):
        if is_device_array(arr):
            assert arr.shape == tuple(shape), (arr.shape, tuple(shape))
            assert arr.dtype == np.dtype(dtype), (arr.dtype, dtype)
            return arr
        a = np.asarray(arr)
        assert a.shape == tuple(shape), (a.shape, tuple(shape))
        return self._staging(tag, shape, dtype).set(a.astype(dtype, copy=False))

    def _dev_out(self, tag, arr, shape, dtype):
        if is_device_array(arr):
            assert arr.shape == tuple(shape), (arr.shape, tuple(shape))
            assert arr.dtype == np.dtype(dtype), (arr.dtype, dtype)
            return arr, None
        a = arr
        assert a.shape == tuple(shape), (a.shape, tuple(shape))
        return self._staging(tag, shape, dtype), a

    def _ensure_mask(self):
        if np.shape(self.dealias) == (0,):
            self.dealias = self.get_dealias_filter()
        stale = not self._mask_set
        fp = None
        if self.dealias_check:
            fp = self._mask_fingerprint()
            stale = stale or fp != self._mask_fp
            if self.num_processes > 1:
                stale = self.comm.allreduce(1.0 if stale else 0.0, MAX) > 0
        if not stale:
            return
        m = np.ascontiguousarray(np.broadcast_to(self.dealias, self.complex_shape()), dtype=np.uint8)
        _lib.call("mfft_plan_set_dealias_mask", self._plan, m.ctypes.data, m.size)
        self._mask_set = True
        self._mask_fp = fp

    def _run(self, forward, src, dst, dealias, src_shape, src_dtype, dst_shape, dst_dtype):
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        code = _DEALIAS[dealias]
        if not self._plan:
            raise _lib.MfftError("this object has no device plan (it was built on a LayoutComm, which serves the shape "
                                 "and mesh helpers only): transforms need a communicator with a GPU behind it")
        self.comm.use_device()
        if code == _lib.DEALIAS_2_3 and not forward:
            self._ensure_mask()
        d_in = self._dev_in("in%d" % forward, src, src_shape, src_dtype)
        d_out, host_out = self._dev_out("out%d" % forward, dst, dst_shape, dst_dtype)
        fn = "mfft_forward" if forward else "mfft_backward"
        # forward with the 2/3-rule is the regular transform (slab.py:355-362)
        _lib.call(fn, self._plan, d_in.ptr, d_out.ptr, code if (code != _lib.DEALIAS_2_3 or not forward) else _lib.DEALIAS_NONE)
        if host_out is not None:
            _lib.call("mfft_plan_sync", self._plan)
            d_out.get(host_out)
            return host_out
        return dst

    def plan_info(self, key):
        """What the plan decided (mfft_plan_get_info): "pruned_route", "comm_cus", "kz_slices", "row_batches", "zfuse",
        "plane_pad"."""
        v = ctypes.c_int64(0)
        _lib.call("mfft_plan_get_info", self._plan, key.encode(), ctypes.byref(v))
        return int(v.value)

    def sync(self):
        """Wait for all transforms enqueued on this object's stream."""
        _lib.call("mfft_plan_sync", self._plan)

    # -- instrumentation (bench.py) ---------------------------------------------
    def enable_timing(self, on=True):
        _lib.call("mfft_plan_timing", self._plan, 1 if on else 0)

    def reset_timing(self):
        _lib.call("mfft_plan_timing_reset", self._plan)

    def stage_times(self):
        """{stage: (total_ms, calls, algorithmic_bytes_per_call)}"""
        n = _lib.call("mfft_plan_timing_get", self._plan, 0, None, None, None, None)
        if n <= 0:
            return {}
        names = ((ctypes.c_char * 32) * n)()
        ms = (ctypes.c_double * n)()
        calls = (ctypes.c_int64 * n)()
        ab = (ctypes.c_double * n)()
        _lib.call("mfft_plan_timing_get", self._plan, n, names, ms, calls, ab)
        return {names[i].value.decode(): (ms[i], calls[i], ab[i]) for i in range(n)}

    def workspace_bytes(self):
        b = ctypes.c_size_t(0)
        _lib.call("mfft_plan_workspace_bytes", self._plan, ctypes.byref(b))
        return b.value

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                _lib.call("mfft_plan_destroy", self._plan)
                self._plan = None
        except Exception:
            pass
