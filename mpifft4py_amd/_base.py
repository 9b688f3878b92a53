"""Shared machinery of the slab / pencil classes: plan handle, host<->device
marshalling, dealias handling.  All arithmetic happens in libmpifft4py_amd.so."""
import ctypes
from collections import defaultdict

import numpy as np

from . import _lib
from .comm import as_comm
from .device import DeviceArray, is_device_array
from .mpibase import datatypes, work_arrays

_DEALIAS = {None: _lib.DEALIAS_NONE, "None": _lib.DEALIAS_NONE,
            "2/3-rule": _lib.DEALIAS_2_3, "3/2-rule": _lib.DEALIAS_3_2}


def default_planner_effort():
    return defaultdict(lambda: "FFTW_MEASURE")


class DistFFTBase(object):
    """Owns one mfft_plan_t.  Subclasses define the shape API of the reference."""

    def _init_common(self, N, L, comm, precision, communication, padsize, threads, planner_effort):
        assert len(L) == 3
        assert len(N) == 3
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
        self.dealias = np.zeros(0)
        self.work_arrays = work_arrays()
        self._plan = None
        self._stage = {}
        if not hasattr(self, "_comm_cus"):
            self._comm_cus = 0

    # The reference reads `self.dealias` on every '2/3-rule' call (slab.py:237-245, pencil.py:455-462), so a caller may
    # replace the filter at any time.  Here the filter lives on the device: assigning to `dealias` marks the device copy
    # stale and the next '2/3-rule' transform uploads the new one (mfft_plan_set_dealias_mask -- COLLECTIVE for plans
    # over more than one rank: every rank must then assign before its next dealiased call, as every rank of the
    # reference would).  Editing the array IN PLACE is not seen: re-assign it (`F.dealias = F.dealias`).
    @property
    def dealias(self):
        return self._dealias

    @dealias.setter
    def dealias(self, value):
        self._dealias = value
        self._mask_set = False

    def _create_plan(self, kind, decomp, p1=0, pipeline=0, drop_nyquist=False, line2d=False):
        d = _lib.PlanDesc()
        for i in range(3):
            d.n[i] = int(self.N[i])
        d.precision = _lib.precision_code(self.precision)
        d.kind = kind
        d.decomp = decomp
        d.p1 = int(p1 or 0)
        d.padsize = float(self.padsize)
        d.pipeline = int(pipeline)
        d.drop_nyquist = 1 if drop_nyquist else 0
        d.line2d = 1 if line2d else 0
        d.comm_cus = int(getattr(self, "_comm_cus", 0) or 0)
        self.comm.use_device()
        h = ctypes.c_void_p()
        _lib.call("mfft_plan_create", self.comm._handle, ctypes.byref(d), ctypes.byref(h))
        self._plan = h.value
        arrs = [(ctypes.c_int64 * 3)() for _ in range(5)]
        grid = (ctypes.c_int64 * 2)()
        sub = (ctypes.c_int64 * 2)()
        _lib.call("mfft_plan_layout", self._plan, arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], grid, sub)
        self._c_real_shape = tuple(arrs[0])
        self._c_complex_shape = tuple(arrs[1])
        self._c_real_start = tuple(arrs[2])
        self._c_complex_start = tuple(arrs[3])
        self._c_real_shape_padded = tuple(arrs[4])
        self._c_grid = tuple(grid)
        self._c_sub = tuple(sub)

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
        if self._mask_set:
            return
        if np.shape(self.dealias) == (0,):
            self.dealias = self.get_dealias_filter()
        m = np.ascontiguousarray(np.broadcast_to(self.dealias, self.complex_shape()), dtype=np.uint8)
        _lib.call("mfft_plan_set_dealias_mask", self._plan, m.ctypes.data, m.size)
        self._mask_set = True

    def _run(self, forward, src, dst, dealias, src_shape, src_dtype, dst_shape, dst_dtype):
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        code = _DEALIAS[dealias]
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
        """What the plan decided (mfft_plan_get_info): "pruned_route", "comm_cus", "kz_slices", "row_batches", "zfuse"."""
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
