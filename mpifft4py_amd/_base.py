"""Shared machinery of the slab / pencil classes: plan handle, host<->device
marshalling, dealias handling.  All arithmetic happens in libmpifft4py_amd.so."""
import ctypes
import threading
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
        self._mask_whole_fp = None
        self._mask_calls = 0
        self._mask_pending = False
        self._mask_serial = 0
        self._whole_job = None
        self.dealias = np.zeros(0)
        self.work_arrays = work_arrays()
        self._plan = None
        self._stage = {}
        if not hasattr(self, "_comm_cus"):
            self._comm_cus = 0

    # The reference reads `self.dealias` on every '2/3-rule' call (slab.py:237-245, pencil.py:455-462), so a caller may
    # replace the filter, or edit it in place, at any time.  Here the filter lives on the device; what keeps the two equal:
    #   * ASSIGNING to `dealias` marks the device copy stale: the next '2/3-rule' call uploads.
    #   * IN-PLACE edits are found by fingerprints of the host array, taken at upload and compared later
    #     (`dealias_check`):
    #       True / "sampled" (default)  every '2/3-rule' call compares a CHEAP fingerprint -- the whole array up to 256 KiB
    #                 (a 64^3 filter: ~10 us), above that 8192 elements at fixed pseudo-random positions (~40 us at 128^3,
    #                 ~50 us for the 540 MB filter of 1024^3; it must stay far below the transform it guards) -- which sees
    #                 planes, bands and blocks at once (an edit of a fraction f goes unnoticed with probability
    #                 (1 - f)^8192: one plane of 1024: 3e-4) but NOT a single changed element of a large mask.  That blind
    #                 spot is closed by a hash of the WHOLE array every `dealias_full_every`-th call (64), computed by a
    #                 background thread (the hash functions release the GIL) and compared when it is ready, at the latest
    #                 at the next such call: ANY edit counts after at most 2 x dealias_full_every (+ dealias_vote_every
    #                 over several ranks) further calls; `F.dealias = F.dealias` makes it count at once.
    #       "full"    every call hashes the whole array (3 - 15 GB/s: 0.3 - 1 ms per 4 MB): every edit counts at the next
    #                 call, as in the reference.
    #       False     no fingerprints, no vote: only assignment uploads (on every rank).
    #     Nothing is copied to take a fingerprint unless the array is not C-contiguous (a broadcast view, a slice), and
    #     then only for the whole-array hashes.
    #   * The upload is COLLECTIVE for plans over more than one rank (mfft_plan_set_dealias_mask: the ranks agree on the
    #     pruned route), and an edit may touch one rank's block only, so the ranks VOTE on "somebody's filter changed" with
    #     one small host all-reduce on every '2/3-rule' call (default, as the reference's per-call read: an assignment or
    #     an edit on ONE rank counts at that rank's next call, and no rank ever enters the collective upload alone).  Where
    #     the all-reduce stays on the host (the IPC and in-process transports: two shared-memory barriers) it costs ~5 us;
    #     over RCCL it is a copy + ncclAllReduce + stream synchronisation that ends the host's asynchronous run-ahead --
    #     codes that mind may set `dealias_vote_every = 16`: a change found or made locally then waits for the next vote
    #     (results use the previous filter until then), still without a rank uploading by itself.
    default_complex_pitch = None          # what `complex_pitch=None` means (tests set "auto" here to run whole suites pitched)
    dealias_check = True
    dealias_full_every = 64
    dealias_vote_every = None             # None / 1: every '2/3-rule' call; n: every n-th call (opt-in)
    _FULL_HASH_BYTES = 256 << 10
    _SAMPLES = 1 << 13
    _sample_index = {}                    # element count -> sorted flat positions; shape -> the same as an index tuple

    @property
    def dealias(self):
        return self._dealias

    @dealias.setter
    def dealias(self, value):
        self._dealias = value
        self._mask_set = False

    @classmethod
    def _samples(cls, a):
        idx = cls._sample_index.get(a.size)
        if idx is None:
            idx = np.sort(np.random.default_rng(0x6d666674).integers(0, a.size, cls._SAMPLES))
            cls._sample_index[a.size] = idx
        if a.flags.c_contiguous:
            return a.reshape(-1)[idx]                      # a view, then a gather of 8192 elements
        key = ("nd",) + a.shape
        tup = cls._sample_index.get(key)
        if tup is None:
            tup = np.unravel_index(idx, a.shape)
            cls._sample_index[key] = tup
        return a[tup]                                      # any strides (broadcast views too), nothing else is touched

    @staticmethod
    def _whole_hash(a):
        """Hash of every element (no copy for C-contiguous arrays)."""
        a = np.asarray(a)
        if not a.flags.c_contiguous:
            a = np.ascontiguousarray(a)
        return _hash_bytes(a.reshape(-1).view(np.uint8))

    def _mask_fingerprint(self, whole=False):
        a = np.asarray(self._dealias)
        if whole or a.nbytes <= self._FULL_HASH_BYTES:
            return (a.shape, a.dtype.str, "whole", self._whole_hash(a))
        return (a.shape, a.dtype.str, "sampled", _hash_bytes(np.ascontiguousarray(self._samples(a)).view(np.uint8)))

    def _large_mask(self):
        return np.asarray(self._dealias).nbytes > self._FULL_HASH_BYTES

    def _whole_hash_start(self):
        """Hash the whole filter in a background thread; `_whole_hash_poll` compares it with the upload's."""
        a = np.asarray(self._dealias)
        job = {"done": threading.Event(), "value": None, "serial": self._mask_serial}

        def work():
            try:
                job["value"] = (a.shape, a.dtype.str, "whole", self._whole_hash(a))
            finally:
                job["done"].set()
        threading.Thread(target=work, daemon=True).start()
        self._whole_job = job

    def _whole_hash_poll(self, wait):
        """True when a finished background hash differs from the one taken at upload."""
        job = self._whole_job
        if job is None or not (wait or job["done"].is_set()):
            return False
        job["done"].wait()
        self._whole_job = None
        return job["serial"] == self._mask_serial and job["value"] != self._mask_whole_fp

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
        req = getattr(self, "_complex_pitch_req", None) or self.default_complex_pitch
        d.complex_pitch = 0 if not req else (-1 if req in ("auto", "line", True) else int(req))
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
        # pitched spectrum (mfft_plan_desc::complex_pitch): elements between the z rows of this rank's complex array
        pitch, alloc = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.call("mfft_layout_complex_pitch", ctypes.byref(d), self.num_processes, self.rank, ctypes.byref(pitch), ctypes.byref(alloc))
        self.complex_pitch = int(pitch.value) if d.complex_pitch else None        # None: compact rows

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
    def _staging(self, tag, shape, dtype, pitch=None):
        key = (tag, tuple(shape), np.dtype(dtype).str, pitch)
        buf = self._stage.get(key)
        if buf is None:
            buf = DeviceArray(shape, dtype, pitch=pitch)
            self._stage[key] = buf
        return buf

    def _check_pitch(self, arr, pitch):
        want = None if (pitch is None or pitch == arr.shape[-1]) else pitch
        if arr.pitch != want:
            raise ValueError("this object's complex arrays have rows %s elements apart (complex_pitch), the array passed has %s: "
                             "allocate it with FFT.empty_complex()" % (want or arr.shape[-1], arr.pitch or arr.shape[-1]))

    def _dev_in(self, tag, arr, shape, dtype, pitch=None):
        if is_device_array(arr):
            assert arr.shape == tuple(shape), (arr.shape, tuple(shape))
            assert arr.dtype == np.dtype(dtype), (arr.dtype, dtype)
            self._check_pitch(arr, pitch)
            return arr
        a = np.asarray(arr)
        assert a.shape == tuple(shape), (a.shape, tuple(shape))
        return self._staging(tag, shape, dtype, pitch).set(a.astype(dtype, copy=False))

    def _dev_out(self, tag, arr, shape, dtype, pitch=None):
        if is_device_array(arr):
            assert arr.shape == tuple(shape), (arr.shape, tuple(shape))
            assert arr.dtype == np.dtype(dtype), (arr.dtype, dtype)
            self._check_pitch(arr, pitch)
            return arr, None
        a = arr
        assert a.shape == tuple(shape), (a.shape, tuple(shape))
        return self._staging(tag, shape, dtype, pitch), a

    def empty_complex(self, components=None):
        """A DeviceArray for this rank's spectrum -- shape complex_shape(), or (components,) + complex_shape() for a vector
        field -- with the row pitch this object was built with (`complex_pitch`): what fftn writes and ifftn reads.  numpy
        arrays keep the reference's compact shapes (slab.py:102-104) and are converted on the way in and out."""
        shape = tuple(int(x) for x in self.complex_shape())
        if components:
            shape = (int(components),) + shape
        return DeviceArray.empty(shape, self.complex, pitch=self.complex_pitch)

    def _ensure_mask(self):
        if np.shape(self.dealias) == (0,):
            self.dealias = self.get_dealias_filter()
        mode = self.dealias_check
        self._mask_calls += 1
        changed = False
        fp = None
        if mode:
            whole = mode == "full"
            fp = self._mask_fingerprint(whole)
            changed = self._mask_set and fp != self._mask_fp
            if not whole and self._mask_set and self._large_mask() and self.dealias_full_every:
                due = self._mask_calls % int(self.dealias_full_every) == 0
                changed = self._whole_hash_poll(wait=due) or changed
                if due and not changed:
                    self._whole_hash_start()
        stale = not self._mask_set
        if mode and self.num_processes > 1:
            # The upload is collective, so no rank may decide on it alone.  The FIRST upload is every rank's first
            # '2/3-rule' call (SPMD by construction); after it, an assignment on this rank (`stale`) and an in-place edit
            # found here (`changed`) both only raise this rank's hand, and all ranks upload together at the next vote --
            # the reference lets one rank assign `F.dealias` by itself (it reads the attribute per rank, slab.py:237-245).
            every = int(self.dealias_vote_every or 1)
            first = self._mask_serial == 0
            self._mask_pending = self._mask_pending or changed or (stale and not first)
            if first:
                pass
            elif every <= 1 or self._mask_calls % every == 0:
                stale = self.comm.allreduce(1.0 if self._mask_pending else 0.0, MAX) > 0
            else:
                stale = False
        else:
            stale = stale or changed
        if not stale:
            return
        m = np.ascontiguousarray(np.broadcast_to(self.dealias, self.complex_shape()), dtype=np.uint8)
        _lib.call("mfft_plan_set_dealias_mask", self._plan, m.ctypes.data, m.size)
        self._mask_set = True
        self._mask_pending = False
        self._mask_serial += 1
        self._whole_job = None
        if mode:
            self._mask_fp = fp
            self._mask_whole_fp = self._mask_fingerprint(True) if (mode != "full" and self._large_mask()) else None
        else:
            self._mask_fp = self._mask_whole_fp = None

    def _run(self, forward, src, dst, dealias, src_shape, src_dtype, dst_shape, dst_dtype):
        assert dealias in ('3/2-rule', '2/3-rule', 'None', None)
        code = _DEALIAS[dealias]
        if not self._plan:
            raise _lib.MfftError("this object has no device plan (it was built on a LayoutComm, which serves the shape "
                                 "and mesh helpers only): transforms need a communicator with a GPU behind it")
        self.comm.use_device()
        if code == _lib.DEALIAS_2_3 and not forward:
            self._ensure_mask()
        # the spectrum (output of fftn, input of ifftn) carries the object's row pitch, the other side is compact
        d_in = self._dev_in("in%d" % forward, src, src_shape, src_dtype, None if forward else self.complex_pitch)
        d_out, host_out = self._dev_out("out%d" % forward, dst, dst_shape, dst_dtype, self.complex_pitch if forward else None)
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
        "plane_pad", "complex_pitch", "complex_pitch_native", "nonlinear_fused_3_2" / "_none" / "_2_3", "nonlinear_bytes"."""
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
