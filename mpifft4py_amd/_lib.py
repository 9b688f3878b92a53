"""ctypes binding of libmpifft4py_amd.so (the C ABI in include/mpifft4py_amd.h).

The product path has NO fallback: if the shared library is missing or a HIP
call fails, an exception is raised.  Nothing in this package imports numpy.fft,
the oracle, or torch.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t,
                    c_uint64, c_uint8, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmpifft4py_amd.so")

SINGLE, DOUBLE = 0, 1
R2C, C2C = 0, 1
SLAB, PENCIL_X, PENCIL_Y = 0, 1, 2
DEALIAS_NONE, DEALIAS_2_3, DEALIAS_3_2 = 0, 1, 2
UNIQUE_ID_BYTES = 128


class PlanDesc(ctypes.Structure):
    _fields_ = [("n", c_int64 * 3), ("precision", c_int), ("kind", c_int), ("decomp", c_int),
                ("p1", c_int), ("padsize", c_double), ("pipeline", c_int), ("drop_nyquist", c_int),
                ("line2d", c_int), ("comm_cus", c_int), ("complex_pitch", c_int), ("reserved", c_int * 3)]


class MfftError(RuntimeError):
    pass


_SIGNATURES = {
    "mfft_version": ([], c_int),
    "mfft_last_error": ([], c_char_p),
    "mfft_device_count": ([POINTER(c_int)], c_int),
    "mfft_set_device": ([c_int], c_int),
    "mfft_get_device": ([POINTER(c_int)], c_int),
    "mfft_device_name": ([c_char_p, c_size_t], c_int),
    "mfft_device_sync": ([], c_int),
    "mfft_device_pci_bus_id": ([c_int, c_char_p, c_size_t], c_int),
    "mfft_malloc": ([POINTER(c_void_p), c_size_t], c_int),
    "mfft_free": ([c_void_p], c_int),
    "mfft_memset": ([c_void_p, c_int, c_size_t], c_int),
    "mfft_memcpy_h2d": ([c_void_p, c_void_p, c_size_t], c_int),
    "mfft_memcpy_d2h": ([c_void_p, c_void_p, c_size_t], c_int),
    "mfft_memcpy_rows_h2d": ([c_void_p, c_size_t, c_void_p, c_size_t, c_size_t], c_int),
    "mfft_memcpy_rows_d2h": ([c_void_p, c_void_p, c_size_t, c_size_t, c_size_t], c_int),
    "mfft_memcpy_d2d": ([c_void_p, c_void_p, c_size_t], c_int),
    "mfft_fill_uniform": ([c_void_p, c_size_t, c_int, c_uint64], c_int),
    "mfft_comm_create_self": ([POINTER(c_void_p)], c_int),
    "mfft_get_unique_id": ([c_void_p], c_int),
    "mfft_comm_create_rccl": ([c_int, c_int, c_void_p, POINTER(c_void_p)], c_int),
    "mfft_comm_create_local": ([c_int, POINTER(c_int), POINTER(c_void_p)], c_int),
    "mfft_comm_size": ([c_void_p, POINTER(c_int)], c_int),
    "mfft_comm_rank": ([c_void_p, POINTER(c_int)], c_int),
    "mfft_comm_barrier": ([c_void_p], c_int),
    "mfft_comm_selftest": ([c_void_p, c_size_t, c_int], c_int),
    "mfft_comm_set_option": ([c_void_p, c_char_p, c_int64], c_int),
    "mfft_comm_get_option": ([c_void_p, c_char_p, POINTER(c_int64)], c_int),
    "mfft_comm_bcast_host": ([c_void_p, c_void_p, c_size_t, c_int], c_int),
    "mfft_comm_allreduce_sum_host": ([c_void_p, POINTER(c_double), c_int], c_int),
    "mfft_comm_allreduce_max_host": ([c_void_p, POINTER(c_double), c_int], c_int),
    "mfft_comm_abort": ([c_void_p], c_int),
    "mfft_comm_destroy": ([c_void_p], c_int),
    "mfft_plan_create": ([c_void_p, POINTER(PlanDesc), POINTER(c_void_p)], c_int),
    "mfft_plan_destroy": ([c_void_p], c_int),
    "mfft_plan_layout": ([c_void_p] + [POINTER(c_int64)] * 7, c_int),
    "mfft_layout_query": ([POINTER(PlanDesc), c_int, c_int] + [POINTER(c_int64)] * 7, c_int),
    "mfft_layout_complex_pitch": ([POINTER(PlanDesc), c_int, c_int, POINTER(c_int64), POINTER(c_int64)], c_int),
    "mfft_plan_workspace_bytes": ([c_void_p, POINTER(c_size_t)], c_int),
    "mfft_plan_exchange_schedule": ([POINTER(PlanDesc), c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int),
                                     POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t), POINTER(c_size_t),
                                     POINTER(c_size_t)], c_int),
    "mfft_plan_exchange_pieces": ([POINTER(PlanDesc), c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int),
                                   POINTER(c_int), POINTER(c_int), POINTER(c_size_t), POINTER(c_size_t),
                                   POINTER(c_size_t), POINTER(c_size_t)], c_int),
    "mfft_plan_relay_schedule": ([POINTER(PlanDesc), c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int),
                                  POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_size_t),
                                  POINTER(c_size_t)], c_int),
    "mfft_forward": ([c_void_p, c_void_p, c_void_p, c_int], c_int),
    "mfft_backward": ([c_void_p, c_void_p, c_void_p, c_int], c_int),
    "mfft_plan_sync": ([c_void_p], c_int),
    "mfft_plan_set_dealias_mask": ([c_void_p, c_void_p, c_size_t], c_int),
    "mfft_plan_get_info": ([c_void_p, c_char_p, POINTER(c_int64)], c_int),
    "mfft_plan_timing": ([c_void_p, c_int], c_int),
    "mfft_plan_timing_reset": ([c_void_p], c_int),
    "mfft_plan_timing_get": ([c_void_p, c_int, c_void_p, POINTER(c_double), POINTER(c_int64), POINTER(c_double)], c_int),
    "mfft_c2c_axis": ([c_void_p, c_void_p, POINTER(c_int64), c_int, c_int, c_int], c_int),
    "mfft_c2c_strided": ([c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, c_int, c_int], c_int),
    "mfft_r2c_last": ([c_void_p, c_void_p, POINTER(c_int64), c_int], c_int),
    "mfft_c2r_last": ([c_void_p, c_void_p, POINTER(c_int64), c_int], c_int),
    "mfft_nlz_rows": ([c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int], c_int),
    "mfft_slab_pack": ([c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int], c_int),
    "mfft_slab_unpack": ([c_void_p, c_void_p, c_int, c_int64, c_int64, c_int64, c_int], c_int),
    "mfft_dealias_filter": ([c_void_p, c_void_p, c_size_t, c_int], c_int),
    "mfft_length_supported": ([c_int64, c_int], c_int),
    "mfft_length_route": ([c_int64, c_int], c_int),
    "mfft_length_route_precision": ([c_int64, c_int, c_int], c_int),
    "mfft_kernel_name": ([c_int, c_int64, c_int, c_int, c_int, c_void_p, c_size_t], c_int),
    "mfft_nonlinear_cross": ([c_void_p, c_void_p, c_void_p, c_void_p, c_int], c_int),
    "mfft_ew_cross": ([c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int], c_int),
    "mfft_ew_curl_hat": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_int], c_int),
    "mfft_ew_ns_rhs": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int64), c_double, c_int], c_int),
    "mfft_ew_axpbz": ([c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_size_t, c_int], c_int),
    "mfft_ew_ns_rk_stage": ([c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int64),
                             c_double, c_double, c_double, c_int, c_int], c_int),
    "mfft_ew_sumsq": ([c_void_p, c_void_p, c_size_t, c_int, POINTER(c_double)], c_int),
    "mfft_ew_dft_bins": ([c_void_p, c_void_p, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), c_int, c_int,
                          POINTER(c_int64), c_int, POINTER(c_double)], c_int),
    "mfft_timer_create": ([POINTER(c_void_p)], c_int),
    "mfft_timer_start": ([c_void_p], c_int),
    "mfft_timer_stop": ([c_void_p, POINTER(c_float)], c_int),
    "mfft_timer_destroy": ([c_void_p], c_int),
}

# functions whose int result is a count, not a status
_COUNT_RESULT = {"mfft_version", "mfft_length_supported", "mfft_length_route", "mfft_length_route_precision", "mfft_plan_timing_get"}

_lib = None


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MfftError(
            "%s not found: build it with `make -C mpifft4py_amd/csrc` (or python -c "
            "'import __graft_entry__ as g; g.build()').  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here means header and library disagree
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc):
    if rc < 0:
        msg = load().mfft_last_error()
        raise MfftError("libmpifft4py_amd error %d: %s" % (rc, msg.decode() if msg else "?"))
    return rc


def call(name, *args):
    rc = getattr(load(), name)(*args)
    if name in _COUNT_RESULT:
        if rc < 0:
            check(rc)
        return rc
    return check(rc)


def exchange_schedule(N, nranks, rank, decomp, which=0, forward=True, padded=False, precision="double",
                      kind=R2C, p1=0, padsize=1.5):
    """Host-only query of the all-to-all-v schedule (no GPU needed): returns
    dict(peers, scount, sdisp, rcount, rdisp) in bytes."""
    d = PlanDesc()
    for i in range(3):
        d.n[i] = int(N[i])
    d.precision = SINGLE if precision == "single" else DOUBLE
    d.kind, d.decomp, d.p1, d.padsize, d.pipeline = kind, decomp, int(p1 or 0), float(padsize), 1
    mx = 64
    n = c_int(0)
    peers = (c_int * mx)()
    arrs = [(c_size_t * mx)() for _ in range(4)]
    call("mfft_plan_exchange_schedule", ctypes.byref(d), nranks, rank, which, 1 if forward else 0,
         1 if padded else 0, mx, ctypes.byref(n), peers, *arrs)
    k = n.value
    return dict(peers=list(peers[:k]), scount=list(arrs[0][:k]), sdisp=list(arrs[1][:k]),
                rcount=list(arrs[2][:k]), rdisp=list(arrs[3][:k]))


def exchange_pieces(N, nranks, rank, decomp, which, forward, pipeline, precision="double", kind=R2C, p1=0):
    """Host-only query of the pipelined exchange selected by `pipeline` (0 = the default): list of piece schedules
    dict(peers, scount, sdisp, rcount, rdisp), displacements relative to the whole buffers."""
    d = PlanDesc()
    for i in range(3):
        d.n[i] = int(N[i])
    d.precision = SINGLE if precision == "single" else DOUBLE
    d.kind, d.decomp, d.p1, d.padsize, d.pipeline = kind, decomp, int(p1 or 0), 1.5, int(pipeline)
    mx = 64
    out, piece, total = [], 0, 1
    while piece < total:
        n, npieces = c_int(0), c_int(0)
        peers = (c_int * mx)()
        arrs = [(c_size_t * mx)() for _ in range(4)]
        call("mfft_plan_exchange_pieces", ctypes.byref(d), nranks, rank, which, 1 if forward else 0, piece, mx,
             ctypes.byref(npieces), ctypes.byref(n), peers, *arrs)
        total, k = npieces.value, n.value
        out.append(dict(peers=list(peers[:k]), scount=list(arrs[0][:k]), sdisp=list(arrs[1][:k]),
                        rcount=list(arrs[2][:k]), rdisp=list(arrs[3][:k])))
        piece += 1
    return out


def relay_schedule(N, nranks, rank, decomp, which, forward=True, precision="double", kind=R2C, p1=0):
    """Host-only query of what `rank` pulls in the relayed form of exchange `which` (mfft_plan_relay_schedule):
    list of dict(phase, kind, frm, msg_src, msg_dst, msg_off, bytes)."""
    d = PlanDesc()
    for i in range(3):
        d.n[i] = int(N[i])
    d.precision = SINGLE if precision == "single" else DOUBLE
    d.kind, d.decomp, d.p1, d.padsize, d.pipeline = kind, decomp, int(p1 or 0), 1.5, 1
    mx = 1024
    n = c_int(0)
    ints = [(c_int * mx)() for _ in range(5)]
    szs = [(c_size_t * mx)() for _ in range(2)]
    call("mfft_plan_relay_schedule", ctypes.byref(d), nranks, rank, which, 1 if forward else 0, mx, ctypes.byref(n),
         *(ints + szs))
    return [dict(phase=ints[0][i], kind=ints[1][i], frm=ints[2][i], msg_src=ints[3][i], msg_dst=ints[4][i],
                 msg_off=szs[0][i], bytes=szs[1][i]) for i in range(n.value)]


def device_count():
    n = c_int(0)
    rc = load().mfft_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def precision_code(dtype_or_name):
    import numpy as np
    if dtype_or_name in ("single", "double"):
        return SINGLE if dtype_or_name == "single" else DOUBLE
    dt = np.dtype(dtype_or_name)
    if dt in (np.dtype(np.float32), np.dtype(np.complex64)):
        return SINGLE
    if dt in (np.dtype(np.float64), np.dtype(np.complex128)):
        return DOUBLE
    raise TypeError("unsupported dtype %s" % dt)
