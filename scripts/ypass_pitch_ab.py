"""The y pass of a Nyquist-holding rank with compact and with line-aligned INPUT rows (plan.hip zrow_pitch), output compact
as the plan needs it; alone on the device (developer tool, round 4).  python scripts/ypass_pitch_ab.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, _lib
from xpass_kernel_ab import timed

_lib.load()


def ypass(label, nouter, n, q, dtype):
    es = np.dtype(dtype).itemsize
    line = 128 // es
    qp = -(-q // line) * line
    prec = _lib.precision_code(dtype)
    A = DeviceArray.random((nouter, n, qp), dtype, seed=1)
    B = DeviceArray.empty((nouter, n, qp), dtype)
    res = []
    for name, ip, op in (("compact -> compact", q, q), ("aligned -> compact", qp, q), ("aligned -> aligned", qp, qp),
                         ("compact -> aligned", q, qp)):
        fn = lambda: _lib.call("mfft_c2c_strided", A.ptr, B.ptr, n, nouter, q, n * ip, ip, n * op, op, 0, prec)
        fn()
        mn, _ = timed(fn)
        res.append("%s %.3f ms (%.0f GB/s)" % (name, mn, 2.0 * nouter * n * q * es / mn / 1e6))
    print("%-44s %s" % (label, "; ".join(res)), flush=True)
    A.free(); B.free()


if __name__ == "__main__":
    c128, c64 = np.complex128, np.complex64
    ypass("x-aligned 4x2, 1024^3: (256, 1024, 257)", 256, 1024, 257, c128)
    ypass("y-aligned 4x2, 1024^3: (512, 1024, 129)", 512, 1024, 129, c128)
    ypass("x-aligned 4x2, 1024^3 fp32: (256, 1024, 257)", 256, 1024, 257, c64)
    ypass("reference: (256, 1024, 256)", 256, 1024, 256, c128)
    ypass("slab 8 ranks kz slice: (128, 1024, 129)", 128, 1024, 129, c128)
    ypass("one rank: (1024, 1024, 513)", 1024, 1024, 513, c128)
    # the x pass of the one-rank inverse if the y pass went first and left rows of 520: in place over 1024 * 520 columns
    from xpass_kernel_ab import run
    run("one rank x pass over (1024, 1024*520)", 1024, 1024 * 520, c128)
    run("one rank x pass over (1024, 1024*513)", 1024, 1024 * 513, c128)
