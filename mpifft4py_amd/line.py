"""Placeholder for mpiFFT4py/line.py (2-D slab transforms, line.py:41-340).

The 2-D class is outside the accelerated hot path; the name
exists so that `from mpifft4py_amd import Line_R2C` -- as the reference's test module does
at import time (tests/test_FFT.py:11) -- works, and fails loudly only when it is used.
"""


class R2C(object):
    def __init__(self, *args, **kwargs):
        raise NotImplementedError(
            "Line_R2C (2-D transforms, mpiFFT4py/line.py) is not part of mpifft4py_amd: only the 3-D slab and "
            "pencil paths are implemented on the GPU")
