"""Counterpart of mpiFFT4py/serialFFT (the reference's backend seam,
serialFFT/__init__.py:1-6): the same free functions, executed by the HIP
kernels of libmpifft4py_amd.so instead of pyFFTW / numpy.fft."""
from .hip_fft import *  # noqa: F401,F403
from .hip_fft import __all__  # noqa: F401
