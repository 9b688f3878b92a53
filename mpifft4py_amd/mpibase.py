"""Counterpart of mpiFFT4py/mpibase.py: dtypes per precision, host array
helpers and the `work_arrays` cache that the reference exports and its demo
uses (demo/spectral_dns_solver.py:19,42,67-68).

Semantics kept from the reference (mpibase.py:61-131): keys are
(shape, dtype, index[, fillzero]) or (ndarray, index[, fillzero]); an array is
created on first use and zero-filled on every access unless fillzero is False.
"""
import collections.abc

import numpy as np

try:                      # only used to hand the matching MPI datatype back to callers
    from mpi4py import MPI as _MPI
except Exception:         # mpi4py is optional
    _MPI = None


def empty(N, dtype=float, bytes=None):
    return np.empty(N, dtype=dtype)


def zeros(N, dtype=float, bytes=None):
    return np.zeros(N, dtype=dtype)


def _parse_work_key(key):
    """Normalise a work-array key to ((shape, dtype, index), fillzero).

    Two spellings name the same buffer (mpibase.py:64-73): `(shape, dtype, index[, fillzero])` with a tuple shape, and
    `(array, index[, fillzero])`, which borrows shape and dtype from an existing array.  Anything else is a TypeError;
    a non-int index or a non-bool fillzero is an AssertionError, as upstream."""
    if not isinstance(key, tuple) or not key:
        raise TypeError("Wrong type of key for work array")
    head, rest = key[0], key[1:]
    if isinstance(head, np.ndarray):
        spec = (head.shape, head.dtype)
    elif isinstance(head, tuple) and rest:
        spec, rest = (head, rest[0]), rest[1:]
    else:
        raise TypeError("Wrong type of key for work array")
    if len(rest) not in (1, 2):
        raise TypeError("Wrong type of key for work array")
    index = rest[0]
    fill = rest[1] if len(rest) == 2 else True
    assert isinstance(fill, bool)
    assert isinstance(index, int)
    return (tuple(int(n) for n in spec[0]), np.dtype(spec[1]), index), fill


class work_arrays(collections.abc.MutableMapping):
    """Cache of host work arrays keyed by (shape, dtype, index): created zeroed on first use, handed out again on every
    later access -- cleared first unless the key says `fillzero=False` (mpibase.py:61-131; the demo fetches its
    scratch fields through it, demo/spectral_dns_solver.py:67-68).  Device-side work buffers are owned by the plans and
    never pass through here."""

    def __init__(self):
        self._arrays = {}
        self.fillzero = True          # what the most recent key asked for (an attribute of the upstream class)

    @property
    def store(self):
        return self._arrays

    def _slot(self, key):
        ident, self.fillzero = _parse_work_key(key)
        return ident

    def __getitem__(self, key):
        ident = self._slot(key)
        arr = self._arrays.get(ident)
        if arr is None:
            arr = self._arrays[ident] = np.zeros(ident[0], dtype=ident[1])
        elif self.fillzero:
            arr.fill(0)
        return arr

    def __setitem__(self, key, value):
        self._arrays[self._slot(key)] = value

    def __delitem__(self, key):
        del self._arrays[self._slot(key)]

    def __iter__(self):
        return iter(self._arrays)

    def __len__(self):
        return len(self._arrays)

    def values(self):
        raise TypeError("Work arrays not iterable")


def datatypes(precision):
    """(real dtype, complex dtype, MPI datatype or name) for 'single'/'double'
    (mpibase.py:133-137)."""
    assert precision in ("single", "double")
    if precision == "single":
        return (np.float32, np.complex64, _MPI.C_FLOAT_COMPLEX if _MPI else "C_FLOAT_COMPLEX")
    return (np.float64, np.complex128, _MPI.C_DOUBLE_COMPLEX if _MPI else "C_DOUBLE_COMPLEX")
