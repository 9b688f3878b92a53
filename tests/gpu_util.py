"""Helpers shared by the GPU parity tests (all compute goes through the C ABI)."""
import numpy as np

from oracle import mpifft_oracle as orc

L = np.array([2 * np.pi] * 3)
TOL = {"double": 1e-10, "single": 1e-5}      # north_star: 1e-10 rel-L2 for fp64; 1e-5 for fp32


def have_gpu():
    from mpifft4py_amd import _lib
    try:
        return _lib.device_count() > 0
    except Exception:
        return False


def rdtype(prec):
    return np.float64 if prec == "double" else np.float32


def cdtype(prec):
    return np.complex128 if prec == "double" else np.complex64


def run_ranks(P, fn):
    """Run fn(comm) on P ranks: SelfComm for P == 1, else a LocalGroup whose
    virtual ranks all live on GPU 0 (exercises the complete distributed path:
    pack / exchange / unpack, only the wire is a device copy instead of RCCL)."""
    from mpifft4py_amd import LocalGroup, SelfComm
    if P == 1:
        return [fn(SelfComm(0))]
    g = LocalGroup(P, devices=[0] * P)
    try:
        return g.run(fn)
    finally:
        g.free()


__all__ = ["orc", "L", "TOL", "have_gpu", "rdtype", "cdtype", "run_ranks"]
