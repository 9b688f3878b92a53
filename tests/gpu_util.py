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


_PEER_PROBE = {}


def peer_probe():
    """On a box with several GPUs: can the in-process transport move data BETWEEN two of them?  One verified
    all-to-all over a LocalGroup whose two ranks live on devices 0 and 1 (hipDeviceEnablePeerAccess + cross-device
    copies and events, csrc/comm.hip).  Returns (ok, text); cached.  tests/test_gpu_zz_multidevice.py asserts it."""
    if "r" not in _PEER_PROBE:
        from mpifft4py_amd import LocalGroup, _lib
        if _lib.device_count() < 2:
            _PEER_PROBE["r"] = (False, "one device")
        else:
            try:
                g = LocalGroup(2, devices=[0, 1])
                try:
                    g.run(lambda comm: comm.selftest(1 << 20, 20000))
                finally:
                    g.free()
                _PEER_PROBE["r"] = (True, "ok")
            except Exception as e:      # noqa: BLE001
                _PEER_PROBE["r"] = (False, "%s: %s" % (type(e).__name__, e))
    return _PEER_PROBE["r"]


def rank_devices(P):
    """Device of every virtual rank: rank r on GPU r % ndev when the box has more than one GPU and the peer probe
    passed (so that the whole -m gpu suite moves its exchanges over xGMI there), else all on GPU 0 -- what a one-GPU
    lease gives.  MFFT_TEST_ONE_DEVICE=1 forces device 0."""
    import os
    from mpifft4py_amd import _lib
    ndev = _lib.device_count()
    if ndev < 2 or os.environ.get("MFFT_TEST_ONE_DEVICE", "0") not in ("", "0") or not peer_probe()[0]:
        return [0] * P
    return [r % ndev for r in range(P)]


def run_ranks(P, fn, devices=None):
    """Run fn(comm) on P ranks: SelfComm for P == 1, else a LocalGroup (exercises the complete distributed path:
    pack / exchange / unpack; the wire is a device copy -- a peer-to-peer one between GPUs where the box has several,
    see rank_devices -- instead of RCCL)."""
    from mpifft4py_amd import LocalGroup, SelfComm
    if P == 1:
        return [fn(SelfComm(0))]
    g = LocalGroup(P, devices=devices if devices is not None else rank_devices(P))
    try:
        return g.run(fn)
    finally:
        g.free()


__all__ = ["orc", "L", "TOL", "have_gpu", "rdtype", "cdtype", "run_ranks", "rank_devices", "peer_probe"]
