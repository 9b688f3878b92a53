"""BASELINE config 5 at FULL size on one GPU: 2048^3 complex64 pencil C2C over 8 ranks (all on this device, exchanging
by device copies; 34 GB per rank).  Size-independent checks: Parseval (device-side reductions) and the round trip on
sampled planes.  python scripts/config5_full.py [n] [P]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Pencil_C2C
from mpifft4py_amd import spectral

import ctypes
_hip = ctypes.CDLL("libamdhip64.so")
_free, _total = ctypes.c_size_t(0), ctypes.c_size_t(0)
_hip.hipMemGetInfo(ctypes.byref(_free), ctypes.byref(_total))
arg = sys.argv[1] if len(sys.argv) > 1 else "2048"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# 2048^3 complex64: four 8.6 GB buffers per rank x 8 ranks = 275 GB.  "auto" = 2048 whenever they fit.
n = (2048 if _free.value > 280e9 else 1024) if arg == "auto" else int(arg)
print("CONFIG5_SIZE n=%d free_hbm_gb=%.1f total_hbm_gb=%.1f" % (n, _free.value / 1e9, _total.value / 1e9), flush=True)
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)

def body(comm):
    # pipeline=1: the exchange pipeline needs a third work buffer per rank, which 8 ranks on ONE device cannot afford
    # at 2048^3 (it is for overlapping real exchanges, of which there are none here)
    F = Pencil_C2C(N, L, comm, "single", alignment="X", pipeline=1 if n >= 2048 else 0)
    r = comm.Get_rank()
    u = DeviceArray.random(F.original_shape(), F.complex, seed=100 + r)
    fu = DeviceArray.empty(F.transformed_shape(), F.complex)
    F.fftn(u, fu); F.sync(); comm.barrier()
    if os.environ.get("CONFIG5_STAGES"):          # per-stage event times of one more pair (not the timed one)
        F.enable_timing(True)
        F.fftn(u, fu); F.ifftn(fu, u); F.sync(); comm.barrier()
        if r == 0:
            print("stages (rank 0, ms): " + " ".join("%s=%.2f" % (k, v[0] / max(v[1], 1)) for k, v in sorted(F.stage_times().items())), flush=True)
        F.enable_timing(False)
    e_u = spectral.sumsq(F, u)            # sum |u|^2 over this rank (device reduction)
    t0 = time.perf_counter()
    F.fftn(u, fu)
    F.ifftn(fu, u)                        # back into u: four 8.6 GB buffers per rank instead of five
    F.sync(); comm.barrier()
    dt = time.perf_counter() - t0
    e_f = spectral.sumsq(F, fu)
    # the synthetic input is a counter-based function of (seed, flat index): its first planes can be regenerated
    a = DeviceArray.random((2,) + tuple(F.original_shape()[1:]), F.complex, seed=100 + r).get()
    b = u.leading(0, 2).get()
    rt = float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(a.ravel()))
    return dt, e_u, e_f, rt, F.original_shape(), F.transformed_shape()

g = LocalGroup(P, devices=[0] * P)
res = g.run(body)
g.free()
dt = max(r[0] for r in res)
eu = sum(r[1] for r in res); ef = sum(r[2] for r in res)
parseval = abs(ef / (eu * float(n) ** 3) - 1.0)
rt = max(r[3] for r in res)
print("2048^3-class check: n=%d P=%d  shapes %s -> %s" % (n, P, res[0][4], res[0][5]))
print("pair time (all %d ranks on ONE GPU, exchanges = device copies): %.1f ms" % (P, dt * 1e3))
print("Parseval |sum|F|^2 / (N^3 sum|u|^2) - 1| = %.2e   round trip rel-L2 (sampled planes) = %.2e" % (parseval, rt))
print("CONFIG5_OK" if parseval < 1e-4 and rt < 1e-5 else "CONFIG5_FAIL")
