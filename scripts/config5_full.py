"""BASELINE config 5 at FULL size on one GPU: 2048^3 complex64 pencil C2C over 8 ranks (all on this device, exchanging
by device copies; 34 GB per rank -- or, on a box with several GPUs, rank r on GPU r % ndev).  Checks: Parseval
(device-side reductions), the round trip on sampled planes, and -- because a consistent permutation of output bins
would pass both -- 16 output bins PER RANK, drawn at random from that rank's block of the spectrum, against the DFT
DEFINITION evaluated on the device in double precision over the whole input (mfft_ew_dft_bins: every rank sums its
block of u for all P x 16 bins, the partial sums are added on the host).  python scripts/config5_full.py [n] [P]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Pencil_C2C
from mpifft4py_amd import spectral, _lib

import ctypes
_hip = ctypes.CDLL("libamdhip64.so")
_free, _total = ctypes.c_size_t(0), ctypes.c_size_t(0)
_hip.hipMemGetInfo(ctypes.byref(_free), ctypes.byref(_total))
arg = sys.argv[1] if len(sys.argv) > 1 else "2048"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
# 2048^3 complex64: four 8.6 GB buffers per rank x 8 ranks = 275 GB.  "auto" = 2048 whenever they fit (with the ranks
# spread over several GPUs every device holds ceil(P / ndev) of them).
ndev = _lib.device_count()
one = os.environ.get("MFFT_TEST_ONE_DEVICE", "0") not in ("", "0")
devices = [0] * P if (ndev < 2 or one) else [r % ndev for r in range(P)]
_need = 34.4e9 * devices.count(0) * (8.0 / P) + 4e9
n = (2048 if _free.value > _need else 1024) if arg == "auto" else int(arg)
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
    # bin-level check, part 1: every rank's partial sums of ALL ranks' bins over its block of the input
    partial = spectral.dft_bins(F, u, ALL_BINS, [s.start or 0 for s in F.original_local_slice()])
    t0 = time.perf_counter()
    F.fftn(u, fu)
    F.ifftn(fu, u)                        # back into u: four 8.6 GB buffers per rank instead of five
    F.sync(); comm.barrier()
    dt = time.perf_counter() - t0
    e_f = spectral.sumsq(F, fu)
    # part 2: the values the transform put at this rank's 16 bins (read back one element at a time)
    ts = F.transformed_local_slice()
    mine = ALL_BINS[16 * r:16 * r + 16]
    got = np.zeros(16, dtype=np.complex128)
    sh = F.transformed_shape()
    for i, (k0, k1, k2) in enumerate(mine):
        l0, l1, l2 = int(k0 - (ts[0].start or 0)), int(k1 - (ts[1].start or 0)), int(k2 - (ts[2].start or 0))
        assert 0 <= l0 < sh[0] and 0 <= l1 < sh[1] and 0 <= l2 < sh[2]
        got[i] = fu.element((l0 * sh[1] + l1) * sh[2] + l2)
    # the synthetic input is a counter-based function of (seed, flat index): its first planes can be regenerated
    a = DeviceArray.random((2,) + tuple(F.original_shape()[1:]), F.complex, seed=100 + r).get()
    b = u.leading(0, 2).get()
    rt = float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(a.ravel()))
    return dt, e_u, e_f, rt, F.original_shape(), F.transformed_shape(), partial, got


# 16 bins per rank inside that rank's block of the spectrum (x-aligned pencil C2C: transformed (N0, N1/P1, N2/P2),
# rank = c0 + P1 c1 owns y in [c0 N1/P1, ...), z in [c1 N2/P2, ...)) -- drawn once, known to every rank
def _all_bins():
    from mpifft4py_amd import LayoutComm
    rng = np.random.default_rng(20261004)
    out = []
    for r in range(P):
        Fl = Pencil_C2C(N, L, LayoutComm(P, r), "single", alignment="X", allow_single=True)
        ts = Fl.transformed_local_slice()
        for _ in range(16):
            out.append([int(rng.integers(s.start or 0, s.stop)) for s in ts])
    return np.array(out, dtype=np.int64)


ALL_BINS = _all_bins()
g = LocalGroup(P, devices=devices)
res = g.run(body)
g.free()
dt = max(r[0] for r in res)
eu = sum(r[1] for r in res); ef = sum(r[2] for r in res)
parseval = abs(ef / (eu * float(n) ** 3) - 1.0)
rt = max(r[3] for r in res)
print("2048^3-class check: n=%d P=%d  shapes %s -> %s" % (n, P, res[0][4], res[0][5]))
print("pair time (%d ranks on %d GPU(s), exchanges = device copies): %.1f ms" % (P, len(set(devices)), dt * 1e3))
print("Parseval |sum|F|^2 / (N^3 sum|u|^2) - 1| = %.2e   round trip rel-L2 (sampled planes) = %.2e" % (parseval, rt))
want = sum(r[6] for r in res)                         # the DFT definition, summed over the ranks' blocks
got = np.concatenate([r[7] for r in res])
rms = float(n) ** 1.5 * np.sqrt(eu / float(n) ** 3)    # rms magnitude of a bin of this input: sqrt(sum |u|^2)
bin_err = float(np.abs(got - want).max() / rms)
dc = float(np.abs(want).max() / rms)
print("bin check: %d bins (16 per rank) against the DFT definition in fp64 on the device: max |F - DFT| / rms|F| = %.2e "
      "(largest checked bin %.1f rms)" % (len(got), bin_err, dc))
print("CONFIG5_OK" if parseval < 1e-4 and rt < 1e-5 and bin_err < 1e-5 else "CONFIG5_FAIL")
