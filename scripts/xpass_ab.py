"""Stage times of an UN-PIPELINED multi-rank pair over P virtual ranks on one GPU, for A/B runs of the x pass behind an
exchange (round 4: out of place into the result, one cache line between x rows where they are 64 KiB multiples apart;
MFFT_XPASS_INPLACE=1 / MFFT_NO_XPAD=1 give the round-3 behaviour).  Developer tool.
python scripts/xpass_ab.py n P slab|slabc2c|pencilX|pencilY|c2cX|c2cY double|single"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Pencil_C2C, Pencil_R2C, Slab_C2C, Slab_R2C

n, P, kind, prec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)


def body(comm):
    if kind == "slab":
        F = Slab_R2C(N, L, comm, prec, pipeline=1)
    elif kind == "slabc2c":
        F = Slab_C2C(N, L, comm, prec, pipeline=1)
    elif kind.startswith("pencil"):
        F = Pencil_R2C(N, L, comm, prec, communication="Alltoallw", alignment=kind[-1], pipeline=1)
    else:
        F = Pencil_C2C(N, L, comm, prec, alignment=kind[-1], pipeline=1)
    cplx = "c2c" in kind
    ishape = F.original_shape() if cplx else F.real_shape()
    oshape = F.transformed_shape() if cplx else F.complex_shape()
    u = DeviceArray.random(ishape, F.complex if cplx else F.float, seed=3 + comm.Get_rank())
    fu = DeviceArray.empty(oshape, F.complex)
    u2 = DeviceArray.empty(ishape, u.dtype)
    F.enable_timing(True)
    for _ in range(2):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync(); comm.barrier(); F.reset_timing()
    t = time.perf_counter()
    for _ in range(5):
        F.fftn(u, fu); F.ifftn(fu, u2)
    F.sync(); comm.barrier()
    dt = (time.perf_counter() - t) / 5
    a = u.leading(0, 1).get(); b = u2.leading(0, 1).get()
    return dt, {k: v[0] / max(v[1], 1) for k, v in F.stage_times().items()}, float(np.linalg.norm((a - b).ravel()) / np.linalg.norm(a.ravel()))


g = LocalGroup(P, devices=[0] * P)
res = g.run(body)
g.free()
env = " ".join("%s=%s" % (k, os.environ[k]) for k in ("MFFT_XPASS_INPLACE", "MFFT_NO_XPAD", "MFFT_FWD_OOP") if k in os.environ)
print("%s %d^3 %s, %d virtual ranks on one GPU, un-pipelined [%s]: %.2f ms per pair (all ranks together), round trip %.1e"
      % (kind, n, prec, P, env or "defaults", max(r[0] for r in res) * 1e3, max(r[2] for r in res)))
print("  rank 0 stages (ms): " + "  ".join("%s %.3f" % kv for kv in sorted(res[0][1].items())))
