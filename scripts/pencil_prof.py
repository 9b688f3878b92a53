"""Stage times of a pencil R2C pair over P virtual ranks on ONE GPU (exchanges = device copies; developer tool):
python scripts/pencil_prof.py [n] [P] [X|Y] [pipeline] [double|single]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Pencil_R2C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
align = sys.argv[3] if len(sys.argv) > 3 else "X"
pipe = int(sys.argv[4]) if len(sys.argv) > 4 else 1
prec = sys.argv[5] if len(sys.argv) > 5 else "double"
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)


def body(comm):
    F = Pencil_R2C(N, L, comm, prec, communication="Alltoallw", alignment=align, pipeline=pipe)
    u = DeviceArray.random(F.real_shape(), F.float, seed=3 + comm.Get_rank())
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
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
dt = max(r[0] for r in res)
print("pencil %s %d^3 %s, %d virtual ranks on one GPU, pipeline %d, zfuse %s: %.2f ms per pair (all ranks together), round trip %.1e"
      % (align, n, "fp64" if prec == "double" else "fp32", P, pipe, "off" if os.environ.get("MFFT_NO_ZFUSE") else "on", dt * 1e3, max(r[2] for r in res)))
print("  rank 0 stages (ms): " + "  ".join("%s %.3f" % kv for kv in sorted(res[0][1].items())))
