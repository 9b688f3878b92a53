"""Dealiased inverse over P virtual ranks on one GPU (developer tool): python scripts/maskprof_ranks.py n P
Times ifftn(fu, u) and ifftn(fu, u, '2/3-rule') of a slab plan (blocking exchange) and prints rank 0's stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Slab_R2C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)

def body(comm):
    F = Slab_R2C(N, L, comm, "double", pipeline=1)
    fu = DeviceArray.random(F.complex_shape(), F.complex, seed=1 + comm.Get_rank())
    u = DeviceArray.empty(F.real_shape(), F.float)
    def run(dealias, reps=6):
        for _ in range(2):
            F.ifftn(fu, u, dealias)
        F.sync(); comm.barrier()
        t = time.perf_counter()
        for _ in range(reps):
            F.ifftn(fu, u, dealias)
        F.sync(); comm.barrier()
        return (time.perf_counter() - t) / reps * 1e3
    a = run(None); b = run("2/3-rule")
    F.enable_timing(True)
    run("2/3-rule", 3)
    return a, b, " ".join("%s=%.3f" % (k, v[0] / max(v[1], 1)) for k, v in sorted(F.stage_times().items()))

g = LocalGroup(P, devices=[0] * P)
res = g.run(body)
g.free()
print("n=%d P=%d ifftn %.3f ms, with the 2/3-rule %.3f ms%s" % (n, P, max(r[0] for r in res), max(r[1] for r in res),
      "  (MFFT_NO_PRUNE)" if os.environ.get("MFFT_NO_PRUNE") else ""))
print(res[0][2])
