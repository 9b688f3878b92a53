"""Pipelined slab transform of an N^3 cube over P VIRTUAL ranks inside one process (LocalGroup: one host thread per rank,
exchanges = device copies), for a single rocprofv3 trace without the process time-slicing of P processes on one GPU:
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d out -- python3 scripts/overlap_local.py 8 1024 4
    python scripts/summarize_overlap.py --by-thread out"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Slab_R2C  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
pipeline = int(sys.argv[3]) if len(sys.argv) > 3 else 4
N = np.array([n] * 3)
L = np.array([2 * np.pi] * 3)


def body(comm):
    F = Slab_R2C(N, L, comm, "double", pipeline=pipeline)
    u = DeviceArray.random(F.real_shape(), F.float, seed=3 + comm.Get_rank())
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    for _ in range(2):
        F.fftn(u, fu)
        F.ifftn(fu, u2)
    F.sync()
    comm.barrier()
    t = time.perf_counter()
    for _ in range(3):
        F.fftn(u, fu)
        F.ifftn(fu, u2)
    F.sync()
    comm.barrier()
    return (time.perf_counter() - t) / 3 * 1e3


g = LocalGroup(P, devices=[0] * P)
res = g.run(body)
g.free()
print("overlap_local: %d virtual ranks, %d^3, pipeline %d: %.3f ms per pair" % (P, n, pipeline, max(res)))
