"""One rank of scripts/overlap_trace.sh: a slab R2C plan of an N^3 cube over WORLD_SIZE ranks with the given exchange
pipeline, W untimed + K timed forward+inverse pairs.  Started once per rank, each under its own rocprofv3."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpifft4py_amd import DeviceArray, Slab_R2C, from_env  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--pipeline", type=int, default=4)
ap.add_argument("--pull", type=int, default=1)
ap.add_argument("--comm-cus", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=2)
a = ap.parse_args()
comm = from_env(None, transport="ipc")
if comm.get_option("ipc_pull") >= 0:
    comm.set_option("ipc_pull", a.pull)
n = a.size
F = Slab_R2C(np.array([n, n, n]), np.array([2 * np.pi] * 3), comm, "double", pipeline=a.pipeline, comm_cus=a.comm_cus)
u = DeviceArray.random(F.real_shape(), F.float, seed=1 + comm.Get_rank())
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
for _ in range(a.warmup):
    F.fftn(u, fu)
    F.ifftn(fu, u2)
F.sync()
comm.barrier()
t0 = time.perf_counter()
for _ in range(a.steps):
    F.fftn(u, fu)
    F.ifftn(fu, u2)
F.sync()
comm.barrier()
dt = (time.perf_counter() - t0) / a.steps
k = min(2, F.real_shape()[0])
x, y = u.leading(0, k).get(), u2.leading(0, k).get()
err = float(np.linalg.norm((x - y).ravel()) / np.linalg.norm(x.ravel()))
print("rank %d: %.3f ms per pair, round trip %.2e" % (comm.Get_rank(), 1e3 * dt, err), flush=True)
assert err < 1e-10
