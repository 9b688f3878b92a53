"""Worker of tests/test_gpu_multiprocess.py::test_process_per_rank_full_size: one PROCESS per rank over the shipped IPC
transport (or the mock RCCL) at the sizes the BASELINE configs name -- 512^3 and 1024^3 fp64 -- every exchange pipeline
flavour of the slab plan and both pencils, against the host's pocketfft (scipy.fft, MP_WORKERS threads) of the same
cube.  (tests/mp_worker.py covers every mode at [32, 64, 128].)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mpifft4py_amd import DeviceArray, Pencil_R2C, Slab_R2C, from_env  # noqa: E402


def rel(x, r):
    return float(np.linalg.norm((x - r).ravel()) / np.linalg.norm(r.ravel()))


def main():
    import faulthandler
    import scipy.fft as sfft
    faulthandler.dump_traceback_later(int(os.environ.get("MP_WORKER_DUMP_AFTER", "200")), exit=True)
    comm = from_env()
    rank, P = comm.Get_rank(), comm.Get_size()
    n = int(os.environ.get("MP_N", "512"))
    workers = int(os.environ.get("MP_WORKERS", "8"))
    N = np.array([n] * 3)
    L = np.array([2 * np.pi] * 3)
    A = np.random.default_rng(1).random(tuple(N))
    B2 = sfft.rfftn(A, workers=workers)
    pipelines = (1, 4, -4) if n >= 1024 else (1, 2, 4, 8, -2, -4, -8)
    for pipeline in pipelines:
        F = Slab_R2C(N, L, comm, "double", pipeline=pipeline)
        u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
        fu = DeviceArray.empty(F.complex_shape(), F.complex)
        u2 = DeviceArray.empty(F.real_shape(), F.float)
        for _ in range(2):
            F.fftn(u, fu)
            F.ifftn(fu, u2)
        F.sync()
        e1, e2 = rel(fu.get(), B2[F.complex_local_slice()]), rel(u2.get(), A[F.real_local_slice()])
        assert e1 < 1e-10 and e2 < 1e-10, ("slab", pipeline, e1, e2)
        del F, u, fu, u2
    if P >= 4:
        for align in "XY":
            for pipeline in (1, 4):
                F = Pencil_R2C(N, L, comm, "double", communication="Alltoallw", alignment=align, pipeline=pipeline)
                u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
                fu = DeviceArray.empty(F.complex_shape(), F.complex)
                u2 = DeviceArray.empty(F.real_shape(), F.float)
                F.fftn(u, fu)
                F.ifftn(fu, u2)
                F.sync()
                e1, e2 = rel(fu.get(), B2[F.complex_local_slice()]), rel(u2.get(), A[F.real_local_slice()])
                assert e1 < 1e-10 and e2 < 1e-10, ("pencil", align, pipeline, e1, e2)
                del F, u, fu, u2
    comm.barrier()
    if rank == 0:
        print("BIG_OK world=%d n=%d" % (P, n))


if __name__ == "__main__":
    main()
