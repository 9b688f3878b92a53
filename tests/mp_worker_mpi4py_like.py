"""Worker of tests/test_gpu_multiprocess.py: the constructors are handed an mpi4py-LIKE communicator (only
Get_rank / Get_size / bcast, what comm.as_comm asks of a real mpi4py one).  The classes wrap it themselves
(comm.from_mpi4py): ranks and the RCCL unique id travel through the object, the data path is the library's."""
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mpifft4py_amd import Pencil_R2C, Slab_R2C  # noqa: E402
from oracle import mpifft_oracle as orc  # noqa: E402


class FileBackedComm(object):
    """Stands in for mpi4py.MPI.COMM_WORLD: a broadcast through a file keyed by the launch."""

    def __init__(self):
        self.rank = int(os.environ["RANK"])
        self.size = int(os.environ["WORLD_SIZE"])
        self.base = "/tmp/mfft_fakempi_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        self.seq = 0

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size

    def bcast(self, obj, root=0):
        path = "%s_%d" % (self.base, self.seq)
        self.seq += 1
        if self.rank == root:
            with open(path + ".tmp", "wb") as f:
                pickle.dump(obj, f)
            os.replace(path + ".tmp", path)
            return obj
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > 120:
                raise RuntimeError("bcast file never appeared")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return pickle.load(f)


def main():
    mpi_like = FileBackedComm()
    N = [32, 64, 128]
    L = np.array([2 * np.pi] * 3)
    A = np.random.default_rng(99).random(N)
    B2 = np.fft.rfftn(A)
    F = Slab_R2C(np.array(N), L, mpi_like, "double")          # wrapped by as_comm -> from_mpi4py -> DistComm
    assert F.num_processes == mpi_like.size and F.rank == mpi_like.rank
    c = F.fftn(np.ascontiguousarray(A[F.real_local_slice()]), np.zeros(F.complex_shape(), dtype=complex))
    b = F.ifftn(c, np.zeros(F.real_shape()))
    assert orc.rel_l2(c, B2[F.complex_local_slice()]) < 1e-10
    assert orc.rel_l2(b, A[F.real_local_slice()]) < 1e-10
    k = F.comm.reduce(float(np.sum(b * b)))                   # the wrapped object offers the mpi4py surface callers use
    if F.rank == 0:
        assert abs(k - float(np.sum(A * A))) < 1e-6 * k
    if mpi_like.size >= 4:
        Fp = Pencil_R2C(np.array(N), L, F.comm, "double", communication="Alltoallw", alignment="X")
        cp = Fp.fftn(np.ascontiguousarray(A[Fp.real_local_slice()]), np.zeros(Fp.complex_shape(), dtype=complex))
        assert orc.rel_l2(cp, B2[Fp.complex_local_slice()]) < 1e-10
    F.comm.barrier()
    if F.rank == 0:
        print("MPI4PY_LIKE_OK", mpi_like.size)
        for f in os.listdir("/tmp"):
            if f.startswith(os.path.basename(mpi_like.base)):
                try:
                    os.remove(os.path.join("/tmp", f))
                except OSError:
                    pass


if __name__ == "__main__":
    main()
