"""`get_subarrays` of the reference (slab.py:199-211, pencil.py:218-246, 971-999) without MPI datatypes.

The reference returns committed `MPI.Datatype.Create_subarray(sizes, subsizes, starts)` objects, one per peer, that
its Alltoallw calls use to pick every peer's box out of the send / receive arrays.  Here the exchange is described by
byte schedules (`mfft_plan_exchange_schedule`), but callers that built their own Alltoallw on top of the reference's
boxes can still ask for them: a `Subarray` carries exactly the three argument lists (and the slices they mean), so
`datatype.Create_subarray(*box.args()).Commit()` rebuilds the MPI object where mpi4py is at hand.
The chunking rules are the reference's: `slab._distribution` spreads a remainder over the first ranks, the pencils'
version (written for power-of-two meshes, pencil.py:76-90) gives a remainder of one to the LAST rank."""


class Subarray(object):
    def __init__(self, sizes, subsizes, starts):
        self.sizes = tuple(int(x) for x in sizes)
        self.subsizes = tuple(int(x) for x in subsizes)
        self.starts = tuple(int(x) for x in starts)

    @property
    def slices(self):
        return tuple(slice(s, s + l) for s, l in zip(self.starts, self.subsizes))

    def args(self):
        return list(self.sizes), list(self.subsizes), list(self.starts)

    def Commit(self):           # the reference commits what it creates; nothing to do here
        return self

    def Free(self):
        pass

    def __eq__(self, other):
        return (self.sizes, self.subsizes, self.starts) == (other.sizes, other.subsizes, other.starts)

    def __repr__(self):
        return "Subarray(sizes=%r, subsizes=%r, starts=%r)" % (self.sizes, self.subsizes, self.starts)


def slab_distribution(N, size):          # slab.py:34-47
    q, r = N // size, N % size
    for i in range(size):
        yield (q + 1, q * i + i) if i < r else (q, q * i + r)


def pencil_subsize(N, size, rank):       # pencil.py:76-78
    return N // size + ((N % size) * (rank == size - 1))


def pencil_distribution(N, size):        # pencil.py:80-90
    q, r = N // size, N % size
    for i in range(size):
        yield (q + 1 if (r == 1 and i + 1 == size) else q, q * i)


def slab_subarrays(N, Np, Nf, P, padsize=1):
    """slab.py:199-211 -> (subarraysA, subarraysB, counts_displs)"""
    A = [Subarray([int(padsize * N[0]), Np[1], Nf], [l, Np[1], Nf], [s, 0, 0]) for l, s in slab_distribution(int(padsize * N[0]), P)]
    B = [Subarray([int(padsize * Np[0]), N[1], Nf], [int(padsize * Np[0]), l, Nf], [0, s, 0]) for l, s in slab_distribution(N[1], P)]
    return A, B, ([1] * P, [0] * P)


def pencil_y_subarrays(N, Nf, P1, P2, c0, c1, padsize=1):
    """pencil.py:218-246 (R2CY)"""
    M, Ny, Q = int(N[0]), int(N[1]), int(Nf)
    m = pencil_subsize(M, P2, c1)
    n = pencil_subsize(int(padsize * Ny), P2, c1)
    q = pencil_subsize(Q, P1, c0)
    s1A = [Subarray([m, int(padsize * Ny), q], [m, l, q], [0, s, 0]) for l, s in pencil_distribution(int(padsize * Ny), P2)]
    s1B = [Subarray([M, n, q], [l, n, q], [s, 0, 0]) for l, s in pencil_distribution(M, P2)]
    m = pencil_subsize(int(padsize * M), P1, c0)
    n = pencil_subsize(int(padsize * Ny), P2, c1)
    s2A = [Subarray([int(padsize * M), n, q], [l, n, q], [s, 0, 0]) for l, s in pencil_distribution(int(padsize * M), P1)]
    s2B = [Subarray([m, n, Q], [m, n, l], [0, 0, s]) for l, s in pencil_distribution(Q, P1)]
    return s1A, s1B, s2A, s2B, ([1] * P2, [0] * P2), ([1] * P1, [0] * P1)


def pencil_x_subarrays(N, Nf, P1, P2, c0, c1, padsize=1):
    """pencil.py:971-999 (R2CX)"""
    M, Ny, Q = int(N[0]), int(N[1]), int(Nf)
    m = pencil_subsize(int(padsize * M), P1, c0)
    n = pencil_subsize(Ny, P1, c0)
    q = pencil_subsize(Q, P2, c1)
    s1A = [Subarray([int(padsize * M), n, q], [l, n, q], [s, 0, 0]) for l, s in pencil_distribution(int(padsize * M), P1)]
    s1B = [Subarray([m, Ny, q], [m, l, q], [0, s, 0]) for l, s in pencil_distribution(Ny, P1)]
    n = pencil_subsize(int(padsize * Ny), P2, c1)
    s2A = [Subarray([m, int(padsize * Ny), q], [m, l, q], [0, s, 0]) for l, s in pencil_distribution(int(padsize * Ny), P2)]
    s2B = [Subarray([m, n, Q], [m, n, l], [0, 0, s]) for l, s in pencil_distribution(Q, P2)]
    return s1A, s1B, s2A, s2B, ([1] * P1, [0] * P1), ([1] * P2, [0] * P2)
