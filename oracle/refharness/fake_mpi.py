"""Threads-as-ranks stand-in for ``mpi4py`` so the *unmodified* reference can be
imported and run in the development container (mpi4py is not installed here).

DEV-CONTAINER ONLY test infrastructure: used by check_oracle_vs_reference.py
and make_golden.py to pin oracle/mpifft_oracle.py and to produce the committed
fixtures under tests/golden/.  Nothing here travels into the product path and
nothing in tests/ or bench.py imports the reference at run time.

Only the MPI surface the reference touches is provided:
Get_size/Get_rank/Split/Bcast/reduce/barrier/Alltoall(+IN_PLACE)/Alltoallw/
Scatter/Send/Recv/Sendrecv_replace, _typedict[...].Create_subarray(...).Commit(),
Compute_dims and a few constants.
"""
import builtins
import collections
import collections.abc
import queue
import sys
import threading
import traceback
import types

import numpy as np

_tls = threading.local()


class _Group:
    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n
        self.lock = threading.Lock()
        self._mail = {}
        self.children = {}
        self.split_seq = [0] * n

    def box(self, key):
        with self.lock:                       # dict-miss insertion must be atomic
            if key not in self._mail:
                self._mail[key] = queue.Queue()
            return self._mail[key]

    def abort(self):
        self.barrier.abort()
        for c in self.children.values():
            c.abort()


class Subarray:
    def __init__(self, sizes, subsizes, starts):
        self.sizes = tuple(int(s) for s in sizes)
        self.slices = tuple(slice(int(s), int(s) + int(l))
                            for s, l in zip(starts, subsizes))

    def Commit(self):
        return self

    def Free(self):
        pass


class _Datatype:
    def Create_subarray(self, sizes, subsizes, starts):
        return Subarray(sizes, subsizes, starts)


IN_PLACE = object()
C_FLOAT_COMPLEX = "C_FLOAT_COMPLEX"
C_DOUBLE_COMPLEX = "C_DOUBLE_COMPLEX"
SUM = "SUM"
MIN = "MIN"
MAX = "MAX"


class Comm:
    def __init__(self, group, rank):
        self.g = group
        self.r = rank

    # -- basics ---------------------------------------------------------
    def Get_size(self):
        return self.g.n

    def Get_rank(self):
        return self.r

    def _allgather(self, obj):
        g = self.g
        g.slots[self.r] = obj
        g.barrier.wait()
        out = list(g.slots)
        g.barrier.wait()
        return out

    def barrier(self):
        self.g.barrier.wait()

    Barrier = barrier

    def Split(self, color=0, key=0):
        color = int(color)          # legacy float colour, pencil.py:192
        g = self.g
        seq = g.split_seq[self.r]
        g.split_seq[self.r] += 1
        info = self._allgather((color, key, self.r))
        members = sorted([(k, r) for (c, k, r) in info if c == color])
        idx = [r for (_, r) in members].index(self.r)
        with g.lock:
            ck = (seq, color)
            if ck not in g.children:
                g.children[ck] = _Group(len(members))
            child = g.children[ck]
        return Comm(child, idx)

    # -- collectives ----------------------------------------------------
    @staticmethod
    def _buf(spec):
        return spec[0] if isinstance(spec, (list, tuple)) else spec

    def Bcast(self, buf, root=0):
        arr = self._buf(buf)
        allb = self._allgather(arr)
        if self.r != root:
            arr[...] = allb[root]
        self.g.barrier.wait()

    def bcast(self, obj, root=0):
        return self._allgather(obj)[root]

    def reduce(self, x, op=SUM, root=0):
        vals = self._allgather(x)
        if self.r != root:
            return None
        if op == MIN:
            return min(vals)
        if op == MAX:
            return max(vals)
        out = vals[0]
        for v in vals[1:]:
            out = out + v
        return out

    def allreduce(self, x, op=SUM):
        vals = self._allgather(x)
        out = vals[0]
        for v in vals[1:]:
            out = out + v
        return out

    def Alltoall(self, send, recv):
        P = self.g.n
        rbuf = self._buf(recv)
        sbuf = rbuf.copy() if send is IN_PLACE else self._buf(send)
        assert sbuf.flags["C_CONTIGUOUS"] and rbuf.flags["C_CONTIGUOUS"]
        allb = self._allgather(sbuf)
        n = sbuf.size // P
        rf = rbuf.reshape(-1)
        for j in range(P):
            rf[j * n:(j + 1) * n] = allb[j].reshape(-1)[self.r * n:(self.r + 1) * n]
        self.g.barrier.wait()

    def Alltoallw(self, send, recv):
        P = self.g.n
        sbuf, _, stypes = send
        rbuf, _, rtypes = recv
        allb = self._allgather((sbuf, stypes))
        for j in range(P):
            sb, st = allb[j]
            src = np.ascontiguousarray(sb[st[self.r].slices])
            dst = rbuf[rtypes[j].slices]
            dst[...] = src.reshape(dst.shape)
        self.g.barrier.wait()

    def Scatter(self, send, recv, root=0):
        rbuf = self._buf(recv)
        allb = self._allgather(self._buf(send) if self.r == root else None)
        n = rbuf.size
        rbuf.reshape(-1)[:] = allb[root].reshape(-1)[self.r * n:(self.r + 1) * n]
        self.g.barrier.wait()

    # -- point to point ---------------------------------------------------
    def Send(self, buf, dest=0, tag=0):
        self.g.box((self.r, dest, tag)).put(np.array(self._buf(buf), copy=True))

    def Recv(self, buf, source=0, tag=0):
        data = self.g.box((source, self.r, tag)).get(timeout=60)
        b = self._buf(buf)
        b[...] = data.reshape(b.shape)

    def Sendrecv_replace(self, buf, dest, sendtag=0, source=None, recvtag=0):
        b = self._buf(buf)
        self.g.box((self.r, dest, sendtag)).put(np.array(b, copy=True))
        data = self.g.box((source, self.r, recvtag)).get(timeout=60)
        b[...] = data.reshape(b.shape)


class _Proxy:
    def __init__(self, name):
        object.__setattr__(self, "_name", name)

    def __getattr__(self, attr):
        return getattr(getattr(_tls, object.__getattribute__(self, "_name")), attr)


def _seed_main_thread():
    _tls.world = Comm(_Group(1), 0)
    _tls.self_ = Comm(_Group(1), 0)


def Compute_dims(nprocs, ndims):
    assert ndims == 2
    best = (nprocs, 1)
    a = 1
    while a * a <= nprocs:
        if nprocs % a == 0:
            best = (nprocs // a, a)
        a += 1
    return list(best)


def install():
    """Register fake ``mpi4py`` / ``mpi4py.MPI`` modules and the numpy/py3
    compatibility shims the 2016-era reference needs."""
    MPI = types.ModuleType("mpi4py.MPI")
    MPI.COMM_WORLD = _Proxy("world")
    MPI.COMM_SELF = _Proxy("self_")
    MPI.IN_PLACE = IN_PLACE
    MPI.C_FLOAT_COMPLEX = C_FLOAT_COMPLEX
    MPI.C_DOUBLE_COMPLEX = C_DOUBLE_COMPLEX
    MPI.SUM, MPI.MIN, MPI.MAX = SUM, MIN, MAX
    MPI._typedict = collections.defaultdict(_Datatype)
    MPI.Compute_dims = Compute_dims
    pkg = types.ModuleType("mpi4py")
    pkg.MPI = MPI
    sys.modules["mpi4py"] = pkg
    sys.modules["mpi4py.MPI"] = MPI
    _seed_main_thread()

    # python / numpy compatibility shims (no reference file is edited)
    if not hasattr(collections, "MutableMapping"):
        collections.MutableMapping = collections.abc.MutableMapping
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    builtins.xrange = range

    _mesh = np.meshgrid
    if not getattr(_mesh, "_listified", False):
        def meshgrid(*a, **k):
            return list(_mesh(*a, **k))
        meshgrid._listified = True
        np.meshgrid = meshgrid

    _og = np.ogrid
    if not getattr(_og, "_listified", False):
        class _OGridList:
            _listified = True

            def __getitem__(self, key):
                return list(_og[key])
        np.ogrid = _OGridList()
    return MPI


def run(P, fn, *args):
    """Run fn(rank, *args) on P threads-as-ranks; returns list of results."""
    group = _Group(P)
    results = [None] * P
    errors = []

    def body(r):
        _tls.world = Comm(group, r)
        _tls.self_ = Comm(_Group(1), 0)
        try:
            results[r] = fn(r, *args)
        except threading.BrokenBarrierError:
            pass
        except BaseException:
            errors.append((r, traceback.format_exc()))
            group.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise RuntimeError("rank %d failed:\n%s" % errors[0])
    return results
