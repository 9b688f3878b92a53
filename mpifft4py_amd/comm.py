"""Communicators for mpifft4py_amd.

The reference receives an mpi4py communicator in its constructors
(mpiFFT4py/slab.py:67-81, pencil.py:167-195) and only ever calls
Get_size / Get_rank / Split / Alltoall(w) / Bcast / reduce / barrier on it.
Here the device-side exchange is RCCL over xGMI (or peer copies inside one
process); these classes carry the handle of the C-ABI communicator plus the
small host-side mpi4py-like surface that tests and demos use.

    SelfComm()                 one rank, no RCCL                       (COMM_SELF)
    DistComm / from_env()      one process per GPU, RCCL               (COMM_WORLD)
    LocalGroup(P).run(fn)      P ranks = P host threads in ONE process, exchanging
                               with peer-to-peer device copies (single-process
                               multi-GPU, or all ranks on one GPU for testing)
    from_mpi4py(comm)          wrap a real mpi4py communicator (RCCL id is
                               broadcast through it)
"""
import ctypes
import os
import pickle
import threading
import time

import numpy as np

from . import _lib

SUM, MAX = "SUM", "MAX"


class _CommBase(object):
    """mpi4py-like host surface on top of an mfft_comm_t handle."""

    def __init__(self, handle, size, rank, device):
        self._handle = handle
        self._size = size
        self._rank = rank
        self.device = device

    # -- mpi4py surface used by the reference's callers ----------------------
    def Get_size(self):
        return self._size

    def Get_rank(self):
        return self._rank

    def selftest(self, bytes_per_peer=1 << 20, timeout_ms=20000):
        """Collective: one small verified all-to-all; raises if the transport does not move data correctly here."""
        _lib.call("mfft_comm_selftest", self._handle, bytes_per_peer, timeout_ms)

    def set_option(self, key, value):
        """Transport knob (mfft_comm_set_option), e.g. ("ipc_pull", 0 | 1 | 2) on the IPC transport."""
        _lib.call("mfft_comm_set_option", self._handle, key.encode(), int(value))

    def get_option(self, key):
        import ctypes
        v = ctypes.c_int64(-1)
        _lib.call("mfft_comm_get_option", self._handle, key.encode(), ctypes.byref(v))
        return int(v.value)

    def barrier(self):
        _lib.call("mfft_comm_barrier", self._handle)

    Barrier = barrier

    def Bcast(self, buf, root=0):
        arr = buf[0] if isinstance(buf, (list, tuple)) else buf
        if self._size == 1:
            return
        if not arr.flags["C_CONTIGUOUS"]:
            raise ValueError("Bcast needs a C-contiguous buffer")
        _lib.call("mfft_comm_bcast_host", self._handle, arr.ctypes.data, arr.nbytes, root)

    def bcast(self, obj, root=0):
        if self._size == 1:
            return obj
        payload = pickle.dumps(obj) if self._rank == root else b""
        n = np.array([len(payload)], dtype=np.int64)
        self.Bcast(n, root)
        buf = np.frombuffer(payload, dtype=np.uint8).copy() if self._rank == root else np.empty(int(n[0]), np.uint8)
        self.Bcast(buf, root)
        return pickle.loads(buf.tobytes())

    def allreduce(self, x, op=SUM):
        v = np.atleast_1d(np.asarray(x, dtype=np.float64)).copy()
        if self._size > 1:
            fn = "mfft_comm_allreduce_max_host" if op == MAX else "mfft_comm_allreduce_sum_host"
            _lib.call(fn, self._handle, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), v.size)
        return float(v[0]) if np.ndim(x) == 0 else v.reshape(np.shape(x))

    def reduce(self, x, op=SUM, root=0):
        r = self.allreduce(x, op)
        return r if self._rank == root else None

    def use_device(self):
        """Make this rank's GPU current on the calling thread."""
        _lib.call("mfft_set_device", self.device)

    def free(self):
        if self._handle:
            _lib.call("mfft_comm_destroy", self._handle)
            self._handle = None


class SubComm(object):
    """Row / column group of a pencil grid (the reference's comm0 / comm1,
    pencil.py:192-195): only rank and size are ever asked of it."""

    def __init__(self, size, rank):
        self._size, self._rank = size, rank

    def Get_size(self):
        return self._size

    def Get_rank(self):
        return self._rank


class SelfComm(_CommBase):
    def __init__(self, device=None):
        lib = _lib.load()
        if device is None:
            d = ctypes.c_int(0)
            _lib.check(lib.mfft_get_device(ctypes.byref(d)))
            device = d.value
        else:
            _lib.call("mfft_set_device", device)
        h = ctypes.c_void_p()
        _lib.call("mfft_comm_create_self", ctypes.byref(h))
        _CommBase.__init__(self, h.value, 1, 0, device)


class DistComm(_CommBase):
    """One process per GPU; device-side exchange through RCCL."""

    def __init__(self, nranks, rank, unique_id, device):
        _lib.call("mfft_set_device", device)
        h = ctypes.c_void_p()
        uid = (ctypes.c_char * _lib.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
        _lib.call("mfft_comm_create_rccl", nranks, rank, uid, ctypes.byref(h))
        _CommBase.__init__(self, h.value, nranks, rank, device)


def get_unique_id():
    buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    _lib.call("mfft_get_unique_id", buf)
    return buf.raw


_RDV_MAGIC = b"MFFTRDV1"
_rdv_count = 0          # rendezvous of this process so far: every from_env() of a launch gets a file name of its own


def _rendezvous_path():
    """Default rendezvous file: a 0700 directory of this user, one name per (MASTER_PORT, launcher pid, run id)."""
    explicit = os.environ.get("MFFT_RENDEZVOUS_FILE")
    if explicit:
        return explicit, False
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mfft-%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise _lib.MfftError("rendezvous directory %s is not private to this user" % d)
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none").replace(os.sep, "_")
    return os.path.join(d, "uid_%s_%d_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), run, _rdv_count)), True


def _publisher_alive(pid, same_parent):
    """A leftover file of a crashed earlier launch names a dead publisher (or one of another launcher)."""
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    except PermissionError:
        return False                       # somebody else's process: not our rank 0
    if same_parent:
        try:
            with open("/proc/%d/stat" % pid) as f:
                ppid = int(f.read().rsplit(")", 1)[1].split()[1])
            return ppid == os.getppid()
        except (OSError, ValueError, IndexError):
            return False
    return True


def _rdv_timeout():
    """Not longer than the transports' own attach / barrier timeout ($MFFT_LOCAL_TIMEOUT, 180 s): a rank that waits here
    for a publisher that has moved on must give up before the publisher's next collective does."""
    try:
        return max(5.0, float(os.environ.get("MFFT_LOCAL_TIMEOUT", "180")) - 10.0)
    except ValueError:
        return 170.0


_UID_FAILED = b"MFFTFAIL"      # rank 0 could not create the id: the rest of the payload is the error text


def _file_bcast(rank, payload, timeout=None):
    """Single-node rendezvous through a file: rank 0 publishes, the rest poll.  The file carries the publisher's pid;
    readers only accept it while that process is alive (and is a child of the same launcher), so a file left behind by
    a crashed earlier launch with the same port and parent is never mistaken for this launch's id."""
    path, same_parent = _rendezvous_path()
    timeout = _rdv_timeout() if timeout is None else timeout
    if rank == 0:
        try:
            os.unlink(path)                # a leftover of an earlier launch
        except FileNotFoundError:
            pass
        tmp = path + ".tmp%d" % os.getpid()
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(_RDV_MAGIC + os.getpid().to_bytes(8, "little") + payload)
        os.replace(tmp, path)
        return payload, path
    t0 = time.time()
    want = len(_RDV_MAGIC) + 8 + _lib.UNIQUE_ID_BYTES
    while True:
        try:
            with open(path, "rb") as f:
                data = f.read()
            if len(data) == want and data[:8] == _RDV_MAGIC and \
                    _publisher_alive(int.from_bytes(data[8:16], "little"), same_parent):
                return data[16:], path
        except FileNotFoundError:
            pass
        if time.time() - t0 > timeout:
            raise _lib.MfftError("rendezvous file %s did not appear" % path)
        time.sleep(0.02)


def from_env(bcast=None, transport=None):
    """Build the world communicator of a `torch.distributed.run` / mpirun style
    launch from RANK / WORLD_SIZE / LOCAL_RANK.  `bcast(obj_or_None) -> obj`
    broadcasts rank 0's unique id (e.g. over a gloo process group); without
    it a file under /tmp is used (single node).  `transport`: "rccl" (grouped send/recv over RCCL) or "ipc"
    (copy-engine pulls through IPC-mapped work buffers, one node); default: $MFFT_TRANSPORT, else rccl.  Rank 0's
    choice counts: the id it creates names the transport."""
    global _rdv_count
    _rdv_count += 1
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    ndev = max(_lib.device_count(), 1)
    device = local % ndev
    if world == 1:
        return SelfComm(device)
    _lib.call("mfft_set_device", device)
    uid, failure = None, None
    if rank == 0:
        saved = os.environ.get("MFFT_TRANSPORT")
        if transport is not None:
            os.environ["MFFT_TRANSPORT"] = transport
        try:
            uid = get_unique_id()
        except Exception as e:      # noqa: BLE001
            # (librccl missing, bad MFFT_TRANSPORT, ...)  The other ranks are about to wait for the id: tell them, so
            # that every rank leaves THIS rendezvous with the same error instead of timing out in a later one
            failure = e
            msg = ("%s: %s" % (type(e).__name__, e)).encode()[:_lib.UNIQUE_ID_BYTES - len(_UID_FAILED)]
            uid = (_UID_FAILED + msg).ljust(_lib.UNIQUE_ID_BYTES, b"\0")
        finally:
            if transport is not None:
                if saved is None:
                    os.environ.pop("MFFT_TRANSPORT", None)
                else:
                    os.environ["MFFT_TRANSPORT"] = saved
    path = None
    if bcast is not None:
        uid = bcast(uid)
    else:
        uid, path = _file_bcast(rank, uid)
    if failure is not None:
        raise failure
    if bytes(uid[:len(_UID_FAILED)]) == _UID_FAILED:
        raise _lib.MfftError("rank 0 could not create the communicator id: %s"
                             % bytes(uid[len(_UID_FAILED):]).rstrip(b"\0").decode(errors="replace"))
    c = DistComm(world, rank, uid, device)
    c.barrier()
    if path and rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return c


def from_mpi4py(comm):
    """Wrap a real mpi4py communicator: ranks and the unique-id broadcast come
    from MPI, the data path is RCCL."""
    rank, world = comm.Get_rank(), comm.Get_size()
    ndev = max(_lib.device_count(), 1)
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("OMPI_COMM_WORLD_LOCAL_RANK", str(rank))))
    device = local % ndev
    if world == 1:
        return SelfComm(device)
    _lib.call("mfft_set_device", device)
    uid = comm.bcast(get_unique_id() if rank == 0 else None, root=0)
    return DistComm(world, rank, uid, device)


class LayoutComm(_CommBase):
    """Rank `rank` of `size` with NO device and no transport behind it: enough for everything the classes answer from
    the decomposition alone -- shapes, slices, `get_local_mesh`, `get_local_wavenumbermesh`, `get_dealias_filter`,
    `get_subarrays` -- e.g. to prepare initial conditions or inspect a layout on a machine without a GPU.  A class
    built on it has no plan: its transforms raise MfftError."""

    def __init__(self, size=1, rank=0):
        if not 0 <= int(rank) < int(size):
            raise ValueError("rank %r is not in [0, %r)" % (rank, size))
        _CommBase.__init__(self, None, int(size), int(rank), None)

    def use_device(self):
        raise _lib.MfftError("a LayoutComm has no device")

    def barrier(self):
        if self._size > 1:
            raise _lib.MfftError("a LayoutComm has no transport")

    Barrier = barrier

    def allreduce(self, x, op=SUM):
        if self._size > 1:
            raise _lib.MfftError("a LayoutComm has no transport")
        return _CommBase.allreduce(self, x, op)

    def free(self):
        pass


def as_comm(comm):
    """Accept None, one of this module's communicators, or an mpi4py one."""
    if comm is None:
        return SelfComm()
    if isinstance(comm, _CommBase):
        return comm
    if hasattr(comm, "Get_size") and hasattr(comm, "bcast"):
        if comm.Get_size() == 1:
            return SelfComm()
        return from_mpi4py(comm)
    raise TypeError("comm must be None, a mpifft4py_amd communicator or an mpi4py communicator")


class LocalComm(_CommBase):
    pass


class LocalGroup(object):
    """P virtual ranks inside this process.  `devices[r]` is the GPU of rank r
    (default: all on the current device)."""

    def __init__(self, nranks, devices=None):
        lib = _lib.load()
        cur = ctypes.c_int(0)
        _lib.check(lib.mfft_get_device(ctypes.byref(cur)))
        self.devices = list(devices) if devices is not None else [cur.value] * nranks
        arr = (ctypes.c_int * nranks)(*self.devices)
        handles = (ctypes.c_void_p * nranks)()
        _lib.call("mfft_comm_create_local", nranks, arr, handles)
        self.comms = [LocalComm(handles[r], nranks, r, self.devices[r]) for r in range(nranks)]
        self.nranks = nranks

    def run(self, fn, *args):
        """Run fn(comm, *args) on one host thread per rank; returns the results."""
        results = [None] * self.nranks
        errors = []

        def body(r):
            try:
                self.comms[r].use_device()
                results[r] = fn(self.comms[r], *args)
            except BaseException as e:      # noqa: BLE001 - re-raised on the caller
                import traceback
                errors.append((r, e, traceback.format_exc()))
                try:                        # peers blocked in a group barrier must not wait for us
                    _lib.call("mfft_comm_abort", self.comms[r]._handle)
                except Exception:           # noqa: BLE001
                    pass

        threads = [threading.Thread(target=body, args=(r,)) for r in range(self.nranks)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            errors.sort(key=lambda t: "never reached the barrier" in str(t[1]))   # root cause first
            r, e, tb = errors[0]
            raise RuntimeError("rank %d failed: %s\n%s" % (r, e, tb))
        return results

    def free(self):
        for c in self.comms:
            c.free()
