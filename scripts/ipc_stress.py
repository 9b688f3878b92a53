"""Soak test of a pull mode of the IPC transport (developer tool, round 3)

    python scripts/ipc_stress.py WORLD [pull mode 0|1|2] [pipeline] [iterations] [size]

Starts WORLD ranks on the visible GPU(s) over the IPC transport.  Iteration k transforms u_k = (k + 1) * u_0 and
compares the forward result with (k + 1) * the first iteration's result and the round trip with u_k, so a chunk pulled
before its sender had written it, or after the sender had overwritten it, cannot hide behind identical data.  (The
per-block factors it prints are of the TRANSFORMED result, in which the x pass has mixed the peers' blocks: they say
that an iteration went wrong, not which peer's chunk it was.)  (Round 3 also ran it with MFFT_IPC_STREAM_FLAGS=1, the
round-2 form of the "streams" mode with the flag operations on the per-peer streams; that form was removed in round 4,
the switch does nothing any more.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def launch():
    import socket
    world = int(sys.argv[1])
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MFFT_TRANSPORT="ipc")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    sys.exit(max(p.wait() for p in procs))


def main():
    import numpy as np
    from mpifft4py_amd import DeviceArray, Slab_R2C, from_env
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    pipeline = int(sys.argv[3]) if len(sys.argv) > 3 else -4
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 128
    comm = from_env(None, transport="ipc")
    comm.set_option("ipc_pull", mode)
    rank, P = comm.Get_rank(), comm.Get_size()
    F = Slab_R2C(np.array([n, n, n]), np.array([2 * np.pi] * 3), comm, "double", pipeline=pipeline)
    u0 = np.random.default_rng(5 + rank).random(F.real_shape())
    u = DeviceArray.from_numpy(u0)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    F.fftn(u, fu)
    F.ifftn(fu, u2)
    F.sync()
    c1 = fu.get()
    Np0 = n // P
    bad = 0
    for k in range(1, iters):
        u.set(u0 * (k + 1))
        F.fftn(u, fu)
        F.ifftn(fu, u2)           # keeps the inverse's buffers in play between forwards
        F.sync()
        c = fu.get()
        for p in range(P):        # rows x of peer p's block
            blk, ref = c[p * Np0:(p + 1) * Np0], c1[p * Np0:(p + 1) * Np0]
            fac = float(np.vdot(ref, blk).real / np.vdot(ref, ref).real)
            err = float(np.linalg.norm(blk - (k + 1) * ref) / np.linalg.norm((k + 1) * ref))
            if err > 1e-9:
                bad += 1
                kind = "EARLY (stale: ready wait)" if abs(fac - k) < 0.2 else "LATE (overwritten: done wait)" if abs(fac - (k + 2)) < 0.2 else "mixed"
                print("rank %d iteration %d: block from rank %d carries factor %.3f instead of %d (rel err %.2e): %s"
                      % (rank, k, p, fac, k + 1, err, kind), flush=True)
        back = u2.get()
        e2 = float(np.linalg.norm(back - (k + 1) * u0) / np.linalg.norm((k + 1) * u0))
        if e2 > 1e-9:
            bad += 1
            print("rank %d iteration %d: round trip rel err %.2e" % (rank, k, e2), flush=True)
        if bad > 20:
            break
    tot = comm.allreduce(float(bad))
    if rank == 0:
        print("IPC_STRESS world=%d pull=%d pipeline=%d iterations=%d flags_on_peer_streams=%s: %d bad blocks"
              % (P, mode, pipeline, iters, "removed", int(tot)), flush=True)


if __name__ == "__main__":
    if "WORLD_SIZE" not in os.environ:
        launch()
    main()
