"""Worker of tests/test_gpu_multiprocess.py: one PROCESS per rank (launched by
torch.distributed.run), all ranks on the box's single GPU, RcclComm over the
shared-memory stand-in for librccl (tests/mock_rccl).  Checks the process-per-GPU
product path -- from_env() rendezvous, DistComm, grouped send/recv all-to-all-v,
the two-stream exchange pipeline -- against the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mpifft4py_amd import DeviceArray, Pencil_R2C, Slab_C2C, Slab_R2C, from_env, _lib  # noqa: E402
from oracle import mpifft_oracle as orc  # noqa: E402

L = np.array([2 * np.pi] * 3)


def main():
    # a hang must say where: every rank dumps its Python stack (and leaves) well before the test's own timeout
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("MP_WORKER_DUMP_AFTER", "200")), exit=True)
    comm = from_env()
    rank, P = comm.Get_rank(), comm.Get_size()

    def stage(name):
        if os.environ.get("MP_WORKER_VERBOSE") and rank == 0:
            sys.stderr.write("[mp_worker] %s\n" % name)
            sys.stderr.flush()
    N = [32, 64, 128]
    A = np.random.default_rng(4242).random(N)
    B2 = np.fft.rfftn(A)

    # host helpers over the communicator
    x = np.arange(5, dtype=np.float64) if rank == 0 else np.zeros(5)
    comm.Bcast(x, root=0)
    assert np.array_equal(x, np.arange(5.0))
    assert comm.allreduce(float(rank)) == sum(range(P))
    assert comm.bcast({"hello": P} if rank == 0 else None)["hello"] == P
    comm.selftest(1 << 18, 30000)           # verified all-to-all of a byte pattern through the transport under test
    if comm.get_option("ipc_pull") < 0:
        # RCCL's entry points: what the LIBRARY says about the communicator (ncclCommCount / ncclCommUserRank / ncclGetVersion
        # behind get_option "rccl_*"; bench.py prints them) must be what the launcher asked for; 9.99.0 is the stand-in
        assert comm.get_option("rccl_nranks") == P and comm.get_option("rccl_rank") == rank, \
            (comm.get_option("rccl_nranks"), comm.get_option("rccl_rank"))
        assert comm.get_option("rccl_version") > 0 and comm.get_option("rccl_device") >= 0
        if os.environ.get("MFFT_RCCL_LIB", "").endswith("libmockrccl.so"):
            assert comm.get_option("rccl_version") == 99900
        buf = __import__("ctypes").create_string_buffer(64)
        _lib.call("mfft_device_pci_bus_id", comm.device, buf, 64)
        assert buf.value.count(b":") == 2, buf.value

    # IPC transport: every way of pulling the chunks (one kernel over all peers, per-peer copy streams, copies one after the
    # other) x CU-masked streams or not must give the SAME bits; the other transports have one mode
    ipc = comm.get_option("ipc_pull") >= 0
    # The per-peer streams mode (ipc_pull = 2) is an option nobody defaults to and the one mode that ever stalled (twice in
    # the closing suites of round 4, ranks sharing a device: profiles/r04_ipc_soak.txt): it is swept only
    # on request (MP_WORKER_STREAMS=1), also with a GPU per rank, so that the first multi-GPU lease cannot go red on it.
    pulls = (1, 2, 0) if os.environ.get("MP_WORKER_STREAMS", "0") not in ("", "0") else (1, 0)
    modes = [(m, cus) for m in pulls for cus in (-1, 16)] if ipc else [(None, 0)]
    shared = int(_lib.device_count()) < P          # several ranks on one device (this pool's one-GPU boxes)
    if ipc and shared:
        # Ranks that SHARE a device: the default (pull kernel) with and without CU-masked streams and the copy-engine
        # mode; at 8 processes without the masks (extra masked streams push the hardware scheduler into time-slicing).
        modes = [(1, -1), (1, 16), (0, -1)] if P <= 4 else [(1, -1), (0, -1)]
    first = {}
    for mode, cus in modes:
        stage("slab pull mode %s comm_cus %s" % (mode, cus))
        if mode is not None:
            comm.set_option("ipc_pull", mode)
        for pipeline in (1, 4, -4):
            F = Slab_R2C(np.array(N), L, comm, "double", pipeline=pipeline, comm_cus=cus)
            u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
            fu = DeviceArray.empty(F.complex_shape(), F.complex)
            u2 = DeviceArray.empty(F.real_shape(), F.float)
            for _ in range(3):
                F.fftn(u, fu)
                F.ifftn(fu, u2)
            F.sync()
            got, back = fu.get(), u2.get()
            assert orc.rel_l2(got, B2[F.complex_local_slice()]) < 1e-10, ("slab", pipeline, mode, cus)
            assert orc.rel_l2(back, A[F.real_local_slice()]) < 1e-10, ("slab", pipeline, mode, cus)
            if pipeline not in first:
                first[pipeline] = (got, back)
            assert np.array_equal(got, first[pipeline][0]) and np.array_equal(back, first[pipeline][1]), ("bits", pipeline, mode, cus)
            del F, u, fu, u2
    if ipc:
        comm.set_option("ipc_pull", int(os.environ.get("MP_WORKER_PULL", "1")))
    # padded + masked paths
    stage("padded + masked slab")
    C0 = B2.copy()
    C0[N[0] // 2] = 0
    C0[:, N[1] // 2] = 0
    C0[:, :, -1] = 0
    lay = orc.SlabLayout(N, P)
    want = orc.slab_r2c_backward_padded(orc.scatter_complex(C0, lay), N)
    F = Slab_R2C(np.array(N), L, comm, "double")
    ap = F.ifftn(np.ascontiguousarray(C0[F.complex_local_slice()]), np.zeros(F.real_shape_padded()), dealias="3/2-rule")
    assert orc.rel_l2(ap, want[rank]) < 4e-10
    cp = F.fftn(ap, np.zeros(F.complex_shape(), dtype=complex), dealias="3/2-rule")
    assert orc.rel_l2(cp, C0[F.complex_local_slice()]) < 4e-10
    # 2/3-rule: the reference's filter (pruned passes, smaller exchange; the ranks agree on the route when the mask is
    # set) with the blocking exchange, and a pipelined plan (mask applied on load)
    for pipeline in (1, 2):
        Fd = Slab_R2C(np.array(N), L, comm, "double", pipeline=pipeline)
        cl = np.ascontiguousarray(B2[Fd.complex_local_slice()])
        mask = np.broadcast_to(Fd.get_dealias_filter(), Fd.complex_shape())
        ud = Fd.ifftn(cl.copy(), np.zeros(Fd.real_shape()), dealias="2/3-rule")
        ur = Fd.ifftn(cl * mask, np.zeros(Fd.real_shape()))
        assert orc.rel_l2(ud, ur) < 1e-12, ("2/3-rule", pipeline)
        # against the oracle's arithmetic on the whole cube, and the route: every rank prunes; at 8 ranks the ky of
        # ranks 2..5 of this mesh (ky 16..47 of 64, removed: 22..42) are partly, of none wholly, removed -> route 1
        gm = orc.dealias_mask(N, np.fft.fftfreq(N[0], 1. / N[0]), np.fft.fftfreq(N[1], 1. / N[1]), np.fft.rfftfreq(N[2], 1. / N[2]))
        want = np.fft.irfftn(B2 * gm, s=N, axes=(0, 1, 2))
        assert orc.rel_l2(ud, want[Fd.real_local_slice()]) < 4e-10, ("2/3-rule vs oracle", pipeline)
        assert Fd.plan_info("pruned_route") == (2 if not mask.any() else 1), (Fd.plan_info("pruned_route"), int(mask.sum()))
    # the nonlinear term as one plan operation (mfft_nonlinear_cross: over several ranks the fused z kernel between blocking
    # exchanges, on buffers the plan takes from the communicator) against six ifftn + cross + three fftn of the same object
    stage("nonlinear cross")
    from mpifft4py_amd import spectral
    for dealias in ("3/2-rule", None):
        Fn = Slab_R2C(np.array(N), L, comm, "double")
        rng_n = np.random.default_rng(900 + rank)
        a, b, got, want_n = (Fn.empty_complex(3) for _ in range(4))
        for x in (a, b):
            for i in range(3):
                Fn.fftn(DeviceArray.from_numpy(rng_n.random(Fn.real_shape()) - 0.5), x.component(i))
        ws = Fn.work_shape(dealias)
        ua, ub, rr = (DeviceArray.empty((3,) + tuple(ws), Fn.float) for _ in range(3))
        for i in range(3):
            Fn.ifftn(a.component(i), ua.component(i), dealias)
            Fn.ifftn(b.component(i), ub.component(i), dealias)
        spectral.cross(Fn, ua, ub, rr)
        for i in range(3):
            Fn.fftn(rr.component(i), want_n.component(i), dealias)
        spectral.cross_transform(Fn, a, b, got, dealias)
        Fn.sync()
        assert Fn.plan_info("nonlinear_fused_3_2" if dealias else "nonlinear_fused_none") == 1
        assert orc.rel_l2(got.get(), want_n.get()) < 1e-13, ("nonlinear", dealias)
        del Fn, a, b, got, want_n, ua, ub, rr
    # C2C
    Ac = A + 1j * np.random.default_rng(7).random(N)
    Fc = Slab_C2C(np.array(N), L, comm, "single")
    c = Fc.fftn(np.ascontiguousarray(Ac[Fc.original_local_slice()]).astype(np.complex64),
                np.zeros(Fc.transformed_shape(), dtype=np.complex64))
    assert orc.rel_l2(c, np.fft.fftn(Ac)[Fc.transformed_local_slice()]) < 1e-5
    # pencils; over the IPC transport once more with relay striping (csrc/relay_plan.h: the sub-group exchanges also use
    # the links to the ranks outside the group, two hops through their memory) -- the same bits must come out
    if P >= 4:
        outs = {}
        for relay in ((0, 1) if ipc else (None,)):
            stage("pencils relay %s" % relay)
            if relay is not None:
                comm.set_option("ipc_relay", relay)
                assert comm.get_option("ipc_relay") == relay
            for align in ("X", "Y"):
                for pipeline in (1, 4):
                    Fp = Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment=align, pipeline=pipeline)
                    for _ in range(2):
                        c = Fp.fftn(np.ascontiguousarray(A[Fp.real_local_slice()]), np.zeros(Fp.complex_shape(), dtype=complex))
                        b = Fp.ifftn(c, np.zeros(Fp.real_shape()))
                    assert orc.rel_l2(c, B2[Fp.complex_local_slice()]) < 1e-10, (align, pipeline, relay)
                    assert orc.rel_l2(b, A[Fp.real_local_slice()]) < 1e-10, (align, pipeline, relay)
                    key = (align, pipeline)
                    if key not in outs:
                        outs[key] = (c, b)
                    assert np.array_equal(c, outs[key][0]) and np.array_equal(b, outs[key][1]), ("bits", align, pipeline, relay)
                    del Fp
            # the 3/2-rule and 2/3-rule pencil paths ride on the same exchanges
            Fp = Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment="X")
            cl = np.ascontiguousarray(B2[Fp.complex_local_slice()])
            up = Fp.ifftn(cl, np.zeros(Fp.real_shape_padded()), dealias="3/2-rule")
            cb = Fp.fftn(up, np.zeros(Fp.complex_shape(), dtype=complex), dealias="3/2-rule")
            ud = Fp.ifftn(cl, np.zeros(Fp.real_shape()), dealias="2/3-rule")
            if "pad" not in outs:
                outs["pad"] = (up, cb, ud)
            assert all(np.array_equal(x, y) for x, y in zip((up, cb, ud), outs["pad"])), ("bits padded / masked", relay)
            del Fp
        if ipc:
            comm.set_option("ipc_relay", 0)
    comm.barrier()
    if rank == 0:
        print("MP_OK world=%d" % P)


if __name__ == "__main__":
    main()
