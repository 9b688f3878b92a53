"""Generate the committed fixtures under tests/golden/ from the REAL reference
(dev container only; the reference never travels to the GPU box).

Every array written here is an *output of the reference classes* run under the
threads-as-ranks fake mpi4py on a seeded input, plus that input.  A fixture is
data (inputs + expected outputs); no reference source is stored.

Usage:  python -m oracle.refharness.make_golden
"""
import json
import os

import numpy as np

from . import fake_mpi
from .ref_import import import_reference

import_reference()
from mpi4py import MPI  # noqa: E402
from mpiFFT4py.slab import R2C as RefSlab, C2C as RefSlabC2C  # noqa: E402
from mpiFFT4py.pencil import R2C as RefPencil  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden")
L = np.array([2 * np.pi] * 3)


def _sl(s):
    return [[int(x.start or 0), int(x.stop)] for x in s]


def layouts():
    table = []
    for N in ([8, 16, 32], [32, 64, 128], [64, 64, 64], [1024, 1024, 1024]):
        for P in (1, 2, 4, 8):
            def slab(rank):
                F = RefSlab(np.array(N), L, MPI.COMM_WORLD, "double")
                return dict(decomp="slab", N=N, P=P, rank=rank,
                            real_shape=list(map(int, F.real_shape())),
                            complex_shape=list(map(int, F.complex_shape())),
                            real_shape_padded=list(map(int, F.real_shape_padded())),
                            real_slice=_sl(F.real_local_slice()),
                            real_slice_padded=_sl(F.real_local_slice(padsize=1.5)),
                            complex_slice=_sl(F.complex_local_slice()))
            table += fake_mpi.run(P, slab)
            if P < 4:
                continue
            for align in ("X", "Y"):
                for P1 in (None, 2):
                    def pen(rank):
                        F = RefPencil(np.array(N), L, MPI.COMM_WORLD, "double", P1=P1,
                                      communication="Alltoallw", alignment=align)
                        return dict(decomp="pencil" + align, N=N, P=P, rank=rank,
                                    P1_arg=P1, P1=int(F.P1), P2=int(F.P2),
                                    c0=int(F.comm0_rank), c1=int(F.comm1_rank),
                                    real_shape=list(map(int, F.real_shape())),
                                    complex_shape=list(map(int, F.complex_shape())),
                                    real_shape_padded=list(map(int, F.real_shape_padded())),
                                    real_slice=_sl(F.real_local_slice()),
                                    real_slice_padded=_sl(F.real_local_slice(padsize=1.5)),
                                    complex_slice=_sl(F.complex_local_slice()))
                    table += fake_mpi.run(P, pen)
    return table


def _gather(parts, slices, shape, dtype):
    G = np.zeros(shape, dtype=dtype)
    for p, s in zip(parts, slices):
        G[s] = p
    return G


def run_case(make, P, A, padded, real_dtype_of):
    """returns gathered (forward-result, backward-result) of the reference."""
    def body(rank):
        F = make()
        if not padded:
            a = np.zeros(F.real_shape(), dtype=real_dtype_of(F))
            a[:] = A[F.real_local_slice()]
            c = np.zeros(F.complex_shape(), dtype=F.complex)
            c = F.fftn(a, c)
            b = np.zeros(F.real_shape(), dtype=real_dtype_of(F))
            b = F.ifftn(c.copy(), b)
            return (F.complex_local_slice(), c.copy(), F.real_local_slice(), b.copy(),
                    F.global_shape() if hasattr(F, "global_shape") else F.global_complex_shape(),
                    tuple(int(n) for n in F.N))
        c = np.zeros(F.complex_shape(), dtype=F.complex)
        c[:] = A[F.complex_local_slice()]
        ap = np.zeros(F.real_shape_padded(), dtype=real_dtype_of(F))
        ap = F.ifftn(c, ap, dealias="3/2-rule")
        cp = np.zeros(F.complex_shape(), dtype=F.complex)
        cp = F.fftn(ap.copy(), cp, dealias="3/2-rule")
        return (F.real_local_slice(padsize=1.5), ap.copy(), F.complex_local_slice(), cp.copy(),
                tuple(int(1.5 * n) for n in F.N),
                F.global_shape() if hasattr(F, "global_shape") else F.global_complex_shape())
    res = fake_mpi.run(P, body)
    g1 = _gather([r[1] for r in res], [r[0] for r in res], res[0][4], res[0][1].dtype)
    g2 = _gather([r[3] for r in res], [r[2] for r in res], res[0][5], res[0][3].dtype)
    return g1, g2


def taylor_green_k(P, dealias):
    """Kinetic energy after 10 RK4 steps of the Taylor-Green vortex at 32^3
    computed WITH THE REFERENCE's slab class (same parameters and update rule
    as the reference demo; known answer 0.124953117517, demo line 103-105)."""
    nu, T, dt = 0.000625, 0.1, 0.01
    N = np.array([32, 32, 32], dtype=int)

    def body(rank):
        F = RefSlab(N, L, MPI.COMM_WORLD, "double")
        U = np.empty((3,) + F.real_shape())
        U_hat = np.empty((3,) + F.complex_shape(), dtype=complex)
        U_hat0, U_hat1, dU = (np.empty_like(U_hat) for _ in range(3))
        X = F.get_local_mesh()
        K = np.array(F.get_local_wavenumbermesh(scaled=True, broadcast=True))
        K2 = np.sum(K * K, 0)
        K_over_K2 = K / np.where(K2 == 0, 1, K2)
        a = [1. / 6., 1. / 3., 1. / 3., 1. / 6.]
        b = [0.5, 0.5, 1.]
        ws = F.work_shape(dealias)
        Ud = np.empty((3,) + ws)
        Cd = np.empty((3,) + ws)

        def rhs_(rhs):
            for i in range(3):
                Ud[i] = F.ifftn(U_hat[i], Ud[i], dealias)
            Cd[2] = F.ifftn(1j * (K[0] * U_hat[1] - K[1] * U_hat[0]), Cd[2], dealias)
            Cd[1] = F.ifftn(1j * (K[2] * U_hat[0] - K[0] * U_hat[2]), Cd[1], dealias)
            Cd[0] = F.ifftn(1j * (K[1] * U_hat[2] - K[2] * U_hat[1]), Cd[0], dealias)
            rhs[0] = F.fftn(Ud[1] * Cd[2] - Ud[2] * Cd[1], rhs[0], dealias)
            rhs[1] = F.fftn(Ud[2] * Cd[0] - Ud[0] * Cd[2], rhs[1], dealias)
            rhs[2] = F.fftn(Ud[0] * Cd[1] - Ud[1] * Cd[0], rhs[2], dealias)
            P_hat = np.sum(rhs * K_over_K2, 0)
            rhs -= P_hat * K
            rhs -= nu * K2 * U_hat
            return rhs

        U[0] = np.sin(X[0]) * np.cos(X[1]) * np.cos(X[2])
        U[1] = -np.cos(X[0]) * np.sin(X[1]) * np.cos(X[2])
        U[2] = 0
        for i in range(3):
            U_hat[i] = F.fftn(U[i], U_hat[i])
        t = 0.0
        while t < T - 1e-8:
            t += dt
            U_hat1[:] = U_hat0[:] = U_hat
            for rk in range(4):
                dU[:] = rhs_(dU)
                if rk < 3:
                    U_hat[:] = U_hat0 + b[rk] * dt * dU
                U_hat1[:] += a[rk] * dt * dU
            U_hat[:] = U_hat1[:]
        for i in range(3):
            U[i] = F.ifftn(U_hat[i], U[i])
        return F.comm.reduce(np.sum(U * U) / N[0] / N[1] / N[2] / 2)
    return fake_mpi.run(P, body)[0]


def line_golden():
    """2-D class (mpiFFT4py/line.py): per-rank inputs and the reference's outputs for the plain transform pair,
    the inverse on an arbitrary spectrum and the 3/2-rule both ways, P = 1, 2, 4."""
    from mpiFFT4py.line import R2C as RefLine
    N = [16, 48]
    L2 = np.array([2 * np.pi, 4 * np.pi])
    for prec, rt, ct in (("double", np.float64, np.complex128), ("single", np.float32, np.complex64)):
        rng = np.random.default_rng(20260211)
        A = rng.random(N).astype(rt)
        Ap = rng.random((int(1.5 * N[0]), int(1.5 * N[1]))).astype(rt)
        out = dict(N=np.array(N), L=L2)
        for P in (1, 2, 4):
            seeds = [np.random.default_rng(1000 + 10 * P + r) for r in range(P)]

            def body(rank):
                F = RefLine(np.array(N), L2, MPI.COMM_WORLD, prec)
                u = np.ascontiguousarray(A[F.real_local_slice()])
                fu = F.fft2(u, np.zeros(F.complex_shape(), dtype=ct)).copy()
                g = seeds[rank]
                crnd = (g.random(F.complex_shape()) + 1j * g.random(F.complex_shape())).astype(ct)
                b = F.ifft2(crnd.copy(), np.zeros(F.real_shape(), dtype=rt)).copy()
                bp = F.ifft2(crnd.copy(), np.zeros(F.real_shape_padded(), dtype=rt), dealias="3/2-rule").copy()
                up = np.ascontiguousarray(Ap[F.real_local_slice(padsize=1.5)])
                cp = F.fft2(up, np.zeros(F.complex_shape(), dtype=ct), dealias="3/2-rule").copy()
                return dict(u=u, fu=fu, crnd=crnd, b=b, bp=bp, up=up, cp=cp)
            for r, d in enumerate(fake_mpi.run(P, body)):
                for k, v in d.items():
                    out["P%d_r%d_%s" % (P, r, k)] = v
        np.savez_compressed(os.path.join(OUT, "line_16x48_%s.npz" % prec), **out)
    print("line fixtures written")


def _boxes(lst):
    return [[list(b.sizes), [int(x.stop - x.start) for x in b.slices], [int(x.start) for x in b.slices]] for b in lst]


def subarrays():
    """get_subarrays of the reference classes (slab.py:199-211, pencil.py:218-246, 971-999): (sizes, subsizes, starts) of
    every box, for padsize 1 and 1.5."""
    table = []
    for N in ([8, 16, 32], [32, 64, 128], [1024, 1024, 1024]):
        for P in (2, 4, 8):
            for pad in (1, 1.5):
                def slab(rank):
                    F = RefSlab(np.array(N), L, MPI.COMM_WORLD, "double")
                    A, B, cd = F.get_subarrays(padsize=pad)
                    return dict(decomp="slab", N=N, P=P, rank=rank, padsize=pad, lists=[_boxes(A), _boxes(B)],
                                counts_displs=[[list(map(int, cd[0])), list(map(int, cd[1]))]])
                table += fake_mpi.run(P, slab)
                if P < 4:
                    continue
                for align in ("X", "Y"):
                    for P1 in (None, 2):
                        def pen(rank):
                            F = RefPencil(np.array(N), L, MPI.COMM_WORLD, "double", P1=P1, communication="Alltoallw", alignment=align)
                            r = F.get_subarrays(padsize=pad)
                            return dict(decomp="pencil" + align, N=N, P=P, rank=rank, padsize=pad, P1_arg=P1,
                                        lists=[_boxes(x) for x in r[:4]],
                                        counts_displs=[[list(map(int, c[0])), list(map(int, c[1]))] for c in r[4:]])
                        table += fake_mpi.run(P, pen)
    return table


def helpers_golden():
    """Host helper API of the reference classes (slab.py:146-197, pencil.py:289-349, 945-957, line.py:105-134,
    mpibase.py:61-131), every rank of every decomposition on a mesh with three different box lengths:
    get_local_mesh, complex_local_wavenumbers, get_local_wavenumbermesh for all 8 (scaled, broadcast,
    eliminate_highest_freq) combinations, get_dealias_filter.  Keys: <class>_P<P>[_P1<p1>]_r<rank>_<what>[_<axis>]."""
    N = [8, 16, 32]
    Lb = np.array([2 * np.pi, 4 * np.pi, 3.0])
    combos = [(s, b, e) for s in (False, True) for b in (False, True) for e in (False, True)]
    for prec in ("double", "single"):
        out = dict(N=np.array(N), L=Lb)

        def put(tag, rank, F, mesh_kw=True, kvec=True):
            pre = "%s_r%d_" % (tag, rank)
            res = {}
            X = F.get_local_mesh()
            for i in range(len(X)):
                res[pre + "mesh_%d" % i] = np.array(X[i])
            if kvec:
                for i, k in enumerate(F.complex_local_wavenumbers()):
                    res[pre + "kvec_%d" % i] = np.array(k)
            if mesh_kw:
                for (s_, b_, e_) in combos:
                    K = F.get_local_wavenumbermesh(scaled=s_, broadcast=b_, eliminate_highest_freq=e_)
                    for i in range(len(K)):
                        res[pre + "K_s%d_b%d_e%d_%d" % (s_, b_, e_, i)] = np.array(K[i])
            res[pre + "dealias"] = np.array(F.get_dealias_filter())
            return res

        for P in (1, 2, 4, 8):
            def slab(rank):
                return put("slab_P%d" % P, rank, RefSlab(np.array(N), Lb.copy(), MPI.COMM_WORLD, prec))
            for d in fake_mpi.run(P, slab):
                out.update(d)
        for P in (1, 2):
            def slabc(rank):
                # upstream's C2C class inherits the R2C wave-number helpers (half-spectrum kz for a full-spectrum array):
                # only the helper that is its own is recorded
                F = RefSlabC2C(np.array(N), Lb.copy(), MPI.COMM_WORLD, prec)
                return {"slabc2c_P%d_r%d_tkvec_%d" % (P, rank, i): np.array(k)
                        for i, k in enumerate(F.transformed_local_wavenumbers())}
            for d in fake_mpi.run(P, slabc):
                out.update(d)
        for P, P1 in ((4, None), (8, None), (8, 2)):
            for align in ("Y", "X"):
                def pen(rank):
                    F = RefPencil(np.array(N), Lb.copy(), MPI.COMM_WORLD, prec, P1=P1, communication="Alltoallw",
                                  alignment=align)
                    # x-aligned: only the mesh, the wave vectors (upstream leaves ky unsliced there: recorded as it is)
                    # and the filter follow the common contract; its get_local_wavenumbermesh has another signature
                    return put("pencil%s_P%d_P1%s" % (align, P, P1), rank, F, mesh_kw=(align == "Y"))
                for d in fake_mpi.run(P, pen):
                    out.update(d)
        from mpiFFT4py.line import R2C as RefLine
        N2, L2 = [16, 48], np.array([2 * np.pi, 5.0])
        out["N_line"], out["L_line"] = np.array(N2), L2
        for P in (1, 2, 4):
            def line(rank):
                return put("line_P%d" % P, rank, RefLine(np.array(N2), L2.copy(), MPI.COMM_WORLD, prec), kvec=False)
            for d in fake_mpi.run(P, line):
                out.update(d)
        np.savez_compressed(os.path.join(OUT, "helpers_%s.npz" % prec), **out)
    print("helper fixtures written")


def main():
    import sys
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "line":
        return line_golden()
    if len(sys.argv) > 1 and sys.argv[1] == "helpers":
        return helpers_golden()
    helpers_golden()
    line_golden()
    with open(os.path.join(OUT, "layouts.json"), "w") as f:
        json.dump(layouts(), f, separators=(",", ":"))
    with open(os.path.join(OUT, "subarrays.json"), "w") as f:
        json.dump(subarrays(), f, separators=(",", ":"))

    N = [8, 16, 32]
    rng = np.random.default_rng(20260210)
    for prec, rt, ct in (("double", np.float64, np.complex128),
                         ("single", np.float32, np.complex64)):
        A = rng.random(N).astype(rt)
        out = dict(A=A)
        fl = lambda F: F.float
        for P in (1, 2, 4):
            for mode in ("Alltoall", "Alltoallw"):
                mk = lambda: RefSlab(np.array(N), L, MPI.COMM_WORLD, prec, communication=mode)
                C, B = run_case(mk, P, A, False, fl)
                out["slab_P%d_%s_fwd" % (P, mode)] = C
                out["slab_P%d_%s_bwd" % (P, mode)] = B
        for P, P1 in ((4, None), (8, None), (8, 2)):
            for align in ("X", "Y"):
                mk = lambda: RefPencil(np.array(N), L, MPI.COMM_WORLD, prec, P1=P1,
                                       communication="Alltoallw", alignment=align)
                C, B = run_case(mk, P, A, False, fl)
                out["pencil%s_P%d_P1%s_fwd" % (align, P, P1)] = C
                out["pencil%s_P%d_P1%s_bwd" % (align, P, P1)] = B
        # 3/2-rule: spectrum with the Nyquist planes removed (tests/test_FFT.py:170-174)
        C0 = np.fft.rfftn(A.astype(np.float64)).astype(ct)
        C0[N[0] // 2] = 0
        C0[:, N[1] // 2] = 0
        C0[:, :, -1] = 0
        out["C0"] = C0
        for P in (1, 2):
            mk = lambda: RefSlab(np.array(N), L, MPI.COMM_WORLD, prec, communication="Alltoallw")
            AP, CP = run_case(mk, P, C0, True, fl)
            out["slab_P%d_pad_bwd" % P] = AP
            out["slab_P%d_pad_fwd" % P] = CP
        for align in ("X", "Y"):
            mk = lambda: RefPencil(np.array(N), L, MPI.COMM_WORLD, prec,
                                   communication="Alltoallw", alignment=align)
            AP, CP = run_case(mk, 4, C0, True, fl)
            out["pencil%s_P4_pad_bwd" % align] = AP
            out["pencil%s_P4_pad_fwd" % align] = CP
        # slab C2C
        Ac = (rng.random(N) + 1j * rng.random(N)).astype(ct)
        out["Ac"] = Ac
        for P in (1, 2):
            mk = lambda: RefSlabC2C(np.array(N), L, MPI.COMM_WORLD, prec)
            C, B = run_case(mk, P, Ac, False, lambda F: F.complex)
            out["slabc2c_P%d_fwd" % P] = C
            out["slabc2c_P%d_bwd" % P] = B
        # slab C2C 3/2-rule (fresh objects per run: padded work arrays are zero, see oracle notes)
        Cc = np.fft.fftn(Ac.astype(np.complex128)).astype(ct)
        out["Cc"] = Cc
        for P in (1, 2):
            mk = lambda: RefSlabC2C(np.array(N), L, MPI.COMM_WORLD, prec)
            AP, CP = run_case(mk, P, Cc, True, lambda F: F.complex)
            if P == 2:
                out["slabc2c_P%d_pad_bwd" % P] = AP
            out["slabc2c_P%d_pad_fwd" % P] = CP
        np.savez_compressed(os.path.join(OUT, "ref_8x16x32_%s.npz" % prec), **out)

    demo = {"k_expected_demo": 0.124953117517}
    for dealias in ("3/2-rule", "2/3-rule", None):
        for P in (1, 2):
            demo["k_P%d_%s" % (P, dealias)] = float(taylor_green_k(P, dealias))
    with open(os.path.join(OUT, "taylor_green.json"), "w") as f:
        json.dump(demo, f, indent=1)
    print("golden fixtures written to", os.path.abspath(OUT))
    print(demo)


if __name__ == "__main__":
    main()
