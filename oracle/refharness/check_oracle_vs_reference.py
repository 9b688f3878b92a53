"""Pin oracle/mpifft_oracle.py against the REAL reference (dev container only).

Runs the unmodified reference classes under the threads-as-ranks fake mpi4py
for every decomposition / communication mode / precision / rank count the
oracle restates, on the same seeded input, and asserts that

  * the layout tables (shapes, slices, grid, sub-ranks) are identical, and
  * every rank's fftn / ifftn result agrees with the oracle's to <= 1e-13
    (double) / 1e-5 (single) relative L2 -- both sides run numpy.fft, so the
    only differences are summation-order free copies.

Usage:  python -m oracle.refharness.check_oracle_vs_reference
"""
import sys

import numpy as np

from . import fake_mpi
from .ref_import import import_reference

import_reference()
from mpi4py import MPI  # noqa: E402  (the fake one)
from mpiFFT4py.slab import R2C as RefSlab, C2C as RefSlabC2C  # noqa: E402
from mpiFFT4py.pencil import R2C as RefPencil  # noqa: E402

from oracle import mpifft_oracle as orc  # noqa: E402

L = np.array([2 * np.pi] * 3)


def _tol(prec):
    return 1e-13 if prec == "double" else 2e-5


def run_ref_slab(N, P, prec, mode, A, padded=False, kind="R2C"):
    def body(rank):
        cls = RefSlab if kind == "R2C" else RefSlabC2C
        F = cls(np.array(N), L, MPI.COMM_WORLD, prec, communication=mode)
        lay = dict(real_shape=tuple(F.real_shape()), complex_shape=tuple(F.complex_shape()),
                   real_slice=F.real_local_slice(), complex_slice=F.complex_local_slice())
        ctype = F.complex
        rtype = F.float if kind == "R2C" else F.complex
        if padded and kind == "C2C":
            c = np.zeros(F.complex_shape(), dtype=ctype)
            c[:] = A[F.complex_local_slice()]
            ap = np.zeros(F.real_shape_padded(), dtype=ctype)
            ap = F.ifftn(c, ap, dealias="3/2-rule")
            cp = np.zeros(F.complex_shape(), dtype=ctype)
            cp = F.fftn(ap.copy(), cp, dealias="3/2-rule")
            return lay, ap.copy(), cp.copy()
        if not padded:
            a = np.zeros(F.real_shape(), dtype=rtype)
            a[:] = A[F.real_local_slice()]
            c = np.zeros(F.complex_shape(), dtype=ctype)
            c = F.fftn(a, c)
            c_in = c.copy()
            b = np.zeros(F.real_shape(), dtype=rtype)
            b = F.ifftn(c, b)
            assert np.array_equal(c, c_in), "reference ifftn modified its input"
            return lay, c.copy(), b.copy()
        # padded: A is the global *spectrum* here
        c = np.zeros(F.complex_shape(), dtype=ctype)
        c[:] = A[F.complex_local_slice()]
        ap = np.zeros(F.real_shape_padded(), dtype=rtype)
        ap = F.ifftn(c, ap, dealias="3/2-rule")
        cp = np.zeros(F.complex_shape(), dtype=ctype)
        cp = F.fftn(ap.copy(), cp, dealias="3/2-rule")
        return lay, ap.copy(), cp.copy()
    return fake_mpi.run(P, body)


def run_ref_pencil(N, P, prec, mode, align, A, P1=None, padded=False):
    def body(rank):
        F = RefPencil(np.array(N), L, MPI.COMM_WORLD, prec, P1=P1,
                      communication=mode, alignment=align)
        lay = dict(real_shape=tuple(F.real_shape()), complex_shape=tuple(F.complex_shape()),
                   real_slice=F.real_local_slice(), complex_slice=F.complex_local_slice(),
                   P1=F.P1, P2=F.P2, c0=F.comm0_rank, c1=F.comm1_rank)
        if not padded:
            a = np.zeros(F.real_shape(), dtype=F.float)
            a[:] = A[F.real_local_slice()]
            c = np.zeros(F.complex_shape(), dtype=F.complex)
            c = F.fftn(a, c)
            b = np.zeros(F.real_shape(), dtype=F.float)
            b = F.ifftn(c.copy(), b)
            return lay, c.copy(), b.copy()
        c = np.zeros(F.complex_shape(), dtype=F.complex)
        c[:] = A[F.complex_local_slice()]
        ap = np.zeros(F.real_shape_padded(), dtype=F.float)
        ap = F.ifftn(c, ap, dealias="3/2-rule")
        cp = np.zeros(F.complex_shape(), dtype=F.complex)
        cp = F.fftn(ap.copy(), cp, dealias="3/2-rule")
        return lay, ap.copy(), cp.copy()
    return fake_mpi.run(P, body)


def _slices_equal(a, b):
    norm = lambda s: tuple((x.start or 0, x.stop) for x in s)
    return norm(a) == norm(b)


def check_slab(N, P, prec, mode, rng):
    rtype, ctype = orc.dtypes(prec)
    A = rng.random(N).astype(rtype)
    ref = run_ref_slab(N, P, prec, mode, A)
    lay = orc.SlabLayout(N, P)
    us = orc.scatter_real(A, lay)
    fus = orc.slab_r2c_forward(us, N, prec)
    back = orc.slab_r2c_backward(fus, N, prec)
    worst = 0.0
    for r in range(P):
        rl, rc, rb = ref[r]
        assert rl["real_shape"] == lay.real_shape()
        assert rl["complex_shape"] == lay.complex_shape()
        assert _slices_equal(rl["real_slice"], lay.real_local_slice(r))
        assert _slices_equal(rl["complex_slice"], lay.complex_local_slice(r))
        worst = max(worst, orc.rel_l2(fus[r], rc), orc.rel_l2(back[r], rb))
    assert worst < _tol(prec), worst
    # padded
    C = np.fft.rfftn(A.astype(np.float64)).astype(ctype)
    C[N[0] // 2] = 0
    C[:, N[1] // 2] = 0
    C[:, :, -1] = 0
    if P <= N[0] // 2:
        refp = run_ref_slab(N, P, prec, mode, C, padded=True)
        cs = orc.scatter_complex(C, lay)
        ap = orc.slab_r2c_backward_padded(cs, N, prec)
        cp = orc.slab_r2c_forward_padded(ap, N, prec)
        for r in range(P):
            worst = max(worst, orc.rel_l2(ap[r], refp[r][1]), orc.rel_l2(cp[r], refp[r][2]))
        assert worst < _tol(prec), worst
    return worst


def check_slab_c2c(N, P, prec, rng):
    rtype, ctype = orc.dtypes(prec)
    A = (rng.random(N) + 1j * rng.random(N)).astype(ctype)
    ref = run_ref_slab(N, P, prec, "Alltoall", A, kind="C2C")
    lay = orc.SlabLayout(N, P, kind="C2C")
    us = orc.scatter_real(A, lay)
    fus = orc.slab_c2c_forward(us, N, prec)
    back = orc.slab_c2c_backward(fus, N, prec)
    worst = 0.0
    for r in range(P):
        rl, rc, rb = ref[r]
        assert rl["complex_shape"] == lay.complex_shape()
        worst = max(worst, orc.rel_l2(fus[r], rc), orc.rel_l2(back[r], rb))
    assert worst < _tol(prec), worst
    # 3/2-rule (fresh reference objects: their padded work arrays are still zero)
    C = np.fft.fftn(A.astype(np.complex128)).astype(ctype)
    if P == 1 or P <= N[0] // 2:
        refp = run_ref_slab(N, P, prec, "Alltoall", C, padded=True, kind="C2C")
        cs = [np.ascontiguousarray(C[lay.complex_local_slice(r)]) for r in range(P)]
        ap = orc.slab_c2c_backward_padded(cs, N, prec)
        cp = orc.slab_c2c_forward_padded(ap, N, prec)
        for r in range(P):
            worst = max(worst, orc.rel_l2(ap[r], refp[r][1]), orc.rel_l2(cp[r], refp[r][2]))
        assert worst < _tol(prec), worst
    return worst


def check_pencil(N, P, prec, mode, align, rng, P1=None):
    rtype, ctype = orc.dtypes(prec)
    A = rng.random(N).astype(rtype)
    ref = run_ref_pencil(N, P, prec, mode, align, A, P1=P1)
    lay = orc.PencilLayout(N, P, P1, align)
    us = orc.scatter_real(A, lay)
    fus = orc.pencil_r2c_forward(us, N, P1, align, prec)
    back = orc.pencil_r2c_backward(fus, N, P1, align, prec)
    worst = 0.0
    for r in range(P):
        rl, rc, rb = ref[r]
        assert (rl["P1"], rl["P2"]) == (lay.P1, lay.P2)
        assert (rl["c0"], rl["c1"]) == lay.ranks(r)
        assert rl["real_shape"] == lay.real_shape()
        assert rl["complex_shape"] == lay.complex_shape(r), (rl["complex_shape"], lay.complex_shape(r))
        assert _slices_equal(rl["real_slice"], lay.real_local_slice(r))
        assert _slices_equal(rl["complex_slice"], lay.complex_local_slice(r))
        worst = max(worst, orc.rel_l2(fus[r], rc), orc.rel_l2(back[r], rb))
    assert worst < _tol(prec), worst
    # padded (3/2-rule)
    C = np.fft.rfftn(A.astype(np.float64)).astype(ctype)
    C[N[0] // 2] = 0
    C[:, N[1] // 2] = 0
    C[:, :, -1] = 0
    refp = run_ref_pencil(N, P, prec, mode, align, C, P1=P1, padded=True)
    cs = orc.scatter_complex(C, lay)
    ap = orc.pencil_r2c_backward_padded(cs, N, P1, align, prec)
    cp = orc.pencil_r2c_forward_padded(ap, N, P1, align, prec)
    for r in range(P):
        worst = max(worst, orc.rel_l2(ap[r], refp[r][1]), orc.rel_l2(cp[r], refp[r][2]))
    assert worst < _tol(prec), worst
    return worst


def check_pencil_n(N, P, prec, align, rng, P1=None):
    """'AlltoallN': input projected to zero z-Nyquist as the reference's test does (tests/test_FFT.py:64-68)."""
    rtype, ctype = orc.dtypes(prec)
    A = rng.random(N).astype(rtype)
    C = np.fft.rfftn(A.astype(np.float64))
    C[:, :, -1] = 0
    A = np.fft.irfftn(C, s=N, axes=(0, 1, 2)).astype(rtype)
    ref = run_ref_pencil(N, P, prec, "AlltoallN", align, A, P1=P1)
    lay = orc.PencilNLayout(N, P, P1, align)
    us = orc.scatter_real(A, lay)
    fus = orc.pencil_r2c_forward_n(us, N, P1, align, prec)
    back = orc.pencil_r2c_backward_n(fus, N, P1, align, prec)
    worst = 0.0
    for r in range(P):
        rl, rc, rb = ref[r]
        assert rl["complex_shape"] == lay.complex_shape(r), (rl["complex_shape"], lay.complex_shape(r))
        assert _slices_equal(rl["complex_slice"], lay.complex_local_slice(r))
        worst = max(worst, orc.rel_l2(fus[r], rc), orc.rel_l2(back[r], rb))
    assert worst < _tol(prec), worst
    return worst


def check_line(N, P, prec, rng):
    """2-D class (line.py): plain pair, inverse of an arbitrary spectrum, 3/2-rule both ways on arbitrary data (P = 1
    no-fold rule, P > 1 Nyquist packing), mask.  The reference's 2/3-rule inverse returns zeros for P > 1 (its masked
    copy aliases a zero-filled work array, line.py:264-266, 297) and is only compared on one rank."""
    from mpi4py import MPI
    from mpiFFT4py.line import R2C as RefLine
    rt, ct = orc.dtypes(prec)
    L2 = np.array([2 * np.pi, 4 * np.pi])
    lay = orc.LineLayout(N, P)
    A = rng.random(N).astype(rt)
    Ap = rng.random((int(1.5 * N[0]), int(1.5 * N[1]))).astype(rt)
    crnd = [(rng.random(lay.complex_shape(r)) + 1j * rng.random(lay.complex_shape(r))).astype(ct) for r in range(P)]

    def body(r):
        F = RefLine(np.array(N), L2, MPI.COMM_WORLD, prec)
        assert tuple(F.real_shape()) == lay.real_shape() and tuple(F.complex_shape()) == lay.complex_shape(r)
        assert _slices_equal(F.real_local_slice(), lay.real_slice(r)) and _slices_equal(F.complex_local_slice(), lay.complex_slice(r))
        c = F.fft2(np.ascontiguousarray(A[F.real_local_slice()]), np.zeros(F.complex_shape(), dtype=ct)).copy()
        b = F.ifft2(crnd[r].copy(), np.zeros(F.real_shape(), dtype=rt)).copy()
        bp = F.ifft2(crnd[r].copy(), np.zeros(F.real_shape_padded(), dtype=rt), dealias="3/2-rule").copy()
        cp = F.fft2(np.ascontiguousarray(Ap[F.real_local_slice(padsize=1.5)]), np.zeros(F.complex_shape(), dtype=ct),
                    dealias="3/2-rule").copy()
        b23 = F.ifft2(crnd[r].copy(), np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule").copy()
        return c, b, bp, cp, b23, F.get_dealias_filter()
    ref = fake_mpi.run(P, body)
    o_c = orc.line_r2c_forward([np.ascontiguousarray(A[lay.real_slice(r)]) for r in range(P)], N, prec)
    o_b = orc.line_r2c_backward(crnd, N, prec)
    o_bp = orc.line_r2c_backward_padded(crnd, N, prec)
    o_cp = orc.line_r2c_forward_padded([np.ascontiguousarray(Ap[lay.real_slice(r, 1.5)]) for r in range(P)], N, prec)
    masks = [orc.line_dealias_mask(N, L2, lay, r) for r in range(P)]
    o_b23 = orc.line_r2c_backward([c * m for c, m in zip(crnd, masks)], N, prec)
    worst = 0.0
    for r in range(P):
        assert np.array_equal(masks[r], ref[r][5])
        worst = max(worst, orc.rel_l2(o_c[r], ref[r][0]), orc.rel_l2(o_b[r], ref[r][1]), orc.rel_l2(o_bp[r], ref[r][2]),
                    orc.rel_l2(o_cp[r], ref[r][3]))
        if P == 1:
            worst = max(worst, orc.rel_l2(o_b23[r], ref[r][4]))
        else:
            assert float(np.abs(ref[r][4]).max()) == 0.0      # the documented upstream behaviour
    assert worst < _tol(prec), worst
    return worst


def main():
    rng = np.random.default_rng(7)
    n = 0
    for N in ([16, 48], [32, 64], [64, 32]):
        for prec in ("double", "single"):
            for P in (1, 2, 4):
                w = check_line(N, P, prec, rng)
                print("line    N=%s P=%d %s           worst rel-L2 %.2e" % (N, P, prec, w))
                n += 1
    for N in ([8, 16, 32], [32, 64, 128]):
        for prec in ("double", "single"):
            for P in (1, 2, 4, 8):
                for mode in ("Alltoall", "Alltoallw"):
                    w = check_slab(N, P, prec, mode, rng)
                    print("slab    N=%s P=%d %s %-9s worst rel-L2 %.2e" % (N, P, prec, mode, w))
                    n += 1
                w = check_slab_c2c(N, P, prec, rng)
                print("slabC2C N=%s P=%d %s           worst rel-L2 %.2e" % (N, P, prec, w))
                n += 1
            for P, P1 in ((4, None), (8, None), (8, 2), (16, None)):
                if N[0] == 8 and P == 16:
                    continue
                for mode in ("Alltoall", "Alltoallw"):
                    for align in ("X", "Y"):
                        w = check_pencil(N, P, prec, mode, align, rng, P1=P1)
                        print("pencil%s N=%s P=%d P1=%s %s %-9s worst rel-L2 %.2e"
                              % (align, N, P, P1, prec, mode, w))
                        n += 1
                for align in ("X", "Y"):
                    w = check_pencil_n(N, P, prec, align, rng, P1=P1)
                    print("pencil%s N=%s P=%d P1=%s %s AlltoallN worst rel-L2 %.2e" % (align, N, P, P1, prec, w))
                    n += 1
    print("OK: %d configurations, oracle == reference" % n)


if __name__ == "__main__":
    sys.exit(main())
