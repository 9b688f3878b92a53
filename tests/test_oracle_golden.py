"""CPU: the oracle (oracle/mpifft_oracle.py) against the fixtures that the REAL
reference produced (tests/golden/*, written by oracle/refharness/make_golden.py).
This is what pins the oracle; the GPU parity tests then compare the HIP path
with the oracle and with the same fixtures."""
import json
import os

import numpy as np
import pytest

from oracle import mpifft_oracle as orc

N = [8, 16, 32]
TOL = {"double": 1e-13, "single": 2e-5}


@pytest.fixture(scope="module", params=["double", "single"])
def gold(request, golden_dir):
    return request.param, np.load(os.path.join(golden_dir, "ref_8x16x32_%s.npz" % request.param))


def test_layout_tables(golden_dir):
    table = json.load(open(os.path.join(golden_dir, "layouts.json")))
    assert len(table) > 200
    for row in table:
        Nn, P, r = row["N"], row["P"], row["rank"]
        if row["decomp"] == "slab":
            lay = orc.SlabLayout(Nn, P)
            cshape = lay.complex_shape()
            rs, cs = lay.real_local_slice(r), lay.complex_local_slice(r)
            rsp = lay.real_local_slice(r, 1.5)
        else:
            lay = orc.PencilLayout(Nn, P, row["P1_arg"], row["decomp"][-1])
            assert (lay.P1, lay.P2) == (row["P1"], row["P2"])
            assert lay.ranks(r) == (row["c0"], row["c1"])
            cshape = lay.complex_shape(r)
            rs, cs = lay.real_local_slice(r), lay.complex_local_slice(r)
            rsp = lay.real_local_slice(r, 1.5)
        assert list(lay.real_shape()) == row["real_shape"]
        assert list(cshape) == row["complex_shape"]
        assert list(lay.real_shape_padded()) == row["real_shape_padded"]
        sl = lambda s: [[int(x.start or 0), int(x.stop)] for x in s]
        assert sl(rs) == row["real_slice"]
        assert sl(cs) == row["complex_slice"]
        assert sl(rsp) == row["real_slice_padded"]


@pytest.mark.parametrize("P", [1, 2, 4])
def test_slab_r2c(gold, P):
    prec, g = gold
    lay = orc.SlabLayout(N, P)
    fus = orc.slab_r2c_forward(orc.scatter_real(g["A"], lay), N, prec)
    C = orc.gather_complex(fus, lay, fus[0].dtype)
    for mode in ("Alltoall", "Alltoallw"):
        assert orc.rel_l2(C, g["slab_P%d_%s_fwd" % (P, mode)]) < TOL[prec]
    back = orc.slab_r2c_backward(fus, N, prec)
    B = orc.gather_real(back, lay, back[0].dtype)
    assert orc.rel_l2(B, g["slab_P%d_Alltoallw_bwd" % P]) < TOL[prec]
    assert orc.rel_l2(B, g["A"]) < 10 * TOL[prec]


@pytest.mark.parametrize("P,P1", [(4, None), (8, None), (8, 2)])
@pytest.mark.parametrize("align", ["X", "Y"])
def test_pencil_r2c(gold, P, P1, align):
    prec, g = gold
    lay = orc.PencilLayout(N, P, P1, align)
    fus = orc.pencil_r2c_forward(orc.scatter_real(g["A"], lay), N, P1, align, prec)
    C = orc.gather_complex(fus, lay, fus[0].dtype)
    assert orc.rel_l2(C, g["pencil%s_P%d_P1%s_fwd" % (align, P, P1)]) < TOL[prec]
    back = orc.pencil_r2c_backward(fus, N, P1, align, prec)
    B = orc.gather_real(back, lay, back[0].dtype)
    assert orc.rel_l2(B, g["pencil%s_P%d_P1%s_bwd" % (align, P, P1)]) < TOL[prec]


@pytest.mark.parametrize("P", [1, 2])
def test_slab_padded(gold, P):
    prec, g = gold
    lay = orc.SlabLayout(N, P)
    ap = orc.slab_r2c_backward_padded(orc.scatter_complex(g["C0"], lay), N, prec)
    AP = orc.gather_real(ap, lay, ap[0].dtype, 1.5)
    assert orc.rel_l2(AP, g["slab_P%d_pad_bwd" % P]) < TOL[prec]
    cp = orc.slab_r2c_forward_padded(ap, N, prec)
    CP = orc.gather_complex(cp, lay, cp[0].dtype)
    assert orc.rel_l2(CP, g["slab_P%d_pad_fwd" % P]) < TOL[prec]
    assert orc.rel_l2(CP, g["C0"]) < 10 * TOL[prec]


@pytest.mark.parametrize("align", ["X", "Y"])
def test_pencil_padded(gold, align):
    prec, g = gold
    lay = orc.PencilLayout(N, 4, None, align)
    ap = orc.pencil_r2c_backward_padded(orc.scatter_complex(g["C0"], lay), N, None, align, prec)
    AP = orc.gather_real(ap, lay, ap[0].dtype, 1.5)
    assert orc.rel_l2(AP, g["pencil%s_P4_pad_bwd" % align]) < TOL[prec]
    cp = orc.pencil_r2c_forward_padded(ap, N, None, align, prec)
    CP = orc.gather_complex(cp, lay, cp[0].dtype)
    assert orc.rel_l2(CP, g["pencil%s_P4_pad_fwd" % align]) < TOL[prec]


@pytest.mark.parametrize("P", [1, 2])
def test_slab_c2c(gold, P):
    prec, g = gold
    lay = orc.SlabLayout(N, P, kind="C2C")
    fus = orc.slab_c2c_forward(orc.scatter_real(g["Ac"], lay), N, prec)
    C = np.zeros(lay.global_complex_shape(), dtype=fus[0].dtype)
    for r, p in enumerate(fus):
        C[lay.complex_local_slice(r)] = p
    assert orc.rel_l2(C, g["slabc2c_P%d_fwd" % P]) < TOL[prec]
    back = orc.slab_c2c_backward(fus, N, prec)
    B = orc.gather_real(back, lay, back[0].dtype)
    assert orc.rel_l2(B, g["slabc2c_P%d_bwd" % P]) < TOL[prec]


def test_forward_is_rfftn(gold):
    """The reference's own oracle (tests/test_FFT.py:66-85): distributed result ==
    serial rfftn on the same data, also with the reference's max-norm criterion."""
    prec, g = gold
    A = g["A"]
    B2 = np.fft.rfftn(A.astype(np.float64))
    for key in ("slab_P4_Alltoallw_fwd", "pencilX_P8_P1None_fwd", "pencilY_P4_P1None_fwd"):
        c = g[key]
        rtol = 1e-8 if prec == "double" else 1e-4
        assert np.all(np.abs((c - B2) / c.max()) < rtol)


@pytest.mark.parametrize("align", ["X", "Y"])
def test_pencil_c2c_extension_is_a_dft(align):
    """Pencil C2C has no reference counterpart (unpinned extension): the oracle for it is
    checked against the DFT definition, numpy.fft.fftn of the gathered array."""
    Nn = [8, 16, 32]
    rng = np.random.default_rng(3)
    A = rng.random(Nn) + 1j * rng.random(Nn)
    for P, P1 in ((4, None), (8, None), (8, 2)):
        lay = orc.PencilC2CLayout(Nn, P, P1, align)
        fus = orc.pencil_c2c_forward(orc.scatter_real(A, lay), Nn, P1, align)
        C = np.zeros(Nn, dtype=complex)
        for r, part in enumerate(fus):
            C[lay.complex_local_slice(r)] = part
        assert orc.rel_l2(C, np.fft.fftn(A)) < 1e-14


@pytest.mark.parametrize("P", [1, 2])
def test_slab_c2c_padded(gold, P):
    prec, g = gold
    lay = orc.SlabLayout(N, P, kind="C2C")
    cs = [np.ascontiguousarray(g["Cc"][lay.complex_local_slice(r)]) for r in range(P)]
    ap = orc.slab_c2c_backward_padded(cs, N, prec)
    cp = orc.slab_c2c_forward_padded(ap, N, prec)
    CP = np.zeros(N, dtype=cp[0].dtype)
    for r, part in enumerate(cp):
        CP[lay.complex_local_slice(r)] = part
    assert orc.rel_l2(CP, g["slabc2c_P%d_pad_fwd" % P]) < TOL[prec]
    if P == 2:
        AP = orc.gather_real(ap, lay, ap[0].dtype, 1.5)
        assert orc.rel_l2(AP, g["slabc2c_P2_pad_bwd"]) < TOL[prec]


def test_oracle_vs_reference_compiled_helpers():
    """oracle/_ref/ref_maths = the reference's own Cython loops (cython/maths.pyx), compiled by
    oracle/build_ref.py from /root/reference.  Bit-exact comparison of the oracle's restatements."""
    from oracle import build_ref
    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref not built (reference not mounted)")
    rng = np.random.default_rng(8)
    for dt in (np.complex128, np.complex64):
        P, Np0, Np1, Nf = 4, 3, 5, 9
        U = (rng.random((P, Np0, Np1, Nf)) + 1j * rng.random((P, Np0, Np1, Nf))).astype(dt)
        T = np.zeros((Np0, P * Np1, Nf), dtype=dt)
        ref.transpose_Uc(T, U, P, Np0, Np1, Nf)                      # maths.pyx:21-31
        assert np.array_equal(T, orc.slab_unpack(U))
        assert np.array_equal(orc.slab_pack(T, P), U)
        fu = (rng.random((6, 7, 5)) + 1j * rng.random((6, 7, 5))).astype(dt)
        mask = (rng.random(fu.shape) > 0.5).astype(np.uint8)
        want = orc.apply_mask(fu, mask).astype(dt)
        got = ref.dealias_filter(fu.copy(), mask)                     # maths.pyx:9-19
        assert np.array_equal(got, want)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4])
def test_line_2d(golden_dir, P, prec):
    """2-D class: the oracle's restatement of line.py against the reference's own outputs
    (tests/golden/line_16x48_*.npz, written by oracle/refharness/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, "line_16x48_%s.npz" % prec))
    Nl = [int(x) for x in g["N"]]
    lay = orc.LineLayout(Nl, P)
    key = lambda r, k: "P%d_r%d_%s" % (P, r, k)
    us = [g[key(r, "u")] for r in range(P)]
    crnd = [g[key(r, "crnd")] for r in range(P)]
    ups = [g[key(r, "up")] for r in range(P)]
    fwd = orc.line_r2c_forward(us, Nl, prec)
    bwd = orc.line_r2c_backward(crnd, Nl, prec)
    bp = orc.line_r2c_backward_padded(crnd, Nl, prec)
    cp = orc.line_r2c_forward_padded(ups, Nl, prec)
    for r in range(P):
        assert us[r].shape == lay.real_shape() and crnd[r].shape == lay.complex_shape(r)
        assert ups[r].shape == lay.real_shape_padded()
        assert orc.rel_l2(fwd[r], g[key(r, "fu")]) < TOL[prec]
        assert orc.rel_l2(bwd[r], g[key(r, "b")]) < TOL[prec]
        assert orc.rel_l2(bp[r], g[key(r, "bp")]) < TOL[prec]
        assert orc.rel_l2(cp[r], g[key(r, "cp")]) < TOL[prec]
