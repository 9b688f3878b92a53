"""GPU parity of the distributed classes against the oracle, the reference's
golden fixtures and numpy.fft: slab R2C / C2C and pencil R2CX / R2CY, P in
{1, 2, 4, 8}, both precisions, un-padded / 3/2-rule / 2/3-rule.  Mirrors
tests/test_FFT.py of the reference (test_FFT, test_FFT_padded, test_FFT_C2C)
with its anisotropic mesh N = [32, 64, 128]."""
import os

import numpy as np
import pytest

from gpu_util import L, TOL, cdtype, have_gpu, orc, rdtype, run_ranks

pytestmark = pytest.mark.gpu

NREF = [32, 64, 128]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not have_gpu():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")


def _slab_roundtrip(N, P, prec, mode, A):
    from mpifft4py_amd import Slab_R2C

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, prec, communication=mode)
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = np.zeros(F.complex_shape(), dtype=F.complex)
        r = F.fftn(a, c)
        assert r is c
        b = np.zeros(F.real_shape(), dtype=F.float)
        b = F.ifftn(c, b)
        return F.complex_local_slice(), c, F.real_local_slice(), b
    return run_ranks(P, body)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("mode", ["Alltoall", "Alltoallw"])
@pytest.mark.parametrize("P", [1, 2, 4, 8])
def test_slab_r2c(P, mode, prec):
    rng = np.random.default_rng(100 + P)
    A = rng.random(NREF).astype(rdtype(prec))
    B2 = np.fft.rfftn(A.astype(np.float64))
    lay = orc.SlabLayout(NREF, P)
    want = orc.slab_r2c_forward(orc.scatter_real(A, lay), NREF, prec)
    res = _slab_roundtrip(NREF, P, prec, mode, A)
    rtol = 1e-8 if prec == "double" else 1e-4
    for r, (cs, c, rs, b) in enumerate(res):
        assert orc.rel_l2(c, want[r]) < TOL[prec]
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        # the reference's own criterion (tests/test_FFT.py:85, 90)
        assert np.all(np.abs((c - B2[cs]) / c.max()) < rtol)
        assert np.all(np.abs((b - A[rs]) / b.max()) < rtol)
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("communication", ["Alltoall", "Alltoallw"])      # 'Alltoall' is the reference factory's default (pencil.py:1479)
@pytest.mark.parametrize("P,P1", [(4, None), (8, None), (8, 2), (16, None)])
def test_pencil_r2c(P, P1, align, prec, communication):
    from mpifft4py_amd import Pencil_R2C
    rng = np.random.default_rng(200 + P)
    A = rng.random(NREF).astype(rdtype(prec))
    B2 = np.fft.rfftn(A.astype(np.float64))
    lay = orc.PencilLayout(NREF, P, P1, align)
    want = orc.pencil_r2c_forward(orc.scatter_real(A, lay), NREF, P1, align, prec)

    def body(comm):
        F = Pencil_R2C(np.array(NREF), L, comm, prec, P1=P1, communication=communication, alignment=align)
        assert F.communication == communication
        assert (F.P1, F.P2) == (lay.P1, lay.P2)
        assert tuple(F.complex_shape()) == tuple(lay.complex_shape(comm.Get_rank()))
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.real_shape(), dtype=F.float))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for r, (cs, c, rs, b) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(c, want[r]) < TOL[prec], (r,)
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4])
def test_slab_c2c(P, prec):
    from mpifft4py_amd import Slab_C2C
    rng = np.random.default_rng(300 + P)
    A = (rng.random(NREF) + 1j * rng.random(NREF)).astype(cdtype(prec))
    B2 = np.fft.fftn(A.astype(np.complex128))

    def body(comm):
        F = Slab_C2C(np.array(NREF), L, comm, prec)
        assert F.global_shape() == tuple(NREF)
        a = np.ascontiguousarray(A[F.original_local_slice()])
        c = F.fftn(a, np.zeros(F.transformed_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.original_shape(), dtype=F.complex))
        return F.transformed_local_slice(), c, F.original_local_slice(), b
    for cs, c, rs, b in run_ranks(P, body):
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


def _padded_case(make, P, C0, prec):
    def body(comm):
        F = make(comm)
        c = np.ascontiguousarray(C0[F.complex_local_slice()])
        ap = F.ifftn(c, np.zeros(F.real_shape_padded(), dtype=F.float), dealias="3/2-rule")
        cp = F.fftn(ap, np.zeros(F.complex_shape(), dtype=F.complex), dealias="3/2-rule")
        return F.real_local_slice(padsize=1.5), ap, F.complex_local_slice(), cp
    return run_ranks(P, body)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4])
def test_slab_padded(P, prec):
    """test_FFT_padded of the reference (tests/test_FFT.py:159-205)."""
    from mpifft4py_amd import Slab_R2C
    rng = np.random.default_rng(400)
    N = NREF
    A = rng.random(N)
    C0 = np.fft.rfftn(A).astype(cdtype(prec))
    C0[N[0] // 2] = 0
    C0[:, N[1] // 2] = 0
    C0[:, :, -1] = 0
    lay = orc.SlabLayout(N, P)
    want = orc.slab_r2c_backward_padded(orc.scatter_complex(C0, lay), N, prec)
    res = _padded_case(lambda comm: Slab_R2C(np.array(N), L, comm, prec), P, C0, prec)
    for r, (rs, ap, cs, cp) in enumerate(res):
        assert orc.rel_l2(ap, want[r]) < 4 * TOL[prec]
        assert orc.rel_l2(cp, C0[cs]) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N", [[32, 64, 128], [64, 32, 1024], [16, 1024, 16], [1024, 8, 48], [20, 12, 44]])
def test_slab_padded_one_rank_line_aligned_intermediates(N, prec, monkeypatch):
    """One rank, fused 3/2-rule transforms (slab.py:250-268, 372-386): the plan's two intermediates keep their z rows on
    cache lines (plan.hip pad_pitch: 513 -> 520 bins), the inverse x pass writes pitched rows per y row, the forward x pass
    tiles the compact result and wraps its input columns (ColParams::in_wrap; [64,32,1024] and [1024,8,48] through the
    three-sub-transform kernels of 1536 in single precision).  Against the oracle, and bit for bit against the compact
    route (MFFT_PAD_ALIGN=0): same arithmetic per column, other addresses."""
    from mpifft4py_amd import SelfComm, Slab_R2C
    A = np.random.default_rng(sum(N)).random(N)
    C0 = np.fft.rfftn(A).astype(cdtype(prec))
    C0[N[0] // 2] = 0
    C0[:, N[1] // 2] = 0
    C0[:, :, -1] = 0
    want = orc.slab_r2c_backward_padded(orc.scatter_complex(C0, orc.SlabLayout(N, 1)), N, prec)[0]
    got = {}
    for mode in ("1", "0"):                         # forced on (the default takes it for double precision rows of 8 KiB only) / off
        monkeypatch.setenv("MFFT_PAD_ALIGN", mode)
        F = Slab_R2C(np.array(N), L, SelfComm(0), prec)
        c = C0.copy()
        up = F.ifftn(c, np.zeros(F.real_shape_padded(), dtype=rdtype(prec)), "3/2-rule")
        assert np.array_equal(c, C0)
        fu = F.fftn(up.copy(), np.zeros(F.complex_shape(), dtype=cdtype(prec)), "3/2-rule")
        assert orc.rel_l2(up, want) < 4 * TOL[prec], (mode, orc.rel_l2(up, want))
        assert orc.rel_l2(fu, C0) < 4 * TOL[prec], (mode, orc.rel_l2(fu, C0))
        got[mode] = (up.copy(), fu.copy())
    if prec == "double":
        assert np.array_equal(got["1"][0], got["0"][0]) and np.array_equal(got["1"][1], got["0"][1])
    else:
        # single precision: the two routes put different columns into the kernels' ragged-lanes path (per-element loads and
        # stores for the last, partial tile of a row), which hipcc contracts into fused multiply-adds differently from the
        # full-lanes path: 1-ulp differences in some rows ([32,64,128]: 12 of 48 x rows, max 3.6e-7; the other meshes: none)
        assert orc.rel_l2(got["1"][0], got["0"][0]) < 1e-6 and orc.rel_l2(got["1"][1], got["0"][1]) < 1e-6


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("P", [4, 8])
def test_pencil_padded(P, align, prec):
    from mpifft4py_amd import Pencil_R2C
    rng = np.random.default_rng(500)
    N = NREF
    A = rng.random(N)
    C0 = np.fft.rfftn(A).astype(cdtype(prec))
    C0[N[0] // 2] = 0
    C0[:, N[1] // 2] = 0
    C0[:, :, -1] = 0
    lay = orc.PencilLayout(N, P, None, align)
    want = orc.pencil_r2c_backward_padded(orc.scatter_complex(C0, lay), N, None, align, prec)
    res = _padded_case(lambda comm: Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw",
                                               alignment=align), P, C0, prec)
    for r, (rs, ap, cs, cp) in enumerate(res):
        assert orc.rel_l2(ap, want[r]) < 4 * TOL[prec]
        assert orc.rel_l2(cp, C0[cs]) < 4 * TOL[prec]


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("decomp,P", [("slab", 1), ("slab", 2), ("slab", 4), ("pencilX", 4), ("pencilY", 4),
                                      ("pencilX", 8), ("pencilY", 8)])
def test_padded_generic_data(decomp, P, prec, fused, monkeypatch):
    """3/2-rule on data that is NOT of the reference test's special form: a random padded real field forward
    (the Nyquist folds of slab.py:480-482 / pencil.py:364-379 matter) and a random spectrum backward, through
    the fused kernels (pad-on-load / truncate-on-store) and through the copy-based path (MFFT_NO_PAD_FUSION)."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    if not fused:
        monkeypatch.setenv("MFFT_NO_PAD_FUSION", "1")
    N = NREF
    rt, ct = rdtype(prec), cdtype(prec)
    rng = np.random.default_rng(510 + P)
    Ap = rng.random([int(1.5 * n) for n in N]).astype(rt)
    Cr = (rng.random((N[0], N[1], N[2] // 2 + 1)) - 0.5 + 1j * (rng.random((N[0], N[1], N[2] // 2 + 1)) - 0.5)).astype(ct)
    if decomp == "slab":
        lay = orc.SlabLayout(N, P)
        want_c = orc.slab_r2c_forward_padded(orc.scatter_real(Ap, lay, 1.5), N, prec)
        want_a = orc.slab_r2c_backward_padded(orc.scatter_complex(Cr, lay), N, prec)
        make = lambda comm: Slab_R2C(np.array(N), L, comm, prec)
    else:
        align = decomp[-1]
        lay = orc.PencilLayout(N, P, None, align)
        want_c = orc.pencil_r2c_forward_padded(orc.scatter_real(Ap, lay, 1.5), N, None, align, prec)
        want_a = orc.pencil_r2c_backward_padded(orc.scatter_complex(Cr, lay), N, None, align, prec)
        make = lambda comm: Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=align)

    def body(comm):
        F = make(comm)
        ap = np.ascontiguousarray(Ap[F.real_local_slice(padsize=1.5)])
        cp = F.fftn(ap, np.zeros(F.complex_shape(), dtype=ct), dealias="3/2-rule")
        c = np.ascontiguousarray(Cr[F.complex_local_slice()])
        a = F.ifftn(c, np.zeros(F.real_shape_padded(), dtype=rt), dealias="3/2-rule")
        return cp, a
    for r, (cp, a) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(cp, want_c[r]) < 4 * TOL[prec], (r, "forward")
        assert orc.rel_l2(a, want_a[r]) < 4 * TOL[prec], (r, "backward")


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4])
@pytest.mark.parametrize("N", [[1200, 24, 10], [24, 1200, 10], [1792, 16, 12], [16, 2048, 12], [1440, 16, 12], [16, 1440, 12]])
def test_wide_and_narrow_tiles_over_ranks(N, P, prec):
    """Round 5's tile shapes (registry.h col_wide: 16 columns per workgroup for 1200 in double precision; col_narrow_f32: 64-byte
    tiles, two workgroups per CU, for 1440 / 1792 / 2048 in single) on ragged column counts (10 / 12 columns: 6 or 7 bins) and
    on the split row maps of 2 and 4 ranks, against the oracle and numpy (the reference's own tolerance, tests/test_FFT.py:85)."""
    rng = np.random.default_rng(sum(N) + P)
    A = rng.random(N).astype(rdtype(prec))
    B2 = np.fft.rfftn(A.astype(np.float64))
    want = orc.slab_r2c_forward(orc.scatter_real(A, orc.SlabLayout(N, P)), N, prec)
    from mpifft4py_amd import Slab_R2C

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, prec)
        c = F.fftn(A[F.real_local_slice()].copy(), np.zeros(F.complex_shape(), dtype=cdtype(prec)))
        b = F.ifftn(c.copy(), np.zeros(F.real_shape(), dtype=rdtype(prec)))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for r, (cs, c, rs, b) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(c, want[r]) < TOL[prec]
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N", [[800, 8, 8], [8, 800, 8], [768, 8, 8], [8, 768, 8], [8, 8, 768], [1024, 8, 8], [8, 1024, 8], [8, 8, 1024], [256, 8, 8],
                               # padded images on the round-3 plans: 1280 -> 1920, 2560 -> 3840, 1600 -> 2400, 2000 -> 3000, 1000 -> 1500
                               [1280, 8, 8], [8, 1280, 8], [8, 8, 1280], [2560, 8, 8], [8, 8, 2560], [8, 1600, 8], [8, 8, 1600],
                               [2000, 8, 8], [8, 8, 2000], [8, 1000, 8], [8, 8, 1000], [8, 500, 8],
                               # round 6: 9 * 2^a meshes on the 27 * 2^a plans: 576 -> 864, 1152 -> 1728, 2304 -> 3456, 288 -> 432
                               [576, 8, 8], [8, 576, 8], [8, 8, 576], [1152, 8, 8], [8, 1152, 8], [8, 8, 1152], [8, 2304, 8], [8, 8, 2304],
                               [8, 288, 8],
                               # ... and the images of groups U and V: 720 -> 1080, 1440 -> 2160, 900 -> 1350, 1800 -> 2700, 432 -> 648, 864 -> 1296
                               [720, 8, 8], [8, 720, 8], [8, 8, 720], [8, 1440, 8], [8, 8, 1440], [900, 8, 8], [8, 8, 900], [8, 1800, 8], [8, 8, 1800],
                               [432, 8, 8], [8, 8, 432], [8, 864, 8], [8, 8, 864], [8, 8, 1728], [8, 8, 500], [8, 8, 300], [1920, 8, 8], [8, 2400, 8],
                               [672, 8, 8], [8, 672, 8], [8, 8, 672], [8, 8, 1344], [8, 336, 8]])
def test_padded_long_axes(N, prec):
    """3/2-rule with one LONG axis: the pad-on-load / truncate-on-store builds of the 1152- and 1536-point strided
    kernels (12 values per thread, register caps, 64-byte tiles in single precision: registry.h col_wgs) and the
    column-limited real kernels of those lengths, which the small meshes of the other tests never reach."""
    from mpifft4py_amd import Slab_R2C
    rt, ct = rdtype(prec), cdtype(prec)
    rng = np.random.default_rng(sum(N) + 17)
    Ap = rng.random([int(1.5 * n) for n in N]).astype(rt)
    Cr = (rng.random((N[0], N[1], N[2] // 2 + 1)) - 0.5 + 1j * (rng.random((N[0], N[1], N[2] // 2 + 1)) - 0.5)).astype(ct)
    lay = orc.SlabLayout(N, 1)
    want_c = orc.slab_r2c_forward_padded(orc.scatter_real(Ap, lay, 1.5), N, prec)
    want_a = orc.slab_r2c_backward_padded(orc.scatter_complex(Cr, lay), N, prec)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, prec)
        cp = F.fftn(Ap.copy(), np.zeros(F.complex_shape(), dtype=ct), dealias="3/2-rule")
        a = F.ifftn(Cr.copy(), np.zeros(F.real_shape_padded(), dtype=rt), dealias="3/2-rule")
        return cp, a
    for cp, a in run_ranks(1, body):
        assert orc.rel_l2(cp, want_c[0]) < 4 * TOL[prec], "forward"
        assert orc.rel_l2(a, want_a[0]) < 4 * TOL[prec], "backward"


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N", [[16, 256, 256], [12, 512, 512]])
def test_one_rank_padded_planes(N, prec, monkeypatch):
    """One rank, real data, a mesh whose own plane pitch N1 * (N2/2 + 1) elements is one the strided x pass reads slowly
    (2^a + 2^(a-7), 2^a + 2^(a-8): the power-of-two meshes 256 and 512; plan.hip p1_plane_pad, a switch: MFFT_P1_XPAD): the
    intermediate's planes lie some cache lines further apart, the forward x pass reads them out of place, the inverse starts with the y pass.
    Route asserted, results against numpy.fft (slab.py:366-370, 247-249), the 2/3-rule with the reference's filter
    (pruned passes) and with an arbitrary one (mask applied by the y pass, now the first one)."""
    from mpifft4py_amd import Slab_R2C, SelfComm
    rng = np.random.default_rng(77)
    ct, rt = cdtype(prec), rdtype(prec)
    monkeypatch.setenv("MFFT_P1_XPAD", "2")
    F = Slab_R2C(np.array(N), L, SelfComm(0), prec)
    assert F.plan_info("plane_pad") == 2 * 128 // np.dtype(ct).itemsize
    A = rng.random(N).astype(rt)
    fu = F.fftn(A, np.zeros(F.complex_shape(), dtype=ct))
    want = np.fft.rfftn(A.astype(np.float64), axes=(0, 1, 2))
    assert orc.rel_l2(fu, want) < 4 * TOL[prec], orc.rel_l2(fu, want)
    C = want.astype(ct)
    c_in = C.copy()
    u = F.ifftn(C, np.zeros(F.real_shape(), dtype=rt))
    assert np.array_equal(C, c_in)
    assert orc.rel_l2(u, A) < 4 * TOL[prec], orc.rel_l2(u, A)
    # the reference's own filter: pruned passes (they do not take the padded route, nothing to pad in rows that are skipped)
    u23 = F.ifftn(C, np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule")
    mask = np.broadcast_to(F.get_dealias_filter(), F.complex_shape())
    w23 = np.fft.irfftn(want * mask, s=N, axes=(0, 1, 2))
    assert orc.rel_l2(u23, w23) < 4 * TOL[prec], orc.rel_l2(u23, w23)
    # an arbitrary filter: applied while the FIRST inverse pass loads -- here the y pass
    M = (rng.random(F.complex_shape()) < 0.6).astype(np.uint8)
    F.dealias = M
    um = F.ifftn(C, np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule")
    assert np.array_equal(C, c_in)
    wm = np.fft.irfftn(want * M, s=N, axes=(0, 1, 2))
    assert orc.rel_l2(um, wm) < 4 * TOL[prec], orc.rel_l2(um, wm)


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("decomp,P,pipeline", [("slab", 1, 0), ("slab", 2, 1), ("slab", 4, 2), ("slab", 4, -2), ("pencilX", 4, 1),
                                               ("pencilX", 8, 2), ("pencilY", 4, 1), ("pencilY", 8, 2), ("c2c", 1, 0), ("c2c", 2, 1)])
def test_two_thirds_rule_mask_on_load(decomp, P, pipeline, prec, fused, monkeypatch):
    """The dealias mask applied by the first inverse pass while it reads the spectrum (ColFft PAD == 3, plan.hip
    fuse_mask) against the masked-copy path (MFFT_NO_MASK_FUSION=1) and against ifftn(fu * mask): blocking and
    pipelined exchanges (the kz-slice and row-batch pipelines read the caller's array at offsets), both precisions, the
    C2C class (on one GPU its inverse starts with the y pass).  The mask here is a random one: nothing in the kernels
    assumes the 2/3-rule's shape."""
    from mpifft4py_amd import Pencil_R2C, Slab_C2C, Slab_R2C
    if not fused:
        monkeypatch.setenv("MFFT_NO_MASK_FUSION", "1")
    N = [64, 64, 64] if decomp == "c2c" else NREF      # 64^3 complex128: power-of-two planes, the y-first route
    rng = np.random.default_rng(640 + P)
    ct = cdtype(prec)
    if decomp == "c2c":
        C = (rng.random(N) - 0.5 + 1j * (rng.random(N) - 0.5)).astype(ct)
        M = (rng.random(N) < 0.6).astype(np.uint8)
    else:
        C = np.fft.rfftn(rng.random(N)).astype(ct)
        M = (rng.random(C.shape) < 0.6).astype(np.uint8)

    def body(comm):
        if decomp == "slab":
            F = Slab_R2C(np.array(N), L, comm, prec, pipeline=pipeline)
        elif decomp == "c2c":
            F = Slab_C2C(np.array(N), L, comm, prec, pipeline=pipeline)
        else:
            F = Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=decomp[-1], pipeline=pipeline)
        sl = F.transformed_local_slice() if decomp == "c2c" else F.complex_local_slice()
        out_shape, out_t = (F.original_shape(), F.complex) if decomp == "c2c" else (F.real_shape(), F.float)
        F.dealias = np.ascontiguousarray(M[sl])          # the attribute the reference computes lazily (slab.py:237-240)
        c = np.ascontiguousarray(C[sl])
        c_in = c.copy()
        u = F.ifftn(c, np.zeros(out_shape, dtype=out_t), dealias="2/3-rule")
        assert np.array_equal(c, c_in)          # input spectrum untouched
        u_ref = F.ifftn((c * M[sl]).astype(ct), np.zeros(out_shape, dtype=out_t))
        return u, u_ref, (F.original_local_slice() if decomp == "c2c" else F.real_local_slice())
    # the oracle's arithmetic on the whole masked spectrum (numpy.fft: slab.py:249 / 237-245)
    CM = C.astype(np.complex128) * M
    want = np.fft.ifftn(CM) if decomp == "c2c" else np.fft.irfftn(CM, s=N, axes=(0, 1, 2))
    for u, u_ref, rsl in run_ranks(P, body):
        assert np.array_equal(u, u_ref)         # same kernels, same values: bit-identical
        assert orc.rel_l2(u, want[rsl]) < 4 * TOL[prec], orc.rel_l2(u, want[rsl])


def _pruned_cases():
    """Every (mesh, ranks, pipeline, precision) of the round-3 matrix: two meshes keep all of it (the one whose ranks 3 and 4
    own nothing but removed ky, and the large one), the others keep one case per rank count; the rest (96 of 176) is
    `slow` (tests/conftest.py: MFFT_TEST_SLOW=1)."""
    out = []
    for N in ([32, 32, 32], [48, 20, 36], [64, 128, 256], [16, 24, 10], [100, 36, 50], [8, 8, 8], [12, 6, 4], [32, 16, 512]):
        for P, pipeline in [(1, 1), (2, 1), (4, 1), (2, 0), (4, 2), (4, 3), (2, -2), (4, -3), (8, 1), (8, 0), (8, -2)]:
            for prec in ("double", "single"):
                full = N in ([32, 32, 32], [64, 128, 256]) or (P, pipeline) in [(1, 1), (2, 1), (8, 1)]
                out.append(pytest.param(N, P, pipeline, prec, marks=() if full else pytest.mark.slow,
                                        id="%s-%d-%d-%s" % ("x".join(map(str, N)), P, pipeline, prec)))
    return out


@pytest.mark.parametrize("N,P,pipeline,prec", _pruned_cases())
def test_two_thirds_rule_pruned(N, P, pipeline, prec, monkeypatch):
    """Real data, the reference's own dealias filter (three 1-D band conditions).  One GPU: the inverse does not load
    the removed rows, skips the tiles of removed columns and reads only the kept bins of every z row; P ranks: the x pass
    writes the kept kz bins only (zeros for removed ky), the exchange carries a2 / Nf of the bytes (plan.hip detect_band,
    ColFft PAD == 4).  Against ifftn(fu * dealias) through the plain kernels and against the same call with
    MFFT_NO_PRUNE=1 (the general masked-load path); the input spectrum stays untouched.  Over 8 ranks some ranks own
    nothing but removed ky ([32,32,32]: ranks 3 and 4, as ranks 3 and 4 of the 1024^3 BASELINE mesh do): they vote
    "compatible", adopt the others' band and contribute zeros; the ROUTE every rank took is asserted
    (mfft_plan_get_info "pruned_route"), and the gathered result is compared with the oracle's arithmetic,
    numpy.fft.irfftn(C * dealias) (slab.py:237-245)."""
    from mpifft4py_amd import Slab_R2C
    if N[0] % P or N[1] % P:
        pytest.skip("mesh does not divide")
    rng = np.random.default_rng(sum(N) + 23)
    ct, rt = cdtype(prec), rdtype(prec)
    C = (rng.random((N[0], N[1], N[2] // 2 + 1)) - 0.5 + 1j * (rng.random((N[0], N[1], N[2] // 2 + 1)) - 0.5)).astype(ct)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, prec, pipeline=pipeline)     # 1 blocking, 0 / > 1 kz slices, < 0 row batches: all pruned
        mask = np.broadcast_to(F.get_dealias_filter(), F.complex_shape())
        c0 = np.ascontiguousarray(C[F.complex_local_slice()])
        c = c0.copy()
        u = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule")
        assert np.array_equal(c, c0)
        u_ref = F.ifftn((c0 * mask).astype(ct), np.zeros(F.real_shape(), dtype=rt))
        os.environ["MFFT_NO_PRUNE"] = "1"        # read at every call; all ranks of this process switch together
        comm.barrier()
        u_gen = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule")
        comm.barrier()
        os.environ.pop("MFFT_NO_PRUNE", None)
        return u, u_ref, u_gen, mask, F.plan_info("pruned_route"), F.complex_local_slice(), F.real_local_slice()
    try:
        res = run_ranks(P, body)
    finally:
        os.environ.pop("MFFT_NO_PRUNE", None)
    M = np.zeros(C.shape, dtype=bool)
    for u, u_ref, u_gen, mask, route, csl, rsl in res:
        assert np.array_equal(u_gen, u_ref)
        # a different build of the same kernels (the compiler may contract other multiply-adds): round-off apart at most
        assert orc.rel_l2(u, u_ref) < 0.05 * TOL[prec], orc.rel_l2(u, u_ref)
        M[csl] = mask
    assert 0 < sum(int(r[3].sum()) for r in res) < C.size
    # the oracle's arithmetic on the gathered arrays
    want = np.fft.irfftn(C.astype(np.complex128) * M, s=N, axes=(0, 1, 2))
    for u, _, _, _, _, _, rsl in res:
        assert orc.rel_l2(u, want[rsl]) < 4 * TOL[prec], orc.rel_l2(u, want[rsl])
    # every rank took the same kind of route; a rank whose ky are all removed says so
    routes = [r[4] for r in res]
    assert all(r > 0 for r in routes) or all(r == 0 for r in routes), routes
    if N in ([32, 32, 32], [64, 128, 256]):
        assert all(r > 0 for r in routes), routes        # radix kernels exist: the pruned route must engage
    for (_, _, _, mask, route, _, _) in res:
        if route:
            assert route == (1 if mask.any() or P == 1 else 2), (route, int(mask.sum()))
    if N == [32, 32, 32] and P == 8:
        assert routes.count(2) == 2, routes              # ranks 3 and 4 own ky 12..19, all of it removed


@pytest.mark.parametrize("P", [1, 2])
@pytest.mark.parametrize("kind", ["ones", "zeros", "kx_band_only", "one_mode", "checker"])
def test_two_thirds_rule_edge_masks(kind, P):
    """Masks at the edges of what detect_band accepts: nothing removed (band form with empty bands), everything removed,
    a band along x only, a single kept mode, a mask that is no product of 1-D conditions -- all must equal ifftn(fu * mask)
    bit for bit or to round-off whichever route they take."""
    from mpifft4py_amd import Slab_R2C
    N = [32, 24, 40]
    C = np.fft.rfftn(np.random.default_rng(77).random(N))
    M = np.ones(C.shape, dtype=np.uint8)
    if kind == "zeros":
        M[:] = 0
    elif kind == "kx_band_only":
        M[10:20] = 0
    elif kind == "one_mode":
        M[:] = 0
        M[0, 0, 0] = 1
    elif kind == "checker":
        M = ((np.indices(C.shape).sum(axis=0) % 3) != 0).astype(np.uint8)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, "double", pipeline=1)
        sl = F.complex_local_slice()
        c = np.ascontiguousarray(C[sl])
        F.dealias = np.ascontiguousarray(M[sl])
        u = F.ifftn(c.copy(), np.zeros(F.real_shape()), dealias="2/3-rule")
        ur = F.ifftn(c * M[sl], np.zeros(F.real_shape()))
        return u, ur, F.real_local_slice()
    want = np.fft.irfftn(C * M, s=N, axes=(0, 1, 2))
    for u, ur, rsl in run_ranks(P, body):
        assert np.abs(u - ur).max() <= 1e-14 * max(np.abs(ur).max(), 1e-300) or np.array_equal(u, ur)
        assert np.abs(u - want[rsl]).max() <= 1e-13 * max(np.abs(want).max(), 1e-300), kind     # the oracle's arithmetic


@pytest.mark.parametrize("decomp,P", [("slab", 1), ("slab", 4), ("pencilY", 4)])
def test_two_thirds_rule_filter_edited_in_place(decomp, P):
    """The reference multiplies by `self.dealias` on every '2/3-rule' call (slab.py:237-245), so an IN-PLACE edit of the
    filter counts from the next call on.  Here the filter lives on the device: the classes fingerprint the host array at
    upload and re-upload -- collectively, after a vote -- when it changed.  Only ONE rank's block is edited (the others
    must follow through the vote), then the lazily built filter is replaced by a bigger one (> 4 MB: the sampled
    fingerprint) and edited by a band."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    N = [32, 64, 128]
    C = np.fft.rfftn(np.random.default_rng(41).random(N))

    def body(comm):
        F = (Slab_R2C(np.array(N), L, comm, "double") if decomp == "slab" else
             Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment="Y"))
        sl = F.complex_local_slice()
        c = np.ascontiguousarray(C[sl])
        out = []
        u0 = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy()      # builds F.dealias lazily
        m0 = np.broadcast_to(F.dealias, F.complex_shape()).copy()
        out.append((u0, m0))
        assert F.dealias.flags.writeable
        if comm.Get_rank() == P - 1:
            F.dealias[:, :, 3:] = 0                                             # in place, on one rank only
        u1 = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy()
        out.append((u1, np.broadcast_to(F.dealias, F.complex_shape()).copy()))
        F.dealias[...] = 1                                                      # "no dealiasing", still in place
        u2 = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy()
        out.append((u2, np.broadcast_to(F.dealias, F.complex_shape()).copy()))
        F.dealias_check = False                                                 # documented opt-out: edits are NOT seen
        F.dealias[...] = 0
        u3 = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy()
        out.append((u3, np.ones(F.complex_shape(), dtype=np.uint8)))
        F.dealias = F.dealias                                                   # ... until the attribute is assigned
        u4 = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy()
        out.append((u4, np.zeros(F.complex_shape(), dtype=np.uint8)))
        return out, sl, F.real_local_slice()
    res = run_ranks(P, body)
    for step in range(5):
        M = np.zeros(C.shape, dtype=np.uint8)
        for out, sl, _ in res:
            M[sl] = out[step][1]
        want = np.fft.irfftn(C * M, s=N, axes=(0, 1, 2))
        for out, _, rsl in res:
            assert np.abs(out[step][0] - want[rsl]).max() <= 1e-13 * max(np.abs(want).max(), 1.0), (step,)
    assert not np.array_equal(res[0][0][0][0], res[0][0][1][0]) or P == 1      # the other ranks' results changed too


@pytest.mark.parametrize("decomp,P", [("slab", 2), ("slab", 4), ("pencilX", 4)])
@pytest.mark.parametrize("vote_every", [None, 3])
def test_two_thirds_rule_filter_assigned_on_one_rank(decomp, P, vote_every):
    """The reference reads `self.dealias` per rank on every call (slab.py:237-245), so ONE rank may assign a new filter by
    itself.  The device upload is collective: the assigning rank must not enter it alone (that was a hang with the
    sampled vote) -- it raises its hand and all ranks upload at the next vote (every call by default; with
    `dealias_vote_every = n` at the latest n calls later, the old filter until then)."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    N = [16, 32, 32]
    C = np.fft.rfftn(np.random.default_rng(43).random(N))

    def body(comm):
        F = (Slab_R2C(np.array(N), L, comm, "double") if decomp == "slab" else
             Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment="X"))
        F.dealias_vote_every = vote_every
        sl = F.complex_local_slice()
        c = np.ascontiguousarray(C[sl])
        u0 = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy()      # first upload: every rank's first call
        m0 = np.broadcast_to(F.dealias, F.complex_shape()).copy()
        m1 = m0
        if comm.Get_rank() == 1:
            m1 = np.ones(F.complex_shape(), dtype=np.uint8)
            m1[..., -2:] = 0
            F.dealias = m1                                                       # assignment on ONE rank
        us = [F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule").copy() for _ in range(4)]
        return u0, m0, us, m1, sl, F.real_local_slice()
    res = run_ranks(P, body)
    M0 = np.zeros(C.shape, dtype=np.uint8)
    M1 = np.zeros(C.shape, dtype=np.uint8)
    for _, m0, _, m1, sl, _ in res:
        M0[sl] = m0
        M1[sl] = m1
    w0 = np.fft.irfftn(C * M0, s=N, axes=(0, 1, 2))
    w1 = np.fft.irfftn(C * M1, s=N, axes=(0, 1, 2))
    tol = 1e-13 * max(np.abs(w0).max(), 1.0)
    for u0, _, us, _, _, rsl in res:
        assert np.abs(u0 - w0[rsl]).max() <= tol
        assert np.abs(us[-1] - w1[rsl]).max() <= tol                              # everybody ends up with the new filter
        if vote_every is None:
            assert np.abs(us[0] - w1[rsl]).max() <= tol                           # ... at the very next call by default
        for u in us:                                                              # and never with a mixture of the two
            assert min(np.abs(u - w0[rsl]).max(), np.abs(u - w1[rsl]).max()) <= tol


def test_two_thirds_rule_large_filter_sampled_fingerprint():
    """A filter above 256 KiB is fingerprinted by 8192 samples: band and plane edits are seen."""
    from mpifft4py_amd import Slab_R2C
    from mpifft4py_amd import SelfComm
    N = [128, 128, 512]
    F = Slab_R2C(np.array(N), L, SelfComm(0), "single")
    assert np.prod(F.complex_shape()) > F._FULL_HASH_BYTES
    C = np.fft.rfftn(np.random.default_rng(5).random(N)).astype(np.complex64)
    u0 = F.ifftn(C, np.zeros(F.real_shape(), dtype=np.float32), dealias="2/3-rule").copy()
    F.dealias[:, 10:20, :] = 0
    M = np.broadcast_to(F.dealias, F.complex_shape())
    u1 = F.ifftn(C, np.zeros(F.real_shape(), dtype=np.float32), dealias="2/3-rule").copy()
    want = np.fft.irfftn(C.astype(np.complex128) * M, s=N, axes=(0, 1, 2))
    assert orc.rel_l2(u1, want) < 4 * TOL["single"]
    assert orc.rel_l2(u0, want) > 1e-3


@pytest.mark.parametrize("P", [1, 2])
@pytest.mark.parametrize("mode", ["sampled", "full"])
def test_two_thirds_rule_one_element_of_a_large_filter(mode, P):
    """ONE element of a 4 MB filter flipped in place (slab.py:237-245 reads the array on every call): the 8192-sample
    fingerprint cannot see it, the whole-array hash of every `dealias_full_every`-th call does -- the new result arrives
    within 2 x dealias_full_every (+ the vote period) further calls, with `dealias_check = "full"` at the next call, and
    on every rank although one rank's block was edited."""
    from mpifft4py_amd import DeviceArray, Slab_R2C
    from mpifft4py_amd._base import DistFFTBase
    N = [128 * P, 128, 512]
    C = np.fft.rfftn(np.random.default_rng(77).random(N)).astype(np.complex64)
    every = 8

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, "single")
        F.dealias_check = mode
        F.dealias_full_every = every
        sl = F.complex_local_slice()
        c = DeviceArray.from_numpy(np.ascontiguousarray(C[sl]))
        u = DeviceArray.empty(F.real_shape(), F.float)
        F.ifftn(c, u, dealias="2/3-rule")
        F.dealias = np.ascontiguousarray(np.broadcast_to(F.dealias, F.complex_shape())).astype(np.uint8)
        assert F.dealias.nbytes > 4 << 20 and F.dealias.nbytes > F._FULL_HASH_BYTES
        F.ifftn(c, u, dealias="2/3-rule")
        u0 = u.get().copy()
        m = F.dealias
        taken = set(DistFFTBase._samples(np.arange(m.size)).tolist())
        flat = m.reshape(-1)
        pos = next(i for i in range(5 * 257 + 3, m.size) if i not in taken and flat[i])   # a kept mode no sample looks at
        if comm.Get_rank() == P - 1:
            flat[pos] = 0
        calls = 0
        bound = 1 if mode == "full" else 2 * every + 1
        M = np.broadcast_to(F.dealias, F.complex_shape()).copy()
        while calls < bound:
            F.ifftn(c, u, dealias="2/3-rule")
            calls += 1
            if not np.array_equal(u.get(), u0):
                break
        return u.get().copy(), u0, M, sl, F.real_local_slice(), calls
    res = run_ranks(P, body)
    M = np.zeros(C.shape, dtype=np.uint8)
    for _, _, m, sl, _, _ in res:
        M[sl] = m
    assert M.size - int(M.sum()) > 0
    want = np.fft.irfftn(C.astype(np.complex128) * M, s=N, axes=(0, 1, 2))
    for u1, u0, _, _, rsl, calls in res:
        assert orc.rel_l2(u1, want[rsl]) < 4 * TOL["single"], (mode, calls)
        assert not np.array_equal(u1, u0), (mode, calls)
        assert calls <= (1 if mode == "full" else 2 * every + 1)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("decomp,P,pipeline", [("slab", 2, 0), ("pencilX", 4, 1), ("pencilX", 4, 4), ("pencilX", 8, 2), ("pencilX", 1, 1),
                                               ("pencilY", 4, 1), ("pencilY", 4, 4), ("pencilY", 8, 1), ("pencilY", 16, 2)])
def test_two_thirds_rule(decomp, P, pipeline, prec):
    """ifftn(dealias='2/3-rule') == ifftn of the masked spectrum (slab.py:237-245, pencil.py:455-462).  With the reference's
    own filter the pencils' first inverse pass takes the band kernel (three 1-D conditions instead of one mask byte per
    element; plan.hip detect_band_local): asserted through mfft_plan_get_info, result against the oracle's arithmetic."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    rng = np.random.default_rng(600)
    N = NREF
    ct, rt = cdtype(prec), rdtype(prec)
    C = np.fft.rfftn(rng.random(N)).astype(ct)

    def body(comm):
        F = (Slab_R2C(np.array(N), L, comm, prec, pipeline=pipeline) if decomp == "slab" else
             Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=decomp[-1], pipeline=pipeline,
                        allow_single=True))
        c = np.ascontiguousarray(C[F.complex_local_slice()])
        c_in = c.copy()
        u = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule")
        assert np.array_equal(c, c_in)          # input spectrum untouched
        mask = np.broadcast_to(F.get_dealias_filter(), F.complex_shape())
        u_ref = F.ifftn((c * mask).astype(ct), np.zeros(F.real_shape(), dtype=rt))
        if decomp != "slab":
            assert F.plan_info("local_band") == 1
        return u, u_ref, F.complex_local_slice(), mask, F.real_local_slice()
    kx = np.fft.fftfreq(N[0], 1. / N[0])
    ky = np.fft.fftfreq(N[1], 1. / N[1])
    kz = np.fft.rfftfreq(N[2], 1. / N[2])
    gmask = orc.dealias_mask(N, kx, ky, kz)
    want = np.fft.irfftn(C.astype(np.complex128) * gmask, s=N, axes=(0, 1, 2))       # the oracle's arithmetic with the oracle's mask
    for u, u_ref, cs, mask, rsl in run_ranks(P, body):
        assert orc.rel_l2(u, u_ref) < 0.05 * TOL[prec]
        assert np.array_equal(mask, gmask[cs])
        assert orc.rel_l2(u, want[rsl]) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
def test_golden_fixtures(prec, golden_dir):
    """Against the arrays the REAL reference produced (tests/golden, N = [8,16,32])."""
    from mpifft4py_amd import Pencil_R2C, Slab_C2C, Slab_R2C
    g = np.load(os.path.join(golden_dir, "ref_8x16x32_%s.npz" % prec))
    N = [8, 16, 32]
    A = g["A"]

    def gather(res, shape, dtype):
        G = np.zeros(shape, dtype=dtype)
        for sl, part in res:
            G[sl] = part
        return G

    for P in (1, 2, 4):
        def body(comm):
            F = Slab_R2C(np.array(N), L, comm, prec)
            c = F.fftn(np.ascontiguousarray(A[F.real_local_slice()]), np.zeros(F.complex_shape(), dtype=F.complex))
            return F.complex_local_slice(), c
        C = gather(run_ranks(P, body), g["slab_P%d_Alltoallw_fwd" % P].shape, cdtype(prec))
        assert orc.rel_l2(C, g["slab_P%d_Alltoallw_fwd" % P]) < TOL[prec]
    for P, P1 in ((4, None), (8, None), (8, 2)):
        for align in ("X", "Y"):
            def body(comm):
                F = Pencil_R2C(np.array(N), L, comm, prec, P1=P1, communication="Alltoallw", alignment=align)
                c = F.fftn(np.ascontiguousarray(A[F.real_local_slice()]), np.zeros(F.complex_shape(), dtype=F.complex))
                return F.complex_local_slice(), c
            key = "pencil%s_P%d_P1%s_fwd" % (align, P, P1)
            C = gather(run_ranks(P, body), g[key].shape, cdtype(prec))
            assert orc.rel_l2(C, g[key]) < TOL[prec], key
    # 3/2-rule, slab P = 2 and pencil P = 4
    C0 = g["C0"]

    def pad_body(make):
        def body(comm):
            F = make(comm)
            ap = F.ifftn(np.ascontiguousarray(C0[F.complex_local_slice()]),
                         np.zeros(F.real_shape_padded(), dtype=F.float), dealias="3/2-rule")
            return F.real_local_slice(padsize=1.5), ap
        return body
    AP = gather(run_ranks(2, pad_body(lambda comm: Slab_R2C(np.array(N), L, comm, prec))), g["slab_P2_pad_bwd"].shape, rdtype(prec))
    assert orc.rel_l2(AP, g["slab_P2_pad_bwd"]) < 4 * TOL[prec]
    for align in ("X", "Y"):
        AP = gather(run_ranks(4, pad_body(lambda comm: Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=align))),
                    g["pencil%s_P4_pad_bwd" % align].shape, rdtype(prec))
        assert orc.rel_l2(AP, g["pencil%s_P4_pad_bwd" % align]) < 4 * TOL[prec]
    # C2C
    Ac = g["Ac"]
    for P in (1, 2):
        def body(comm):
            F = Slab_C2C(np.array(N), L, comm, prec)
            c = F.fftn(np.ascontiguousarray(Ac[F.original_local_slice()]), np.zeros(F.transformed_shape(), dtype=F.complex))
            return F.transformed_local_slice(), c
        C = gather(run_ranks(P, body), g["slabc2c_P%d_fwd" % P].shape, cdtype(prec))
        assert orc.rel_l2(C, g["slabc2c_P%d_fwd" % P]) < TOL[prec]


def test_device_arrays_and_cubic_sizes():
    """Device-resident in/out (the benchmark path) at 64^3 and 128^3, P = 1 and 4."""
    from mpifft4py_amd import DeviceArray, Slab_R2C
    for n, P in ((64, 1), (128, 4), (256, 2)):
        N = [n, n, n]
        rng = np.random.default_rng(n)
        A = rng.random(N)
        B2 = np.fft.rfftn(A)

        def body(comm):
            F = Slab_R2C(np.array(N), L, comm, "double")
            u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
            fu = DeviceArray.empty(F.complex_shape(), F.complex)
            u2 = DeviceArray.empty(F.real_shape(), F.float)
            assert F.fftn(u, fu) is fu
            F.ifftn(fu, u2)
            F.sync()
            assert np.array_equal(u.get(), A[F.real_local_slice()])       # input untouched
            return F.complex_local_slice(), fu.get(), F.real_local_slice(), u2.get()
        for cs, c, rs, b in run_ranks(P, body):
            assert orc.rel_l2(c, B2[cs]) < 1e-10
            assert orc.rel_l2(b, A[rs]) < 1e-10


def test_errors_match_the_reference():
    from mpifft4py_amd import LocalGroup, Pencil_R2C, SelfComm, Slab_R2C
    N = np.array([8, 16, 32])
    with pytest.raises(AssertionError):
        Slab_R2C(np.array([8, 16]), L, SelfComm(0), "double")
    with pytest.raises(AssertionError):
        Slab_R2C(N, L, SelfComm(0), "half")
    with pytest.raises(AssertionError):        # pencil needs more than one rank (pencil.py:176)
        Pencil_R2C(N, L, SelfComm(0), "double")
    F = Slab_R2C(N, L, SelfComm(0), "double")
    with pytest.raises(AssertionError):
        F.fftn(np.zeros(F.real_shape()), np.zeros(F.complex_shape(), dtype=complex), dealias="bogus")
    with pytest.raises(AssertionError):
        F.fftn(np.zeros((3, 3, 3)), np.zeros(F.complex_shape(), dtype=complex))
    g = LocalGroup(3, devices=[0, 0, 0])
    try:
        with pytest.raises(RuntimeError):      # IOError("Number of cpus must be in ...") on every rank
            g.run(lambda comm: Slab_R2C(N, L, comm, "double"))
    finally:
        g.free()


@pytest.mark.parametrize("pipeline", [1, 2, 4, -2, -3, -4, -64])
def test_slab_exchange_pipeline(pipeline):
    """The exchange pipelines (compute stream + communication stream; positive: kz slices, negative: batches of
    local x rows) give the same numbers as the un-pipelined path."""
    from mpifft4py_amd import DeviceArray, Slab_R2C
    N = [64, 64, 128]
    P = 4
    rng = np.random.default_rng(77)
    A = rng.random(N)
    B2 = np.fft.rfftn(A)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, "double", pipeline=pipeline)
        u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
        fu = DeviceArray.empty(F.complex_shape(), F.complex)
        u2 = DeviceArray.empty(F.real_shape(), F.float)
        for _ in range(3):                      # back-to-back calls reuse the send/recv buffers
            F.fftn(u, fu)
            F.ifftn(fu, u2)
        F.sync()
        c = fu.get()
        m = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule")
        mref = F.ifftn(c * np.broadcast_to(F.get_dealias_filter(), c.shape), np.zeros(F.real_shape()))
        return F.complex_local_slice(), c, F.real_local_slice(), u2.get(), m, mref
    for cs, c, rs, b, m, mref in run_ranks(P, body):
        assert orc.rel_l2(c, B2[cs]) < 1e-10
        assert orc.rel_l2(b, A[rs]) < 1e-10
        assert orc.rel_l2(m, mref) < 1e-12


def test_rccl_communicator_single_rank():
    """RCCL is dlopen'ed and usable: unique id, ncclCommInitRank, broadcast, all-reduce
    (one rank is all a gpurun box has; the multi-rank wire is exercised by bench.py --gpus N)."""
    from mpifft4py_amd import Pencil_R2C
    from mpifft4py_amd.comm import MAX, DistComm, get_unique_id
    uid = get_unique_id()
    assert len(uid) == 128
    c = DistComm(1, 0, uid, 0)
    assert (c.Get_size(), c.Get_rank()) == (1, 0)
    c.barrier()
    assert c.allreduce(3.5) == 3.5
    assert c.allreduce(2.0, op=MAX) == 2.0
    # a pencil plan on the 1x1 grid drives comm->alltoallv (self chunks) on the RCCL communicator
    N = np.array([16, 32, 64])
    rng = np.random.default_rng(1)
    A = rng.random(N)
    for align in ("X", "Y"):
        F = Pencil_R2C(N, L, c, "double", communication="Alltoallw", alignment=align, allow_single=True)
        cc = F.fftn(A, np.zeros(F.complex_shape(), dtype=complex))
        assert orc.rel_l2(cc, np.fft.rfftn(A)) < 1e-10
        assert orc.rel_l2(F.ifftn(cc, np.zeros(F.real_shape())), A) < 1e-10
    c.free()


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("P,P1", [(4, None), (8, None), (8, 2)])
def test_pencil_c2c_extension(P, P1, align, prec):
    """Pencil C2C (BASELINE config 5; no reference implementation exists) against the
    oracle's restatement and numpy.fft.fftn."""
    from mpifft4py_amd import Pencil_C2C
    rng = np.random.default_rng(900 + P)
    A = (rng.random(NREF) + 1j * rng.random(NREF)).astype(cdtype(prec))
    B2 = np.fft.fftn(A.astype(np.complex128))
    lay = orc.PencilC2CLayout(NREF, P, P1, align)
    want = orc.pencil_c2c_forward(orc.scatter_real(A, lay), NREF, P1, align, prec)

    def body(comm):
        F = Pencil_C2C(np.array(NREF), L, comm, prec, P1=P1, alignment=align)
        assert F.global_shape() == tuple(NREF)
        a = np.ascontiguousarray(A[F.original_local_slice()])
        c = F.fftn(a, np.zeros(F.transformed_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.original_shape(), dtype=F.complex))
        return F.transformed_local_slice(), c, F.original_local_slice(), b
    for r, (cs, c, rs, b) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(c, want[r]) < TOL[prec]
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


@pytest.mark.parametrize("kind,N", [("c2c", [64, 256, 512]), ("r2c", [64, 256, 1024])])
@pytest.mark.parametrize("cls,P,pipeline", [("slab", 2, 1), ("slab", 2, 0), ("slab", 4, 0), ("pencilX", 4, 1), ("pencilX", 4, 0),
                                            ("pencilY", 4, 1), ("pencilY", 4, 0)])
def test_exchanged_layouts_with_padded_x_rows(cls, P, pipeline, kind, N):
    """Round 4: where the x rows of an exchanged layout lie a slow pitch apart (a power of two, 2^a + 2^(a-7..9): plan.hip
    xplane_pad / slice_pitch) the chunks carry one cache line between x rows and the x pass reads them out of place.
    Meshes on which that is the case for the slab (un-pipelined and kz slices), the x-aligned pencil (forward) and the
    y-aligned one (inverse), un-pipelined and pipelined, complex and real data -- asserted through the device-free
    schedule query -- against numpy.fft on the gathered array."""
    from mpifft4py_amd import Pencil_C2C, Pencil_R2C, Slab_C2C, Slab_R2C, _lib
    dec = {"slab": _lib.SLAB, "pencilX": _lib.PENCIL_X, "pencilY": _lib.PENCIL_Y}[cls]
    knd = _lib.C2C if kind == "c2c" else _lib.R2C
    Nf = N[2] if kind == "c2c" else N[2] // 2 + 1
    # the pad is there: some chunk of the exchange in front of the x pass is larger than the compact one
    if cls == "slab":
        if pipeline == 1:
            sc = _lib.exchange_schedule(N, P, 0, dec, 0, True, kind=knd)["scount"][0]
            assert sc == (N[0] // P) * ((N[1] // P) * Nf + 8) * 16
        else:
            pcs = _lib.exchange_pieces(N, P, 0, dec, 0, True, 0, kind=knd)
            assert len(pcs) == 4 and sum(pc["scount"][0] for pc in pcs) > (N[0] // P) * (N[1] // P) * Nf * 16
    else:
        fwd = cls == "pencilX"
        sc = _lib.exchange_schedule(N, P, 0, dec, 1, fwd, kind=knd)["scount"][0]
        q = Nf // 2
        rows, plane = (N[0] // 2, (N[1] // 2) * q)
        assert sc == rows * (plane + 8) * 16, (sc, rows, plane)
    rng = np.random.default_rng(77 + P)
    if kind == "c2c":
        A = rng.random(N) - 0.5 + 1j * (rng.random(N) - 0.5)
        B = np.fft.fftn(A)
    else:
        A = rng.random(N)
        B = np.fft.rfftn(A)

    def body(comm):
        if cls == "slab":
            F = (Slab_C2C if kind == "c2c" else Slab_R2C)(np.array(N), L, comm, "double", pipeline=pipeline)
        elif kind == "c2c":
            F = Pencil_C2C(np.array(N), L, comm, "double", alignment=cls[-1], pipeline=pipeline)
        else:
            F = Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment=cls[-1], pipeline=pipeline)
        isl = F.original_local_slice() if kind == "c2c" else F.real_local_slice()
        osl = F.transformed_local_slice() if kind == "c2c" else F.complex_local_slice()
        a = np.ascontiguousarray(A[isl])
        c = F.fftn(a, np.zeros(B[osl].shape, dtype=complex))
        b = F.ifftn(c, np.zeros(a.shape, dtype=a.dtype))
        c2 = F.fftn(a, np.zeros(B[osl].shape, dtype=complex))            # the work buffers have changed roles once
        return orc.rel_l2(c, B[osl]), orc.rel_l2(b, a), np.array_equal(c, c2)
    for e_f, e_b, same in run_ranks(P, body):
        assert e_f < 1e-10 and e_b < 1e-10 and same, (e_f, e_b, same)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align,P,P1,pipeline", [("X", 4, None, 1), ("X", 4, None, 0), ("X", 8, None, 0), ("X", 8, 2, 1),
                                                 ("Y", 4, None, 1), ("Y", 4, None, 0), ("Y", 8, None, 0), ("Y", 8, 2, 1)])
def test_pencil_forward_z_blocks_with_line_aligned_rows(align, P, P1, pipeline, prec):
    """Round 4: in the forward z-splitting exchange the rows of a chunk of 64 columns and more lie a whole number of cache
    lines apart in the y-aligned plans (plan.hip zrow_pitch, fft_kernels.h ZSplit pitch): the rank that holds the Nyquist
    column (129 / 65 columns here) runs its x pass off the 2^a + 2^(a-7) pitch and carries the row pitch through the second
    exchange; the x-aligned plans keep compact rows (same test, same meshes).  [32, 64, 512]: Nf = 257; asserted through the schedule query; against numpy.fft."""
    from mpifft4py_amd import Pencil_R2C, _lib
    N = [32, 64, 512]
    dec = _lib.PENCIL_X if align == "X" else _lib.PENCIL_Y
    s0 = _lib.exchange_schedule(N, P, P - 1, dec, 0, True, p1=P1 or 0, precision=prec)
    lay = orc.PencilLayout(N, P, P1, align)
    m, n = int(lay.N1[0]), int(lay.N2[1])
    es = 16 if prec == "double" else 8
    q_last = lay.complex_shape(P - 1)[2]
    per_line = 128 // es
    want_pitch = -(-q_last // per_line) * per_line if align == "Y" else q_last      # y-aligned plans only (plan.hip zrow_pitch)
    assert q_last % 2 == 1 and s0["rcount"][0] == m * n * want_pitch * es, (q_last, s0)
    rng = np.random.default_rng(31 + P)
    A = rng.random(N).astype(rdtype(prec))
    B = np.fft.rfftn(A.astype(np.float64))

    def body(comm):
        F = Pencil_R2C(np.array(N), L, comm, prec, P1=P1, communication="Alltoallw", alignment=align, pipeline=pipeline)
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.real_shape(), dtype=F.float))
        c2 = F.fftn(a, np.zeros(F.complex_shape(), dtype=F.complex))
        return orc.rel_l2(c, B[F.complex_local_slice()]), orc.rel_l2(b, a), np.array_equal(c, c2)
    for e_f, e_b, same in run_ranks(P, body):
        assert e_f < TOL[prec] and e_b < 4 * TOL[prec] and same, (e_f, e_b, same)


@pytest.mark.parametrize("kind", ["r2c", "c2c"])
@pytest.mark.parametrize("align,P1,pipeline", [("X", 4, 1), ("X", 4, 0), ("X", 1, 1), ("X", 1, 0), ("Y", 1, 1), ("Y", 1, 0), ("Y", 4, 1), ("Y", 4, 0)])
def test_one_dimensional_process_grids_with_padded_layouts(align, P1, pipeline, kind):
    """P x 1 and 1 x P grids (the C ABI takes them; bench.py times them) on meshes where the padded exchange layouts of
    round 4 are active: a group of one rank exchanges nothing, so the buffers change roles differently (plan.hip:
    zsolo / g2solo) -- x-aligned 4 x 1: no z exchange but a padded second one; y-aligned 1 x 4: a z exchange with pitched
    rows and no second one; and the two grids on which neither pad applies."""
    from mpifft4py_amd import Pencil_C2C, Pencil_R2C
    N = [64, 256, 1024] if kind == "r2c" else [64, 256, 512]
    P = 4
    rng = np.random.default_rng(5 + P1)
    if kind == "c2c":
        A = rng.random(N) - 0.5 + 1j * (rng.random(N) - 0.5)
        B = np.fft.fftn(A)
    else:
        A = rng.random(N)
        B = np.fft.rfftn(A)

    def body(comm):
        if kind == "c2c":
            F = Pencil_C2C(np.array(N), L, comm, "double", P1=P1, alignment=align, pipeline=pipeline, allow_odd_grid=True)
            isl, osl = F.original_local_slice(), F.transformed_local_slice()
        else:
            F = Pencil_R2C(np.array(N), L, comm, "double", P1=P1, communication="Alltoallw", alignment=align, pipeline=pipeline,
                           allow_odd_grid=True)
            isl, osl = F.real_local_slice(), F.complex_local_slice()
        a = np.ascontiguousarray(A[isl])
        c = F.fftn(a, np.zeros(B[osl].shape, dtype=complex))
        b = F.ifftn(c, np.zeros(a.shape, dtype=a.dtype))
        return orc.rel_l2(c, B[osl]), orc.rel_l2(b, a)
    for e_f, e_b in run_ranks(P, body):
        assert e_f < 1e-10 and e_b < 1e-10, (e_f, e_b)


@pytest.mark.parametrize("pipeline", [1, 0])
def test_pencil_c2c_y_with_pitched_z_chunks(pipeline):
    """The y-aligned pencil's row pitch (plan.hip zrow_pitch) on COMPLEX data: chunks of 72 complex64 columns (576 bytes)
    travel 80 apart; the contiguous-axis c2c kernel writes them (RowFft CHUNK through ZSplit)."""
    from mpifft4py_amd import Pencil_C2C, _lib
    N, P = [32, 64, 144], 4
    s0 = _lib.exchange_schedule(N, P, 0, _lib.PENCIL_Y, 0, True, kind=_lib.C2C, precision="single")
    assert s0["scount"][0] == (32 // 2) * (64 // 2) * 80 * 8, s0
    rng = np.random.default_rng(12)
    A = (rng.random(N) - 0.5 + 1j * (rng.random(N) - 0.5)).astype(np.complex64)
    B = np.fft.fftn(A.astype(np.complex128))

    def body(comm):
        F = Pencil_C2C(np.array(N), L, comm, "single", alignment="Y", pipeline=pipeline)
        a = np.ascontiguousarray(A[F.original_local_slice()])
        c = F.fftn(a, np.zeros(F.transformed_shape(), dtype=np.complex64))
        b = F.ifftn(c, np.zeros(F.original_shape(), dtype=np.complex64))
        return orc.rel_l2(c, B[F.transformed_local_slice()]), orc.rel_l2(b, a)
    for e_f, e_b in run_ranks(P, body):
        assert e_f < TOL["single"] and e_b < 4 * TOL["single"], (e_f, e_b)


CONFIG5_MESHES = [[2048, 64, 32], [64, 2048, 32], [32, 64, 2048], [4096, 32, 16]]


@pytest.mark.parametrize("N", CONFIG5_MESHES, ids=lambda n: "x".join(map(str, n)))
@pytest.mark.parametrize("cls,P,P1", [("pencilX", 4, None), ("pencilX", 8, None), ("pencilY", 4, None), ("pencilY", 8, None),
                                      ("pencilX", 8, 2), ("slab", 1, None), ("slab", 4, None), ("slab", 8, None)])
def test_config5_kernels_mid_size_vs_dft(cls, P, P1, N):
    """BASELINE config 5 (2048^3 complex64 pencil C2C) one size class down: meshes with ONE long axis of 2048 / 4096 so that
    the kernels only that config reaches at scale -- the fp32 strided kernels of length 2048 / 4096 (plans.h
    MFFT_COLPLANS_F32_C: 32 values per thread, 1024 threads) on whole 128-byte tiles, several tiles per XCD, the nt
    variants, the contiguous-axis c2c kernels of those lengths -- are compared bin by bin with numpy.fft.fftn of the
    gathered array (the full-size run checks properties only, a consistent permutation of bins
    would pass them).  Slab C2C (slab.py:743-772) on the same meshes."""
    from mpifft4py_amd import Pencil_C2C, Slab_C2C
    if cls != "slab":
        lay_p1 = P1 if P1 else {4: 2, 8: 4}[P]
        p2 = P // lay_p1
        if N[0] % lay_p1 or N[1] % p2 or ((N[1] % lay_p1 or N[2] % p2) if cls == "pencilX" else (N[0] % p2 or N[2] % lay_p1)):
            pytest.skip("mesh does not divide over the %dx%d grid" % (lay_p1, p2))
    elif N[0] % P or N[1] % P:
        pytest.skip("mesh does not divide")
    rng = np.random.default_rng(5000 + sum(N) + P)
    A = (rng.random(N) - 0.5 + 1j * (rng.random(N) - 0.5)).astype(np.complex64)
    B = np.fft.fftn(A.astype(np.complex128))

    def body(comm):
        F = (Slab_C2C(np.array(N), L, comm, "single") if cls == "slab" else
             Pencil_C2C(np.array(N), L, comm, "single", P1=P1, alignment=cls[-1], allow_single=True))
        a = np.ascontiguousarray(A[F.original_local_slice()])
        c = F.fftn(a, np.zeros(F.transformed_shape(), dtype=np.complex64))
        b = F.ifftn(c, np.zeros(F.original_shape(), dtype=np.complex64))
        return F.transformed_local_slice(), c, F.original_local_slice(), b
    res = run_ranks(P, body)
    G = np.zeros(N, dtype=np.complex64)
    for cs, c, rs, b in res:
        G[cs] = c
        assert orc.rel_l2(b, A[rs]) < 4 * TOL["single"]
    assert orc.rel_l2(G, B) < TOL["single"]
    # bin level: the largest single-bin deviation, relative to the rms magnitude of the spectrum
    assert np.abs(G - B).max() < 50 * TOL["single"] * np.sqrt(np.mean(np.abs(B) ** 2))


@pytest.mark.parametrize("N,P", [([48, 96, 80], 1), ([48, 96, 80], 2), ([24, 40, 12], 4), ([96, 20, 192], 4),
                                 ([6, 10, 8], 2), ([2, 2, 4], 1), ([2, 2, 4], 2), ([4, 4, 4], 4)])
def test_slab_non_power_of_two_and_tiny_meshes(N, P):
    """Ragged / 3- and 5-smooth / minimum-size meshes (the reference only requires P = 2^i <= N[0])."""
    from mpifft4py_amd import Slab_R2C
    rng = np.random.default_rng(sum(N))
    A = rng.random(N)
    B2 = np.fft.rfftn(A)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, "double")
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=complex))
        b = F.ifftn(c, np.zeros(F.real_shape()))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for cs, c, rs, b in run_ranks(P, body):
        assert orc.rel_l2(c, B2[cs]) < 1e-10
        assert orc.rel_l2(b, A[rs]) < 1e-10


@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("N,P", [([48, 96, 80], 4), ([16, 16, 8], 4), ([40, 24, 48], 4)])
def test_pencil_non_power_of_two_meshes(N, P, align):
    from mpifft4py_amd import Pencil_R2C
    rng = np.random.default_rng(sum(N) + 1)
    A = rng.random(N)
    B2 = np.fft.rfftn(A)

    def body(comm):
        F = Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment=align)
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=complex))
        b = F.ifftn(c, np.zeros(F.real_shape()))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for cs, c, rs, b in run_ranks(P, body):
        assert orc.rel_l2(c, B2[cs]) < 1e-10
        assert orc.rel_l2(b, A[rs]) < 1e-10


def test_unsupported_mesh_raises_cleanly():
    from mpifft4py_amd import SelfComm, Slab_R2C, _lib
    with pytest.raises(_lib.MfftError):          # beyond the supported lengths (1 ... 2^20: mfft_length_route)
        Slab_R2C(np.array([(1 << 20) + 2, 2, 2]), L, SelfComm(0), "double")
    with pytest.raises(_lib.MfftError):
        Slab_R2C(np.array([2, 2, (1 << 21) + 4]), L, SelfComm(0), "double")
    with pytest.raises(_lib.MfftError):          # odd real axis
        Slab_R2C(np.array([8, 8, 9]), L, SelfComm(0), "double")


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4])
def test_slab_c2c_padded(P, prec, golden_dir):
    """3/2-rule for the C2C class (the padded half of the reference's test_FFT_C2C,
    tests/test_FFT.py:213-258), against the oracle (pinned to the reference, including its
    different Nyquist handling for P == 1 and P > 1) and against the reference's own arrays."""
    from mpifft4py_amd import Slab_C2C
    rng = np.random.default_rng(700 + P)
    N = NREF
    A = (rng.random(N) + 1j * rng.random(N))
    C = np.fft.fftn(A).astype(cdtype(prec))
    lay = orc.SlabLayout(N, P, kind="C2C")
    cs_in = [np.ascontiguousarray(C[lay.complex_local_slice(r)]) for r in range(P)]
    want_ap = orc.slab_c2c_backward_padded(cs_in, N, prec)
    want_cp = orc.slab_c2c_forward_padded(want_ap, N, prec)

    def body(comm):
        F = Slab_C2C(np.array(N), L, comm, prec)
        r = comm.Get_rank()
        ap = F.ifftn(cs_in[r], np.zeros(F.original_shape_padded(), dtype=F.complex), dealias="3/2-rule")
        ap2 = F.ifftn(cs_in[r], np.zeros(F.original_shape_padded(), dtype=F.complex), dealias="3/2-rule")
        assert np.array_equal(ap, ap2)          # repeatable (the reference's cached work arrays are not)
        cp = F.fftn(ap, np.zeros(F.transformed_shape(), dtype=F.complex), dealias="3/2-rule")
        return ap, cp
    for r, (ap, cp) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(ap, want_ap[r]) < 4 * TOL[prec]
        assert orc.rel_l2(cp, want_cp[r]) < 4 * TOL[prec]
    # the reference's arrays at N = [8, 16, 32]
    g = np.load(os.path.join(golden_dir, "ref_8x16x32_%s.npz" % prec))
    if P <= 2:
        Ng = [8, 16, 32]
        Cc = g["Cc"]

        def body2(comm):
            F = Slab_C2C(np.array(Ng), L, comm, prec)
            c = np.ascontiguousarray(Cc[F.transformed_local_slice()])
            ap = F.ifftn(c, np.zeros(F.original_shape_padded(), dtype=F.complex), dealias="3/2-rule")
            cp = F.fftn(ap, np.zeros(F.transformed_shape(), dtype=F.complex), dealias="3/2-rule")
            return F.original_local_slice(padsize=1.5), ap, F.transformed_local_slice(), cp
        res = run_ranks(P, body2)
        CP = np.zeros(Ng, dtype=cdtype(prec))
        for _, _, cs, cp in res:
            CP[cs] = cp
        assert orc.rel_l2(CP, g["slabc2c_P%d_pad_fwd" % P]) < 4 * TOL[prec]
        if P == 2:
            AP = np.zeros([12, 24, 48], dtype=cdtype(prec))
            for rs, ap, _, _ in res:
                AP[rs] = ap
            assert orc.rel_l2(AP, g["slabc2c_P2_pad_bwd"]) < 4 * TOL[prec]


def test_local_group_does_not_hang_when_one_rank_fails():
    """A failing rank aborts the in-process group: its peers get an error instead of waiting forever."""
    import time
    from mpifft4py_amd import LocalGroup, Slab_R2C
    N = np.array([16, 16, 32])
    g = LocalGroup(2, devices=[0, 0])

    def body(comm):
        F = Slab_R2C(N, L, comm, "double")
        if comm.Get_rank() == 1:
            raise ValueError("boom")
        return F.fftn(np.zeros(F.real_shape()), np.zeros(F.complex_shape(), dtype=complex))
    t0 = time.time()
    with pytest.raises(RuntimeError, match="boom"):
        g.run(body)
    assert time.time() - t0 < 60


def test_config2_512_cubed_one_gpu():
    """BASELINE config 2: 512^3 fp64 slab R2C on one GPU, device-resident (single-GPU kernels, no exchange).  Forward
    spectrum against the host's pocketfft on the SAME input (rel-L2 <= 1e-10, and the reference's own max-norm
    criterion tests/test_FFT.py:85), round trip <= 1e-10, input preserved by both transforms."""
    import os
    import scipy.fft as sfft
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    N = np.array([512] * 3)
    F = Slab_R2C(N, L, SelfComm(0), "double")
    assert F.num_processes == 1 and tuple(F.real_shape()) == (512, 512, 512) and tuple(F.complex_shape()) == (512, 512, 257)
    A = np.random.default_rng(512).random(tuple(N))
    u = DeviceArray.from_numpy(A)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    F.fftn(u, fu)
    F.sync()
    c = fu.get()
    F.ifftn(fu, u2)
    F.sync()
    C = sfft.rfftn(A, workers=os.cpu_count())
    assert orc.rel_l2(c, C) < 1e-10                                           # forward vs pocketfft
    assert np.abs(c - C).max() / np.abs(C).max() < 1e-8                       # the reference's criterion
    assert orc.rel_l2(u2.get(), A) < 1e-10                                    # round trip
    assert np.array_equal(u.get(), A) and np.array_equal(fu.get(), c)        # inputs untouched


@pytest.mark.parametrize("N", [[500, 500, 500], [384, 384, 384], [1000, 250, 200], [96, 1152, 320], [320, 96, 2304]])
def test_round2_plans_against_pocketfft(N):
    """The lengths whose kernels changed in round 2 -- the 125*2^a plans (250 / 500 / 1000 / real 2000), 384 and 1152 with
    12 values per thread, the row kernels without LDS twiddles and with the split exchange (real 2304 = 1152 complex,
    640-complex rows of 1280 are in the stage tests) -- at sizes where whole workgroups and ragged tiles both occur,
    device-resident, against the host's pocketfft on the same input; plus the pruned 2/3-rule inverse against
    irfftn(C * dealias)."""
    import os
    import scipy.fft as sfft
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    F = Slab_R2C(np.array(N), L, SelfComm(0), "double")
    A = np.random.default_rng(sum(N)).random(tuple(N))
    u = DeviceArray.from_numpy(A)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    u3 = DeviceArray.empty(F.real_shape(), F.float)
    F.fftn(u, fu)
    F.ifftn(fu, u2)
    F.ifftn(fu, u3, dealias="2/3-rule")
    F.sync()
    C = sfft.rfftn(A, workers=os.cpu_count())
    assert orc.rel_l2(fu.get(), C) < 1e-10
    assert orc.rel_l2(u2.get(), A) < 1e-10
    mask = np.broadcast_to(F.get_dealias_filter(), F.complex_shape())
    want = sfft.irfftn(C * mask, s=tuple(N), workers=os.cpu_count())
    assert orc.rel_l2(u3.get(), want) < 1e-10


@pytest.mark.parametrize("N", [[480, 480, 480], [240, 720, 360], [900, 60, 1200], [90, 1440, 300], [150, 180, 3600], [960, 30, 120]])
def test_round3_plans_against_pocketfft(N):
    """The lengths with both 3 and 5 among their factors (plans.h groups L and M, 30 values per thread; chirp-z before
    round 3), every axis of the slab transform: 240 and 480 complex rows are the two contiguous-axis kernels hipcc
    miscompiled before fft_kernels.h row_thread_index (480 also as the real length 960 and 240 as 480) -- against the
    host's pocketfft on the same input, with the pruned 2/3-rule inverse and the 3/2-rule padded pair."""
    import os
    import scipy.fft as sfft
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    F = Slab_R2C(np.array(N), L, SelfComm(0), "double")
    A = np.random.default_rng(sum(N)).random(tuple(N))
    u = DeviceArray.from_numpy(A)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    u3 = DeviceArray.empty(F.real_shape(), F.float)
    F.fftn(u, fu)
    F.ifftn(fu, u2)
    F.ifftn(fu, u3, dealias="2/3-rule")
    F.sync()
    C = sfft.rfftn(A, workers=os.cpu_count())
    assert orc.rel_l2(fu.get(), C) < 1e-10
    assert orc.rel_l2(u2.get(), A) < 1e-10
    mask = np.broadcast_to(F.get_dealias_filter(), F.complex_shape())
    want = sfft.irfftn(C * mask, s=tuple(N), workers=os.cpu_count())
    assert orc.rel_l2(u3.get(), want) < 1e-10
    del u3, want
    if max(N) <= 1200:                               # padded lengths 3N/2 must have kernels: 1800 complex, real 3600
        up = DeviceArray.empty(F.real_shape_padded(), F.float)
        fu2 = DeviceArray.empty(F.complex_shape(), F.complex)
        F.ifftn(fu, up, dealias="3/2-rule")
        F.fftn(up, fu2, dealias="3/2-rule")
        F.sync()
        want_p = orc.slab_r2c_backward_padded([C], N, "double")[0]
        assert orc.rel_l2(up.get(), want_p) < 1e-10
        assert orc.rel_l2(fu2.get(), orc.slab_r2c_forward_padded([want_p], N, "double")[0]) < 1e-10


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("N,P", [([60, 120, 240], 4), ([120, 180, 480], 8), ([240, 60, 360], 4)])
def test_pencil_round3_lengths(N, P, align, prec):
    """Pencil transforms whose every axis is one of the 15-smooth radix lengths of round 3 (plans.h groups L, M): the z-chunk
    (CHUNK), column-limited (LIMIT) and band variants of those kernels, which the slab tests do not reach -- plain pair
    against numpy's rfftn, 2/3-rule against irfftn(C * mask), 3/2-rule pair against the oracle (pencil.py:511-632, 758-883)."""
    from mpifft4py_amd import Pencil_R2C
    rng = np.random.default_rng(sum(N) + P)
    ct, rt = cdtype(prec), rdtype(prec)
    A = rng.random(N)
    C = np.fft.rfftn(A)
    C0 = C.astype(ct).copy()
    C0[N[0] // 2] = 0
    C0[:, N[1] // 2] = 0
    C0[:, :, -1] = 0
    lay = orc.PencilLayout(N, P, None, align)
    want_pad = orc.pencil_r2c_backward_padded(orc.scatter_complex(C0, lay), N, None, align, prec)
    kx = np.fft.fftfreq(N[0], 1. / N[0])
    ky = np.fft.fftfreq(N[1], 1. / N[1])
    kz = np.fft.rfftfreq(N[2], 1. / N[2])
    want_23 = np.fft.irfftn(C.astype(ct).astype(np.complex128) * orc.dealias_mask(N, kx, ky, kz), s=N, axes=(0, 1, 2))

    def body(comm):
        F = Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=align)
        a = np.ascontiguousarray(A[F.real_local_slice()]).astype(rt)
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=ct))
        b = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt))
        cg = np.ascontiguousarray(C[F.complex_local_slice()]).astype(ct)
        u23 = F.ifftn(cg, np.zeros(F.real_shape(), dtype=rt), dealias="2/3-rule")
        c0 = np.ascontiguousarray(C0[F.complex_local_slice()])
        ap = F.ifftn(c0, np.zeros(F.real_shape_padded(), dtype=rt), dealias="3/2-rule")
        cp = F.fftn(ap, np.zeros(F.complex_shape(), dtype=ct), dealias="3/2-rule")
        return F.complex_local_slice(), F.real_local_slice(), c, b, u23, ap, cp, F.plan_info("zfuse"), F.plan_info("local_band")
    for r, (cs, rs, c, b, u23, ap, cp, zfuse, lband) in enumerate(run_ranks(P, body)):
        assert zfuse == 1 and lband == 1        # the fused z-chunk kernels and the band route ran, not their copy-based fallbacks
        assert orc.rel_l2(c, C[cs]) < 4 * TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]
        assert orc.rel_l2(u23, want_23[rs]) < 4 * TOL[prec]
        assert orc.rel_l2(ap, want_pad[r]) < 4 * TOL[prec]
        assert orc.rel_l2(cp, C0[cs]) < 4 * TOL[prec]


def test_full_size_1024_cubed():
    """BASELINE workload at full size: 1024^3 fp64, device-resident.  Forward spectrum against
    the host's pocketfft (scipy.fft, all cores) on the SAME input, round trip, input preserved.
    Needs ~45 GB of host RAM and ~35 GB of HBM; skipped if the host is too small."""
    import os
    import scipy.fft as sfft
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    try:
        avail_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
    except Exception:  # noqa: BLE001
        avail_kb = 0
    if avail_kb < 60 * 1024 * 1024:
        pytest.skip("host RAM too small for the full-size reference")
    N = np.array([1024] * 3)
    F = Slab_R2C(N, L, SelfComm(0), "double")
    u = DeviceArray.random(F.real_shape(), F.float, seed=7)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    A = u.get()
    F.fftn(u, fu)
    F.ifftn(fu, u2)
    F.sync()

    def rel(X, Y, step):
        num = den = 0.0
        for i in range(0, X.shape[0], step):
            d = X[i:i + step] - Y[i:i + step]
            num += float(np.vdot(d, d).real)
            den += float(np.vdot(Y[i:i + step], Y[i:i + step]).real)
        return (num / den) ** 0.5
    assert 0.0 <= A.min() and A.max() < 1.0 and abs(A.mean() - 0.5) < 1e-3      # U[0,1) synthetic input
    assert rel(u2.get(), A, 64) < 1e-10                                          # round trip
    assert np.array_equal(u.leading(0, 4).get(), A[:4])                          # input untouched
    C = sfft.rfftn(A, workers=os.cpu_count())
    assert rel(fu.get(), C, 32) < 1e-10                                          # forward vs pocketfft


def test_full_size_1024_cubed_eight_ranks():
    """BASELINE configs 3 and 4 at full size with the decomposition of an 8-GPU node: 1024^3 fp64 over 8 ranks
    (all on this GPU, exchanging by device copies) as slab (pipelined exchange), pencil X and pencil Y (4x2).
    Every rank's spectrum block against the host's pocketfft on the same global input, and the round trip."""
    import os
    import scipy.fft as sfft
    from mpifft4py_amd import DeviceArray, Pencil_R2C, Slab_R2C
    try:
        avail_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
    except Exception:  # noqa: BLE001
        avail_kb = 0
    if avail_kb < 60 * 1024 * 1024:
        pytest.skip("host RAM too small for the full-size reference")
    N = np.array([1024] * 3)
    A = np.random.default_rng(77).random(tuple(N))
    C = sfft.rfftn(A, workers=os.cpu_count())

    def rel(X, Y):
        d = X - Y
        return (float(np.vdot(d, d).real) / float(np.vdot(Y, Y).real)) ** 0.5

    def make(kind):
        def body(comm):
            if kind == "slab":
                F = Slab_R2C(N, L, comm, "double")
            else:
                F = Pencil_R2C(N, L, comm, "double", communication="Alltoallw", alignment=kind)
            u = DeviceArray.from_numpy(np.ascontiguousarray(A[F.real_local_slice()]))
            fu = DeviceArray.empty(F.complex_shape(), F.complex)
            u2 = DeviceArray.empty(F.real_shape(), F.float)
            F.fftn(u, fu)
            F.ifftn(fu, u2)
            F.sync()
            e1 = rel(fu.get(), C[F.complex_local_slice()])
            e2 = rel(u2.get(), A[F.real_local_slice()])
            return F.complex_shape(), e1, e2
        return body
    for kind in ("slab", "X", "Y"):
        res = run_ranks(8, make(kind))
        shapes = [r[0] for r in res]
        if kind == "slab":
            assert shapes == [(1024, 128, 513)] * 8
        elif kind == "X":
            assert shapes == [(1024, 256, 256)] * 4 + [(1024, 256, 257)] * 4
        else:
            assert shapes == [(512, 1024, 128)] * 3 + [(512, 1024, 129)] + [(512, 1024, 128)] * 3 + [(512, 1024, 129)]
        for _, e1, e2 in res:
            assert e1 < 1e-10 and e2 < 1e-10, (kind, e1, e2)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("P,P1", [(4, None), (8, None), (8, 2)])
def test_pencil_alltoalln(P, P1, align, prec):
    """communication='AlltoallN' (pencil.py:410-432, 647-668): the z-Nyquist column is neglected.
    Same protocol as the reference's test (tests/test_FFT.py:64-68): the input is first projected
    onto fields without that mode."""
    from mpifft4py_amd import Pencil_R2C
    rng = np.random.default_rng(800 + P)
    C = np.fft.rfftn(rng.random(NREF))
    C[:, :, -1] = 0
    A = np.fft.irfftn(C, s=NREF, axes=(0, 1, 2)).astype(rdtype(prec))
    lay = orc.PencilNLayout(NREF, P, P1, align)
    want = orc.pencil_r2c_forward_n(orc.scatter_real(A, lay), NREF, P1, align, prec)

    def body(comm):
        F = Pencil_R2C(np.array(NREF), L, comm, prec, P1=P1, communication="AlltoallN", alignment=align)
        assert tuple(F.complex_shape()) == tuple(lay.complex_shape(comm.Get_rank()))
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.real_shape(), dtype=F.float))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for r, (cs, c, rs, b) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(c, want[r]) < TOL[prec]
        assert orc.rel_l2(c, C[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


def test_hipgraph_replay_opt_in(monkeypatch):
    """MFFT_GRAPH=1: the kernel sequence of a (direction, buffers, dealias) combination is captured on its
    second use and replayed afterwards; results are identical and survive work-buffer growth."""
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    monkeypatch.setenv("MFFT_GRAPH", "1")
    N = np.array([32, 64, 16])
    rng = np.random.default_rng(21)
    A = rng.random(tuple(N))
    F = Slab_R2C(N, L, SelfComm(0), "double")
    u = DeviceArray.from_numpy(A)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    up = DeviceArray.empty(F.real_shape_padded(), F.float)
    fu2 = DeviceArray.empty(F.complex_shape(), F.complex)
    ref = np.fft.rfftn(A)
    for it in range(5):                      # direct, capture, replay, replay, replay
        F.fftn(u, fu)
        F.ifftn(fu, u2)
        F.sync()
        assert orc.rel_l2(fu.get(), ref) < 1e-10, it
        assert orc.rel_l2(u2.get(), A) < 1e-10, it
        if it == 2:                          # padded path grows the work buffers: graphs must be rebuilt
            F.ifftn(fu, up, "3/2-rule")
            F.fftn(up, fu2, "3/2-rule")
            F.sync()
    # new data through the same buffers goes through the replayed graph
    B = rng.random(tuple(N))
    u.set(B)
    F.fftn(u, fu)
    F.sync()
    assert orc.rel_l2(fu.get(), np.fft.rfftn(B)) < 1e-10


# ---- arbitrary lengths: chirp-z (Bluestein) kernels, csrc/fft_chirpz.h ---------------------------------
@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N,P", [([7, 9, 22], 1), ([36, 60, 100], 1), ([36, 60, 100], 2), ([36, 60, 100], 4),
                                 ([13, 17, 38], 1), ([130, 66, 258], 2), ([1025, 3, 6], 1), ([4, 1026, 14], 2)])
def test_slab_r2c_arbitrary_lengths(N, P, prec):
    """numpy/FFTW take any mesh (numpy_fft.py:25-107); lengths without a radix plan go through the chirp-z
    kernels with the same fused pack / unpack row maps."""
    from mpifft4py_amd import Slab_R2C
    rng = np.random.default_rng(sum(N))
    A = rng.random(N).astype(rdtype(prec))
    B2 = np.fft.rfftn(A.astype(np.float64))
    lay = orc.SlabLayout(N, P)
    want = orc.slab_r2c_forward(orc.scatter_real(A, lay), N, prec)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, prec)
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.real_shape(), dtype=F.float))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for r, (cs, c, rs, b) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(c, want[r]) < TOL[prec]
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


@pytest.mark.parametrize("align", ["X", "Y"])
@pytest.mark.parametrize("N,P", [([36, 60, 100], 4), ([28, 44, 72], 4), ([72, 56, 200], 8)])
def test_pencil_r2c_arbitrary_lengths(N, P, align):
    from mpifft4py_amd import Pencil_R2C
    rng = np.random.default_rng(sum(N) + 5)
    A = rng.random(N)
    B2 = np.fft.rfftn(A)

    def body(comm):
        F = Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment=align)
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=complex))
        b = F.ifftn(c, np.zeros(F.real_shape()))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for cs, c, rs, b in run_ranks(P, body):
        assert orc.rel_l2(c, B2[cs]) < 1e-10
        assert orc.rel_l2(b, A[rs]) < 1e-10


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N,P", [([15, 21, 25], 1), ([18, 14, 11], 2), ([36, 60, 49], 4)])
def test_slab_c2c_arbitrary_lengths(N, P, prec):
    from mpifft4py_amd.slab import C2C
    rng = np.random.default_rng(sum(N) + 9)
    A = (rng.random(N) + 1j * rng.random(N)).astype(cdtype(prec))
    B2 = np.fft.fftn(A.astype(np.complex128))

    def body(comm):
        F = C2C(np.array(N), L, comm, prec)
        a = np.ascontiguousarray(A[F.original_local_slice()])
        c = F.fftn(a, np.zeros(F.transformed_shape(), dtype=F.complex))
        b = F.ifftn(c, np.zeros(F.original_shape(), dtype=F.complex))
        return F.transformed_local_slice(), c, F.original_local_slice(), b
    for cs, c, rs, b in run_ranks(P, body):
        assert orc.rel_l2(c, B2[cs]) < TOL[prec]
        assert orc.rel_l2(b, A[rs]) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N", [[64, 64, 64], [24, 128, 64], [40, 32, 256], [16, 4096, 2], [12, 64, 2048]])
def test_slab_c2c_power_of_two_planes(N, prec):
    """One GPU, complex data whose y-z planes are a multiple of 64 KiB: the transforms go through an intermediate with
    padded planes (plan.hip plane_pad), the inverse in the order y, x, z.  The 2/3-rule inverse and the input's
    preservation ride along."""
    from mpifft4py_amd.slab import C2C
    rng = np.random.default_rng(sum(N) + 13)
    A = (rng.random(N) + 1j * rng.random(N)).astype(cdtype(prec))
    B2 = np.fft.fftn(A.astype(np.complex128))

    def body(comm):
        F = C2C(np.array(N), L, comm, prec)
        c = F.fftn(A.copy(), np.zeros(F.transformed_shape(), dtype=F.complex))
        c_in = c.copy()
        b = F.ifftn(c_in, np.zeros(F.original_shape(), dtype=F.complex))
        assert np.array_equal(c_in, c)
        return c, b
    for c, b in run_ranks(1, body):
        assert orc.rel_l2(c, B2) < TOL[prec]
        assert orc.rel_l2(b, A) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("N", [[8, 256, 30], [16, 512, 62], [4, 1024, 14]])
def test_slab_r2c_power_of_two_planes(N, prec):
    """Real data whose half-spectrum planes are a multiple of 64 KiB (N2 = 2 mod 4 makes N2/2+1 even): the forward
    transform takes the padded-plane route too (y out of place into the work buffer, x back into the result)."""
    from mpifft4py_amd import Slab_R2C
    rng = np.random.default_rng(sum(N) + 29)
    A = rng.random(N).astype(rdtype(prec))
    B2 = np.fft.rfftn(A.astype(np.float64))

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, prec)
        c = F.fftn(A.copy(), np.zeros(F.complex_shape(), dtype=F.complex))
        b = F.ifftn(c.copy(), np.zeros(F.real_shape(), dtype=F.float))
        d = F.ifftn(c.copy(), np.zeros(F.real_shape(), dtype=F.float), dealias="2/3-rule")
        mask = np.broadcast_to(F.get_dealias_filter(), F.complex_shape())
        d_ref = F.ifftn((c * mask).astype(F.complex), np.zeros(F.real_shape(), dtype=F.float))
        return c, b, d, d_ref
    for c, b, d, d_ref in run_ranks(1, body):
        assert orc.rel_l2(c, B2) < TOL[prec]
        assert orc.rel_l2(b, A) < 4 * TOL[prec]
        assert orc.rel_l2(d, d_ref) < 0.05 * TOL[prec]


@pytest.mark.parametrize("N,P", [([24, 40, 20], 1), ([24, 40, 20], 2), ([12, 28, 36], 1), ([40, 24, 28], 4),
                                 ([96, 48, 192], 1), ([96, 48, 192], 2), ([24, 48, 96], 4)])
def test_slab_padded_arbitrary_lengths(N, P):
    """3/2-rule on meshes whose padded lengths have no radix plan (60, 30, 42, 54: chirp-z, copy-based pad)
    and on 3*2^a meshes, whose padded lengths 9*2^a do (fused pad / truncate)."""
    from mpifft4py_amd import Slab_R2C
    rng = np.random.default_rng(sum(N) + 11)
    A = rng.random(N)
    lay = orc.SlabLayout(N, P)
    fu_ranks = orc.slab_r2c_forward(orc.scatter_real(A, lay), N, "double")
    want_up = orc.slab_r2c_backward_padded(fu_ranks, N, "double")
    want_fu = orc.slab_r2c_forward_padded(want_up, N, "double")

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, "double")
        r = comm.Get_rank()
        up = F.ifftn(fu_ranks[r].copy(), np.zeros(F.real_shape_padded()), "3/2-rule")
        fu = F.fftn(up, np.zeros(F.complex_shape(), dtype=complex), "3/2-rule")
        return up, fu
    for r, (up, fu) in enumerate(run_ranks(P, body)):
        assert orc.rel_l2(up, want_up[r]) < 1e-10
        assert orc.rel_l2(fu, want_fu[r]) < 1e-10


@pytest.mark.parametrize("N,P,decomp", [([4100, 8, 6], 1, "slab"), ([8, 5000, 6], 2, "slab"), ([8, 4, 8194], 2, "slab"),
                                        ([8, 4, 8190], 1, "slab"), ([16, 4104, 12], 4, "pencilY"), ([4098, 8, 12], 4, "pencilX"),
                                        ([6, 10, 9002], 1, "slab"), ([4608, 4, 6], 1, "slab"), ([4, 6144, 10240], 2, "slab")])
def test_meshes_with_an_axis_beyond_the_radix_plans(N, P, decomp):
    """numpy / FFTW take every mesh (numpy_fft.py:25-107); axes of 4100, 5000, 8194 and 9002 (real) points go through
    the scratch-buffer fallback (csrc/bigfft.hip) inside the slab and pencil plans -- plain transforms, the 2/3-rule through the
    masked copy and the 3/2-rule through the copy-based pad --, 4608 / 6144 / 10240 (real) through the radix plans of round 5."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    A = np.random.default_rng(sum(N)).random(N)
    B = np.fft.rfftn(A)

    def body(comm):
        F = (Slab_R2C(np.array(N), L, comm, "double") if decomp == "slab" else
             Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment=decomp[-1]))
        a = np.ascontiguousarray(A[F.real_local_slice()])
        c = F.fftn(a, np.zeros(F.complex_shape(), dtype=complex))
        b = F.ifftn(c, np.zeros(F.real_shape()))
        m = F.ifftn(c, np.zeros(F.real_shape()), dealias="2/3-rule")
        mask = np.broadcast_to(F.dealias, F.complex_shape()).copy()
        return F.complex_local_slice(), c, F.real_local_slice(), b, m, mask
    res = run_ranks(P, body)
    M = np.zeros(B.shape, dtype=np.uint8)
    for cs, c, rs, b, m, mask in res:
        M[cs] = mask
    want_m = np.fft.irfftn(B * M, s=N, axes=(0, 1, 2))
    for cs, c, rs, b, m, mask in res:
        assert orc.rel_l2(c, B[cs]) < 1e-10, orc.rel_l2(c, B[cs])
        assert orc.rel_l2(b, A[rs]) < 1e-10
        assert orc.rel_l2(m, want_m[rs]) < 1e-10


@pytest.mark.parametrize("N,P", [([1, 8, 8], 1), ([8, 1, 8], 1), ([1, 1, 8], 1), ([2, 1, 4], 1), ([1, 16, 2], 1),
                                 ([4, 4, 2], 2), ([1, 1, 2], 1)])
def test_meshes_with_unit_axes(N, P):
    """Length-1 axes (a transform of length 1 is a copy) and the shortest real axis."""
    from mpifft4py_amd import Slab_R2C
    A = np.random.default_rng(1).random(N)
    B = np.fft.rfftn(A)

    def body(comm):
        F = Slab_R2C(np.array(N), L, comm, "double")
        c = F.fftn(np.ascontiguousarray(A[F.real_local_slice()]), np.zeros(F.complex_shape(), dtype=complex))
        b = F.ifftn(c, np.zeros(F.real_shape()))
        return F.complex_local_slice(), c, F.real_local_slice(), b
    for cs, c, rs, b in run_ranks(P, body):
        assert orc.rel_l2(c, B[cs]) < 1e-12 and orc.rel_l2(b, A[rs]) < 1e-12


@pytest.mark.parametrize("N", [[1, 8, 5], [8, 1, 3], [1, 1, 7], [3, 1, 1]])
def test_c2c_meshes_with_unit_axes(N):
    from mpifft4py_amd import SelfComm
    from mpifft4py_amd.slab import C2C
    A = np.random.default_rng(2).random(N) + 1j * np.random.default_rng(3).random(N)
    F = C2C(np.array(N), L, SelfComm(0), "double")
    c = F.fftn(A.copy(), np.zeros(F.transformed_shape(), dtype=complex))
    b = F.ifftn(c, np.zeros(F.original_shape(), dtype=complex))
    assert orc.rel_l2(c, np.fft.fftn(A)) < 1e-12 and orc.rel_l2(b, A) < 1e-12


@pytest.mark.parametrize("ps", [2.0, 1.25, 1.75])
@pytest.mark.parametrize("kind,N,P", [("slab", [16, 32, 24], 1), ("slab", [16, 32, 24], 2), ("slab", [32, 16, 64], 4),
                                      ("X", [32, 32, 64], 4), ("Y", [32, 32, 64], 4)])
def test_other_pad_factors(kind, N, P, ps):
    """`padsize` other than 1.5 (constructor argument of every class, slab.py:68, pencil.py:168): copy-based pad /
    truncate, padded lengths of any kind (28, 30, 56, ...)."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    A = np.random.default_rng(3).random(N)
    if kind == "slab":
        lay = orc.SlabLayout(N, P, padsize=ps)
        fus = orc.slab_r2c_forward(orc.scatter_real(A, lay), N, "double")
        wb = orc.slab_r2c_backward_padded(fus, N, "double", ps)
        wc = orc.slab_r2c_forward_padded(wb, N, "double", ps)
        make = lambda c: Slab_R2C(np.array(N), L, c, "double", padsize=ps)
    else:
        lay = orc.PencilLayout(N, P, None, kind)
        fus = orc.pencil_r2c_forward(orc.scatter_real(A, lay), N, None, kind, "double")
        wb = orc.pencil_r2c_backward_padded(fus, N, None, kind, "double", ps)
        wc = orc.pencil_r2c_forward_padded(wb, N, None, kind, "double", ps)
        make = lambda c: Pencil_R2C(np.array(N), L, c, "double", communication="Alltoallw", alignment=kind, padsize=ps)

    def body(c):
        F = make(c)
        r = c.Get_rank()
        b = F.ifftn(fus[r].copy(), np.zeros(wb[r].shape), "3/2-rule")
        cc = F.fftn(b, np.zeros(wc[r].shape, dtype=complex), "3/2-rule")
        return orc.rel_l2(b, wb[r]), orc.rel_l2(cc, wc[r])
    for e1, e2 in run_ranks(P, body):
        assert e1 < 1e-10 and e2 < 1e-10


def _pipeline_cases(grids):
    """grids x depths x precisions; sixteen virtual ranks on one device take a second per case: depths 3 and 16 of the
    16-rank grid are `slow` (they run at 4 and 8 ranks), depths 2 and 4 stay."""
    return [pytest.param(P, P1, depth, prec, marks=pytest.mark.slow if (P == 16 and depth in (3, 16)) else (),
                         id="%d-%s-%d-%s" % (P, P1, depth, prec))
            for (P, P1) in grids for depth in (2, 3, 4, 16) for prec in ("double", "single")]


@pytest.mark.parametrize("P,P1,depth,prec", _pipeline_cases([(4, None), (8, None), (8, 2), (16, None)]))
def test_pencil_y_exchange_pipeline(P, P1, depth, prec):
    """Exchange pipeline of the y-aligned pencil (the reference class's default alignment; pencil.py:730-754, 483-507):
    z stage and z-splitting exchange in batches of local x rows, then the x transform, then the x-chunk exchange and the
    y transform in batches of the rows owned afterwards.  Same numbers as the un-pipelined path, R2C plain / 2/3-rule
    and C2C; the default depth (pipeline=0) is pipelined too."""
    from mpifft4py_amd.pencil import C2CY, R2CY
    N = [32, 64, 128]
    rt, ct = rdtype(prec), cdtype(prec)
    rng = np.random.default_rng(950 + P + depth)
    A = rng.random(N).astype(rt)
    Ac = (rng.random(N) + 1j * rng.random(N)).astype(ct)

    def body(comm):
        res = []
        for pipe in (1, depth, 0):
            F = R2CY(np.array(N), L, comm, prec, P1=P1, communication="Alltoallw", pipeline=pipe)
            a = np.ascontiguousarray(A[F.real_local_slice()])
            c = F.fftn(a, np.zeros(F.complex_shape(), dtype=ct))
            b = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt))
            b23 = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt), "2/3-rule")
            G = C2CY(np.array(N), L, comm, prec, P1=P1, pipeline=pipe)
            ac = np.ascontiguousarray(Ac[G.original_local_slice()])
            cc = G.fftn(ac, np.zeros(G.transformed_shape(), dtype=ct))
            bc = G.ifftn(cc, np.zeros(G.original_shape(), dtype=ct))
            res.append((c, b, b23, cc, bc, F.complex_local_slice(), F.real_local_slice(), ac,
                        sorted(k for k in F.stage_times())))
        return res
    B2 = np.fft.rfftn(A.astype(np.float64))
    for plain, piped, dflt in run_ranks(P, body):
        for other in (piped, dflt):
            for x, y in zip(plain[:5], other[:5]):
                assert np.array_equal(x, y)                   # same kernels, same order of operations per element
        assert orc.rel_l2(piped[0], B2[piped[5]]) < TOL[prec]
        assert orc.rel_l2(piped[1], A[piped[6]]) < 4 * TOL[prec]
        assert orc.rel_l2(piped[4], piped[7]) < 4 * TOL[prec]


@pytest.mark.parametrize("P,P1,depth,prec", _pipeline_cases([(4, None), (8, None), (8, 2), (16, None), (2, 1), (4, 1), (8, 1), (2, 2), (4, 4)]))
def test_pencil_x_exchange_pipeline(P, P1, depth, prec):
    """Opt-in exchange pipeline of the x-aligned pencil (batches of local x rows through both exchanges, compute and
    communication streams): same numbers as the un-pipelined path, R2C plain / 2/3-rule and C2C.  The 1 x P2, P1 x 1
    and odd grids are not offered by the reference (pencil.py:204-208) but by the C ABI (any P1 dividing P): on a
    1 x P2 grid the inverse's z exchange must not deliver into the buffer the y transforms of later batches read."""
    from mpifft4py_amd.pencil import C2CX, R2CX
    N = [32, 64, 128]
    rt, ct = rdtype(prec), cdtype(prec)
    rng = np.random.default_rng(900 + P + depth)
    A = rng.random(N).astype(rt)
    Ac = (rng.random(N) + 1j * rng.random(N)).astype(ct)

    def body(comm):
        res = []
        for pipe in (1, depth):                               # 1 = no pipeline (0 would be the default depth, 4)
            F = R2CX(np.array(N), L, comm, prec, P1=P1, communication="Alltoallw", allow_single=True, pipeline=pipe,
                     allow_odd_grid=True)
            a = np.ascontiguousarray(A[F.real_local_slice()])
            c = F.fftn(a, np.zeros(F.complex_shape(), dtype=ct))
            b = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt))
            b23 = F.ifftn(c, np.zeros(F.real_shape(), dtype=rt), "2/3-rule")
            G = C2CX(np.array(N), L, comm, prec, P1=P1, allow_single=True, pipeline=pipe, allow_odd_grid=True)
            ac = np.ascontiguousarray(Ac[G.original_local_slice()])
            cc = G.fftn(ac, np.zeros(G.transformed_shape(), dtype=ct))
            bc = G.ifftn(cc, np.zeros(G.original_shape(), dtype=ct))
            res.append((c, b, b23, cc, bc, F.complex_local_slice(), F.real_local_slice(), ac))
        return res
    B2 = np.fft.rfftn(A.astype(np.float64))
    for plain, piped in run_ranks(P, body):
        for x, y in zip(plain[:5], piped[:5]):
            assert np.array_equal(x, y)                       # same kernels, same order of operations per element
        assert orc.rel_l2(piped[0], B2[piped[5]]) < TOL[prec]
        assert orc.rel_l2(piped[1], A[piped[6]]) < 4 * TOL[prec]
        assert orc.rel_l2(piped[4], piped[7]) < 4 * TOL[prec]


def test_get_subarrays_methods(golden_dir):
    """The classes' get_subarrays (slab.py:199-211, pencil.py:218-246, 971-999) against the reference's own boxes
    (tests/golden/subarrays.json), rank by rank on 4 and 8 ranks."""
    import json
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    table = json.load(open(os.path.join(golden_dir, "subarrays.json")))
    N = [32, 64, 128]
    for P in (4, 8):
        for decomp in ("slab", "pencilX", "pencilY"):
            def body(comm):
                F = (Slab_R2C(np.array(N), L, comm, "double") if decomp == "slab" else
                     Pencil_R2C(np.array(N), L, comm, "double", communication="Alltoallw", alignment=decomp[-1]))
                out = {}
                for pad in (1, 1.5):
                    r = F.get_subarrays(padsize=pad)
                    nl = 2 if decomp == "slab" else 4
                    out[pad] = ([[list(map(list, b.args())) for b in lst] for lst in r[:nl]],
                                [[list(c[0]), list(c[1])] for c in r[nl:]])
                return comm.Get_rank(), out
            for rank, out in run_ranks(P, body):
                for pad in (1, 1.5):
                    rec = [t for t in table if t["decomp"] == decomp and t["N"] == N and t["P"] == P and t["rank"] == rank
                           and t["padsize"] == pad and t.get("P1_arg") is None]
                    assert len(rec) == 1
                    assert out[pad][0] == rec[0]["lists"] and out[pad][1] == rec[0]["counts_displs"], (decomp, P, rank, pad)
