"""GPU: the pitched device spectrum (mfft_plan_desc::complex_pitch; SURVEY 7 "padded device pitch internally, exact Nf at
the API boundary").  The logical shapes stay the reference's (slab.py:102-104, pencil.py:248-287); only the rows of the
device-resident complex array lie a whole number of cache lines apart.  Checked against the oracle / numpy.fft through
numpy arrays (converted at the boundary) and through DeviceArrays (FFT.empty_complex), on one rank (every pass runs on
the pitched rows) and on several (the plan converts), and -- the parity suite once more -- by running the slab / pencil /
padded / 2/3-rule / golden-fixture tests of tests/test_gpu_parity.py with every object built pitched."""
import numpy as np
import pytest

from gpu_util import L, TOL, cdtype, have_gpu, orc, rdtype, run_ranks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not have_gpu():
        pytest.fail("no GPU visible")


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("pitch", ["auto", "plus3"])
@pytest.mark.parametrize("N", [[32, 64, 128], [64, 32, 32], [8, 16, 32], [128, 128, 128], [20, 24, 40], [16, 512, 256]])
def test_one_rank_native(N, pitch, prec):
    """One rank: r2c / c2r, both strided passes, the 3/2-rule passes and the masked 2/3-rule passes all on rows `pitch`
    apart; DeviceArrays in and out; the elements between the rows never leak into a result."""
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    N = np.array(N)
    nf = int(N[2]) // 2 + 1
    F = Slab_R2C(N, L, SelfComm(0), prec, complex_pitch="auto" if pitch == "auto" else nf + 3)
    line = 128 // np.dtype(F.complex).itemsize
    want_pitch = (nf + line - 1) // line * line if pitch == "auto" else nf + 3
    assert F.complex_pitch == want_pitch and tuple(F.complex_shape()) == (int(N[0]), int(N[1]), nf)
    assert F.plan_info("complex_pitch") == want_pitch and F.plan_info("complex_pitch_native") == 1
    rng = np.random.default_rng(5 + int(N[0]))
    A = rng.random(tuple(N)).astype(rdtype(prec))
    ref = np.fft.rfftn(A.astype(np.float64))
    fu = F.empty_complex()
    assert fu.pitch == (None if want_pitch == nf else want_pitch) and fu.shape == tuple(F.complex_shape())
    # poison the whole allocation (the gaps included): NaNs there must stay there
    poison = DeviceArray(fu.shape[:-1] + (want_pitch,), fu.dtype, ptr=fu.ptr, owner=False)
    poison.set(np.full(poison.shape, np.nan + 1j * np.nan, dtype=fu.dtype))
    u = DeviceArray.from_numpy(A)
    F.fftn(u, fu)
    F.sync()
    got = fu.get()
    assert got.shape == ref.shape and orc.rel_l2(got, ref) < TOL[prec]
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    F.ifftn(fu, u2)
    F.sync()
    assert orc.rel_l2(u2.get(), A) < 4 * TOL[prec]
    assert np.array_equal(fu.get(), got)                                    # the spectrum is preserved
    # numpy in / out: converted at the boundary
    c = F.fftn(A, np.zeros(F.complex_shape(), dtype=F.complex))
    assert orc.rel_l2(c, ref) < TOL[prec]
    b = F.ifftn(c, np.zeros(F.real_shape(), dtype=F.float))
    assert orc.rel_l2(b, A) < 4 * TOL[prec]
    # 3/2-rule and 2/3-rule against the oracle
    C0 = ref.astype(cdtype(prec))
    ap = F.ifftn(C0, np.zeros(F.real_shape_padded(), dtype=F.float), "3/2-rule")
    assert orc.rel_l2(ap, orc.slab_r2c_backward_padded([C0], N, prec)[0]) < 4 * TOL[prec]
    cp = F.fftn(ap, np.zeros(F.complex_shape(), dtype=F.complex), "3/2-rule")
    assert orc.rel_l2(cp, orc.slab_r2c_forward_padded([ap], N, prec)[0]) < 4 * TOL[prec]
    am = F.ifftn(C0, np.zeros(F.real_shape(), dtype=F.float), "2/3-rule")
    assert orc.rel_l2(am, orc.slab_r2c_backward([orc.apply_mask(C0, F.get_dealias_filter())], N, prec)[0]) < 4 * TOL[prec]


@pytest.mark.parametrize("dealias", ["3/2-rule", "2/3-rule", None])
@pytest.mark.parametrize("decomp,P", [("slab", 2), ("slab", 4), ("pencilX", 4), ("pencilY", 4), ("slabC2C", 1), ("slabC2C", 2)])
def test_ranks_and_pencils_convert(decomp, P, dealias):
    """Plans whose routes want compact rows (several ranks, pencils, complex data) take pitched arrays through a compact
    copy of their own: bit-identical to the same object built compact."""
    from mpifft4py_amd import DeviceArray, Slab_C2C
    from mpifft4py_amd.pencil import R2C as Pencil_R2C
    from mpifft4py_amd.slab import R2C as Slab_R2C
    N = np.array([16, 32, 32])
    if decomp == "slabC2C" and dealias == "2/3-rule":
        pytest.skip("the reference's C2C class has no 2/3-rule filter")

    def work(comm):
        out = []
        for pitch in (None, "auto"):
            if decomp == "slab":
                F = Slab_R2C(N, L, comm, "double", complex_pitch=pitch)
            elif decomp == "slabC2C":
                F = Slab_C2C(N, L, comm, "double", complex_pitch=pitch)
            else:
                F = Pencil_R2C(N, L, comm, "double", communication="Alltoallw", alignment=decomp[-1], complex_pitch=pitch)
            if pitch:
                assert F.plan_info("complex_pitch") > 0 and F.plan_info("complex_pitch_native") == 0
            rng = np.random.default_rng(9 + comm.Get_rank())
            if decomp == "slabC2C":
                a = (rng.random(F.original_shape()) + 1j * rng.random(F.original_shape())).astype(F.complex)
                fu = F.empty_complex()
                F.fftn(DeviceArray.from_numpy(a), fu)
                back = DeviceArray.empty(F.original_shape(), F.complex)
                F.ifftn(fu, back)
            else:
                a = rng.random(F.real_shape())
                fu = F.empty_complex()
                F.fftn(DeviceArray.from_numpy(a), fu)
                back = DeviceArray.empty(F.work_shape(dealias), F.float)
                F.ifftn(fu, back, dealias)
            F.sync()
            out.append((fu.get(), back.get()))
        return bool(np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]))

    assert all(run_ranks(P, work))


@pytest.mark.parametrize("dealias", ["3/2-rule", None])
@pytest.mark.parametrize("decomp,P", [("slab", 2), ("pencilX", 4)])
def test_nonlinear_cross_pitched_over_ranks(decomp, P, dealias):
    """mfft_nonlinear_cross on pitched vector fields where the plan does not run on pitched rows (several ranks, pencils): the
    composition inside converts every transform at the boundary; same results as the compact object, bit for bit."""
    from mpifft4py_amd import DeviceArray, spectral
    from mpifft4py_amd.pencil import R2C as Pencil_R2C
    from mpifft4py_amd.slab import R2C as Slab_R2C
    N = np.array([16, 32, 32])

    def work(comm):
        res = []
        for pitch in (None, "auto"):
            F = (Slab_R2C(N, L, comm, "double", complex_pitch=pitch) if decomp == "slab" else
                 Pencil_R2C(N, L, comm, "double", communication="Alltoallw", alignment="X", complex_pitch=pitch))
            rng = np.random.default_rng(17 + comm.Get_rank())
            a, b, out = F.empty_complex(3), F.empty_complex(3), F.empty_complex(3)
            for x in (a, b):
                for i in range(3):
                    F.fftn(DeviceArray.from_numpy(rng.random(F.real_shape()) - 0.5), x.component(i))
            spectral.cross_transform(F, a, b, out, dealias)
            F.sync()
            res.append(out.get())
        return float(np.abs(res[0] - res[1]).max() / np.abs(res[0]).max())

    assert max(run_ranks(P, work)) < 1e-13


def test_wrong_pitch_is_refused():
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_R2C
    F = Slab_R2C(np.array([16, 16, 32]), L, SelfComm(0), "double", complex_pitch="auto")
    u = DeviceArray.from_numpy(np.random.default_rng(0).random(F.real_shape()))
    with pytest.raises(ValueError):
        F.fftn(u, DeviceArray.empty(F.complex_shape(), F.complex))            # a compact array on a pitched object
    G = Slab_R2C(np.array([16, 16, 32]), L, SelfComm(0), "double")
    with pytest.raises(ValueError):
        G.fftn(u, F.empty_complex())                                           # and the other way round
    with pytest.raises(Exception):
        Slab_R2C(np.array([16, 16, 32]), L, SelfComm(0), "double", complex_pitch=5)      # shorter than the 17 bins of a row


@pytest.mark.parametrize("dealias", ["3/2-rule", "2/3-rule", None])
def test_nonlinear_and_solver_pitched(dealias, golden_dir):
    """mfft_nonlinear_cross and the one-sweep Runge-Kutta stage on pitched vector fields: the Taylor-Green known answer,
    equal to the compact run to rounding."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    import spectral_dns_device as demo
    from mpifft4py_amd import SelfComm
    gold = json.load(open(os.path.join(golden_dir, "taylor_green.json")))
    kp = demo.solve(SelfComm(0), dealias=dealias, complex_pitch="auto")
    kc = demo.solve(SelfComm(0), dealias=dealias, complex_pitch=None)
    assert round(kp - gold["k_expected_demo"], 7) == 0 and abs(kp - gold["k_P1_%s" % dealias]) < 1e-11
    assert abs(kp - kc) < 1e-13


# ---- the parity suite once more, every object pitched ----------------------------------------------------------------
def _pitched_rerun(monkeypatch, fn, *args, **kw):
    from mpifft4py_amd._base import DistFFTBase
    monkeypatch.setattr(DistFFTBase, "default_complex_pitch", "auto")
    return fn(*args, **kw)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4, 8])
def test_parity_slab_r2c_pitched(P, prec, monkeypatch):
    import test_gpu_parity as t
    _pitched_rerun(monkeypatch, t.test_slab_r2c, P, "Alltoallw", prec)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("P", [1, 2, 4])
def test_parity_slab_padded_pitched(P, prec, monkeypatch):
    import test_gpu_parity as t
    _pitched_rerun(monkeypatch, t.test_slab_padded, P, prec)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("align", ["X", "Y"])
def test_parity_pencil_pitched(align, prec, monkeypatch):
    import test_gpu_parity as t
    _pitched_rerun(monkeypatch, t.test_pencil_r2c, 4, None, align, prec, "Alltoallw")
    _pitched_rerun(monkeypatch, t.test_pencil_padded, 4, align, prec)


@pytest.mark.parametrize("prec", ["double", "single"])
def test_parity_golden_fixtures_pitched(prec, golden_dir, monkeypatch):
    import test_gpu_parity as t
    _pitched_rerun(monkeypatch, t.test_golden_fixtures, prec, golden_dir)
