"""CPU: the host helper API of the classes against outputs of the REFERENCE classes (tests/golden/helpers_*.npz, written
by oracle/refharness/make_golden.py from slab.py:146-197, pencil.py:289-349, 945-957, line.py:105-134), value AND dtype,
for every rank of every decomposition.  The classes are built on a LayoutComm: no device, no plan."""
import os

import numpy as np
import pytest

from mpifft4py_amd import LayoutComm, Line_R2C, Pencil_R2C, Slab_C2C, Slab_R2C, work_arrays
from mpifft4py_amd import _lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMBOS = [(s, b, e) for s in (False, True) for b in (False, True) for e in (False, True)]


@pytest.fixture(scope="module", params=["double", "single"])
def gold(request):
    return request.param, np.load(os.path.join(GOLDEN, "helpers_%s.npz" % request.param))


def same(got, want, what):
    got = np.asarray(got)
    assert got.dtype == want.dtype, (what, got.dtype, want.dtype)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(got, want), what


def check(G, pre, F, mesh_kw=True, kvec=True, ky_unsliced=False):
    X = F.get_local_mesh()
    assert len(X) == len(F.real_shape())
    for i in range(len(X)):
        same(X[i], G[pre + "mesh_%d" % i], pre + "mesh")
    if kvec:
        K = F.complex_local_wavenumbers()
        for i in range(3):
            want = G[pre + "kvec_%d" % i]
            if ky_unsliced and i == 1:            # upstream's x-aligned class returns all of ky
                want = want[F.complex_local_slice()[1]]
            same(K[i], want, pre + "kvec")
    if mesh_kw:
        for (s, b, e) in COMBOS:
            K = F.get_local_wavenumbermesh(scaled=s, broadcast=b, eliminate_highest_freq=e)
            for i in range(len(K)):
                same(K[i], G[pre + "K_s%d_b%d_e%d_%d" % (s, b, e, i)], pre + "K%s" % ((s, b, e),))
    got, want = F.get_dealias_filter(), G[pre + "dealias"]
    assert got.shape == tuple(F.complex_shape()) and got.dtype == np.uint8
    if ky_unsliced and want.shape != got.shape:
        # upstream's x-aligned filter comes out of its own mesh helper, which drops the Nyquist column on the ranks
        # that hold it (pencil.py:958-969: the filter does not even broadcast against the spectrum there); compare the
        # columns it has, the missing one is kz = N2/2 > 2/3 (N2/2 + 1): removed
        assert want.shape[:2] == got.shape[:2] and want.shape[2] == got.shape[2] - 1
        assert not got[:, :, -1].any()
        got = got[:, :, :-1]
    same(got, want, pre + "dealias")


@pytest.mark.parametrize("P", [1, 2, 4, 8])
def test_slab_helpers_match_reference(gold, P):
    prec, G = gold
    for r in range(P):
        F = Slab_R2C(G["N"], G["L"], LayoutComm(P, r), prec)
        check(G, "slab_P%d_r%d_" % (P, r), F)


@pytest.mark.parametrize("P,P1", [(4, None), (8, None), (8, 2)])
@pytest.mark.parametrize("align", ["Y", "X"])
def test_pencil_helpers_match_reference(gold, P, P1, align):
    prec, G = gold
    for r in range(P):
        F = Pencil_R2C(G["N"], G["L"], LayoutComm(P, r), prec, P1=P1, alignment=align)
        check(G, "pencil%s_P%d_P1%s_r%d_" % (align, P, P1, r), F, mesh_kw=(align == "Y"), ky_unsliced=(align == "X"))
        if align == "X":
            # the common contract on the x-aligned class: the grid is the open product of its own wave vectors
            kv = F.complex_local_wavenumbers()
            K = F.get_local_wavenumbermesh(scaled=True, broadcast=True)
            Lp = 2 * np.pi / F.L
            for i in range(3):
                assert K[i].shape == tuple(F.complex_shape()) and K[i].dtype == F.float
                sh = [1, 1, 1]
                sh[i] = -1
                assert np.array_equal(K[i], np.broadcast_to((kv[i] * Lp[i]).astype(F.float).reshape(sh), K[i].shape))


@pytest.mark.parametrize("P", [1, 2, 4])
def test_line_helpers_match_reference(gold, P):
    prec, G = gold
    for r in range(P):
        F = Line_R2C(G["N_line"], G["L_line"], LayoutComm(P, r), prec)
        check(G, "line_P%d_r%d_" % (P, r), F, kvec=False)


@pytest.mark.parametrize("P", [1, 2])
def test_slab_c2c_wavenumbers(gold, P):
    prec, G = gold
    for r in range(P):
        F = Slab_C2C(G["N"], G["L"], LayoutComm(P, r), prec)
        for i, k in enumerate(F.transformed_local_wavenumbers()):
            same(k, G["slabc2c_P%d_r%d_tkvec_%d" % (P, r, i)], "tkvec")
        # full-spectrum kz for the helpers too (upstream inherits the half-spectrum ones: not reproduced)
        assert [len(k) for k in F.complex_local_wavenumbers()] == list(F.complex_shape())
        assert F.get_dealias_filter().shape == tuple(F.complex_shape())


def test_layout_only_objects_refuse_to_transform():
    F = Slab_R2C(np.array([8, 8, 8]), np.array([1., 1., 1.]), LayoutComm(2, 1), "double")
    assert F.real_shape() == (4, 8, 8) and F.complex_local_slice()[1] == slice(4, 8, 1)
    with pytest.raises(_lib.MfftError):
        F.fftn(np.zeros(F.real_shape()), np.zeros(F.complex_shape(), dtype=complex))
    with pytest.raises(_lib.MfftError):        # what plan creation rejects is rejected without a device too
        Slab_R2C(np.array([8, 8, 7]), np.array([1., 1., 1.]), LayoutComm(1, 0), "double")
    with pytest.raises(ValueError):
        LayoutComm(2, 2)


def test_work_arrays_contract():
    """mpibase.py:61-131: two spellings of a key, zero-fill on access unless told otherwise, errors."""
    w = work_arrays()
    a = w[((3, 4), float, 0)]
    assert a.shape == (3, 4) and a.dtype == np.float64 and not a.any()
    a[:] = 1
    b = w[(a, 1)]
    assert b is not a and b.shape == a.shape and b.dtype == a.dtype
    assert w[(a, 0, False)] is a and a.sum() == 12               # fillzero=False keeps the contents
    assert w.fillzero is False
    assert w[((3, 4), np.float64, 0)] is a and not a.any()        # default: cleared
    assert w.fillzero is True
    c = w[((2,), np.complex64, 0, False)]
    assert c.dtype == np.complex64 and len(w) == 3
    w[(c, 5)] = np.ones(2, dtype=np.complex64)
    assert w[(c, 5, False)].sum() == 2
    del w[(c, 5)]
    assert len(w) == 3 and len(list(iter(w))) == 3
    for bad in [(1, 2, 3), ([3], float, 0), ((3,), float), ((3,), float, 0, True, 1), (a,)]:
        with pytest.raises(TypeError):
            w[bad]
    with pytest.raises(AssertionError):
        w[((3,), float, 0.5)]
    with pytest.raises(AssertionError):
        w[((3,), float, 0, 1)]
    with pytest.raises(TypeError):
        w.values()


def test_dealias_fingerprints_without_copies_and_per_size_cache():
    """`F.dealias` fingerprints (slab.py:237-245 reads the attribute on every call; _base.py compares fingerprints instead):
    the whole-array hash is the same for equal content whatever the strides, the sampled one reads 8192 elements through
    the array's own strides (a broadcast view is not expanded), the sample positions are cached per size -- two plans of
    different sizes do not evict each other -- and a single changed element is seen by the whole hash."""
    from mpifft4py_amd._base import DistFFTBase
    F = Slab_R2C(np.array([32, 32, 32]), np.array([2 * np.pi] * 3), LayoutComm(1, 0), "double")
    DistFFTBase._sample_index.clear()
    a = np.ones((96, 96, 49), dtype=np.uint8)                         # 451 584 B > 256 KiB: sampled
    b = np.ones((128, 64, 65), dtype=bool)
    row = np.ones((1, 1, 49), dtype=np.uint8)
    view = np.broadcast_to(row, a.shape)                               # strides (0, 0, 1): 49 bytes of memory
    F._dealias = a
    fa = F._mask_fingerprint()
    assert fa[2] == "sampled"
    F._dealias = b
    fb = F._mask_fingerprint()
    assert a.size in DistFFTBase._sample_index and b.size in DistFFTBase._sample_index
    idx_a = DistFFTBase._sample_index[a.size]
    F._dealias = a
    assert F._mask_fingerprint() == fa and DistFFTBase._sample_index[a.size] is idx_a
    F._dealias = view
    fv = F._mask_fingerprint()
    assert fv == fa                                                    # same content, other strides
    assert ("nd",) + a.shape in DistFFTBase._sample_index
    assert F._mask_fingerprint(True)[3] == DistFFTBase._whole_hash(a)
    # one element of the large array, at a position the samples do not hold
    taken = set(idx_a.tolist())
    pos = next(i for i in range(a.size) if i not in taken)
    a.reshape(-1)[pos] = 0
    F._dealias = a
    assert F._mask_fingerprint() == fa                                 # the documented blind spot of the cheap fingerprint
    assert F._mask_fingerprint(True)[3] != DistFFTBase._whole_hash(np.ones_like(a))
    a.reshape(-1)[int(idx_a[0])] = 0
    assert F._mask_fingerprint() != fa
    small = np.ones((8, 8, 5))
    F._dealias = small
    assert F._mask_fingerprint()[2] == "whole"
    assert fb[2] == "sampled"
