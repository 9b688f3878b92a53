"""GPU parity, stage level: the serialFFT seam (fft/ifft/rfft/irfft/... of
mpifft4py_amd.serialFFT, i.e. mfft_c2c_axis / mfft_r2c_last / mfft_c2r_last),
slab pack/unpack and the dealias mask, against numpy.fft (the arithmetic the
reference's numpy backend uses, numpy_fft.py:25-107)."""
import ctypes

import numpy as np
import pytest

from gpu_util import TOL, cdtype, have_gpu, orc, rdtype

pytestmark = pytest.mark.gpu

LENGTHS = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096,
           6, 12, 24, 48, 96, 192, 384, 768, 1536, 3072,
           10, 20, 40, 80, 160, 320, 640, 1280, 2560,
           18, 36, 72, 144, 288, 576, 1152, 2304, 50, 100, 200, 400, 800, 1600, 250, 500, 1000, 2000,
           # lengths with both 3 and 5 among their factors (plans.h groups L, M: 30 values per thread)
           30, 60, 90, 120, 150, 180, 240, 300, 360, 450, 480, 600, 720, 900, 960, 1200, 1440, 1800,
           750, 1500, 1920, 2400, 3000, 3840,
           # round 4: 7 * 2^a (plans.h group O: 28 values per thread, radix 28 = 7 x 4) and 8192
           14, 28, 56, 112, 224, 448, 896, 1792, 3584, 8192,
           # round 5: the radix plans between 4096 and 8192 (plans.h group Q) and 21 * 2^a (group R: 42 values per thread)
           4608, 5120, 6144, 7168, 42, 84, 168, 336, 672, 1344, 2688,
           # round 6: 35 * 2^a (plans.h group S: 70 values per thread, radix 70 = 7 x 10, single precision; double: chirp-z)
           70, 140, 280, 560, 1120, 2240,
           # round 6: 27 * 2^a (plans.h group T: the 3/2-rule images of the 9 * 2^a meshes)
           54, 108, 216, 432, 864, 1728, 3456,
           # round 6: the other 3/2-rule images (groups U, V): 135 * 2^a, 1350 / 2700 / 2250, the odd 675 / 1125 (15 values per thread), 81 * 2^a
           270, 540, 1080, 2160, 1350, 2700, 2250, 675, 1125, 162, 324, 648, 1296, 2592, 75, 135, 225, 375, 2880, 3600,
           126, 252, 504, 1008, 2016]      # group W: 63 * 2^a


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not have_gpu():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("n", LENGTHS)
def test_c2c_every_length_every_axis(n, prec):
    from mpifft4py_amd import fft, ifft
    rng = np.random.default_rng(n)
    for axis in (0, 1, 2):
        shape = [3, 5, 7]
        shape[axis] = n
        a = (rng.random(shape) - 0.5 + 1j * (rng.random(shape) - 0.5)).astype(cdtype(prec))
        ref = np.fft.fft(a.astype(np.complex128), axis=axis)
        got = fft(a, axis=axis)
        assert got.dtype == cdtype(prec)
        assert orc.rel_l2(got, ref) < TOL[prec], (n, axis)
        refi = np.fft.ifft(a.astype(np.complex128), axis=axis)
        goti = ifft(a, axis=axis)
        assert orc.rel_l2(goti, refi) < TOL[prec], (n, axis)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("n", [2 * m for m in LENGTHS])
def test_rfft_irfft_every_length(n, prec):
    from mpifft4py_amd import rfft, irfft
    rng = np.random.default_rng(n + 1)
    a = (rng.random((3, 5, n)) - 0.5).astype(rdtype(prec))
    ref = np.fft.rfft(a.astype(np.float64), axis=2)
    got = rfft(a, axis=2)
    assert got.shape == ref.shape
    assert orc.rel_l2(got, ref) < TOL[prec]
    # c2r convention: imaginary parts of the k=0 and k=n/2 bins are ignored
    c = ref.astype(cdtype(prec)).copy()
    c[..., 0] += 1j * 0.7
    c[..., -1] -= 1j * 0.3
    back = irfft(c, axis=2)
    assert back.shape == a.shape
    assert orc.rel_l2(back, np.fft.irfft(c.astype(np.complex128), n=n, axis=2)) < TOL[prec]
    assert orc.rel_l2(back, a) < 4 * TOL[prec]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("shape", [(32, 64, 128), (8, 16, 32), (48, 24, 96), (1, 64, 64), (64, 1, 20)])
def test_multi_axis_transforms(shape, prec):
    import mpifft4py_amd as m
    rng = np.random.default_rng(5)
    a = rng.random(shape).astype(rdtype(prec))
    a64 = a.astype(np.float64)
    assert orc.rel_l2(m.rfftn(a, axes=(0, 1, 2)), np.fft.rfftn(a64)) < TOL[prec]
    assert orc.rel_l2(m.rfft2(a, axes=(1, 2)), np.fft.rfft2(a64, axes=(1, 2))) < TOL[prec]
    c = np.fft.rfftn(a64).astype(cdtype(prec))
    assert orc.rel_l2(m.irfftn(c, axes=(0, 1, 2)), a64) < 4 * TOL[prec]
    c2 = np.fft.rfft2(a64, axes=(1, 2)).astype(cdtype(prec))
    assert orc.rel_l2(m.irfft2(c2, axes=(1, 2)), a64) < 4 * TOL[prec]
    z = (a + 1j * rng.random(shape)).astype(cdtype(prec))
    z128 = z.astype(np.complex128)
    assert orc.rel_l2(m.fftn(z), np.fft.fftn(z128)) < TOL[prec]
    assert orc.rel_l2(m.ifftn(z), np.fft.ifftn(z128)) < TOL[prec]
    assert orc.rel_l2(m.fft2(z, axes=(1, 2)), np.fft.fft2(z128, axes=(1, 2))) < TOL[prec]
    assert orc.rel_l2(m.ifft2(z, axes=(0, 1)), np.fft.ifft2(z128, axes=(0, 1))) < TOL[prec]
    # output-array form: b is filled and returned
    b = np.zeros(np.fft.rfftn(a64).shape, dtype=cdtype(prec))
    r = m.rfftn(a, b, axes=(0, 1, 2))
    assert r is b and orc.rel_l2(b, np.fft.rfftn(a64)) < TOL[prec]


def test_input_is_not_modified():
    import mpifft4py_amd as m
    from mpifft4py_amd import DeviceArray
    rng = np.random.default_rng(9)
    z = (rng.random((16, 32, 64)) + 1j * rng.random((16, 32, 64)))
    d = DeviceArray.from_numpy(z)
    out = m.fftn(d)
    assert np.array_equal(d.get(), z)
    assert orc.rel_l2(out, np.fft.fftn(z)) < 1e-10


def test_linearity_and_parseval_large():
    """Size-independent properties at a size the O(N log N) host check would be slow for."""
    import mpifft4py_amd as m
    rng = np.random.default_rng(11)
    shape = (64, 256, 512)
    a = rng.random(shape)
    b = rng.random(shape)
    fa, fb = m.rfftn(a), m.rfftn(b)
    fab = m.rfftn(2.0 * a - 3.0 * b)
    assert orc.rel_l2(fab, 2.0 * fa - 3.0 * fb) < 1e-12
    # Parseval for the half spectrum
    w = np.full(shape[2] // 2 + 1, 2.0)
    w[0] = w[-1] = 1.0
    e_spec = float(np.sum((np.abs(fa) ** 2) * w)) / a.size
    assert abs(e_spec - float(np.sum(a * a))) / float(np.sum(a * a)) < 1e-12


@pytest.mark.parametrize("prec", ["double", "single"])
def test_slab_pack_unpack(prec):
    """mfft_slab_pack == slab.py:403, mfft_slab_unpack == transpose_Uc (maths.pyx:21-31); bit exact."""
    from mpifft4py_amd import DeviceArray, _lib
    rng = np.random.default_rng(3)
    P, Np0, Np1, Nf = 4, 6, 5, 17
    T = (rng.random((Np0, P * Np1, Nf)) + 1j * rng.random((Np0, P * Np1, Nf))).astype(cdtype(prec))
    dT = DeviceArray.from_numpy(T)
    dM = DeviceArray.empty((P, Np0, Np1, Nf), T.dtype)
    _lib.call("mfft_slab_pack", dT.ptr, dM.ptr, P, Np0, Np1, Nf, _lib.precision_code(prec))
    assert np.array_equal(dM.get(), orc.slab_pack(T, P))
    dT2 = DeviceArray.zeros(T.shape, T.dtype)
    _lib.call("mfft_slab_unpack", dM.ptr, dT2.ptr, P, Np0, Np1, Nf, _lib.precision_code(prec))
    assert np.array_equal(dT2.get(), T)
    assert np.array_equal(orc.slab_unpack(orc.slab_pack(T, P)), T)


@pytest.mark.parametrize("prec", ["double", "single"])
def test_dealias_filter(prec):
    from mpifft4py_amd import DeviceArray, _lib
    rng = np.random.default_rng(4)
    fu = (rng.random((9, 10, 11)) + 1j * rng.random((9, 10, 11))).astype(cdtype(prec))
    mask = (rng.random(fu.shape) > 0.4).astype(np.uint8)
    d = DeviceArray.from_numpy(fu)
    dm = DeviceArray.from_numpy(mask)
    _lib.call("mfft_dealias_filter", d.ptr, dm.ptr, fu.size, _lib.precision_code(prec))
    assert np.array_equal(d.get(), orc.apply_mask(fu, mask).astype(fu.dtype))


def test_unsupported_length_raises():
    """Lengths run to 2^20 (include/mpifft4py_amd.h mfft_length_route); beyond that the call fails loudly."""
    import mpifft4py_amd as m
    from mpifft4py_amd import _lib
    with pytest.raises(_lib.MfftError):
        m.fft(np.zeros(((1 << 20) + 1, 1, 1), dtype=np.complex64), axis=0)
    with pytest.raises(_lib.MfftError):
        m.rfft(np.zeros((1, 1, (1 << 20) + 2), dtype=np.float32), axis=2)


# lengths without a radix plan: chirp-z kernels (csrc/fft_chirpz.h); primes, prime powers, 7-smooth,
# odd 15-smooth ones, and the range ends of several convolution lengths
CHIRPZ = [3, 5, 7, 9, 11, 13, 15, 17, 25, 27, 31, 33, 45, 49, 75, 84, 127, 129, 255, 257, 504,
          675, 729, 1008, 1023, 1025, 1201, 1537, 2047,
          # round 4: convolution length 8192 -- everything up to 4096 (7-smooth meshes like 2240 = 35 * 64, primes, range ends; 2688 has a
          # radix plan since round 5)
          2049, 2100, 2240, 2688, 3125, 3600, 4093, 4095]


# round 5: lengths without a radix plan beyond the one-workgroup chirp-z range (complex n > 4096, odd real n > 4096, anything
# above 8192): Bluestein over a four-step power-of-two transform in a scratch buffer (csrc/bigfft.hip, route 3 of
# mfft_length_route) -- primes, range ends, composite lengths, M = 16384 ... 2^18
BIG = [4097, 4099, 5000, 6561, 8191, 8193, 10000, 16385, 30011, 65537, 100003,
       # composite lengths n = n1 * n2 with radix plans for both factors: the four-step transform at length n itself, no Bluestein
       6400, 12288, 16384, 20000, 65536, 1 << 20]


def _slow_above(lengths, limit, also=()):
    """the longest lengths (host FFTs of 15 - 30 million points per case) and some of the others only with MFFT_TEST_SLOW=1
    (tests/conftest.py): every mechanism -- Bluestein at M = 2^14 ... 2^18, the composite route, primes, range ends -- keeps
    cases in the default run"""
    return [pytest.param(n, marks=pytest.mark.slow) if (n > limit or n in also) else n for n in lengths]


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("n", _slow_above(BIG, 65537, also=(4099, 6561, 8193, 16385, 30011, 12288, 20000)))
def test_c2c_lengths_through_the_scratch_buffer_fallback(n, prec):
    """numpy_fft.py:25-37 takes every n: so does mfft_c2c_axis, along every axis, forward and inverse."""
    from mpifft4py_amd import _lib, fft, ifft
    assert _lib.load().mfft_length_route(n, 0) == 3
    rng = np.random.default_rng(n)
    for axis in (0, 1, 2):
        shape = [2, 3, 5]
        shape[axis] = n
        a = (rng.random(shape) - 0.5 + 1j * (rng.random(shape) - 0.5)).astype(cdtype(prec))
        got = fft(a, axis=axis)
        assert orc.rel_l2(got, np.fft.fft(a.astype(np.complex128), axis=axis)) < 2 * TOL[prec], (n, axis)
        goti = ifft(a, axis=axis)
        assert orc.rel_l2(goti, np.fft.ifft(a.astype(np.complex128), axis=axis)) < 2 * TOL[prec], (n, axis)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("n", _slow_above([4099, 8193, 8194, 10000, 16386, 20001, 65538, 131071, 12800, 32768, 200000], 65538, also=(8193, 16386, 20001)))
def test_rfft_irfft_lengths_through_the_scratch_buffer_fallback(n, prec):
    from mpifft4py_amd import _lib, rfft, irfft
    assert _lib.load().mfft_length_route(n, 1) == 3
    rng = np.random.default_rng(n + 1)
    a = (rng.random((2, 3, n)) - 0.5).astype(rdtype(prec))
    ref = np.fft.rfft(a.astype(np.float64), axis=2)
    got = rfft(a, axis=2)
    assert got.shape == ref.shape
    assert orc.rel_l2(got, ref) < 2 * TOL[prec]
    c = ref.astype(cdtype(prec)).copy()
    c[..., 0] += 1j * 0.7                      # ignored by c2r
    if n % 2 == 0:
        c[..., -1] -= 1j * 0.3
    back = irfft(c, np.zeros(a.shape, dtype=a.dtype), axis=2)
    assert orc.rel_l2(back, np.fft.irfft(c.astype(np.complex128), n=n, axis=2)) < 2 * TOL[prec]
    assert orc.rel_l2(back, a) < 4 * TOL[prec]


def test_many_vectors_through_the_scratch_buffer_fallback_in_chunks():
    """More vectors than the 256 MiB scratch buffer holds at once (M = 16384, 16 bytes: 1024 vectors per chunk): rows in
    several chunks, and strided columns of one batch wider than the buffer (runs of columns) and of many batches."""
    from mpifft4py_amd import fft
    rng = np.random.default_rng(8)
    n = 4100
    a = (rng.random((40, 30, n)) - 0.5 + 1j * (rng.random((40, 30, n)) - 0.5))            # 1200 rows
    assert orc.rel_l2(fft(a, axis=2), np.fft.fft(a, axis=2)) < 2e-10
    b = np.ascontiguousarray(np.moveaxis(a, 2, 0))                                        # (n, 40, 30): one batch of 1200 columns
    assert orc.rel_l2(fft(b, axis=0), np.fft.fft(b, axis=0)) < 2e-10
    c = np.ascontiguousarray(np.moveaxis(a, 2, 1))                                        # (40, n, 30): 40 batches of 30 columns
    assert orc.rel_l2(fft(c, axis=1), np.fft.fft(c, axis=1)) < 2e-10


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("n", CHIRPZ)
def test_c2c_arbitrary_length_every_axis(n, prec):
    from mpifft4py_amd import fft, ifft
    rng = np.random.default_rng(n)
    for axis in (0, 1, 2):
        shape = [3, 5, 7]
        shape[axis] = n
        a = (rng.random(shape) - 0.5 + 1j * (rng.random(shape) - 0.5)).astype(cdtype(prec))
        got = fft(a, axis=axis)
        assert orc.rel_l2(got, np.fft.fft(a.astype(np.complex128), axis=axis)) < TOL[prec], (n, axis)
        goti = ifft(a, axis=axis)
        assert orc.rel_l2(goti, np.fft.ifft(a.astype(np.complex128), axis=axis)) < TOL[prec], (n, axis)


@pytest.mark.parametrize("prec", ["double", "single"])
@pytest.mark.parametrize("n", CHIRPZ + [22, 126, 258, 1026, 2046, 2050, 4094, 4098, 5000, 8190])
def test_rfft_irfft_arbitrary_length(n, prec):
    """Real transforms of any length, odd ones included (numpy_fft.py:39-51 with n given by the output)."""
    from mpifft4py_amd import rfft, irfft
    rng = np.random.default_rng(n + 1)
    a = (rng.random((3, 5, n)) - 0.5).astype(rdtype(prec))
    ref = np.fft.rfft(a.astype(np.float64), axis=2)
    got = rfft(a, axis=2)
    assert got.shape == ref.shape
    assert orc.rel_l2(got, ref) < TOL[prec]
    c = ref.astype(cdtype(prec)).copy()
    c[..., 0] += 1j * 0.7                      # ignored by c2r
    if n % 2 == 0:
        c[..., -1] -= 1j * 0.3
    back = irfft(c, np.zeros(a.shape, dtype=a.dtype), axis=2)
    assert orc.rel_l2(back, np.fft.irfft(c.astype(np.complex128), n=n, axis=2)) < TOL[prec]
    assert orc.rel_l2(back, a) < 4 * TOL[prec]


def test_unpack_and_mask_vs_reference_compiled_loops():
    """mfft_slab_unpack / mfft_dealias_filter against the reference's own compiled Cython loops
    (oracle/_ref, built from /root/reference/mpiFFT4py/cython/maths.pyx in the dev container)."""
    from oracle import build_ref
    from mpifft4py_amd import DeviceArray, _lib
    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref not present")
    rng = np.random.default_rng(12)
    for prec in ("double", "single"):
        dt = cdtype(prec)
        P, Np0, Np1, Nf = 8, 4, 6, 33
        U = (rng.random((P, Np0, Np1, Nf)) + 1j * rng.random((P, Np0, Np1, Nf))).astype(dt)
        T = np.zeros((Np0, P * Np1, Nf), dtype=dt)
        ref.transpose_Uc(T, U, P, Np0, Np1, Nf)
        dU = DeviceArray.from_numpy(U)
        dT = DeviceArray.zeros(T.shape, dt)
        _lib.call("mfft_slab_unpack", dU.ptr, dT.ptr, P, Np0, Np1, Nf, _lib.precision_code(prec))
        assert np.array_equal(dT.get(), T)
        fu = (rng.random((16, 8, 33)) + 1j * rng.random((16, 8, 33))).astype(dt)
        mask = (rng.random(fu.shape) > 0.5).astype(np.uint8)
        want = ref.dealias_filter(fu.copy(), mask)
        d = DeviceArray.from_numpy(fu)
        dm = DeviceArray.from_numpy(mask)
        _lib.call("mfft_dealias_filter", d.ptr, dm.ptr, fu.size, _lib.precision_code(prec))
        assert np.array_equal(d.get(), want)


@pytest.mark.parametrize("n", [8, 7, 64, 100, 129])
@pytest.mark.parametrize("type_", [2, 3])
def test_dct(n, type_):
    """serialFFT.dct (numpy_fft.py:11-22): scipy.fftpack convention, real and complex input."""
    from scipy.fftpack import dct as sdct
    from mpifft4py_amd import dct
    rng = np.random.default_rng(n)
    for axis in (0, 1):
        shape = [5, 6]
        shape[axis] = n
        a = rng.random(shape)
        assert orc.rel_l2(dct(a, type=type_, axis=axis), sdct(a, type=type_, axis=axis)) < 1e-12
        c = a + 1j * rng.random(shape)
        b = np.zeros(shape, dtype=complex)
        assert dct(c, b, type=type_, axis=axis) is b
        assert orc.rel_l2(b, sdct(c.real, type=type_, axis=axis) + 1j * sdct(c.imag, type=type_, axis=axis)) < 1e-12
    with pytest.raises(NotImplementedError):
        dct(np.zeros(8), type=1)


@pytest.mark.parametrize("prec", ["double", "single"])
def test_dft_bins_checker_against_numpy(prec):
    """mfft_ew_dft_bins (the bin-level checker of the full-size config-5 run): partial sums over two blocks of a mesh add up
    to numpy.fft.fftn's bins -- complex and real input, forward and inverse sign, more bins than one launch takes."""
    from mpifft4py_amd import DeviceArray, SelfComm, Slab_C2C, Slab_R2C, spectral
    N = [12, 10, 16]
    rng = np.random.default_rng(8)
    A = (rng.random(N) - 0.5 + 1j * (rng.random(N) - 0.5)).astype(cdtype(prec))
    R = (rng.random(N) - 0.5).astype(rdtype(prec))
    bins = np.stack([rng.integers(0, n, 37) for n in N], axis=1)
    Fc = Slab_C2C(np.array(N), np.array([1., 1., 1.]), SelfComm(0), prec)
    Fr = Slab_R2C(np.array(N), np.array([1., 1., 1.]), SelfComm(0), prec)
    for F, X in ((Fc, A), (Fr, R)):
        for inverse in (False, True):
            tot = np.zeros(len(bins), dtype=complex)
            for (x0, x1) in ((0, 5), (5, 12)):                    # two "ranks": blocks of x planes
                blk = DeviceArray.from_numpy(np.ascontiguousarray(X[x0:x1]))
                tot += spectral.dft_bins(F, blk, bins, [x0, 0, 0], inverse=inverse)
            full = np.fft.ifftn(X.astype(np.complex128)) * np.prod(N) if inverse else np.fft.fftn(X.astype(np.complex128))
            want = full[bins[:, 0], bins[:, 1], bins[:, 2]]
            assert np.abs(tot - want).max() < 1e-12 * np.sqrt(np.prod(N)), (prec, inverse)


@pytest.mark.parametrize("mode,third", [("1", "0"), ("0", "0"), ("0", "2"), ("1", "1")])
def test_three_sub_transform_kernels_of_1536(mode, third):
    """fft_col3.h ColFft3 (1536 = 3 x 512 per workgroup) forced on (MFFT_COL3=1) and off (=0), in a child process (the
    switch is read once): c2c of length 1536 on every axis, both precisions, in place and out of place (serialFFT), and the
    3/2-rule pairs of a [1024, 16, 32] and a [16, 1024, 32] mesh, whose padded x / y axis is 1536 (pad-on-load inverse,
    truncate-on-store forward).  `third`: MFFT_COL3S -- the pad-on-load inverse with one third of a tile's transform per
    workgroup (ColFft3S) never (0: the ColFft / ColFft3 kernels it replaces by default stay covered), by default (1), on
    every pass (2: also the y pass, an outer batch, in double precision)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import mpifft4py_amd as m
from mpifft4py_amd import Slab_R2C, SelfComm
rng = np.random.default_rng(3)
for prec, ct, tol in (("double", np.complex128, 1e-10), ("single", np.complex64, 1e-5)):
    for axis, shape in ((0, (1536, 3, 40)), (1, (3, 1536, 24)), (0, (1536, 1, 9))):
        a = (rng.random(shape) - 0.5 + 1j * (rng.random(shape) - 0.5)).astype(ct)
        for f, g in ((m.fft, np.fft.fft), (m.ifft, np.fft.ifft)):
            ref = g(a.astype(np.complex128), axis=axis)
            got = f(a, axis=axis)
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < tol, (prec, axis, shape)
    for N in ([1024, 16, 32], [16, 1024, 32]):
        F = Slab_R2C(np.array(N), np.array([2 * np.pi] * 3), SelfComm(0), prec)
        C = np.fft.rfftn(rng.random(N)).astype(ct)
        C[N[0] // 2] = 0; C[:, N[1] // 2] = 0; C[:, :, -1] = 0
        up = F.ifftn(C, np.zeros(F.real_shape_padded(), dtype=F.float), dealias="3/2-rule")
        M0, M1, h0, h1 = 3 * N[0] // 2, 3 * N[1] // 2, N[0] // 2, N[1] // 2
        Cp = np.zeros((M0, M1, 25), dtype=np.complex128)
        Cp[:h0, :h1, :17] = C[:h0, :h1]; Cp[:h0, -h1:, :17] = C[:h0, h1:]
        Cp[-h0:, :h1, :17] = C[h0:, :h1]; Cp[-h0:, -h1:, :17] = C[h0:, h1:]
        want = np.fft.irfftn(Cp, s=(M0, M1, 48), axes=(0, 1, 2)) * 1.5 ** 3
        assert np.linalg.norm(up - want) / np.linalg.norm(want) < 4 * tol, (prec, N)
        back = F.fftn(up, np.zeros(F.complex_shape(), dtype=ct), dealias="3/2-rule")
        assert np.linalg.norm(back - C) / np.linalg.norm(C) < 4 * tol, (prec, N)
print("COL3_OK")
""" % root
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, MFFT_COL3=mode, MFFT_COL3S=third), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=240)
    assert p.returncode == 0 and b"COL3_OK" in p.stdout, p.stdout.decode()[-3000:]


@pytest.mark.parametrize("prec", ["double", "single"])
def test_c2c_strided_entry_point(prec):
    """mfft_c2c_strided: the strided transform with every stride spelled out (rows further apart than the columns they
    hold, different pitches on the two sides, in place and out of place) against numpy.fft on the same views."""
    from mpifft4py_amd import DeviceArray, _lib
    rng = np.random.default_rng(21)
    ct = cdtype(prec)
    for n, nouter, ncols, ip, op in ((96, 3, 33, 40, 33), (128, 2, 65, 65, 72), (60, 1, 500, 512, 500), (1024, 2, 129, 136, 129)):
        A = (rng.random((nouter, n, ip)) - 0.5 + 1j * (rng.random((nouter, n, ip)) - 0.5)).astype(ct)
        dA = DeviceArray.from_numpy(A)
        dB = DeviceArray.zeros((nouter, n, op), ct)
        for inverse in (0, 1):
            _lib.call("mfft_c2c_strided", dA.ptr, dB.ptr, n, nouter, ncols, n * ip, ip, n * op, op, inverse, _lib.precision_code(prec))
            f = np.fft.ifft if inverse else np.fft.fft
            want = f(A[:, :, :ncols].astype(np.complex128), axis=1)
            assert orc.rel_l2(dB.get()[:, :, :ncols], want) < TOL[prec], (n, inverse)
        _lib.call("mfft_c2c_strided", dA.ptr, dA.ptr, n, nouter, ncols, n * ip, ip, n * ip, ip, 0, _lib.precision_code(prec))
        got = dA.get()
        assert orc.rel_l2(got[:, :, :ncols], np.fft.fft(A[:, :, :ncols].astype(np.complex128), axis=1)) < TOL[prec]
        assert np.array_equal(got[:, :, ncols:], A[:, :, ncols:])          # the columns between the rows are not touched
    with pytest.raises(_lib.MfftError):
        _lib.call("mfft_c2c_strided", dA.ptr, dA.ptr, 8, 1, 16, 0, 8, 0, 16, 0, 1)
