"""CPU oracle for the slab / pencil distributed 3-D FFT hot path.

TEST INFRASTRUCTURE ONLY.  This file is a numpy restatement of the algorithm
that spectralDNS/mpiFFT4py runs on its hot path.  It exists so that the HIP
path can be checked against something that is (a) independent of the HIP code
and (b) pinned against the real reference.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product package ``mpifft4py_amd`` never does.

Pinning status: PINNED.  ``oracle/refharness/check_oracle_vs_reference.py``
imports the real reference from /root/reference (dev container only, through a
threads-as-ranks fake mpi4py) and checks every function below against it for
slab/pencil x Alltoall/Alltoallw x P in {1,2,4,8}; the committed fixtures under
``tests/golden/`` were produced from the *reference*, not from this file
(``oracle/refharness/make_golden.py``), and ``tests/test_oracle_golden.py``
re-checks this file against them on every run.

The arithmetic of the path lives in a third-party dependency of the reference
(FFTW via pyfftw, or numpy.fft/pocketfft; un-pinned in the reference's
requirements.txt:3 / conf/conda/meta.yaml:24-25).  The DFT values are therefore
pinned by numpy.fft (pocketfft) itself, which is what the reference's own
tests compare against (tests/test_FFT.py:66-91).

Formulation: "world" style.  Every function takes the list of all ranks' local
arrays and returns the list of all ranks' results; the exchange steps are
explicit array copies.  No threads, no MPI.

Reference map (all paths relative to /root/reference):
  slab layout            mpiFFT4py/slab.py:75-144
  slab forward           mpiFFT4py/slab.py:349-443   (rfft2 -> pack -> Alltoall -> fft x)
  slab inverse           mpiFFT4py/slab.py:214-308   (ifft x -> Alltoall -> unpack -> irfft2)
  slab 3/2-rule          mpiFFT4py/slab.py:250-268, 310-344, 372-386, 445-483, 516-536
  slab C2C               mpiFFT4py/slab.py:538-825
  pencil chunk rule      mpiFFT4py/pencil.py:76-90
  pencil layouts         mpiFFT4py/pencil.py:190-199, 248-287, 908-943
  pencil Y forward/inv   mpiFFT4py/pencil.py:730-754 / 483-507
  pencil X forward/inv   mpiFFT4py/pencil.py:1312-1337 / 1082-1105
  pencil 3/2-rule        mpiFFT4py/pencil.py:351-379, 604-632, 858-883, 1196-1224, 1440-1475
  line (2-D) class       mpiFFT4py/line.py:41-340
  transpose_Uc           mpiFFT4py/cython/maths.pyx:21-31
  dealias_filter         mpiFFT4py/cython/maths.pyx:9-19
"""
import numpy as np

# --------------------------------------------------------------------------
# dtypes (mpibase.py:133-137)
# --------------------------------------------------------------------------

def dtypes(precision):
    assert precision in ("single", "double")
    return ((np.float32, np.complex64) if precision == "single"
            else (np.float64, np.complex128))


# --------------------------------------------------------------------------
# chunking rules
# --------------------------------------------------------------------------

def pencil_chunks(n, size):
    """(length, start) per rank; remainder (only 1 is handled upstream) goes to
    the LAST rank.  pencil.py:80-90."""
    q, r = divmod(n, size)
    out = []
    for i in range(size):
        ln = q + (1 if (r == 1 and i == size - 1) else 0)
        out.append((ln, q * i))
    return out


def slab_chunks(n, size):
    """(length, start) per rank, remainder spread over the first ranks.
    slab.py:34-47."""
    q, r = divmod(n, size)
    out = []
    for i in range(size):
        if i < r:
            out.append((q + 1, q * i + i))
        else:
            out.append((q, q * i + r))
    return out


def compute_dims(nprocs):
    """Balanced non-increasing 2-factorisation, as MPI.Compute_dims(n, 2)
    returns for the power-of-two sizes the reference accepts (pencil.py:185)."""
    best = (nprocs, 1)
    a = 1
    while a * a <= nprocs:
        if nprocs % a == 0:
            best = (nprocs // a, a)
        a += 1
    return best


# --------------------------------------------------------------------------
# slab layout (slab.py:98-144)
# --------------------------------------------------------------------------

class SlabLayout:
    def __init__(self, N, P, kind="R2C", padsize=1.5):
        N = np.asarray(N, dtype=int)
        assert len(N) == 3
        # slab.py:89-91 : P must be a power of two not larger than N[0]
        if P not in [2 ** i for i in range(int(np.log2(N[0])) + 1)]:
            raise IOError("Number of cpus must be a power of two <= N[0]")
        self.N, self.P, self.kind, self.padsize = N, P, kind, padsize
        self.Np = N // P
        self.Nf = int(N[2] // 2 + 1) if kind == "R2C" else int(N[2])
        self.Nfp = (int(padsize * N[2] // 2 + 1) if kind == "R2C"
                    else int(padsize * N[2]))

    def real_shape(self):
        return (int(self.Np[0]), int(self.N[1]), int(self.N[2]))

    def complex_shape(self):
        return (int(self.N[0]), int(self.Np[1]), self.Nf)

    def complex_shape_T(self):
        return (int(self.Np[0]), int(self.N[1]), self.Nf)

    def real_shape_padded(self):
        p = self.padsize
        return (int(p * self.Np[0]), int(p * self.N[1]), int(p * self.N[2]))

    def global_real_shape(self):
        return tuple(int(n) for n in self.N)

    def global_complex_shape(self, padsize=1.0):
        if self.kind == "R2C":
            return (int(padsize * self.N[0]), int(padsize * self.N[1]),
                    int(padsize * self.N[2] // 2 + 1))
        return (int(padsize * self.N[0]), int(padsize * self.N[1]),
                int(padsize * self.N[2]))

    def real_local_slice(self, rank, padsize=1):
        return (slice(int(padsize * rank * self.Np[0]),
                      int(padsize * (rank + 1) * self.Np[0]), 1),
                slice(0, int(padsize * self.N[1]), 1),
                slice(0, int(padsize * self.N[2]), 1))

    def complex_local_slice(self, rank):
        return (slice(0, int(self.N[0]), 1),
                slice(int(rank * self.Np[1]), int((rank + 1) * self.Np[1]), 1),
                slice(0, self.Nf, 1))


# --------------------------------------------------------------------------
# slab pack / unpack (slab.py:403, maths.pyx:21-31)
# --------------------------------------------------------------------------

def slab_pack(Uc_hatT, P):
    """U_mpi[p, i, j, k] = Uc_hatT[i, p*Np1 + j, k]   (slab.py:403)."""
    Np0, N1, Nf = Uc_hatT.shape
    Np1 = N1 // P
    return np.ascontiguousarray(
        Uc_hatT.reshape(Np0, P, Np1, Nf).transpose(1, 0, 2, 3))


def slab_unpack(U_mpi):
    """Uc_hatT[i, p*Np1 + j, k] = U_mpi[p, i, j, k]   (maths.pyx:21-31)."""
    P, Np0, Np1, Nf = U_mpi.shape
    return np.ascontiguousarray(
        U_mpi.transpose(1, 0, 2, 3)).reshape(Np0, P * Np1, Nf)


def alltoall_world(send):
    """Equal-chunk all-to-all on flat buffers: recv[r] chunk j = send[j] chunk r."""
    P = len(send)
    flat = [s.reshape(P, -1) for s in send]
    return [np.concatenate([flat[j][r] for j in range(P)]) for r in range(P)]


# --------------------------------------------------------------------------
# slab R2C forward / inverse, un-padded
# --------------------------------------------------------------------------

def slab_r2c_forward(us, N, precision="double", trace=None):
    """us[r] has real_shape; returns fu[r] of complex_shape.  slab.py:349-443."""
    P = len(us)
    lay = SlabLayout(N, P)
    _, ctype = dtypes(precision)
    if P == 1:
        return [np.fft.rfftn(us[0], axes=(0, 1, 2)).astype(ctype)]   # slab.py:369
    Uc_hatT = [np.fft.rfft2(u, axes=(1, 2)).astype(ctype) for u in us]  # :400
    packed = [slab_pack(a, P) for a in Uc_hatT]                         # :403
    recv = alltoall_world(packed)                                       # :406
    Uc_hat = [r.reshape(lay.complex_shape()) for r in recv]
    if trace is not None:
        trace.update(Uc_hatT=Uc_hatT, packed=packed, Uc_hat=Uc_hat)
    return [np.fft.fft(a, axis=0).astype(ctype) for a in Uc_hat]        # :442


def slab_r2c_backward(fus, N, precision="double", trace=None):
    """fus[r] has complex_shape; returns u[r] of real_shape.  slab.py:214-308."""
    P = len(fus)
    lay = SlabLayout(N, P)
    rtype, ctype = dtypes(precision)
    if P == 1:
        return [np.fft.irfftn(fus[0], s=lay.global_real_shape(),
                              axes=(0, 1, 2)).astype(rtype)]            # :249
    Uc_hat = [np.fft.ifft(f, axis=0).astype(ctype) for f in fus]        # :275
    recv = alltoall_world(Uc_hat)                                       # :281
    Np0, Np1, Nf = int(lay.Np[0]), int(lay.Np[1]), lay.Nf
    Uc_hatT = [slab_unpack(r.reshape(P, Np0, Np1, Nf)) for r in recv]   # :284
    if trace is not None:
        trace.update(Uc_hat=Uc_hat, Uc_hatT=Uc_hatT)
    return [np.fft.irfft2(a, s=(int(N[1]), int(N[2])), axes=(1, 2)).astype(rtype)
            for a in Uc_hatT]                                           # :306


# --------------------------------------------------------------------------
# slab C2C (slab.py:638-669, 743-772)
# --------------------------------------------------------------------------

def slab_c2c_forward(us, N, precision="double"):
    P = len(us)
    lay = SlabLayout(N, P, kind="C2C")
    _, ctype = dtypes(precision)
    if P == 1:
        return [np.fft.fftn(us[0], axes=(0, 1, 2)).astype(ctype)]
    UT = [np.fft.fft2(u, axes=(1, 2)).astype(ctype) for u in us]
    recv = alltoall_world([slab_pack(a, P) for a in UT])
    return [np.fft.fft(r.reshape(lay.complex_shape()), axis=0).astype(ctype)
            for r in recv]


def slab_c2c_backward(fus, N, precision="double"):
    P = len(fus)
    lay = SlabLayout(N, P, kind="C2C")
    _, ctype = dtypes(precision)
    if P == 1:
        return [np.fft.ifftn(fus[0], axes=(0, 1, 2)).astype(ctype)]
    Uc_hat = [np.fft.ifft(f, axis=0).astype(ctype) for f in fus]
    recv = alltoall_world(Uc_hat)
    Np0, Np1, Nf = int(lay.Np[0]), int(lay.Np[1]), lay.Nf
    UT = [slab_unpack(r.reshape(P, Np0, Np1, Nf)) for r in recv]
    return [np.fft.ifft2(a, axes=(1, 2)).astype(ctype) for a in UT]


# --------------------------------------------------------------------------
# 3/2-rule helpers for the R2C classes (slab.py:516-536, pencil.py:351-379)
# --------------------------------------------------------------------------

def pad_axis(fu, npad, n, axis):
    """Zero-pad the two-sided spectrum axis of original length n to npad:
    low half to the front, high half to the back (slab.py:518-523)."""
    shp = list(fu.shape)
    shp[axis] = npad
    fp = np.zeros(shp, dtype=fu.dtype)
    lo = [slice(None)] * fu.ndim
    hi_src = [slice(None)] * fu.ndim
    hi_dst = [slice(None)] * fu.ndim
    lo[axis] = slice(0, n // 2)
    hi_src[axis] = slice(n // 2, n)
    hi_dst[axis] = slice(npad - n // 2, npad)
    fp[tuple(lo)] = fu[tuple(lo)]
    fp[tuple(hi_dst)] = fu[tuple(hi_src)]
    return fp


def trunc_axis(fp, n, axis):
    """Inverse of pad_axis with the Nyquist fold: out[:n/2+1] = fp[:n/2+1];
    out[n/2:] += fp[-n/2:]   (slab.py:529-533, pencil.py:367-379)."""
    shp = list(fp.shape)
    npad = shp[axis]
    shp[axis] = n
    fu = np.zeros(shp, dtype=fp.dtype)
    a = [slice(None)] * fp.ndim
    b_dst = [slice(None)] * fp.ndim
    b_src = [slice(None)] * fp.ndim
    a[axis] = slice(0, n // 2 + 1)
    b_dst[axis] = slice(n // 2, n)
    b_src[axis] = slice(npad - n // 2, npad)
    fu[tuple(a)] = fp[tuple(a)]
    fu[tuple(b_dst)] += fp[tuple(b_src)]
    return fu


def pad_z(fu, nfp):
    """fp[:, :, :Nf] = fu (slab.py:524-525)."""
    fp = np.zeros(fu.shape[:2] + (nfp,), dtype=fu.dtype)
    fp[:, :, :fu.shape[2]] = fu
    return fp


# --------------------------------------------------------------------------
# slab R2C, 3/2-rule (slab.py:250-268, 310-344, 372-386, 445-483)
# --------------------------------------------------------------------------

def slab_r2c_backward_padded(fus, N, precision="double", padsize=1.5):
    P = len(fus)
    lay = SlabLayout(N, P, padsize=padsize)
    rtype, ctype = dtypes(precision)
    N = lay.N
    M0, M1, M2 = (int(padsize * n) for n in N)
    if P == 1:
        f = fus[0] * padsize ** 3
        f = pad_axis(f, M0, int(N[0]), 0)
        f = pad_axis(f, M1, int(N[1]), 1)
        f = pad_z(f, lay.Nfp)
        return [np.fft.irfftn(f, s=(M0, M1, M2), axes=(0, 1, 2)).astype(rtype)]
    assert P <= N[0] // 2
    # pad in x, ifft x (slab.py:321-323)
    a = [np.fft.ifft(pad_axis(f * padsize ** 3, M0, int(N[0]), 0), axis=0).astype(ctype)
         for f in fus]
    # exchange: x split in P chunks of padsize*Np0, gather y (slab.py:325-334)
    Mp0 = int(padsize * lay.Np[0])
    Np1 = int(lay.Np[1])
    recv = alltoall_world(a)
    b = [slab_unpack(r.reshape(P, Mp0, Np1, lay.Nf)) for r in recv]
    # pad y, ifft y (slab.py:337-339); pad z, irfft z (slab.py:342-344)
    out = []
    for x in b:
        x = np.fft.ifft(pad_axis(x, M1, int(N[1]), 1), axis=1).astype(ctype)
        x = pad_z(x, lay.Nfp)
        out.append(np.fft.irfft(x, n=M2, axis=2).astype(rtype))
    return out


def slab_r2c_forward_padded(us, N, precision="double", padsize=1.5):
    P = len(us)
    lay = SlabLayout(N, P, padsize=padsize)
    _, ctype = dtypes(precision)
    N = lay.N
    Nf = lay.Nf
    if P == 1:
        fp = np.fft.rfftn(us[0], axes=(0, 1, 2)).astype(ctype)
        fp = fp[:, :, :Nf]
        fp = trunc_axis(fp, int(N[1]), 1)
        fp = trunc_axis(fp, int(N[0]), 0)
        return [(fp / padsize ** 3).astype(ctype)]
    assert P <= N[0] // 2
    Mp0 = int(padsize * lay.Np[0])
    a = []
    for u in us:
        x = np.fft.rfft2(u, axes=(1, 2)).astype(ctype)          # slab.py:455
        x = trunc_axis(x[:, :, :Nf], int(N[1]), 1)              # :459 (copy_from_padded axis 1)
        a.append(slab_pack(x, P))                               # :463
    recv = alltoall_world(a)                                    # :465
    out = []
    for r in recv:
        x = np.fft.fft(r.reshape(P * Mp0, int(lay.Np[1]), Nf), axis=0).astype(ctype)  # :476
        out.append((trunc_axis(x, int(N[0]), 0) / padsize ** 3).astype(ctype))        # :480-483
    return out


# --------------------------------------------------------------------------
# slab C2C, 3/2-rule (slab.py:618-632, 671-698, 727-741, 774-799).  The reference
# treats the Nyquist modes differently for P == 1 (plain corner copies, no fold)
# and P > 1 (copy_from_padded folds y and z, the final x truncation does not);
# both branches are restated as they are.  Padded work arrays are zero outside
# the copied corners (first-use state of the reference's work-array cache).
# --------------------------------------------------------------------------

def trunc_axis_nofold(fp, n, axis):
    """out[:n/2] = fp[:n/2]; out[n/2:] = fp[-n/2:]   (slab.py:736-739, 796-797)."""
    npad = fp.shape[axis]
    lo = [slice(None)] * fp.ndim
    hi = [slice(None)] * fp.ndim
    lo[axis] = slice(0, n // 2)
    hi[axis] = slice(npad - n // 2, npad)
    return np.concatenate([fp[tuple(lo)], fp[tuple(hi)]], axis=axis)


def slab_c2c_backward_padded(fus, N, precision="double", padsize=1.5):
    P = len(fus)
    lay = SlabLayout(N, P, kind="C2C", padsize=padsize)
    _, ctype = dtypes(precision)
    N = lay.N
    M0, M1, M2 = (int(padsize * n) for n in N)
    if P == 1:
        f = fus[0] * padsize ** 3
        f = pad_axis(pad_axis(pad_axis(f, M0, int(N[0]), 0), M1, int(N[1]), 1), M2, int(N[2]), 2)
        return [np.fft.ifftn(f, axes=(0, 1, 2)).astype(ctype)]
    Mp0, Np1, Nf = int(padsize * lay.Np[0]), int(lay.Np[1]), lay.Nf
    a = [np.fft.ifft(pad_axis(f * padsize ** 3, M0, int(N[0]), 0), axis=0).astype(ctype) for f in fus]
    recv = alltoall_world(a)
    out = []
    for r in recv:
        x = slab_unpack(r.reshape(P, Mp0, Np1, Nf))
        x = np.fft.ifft(pad_axis(x, M1, int(N[1]), 1), axis=1).astype(ctype)
        x = np.fft.ifft(pad_axis(x, M2, int(N[2]), 2), axis=2).astype(ctype)
        out.append(x)
    return out


def slab_c2c_forward_padded(us, N, precision="double", padsize=1.5):
    P = len(us)
    lay = SlabLayout(N, P, kind="C2C", padsize=padsize)
    _, ctype = dtypes(precision)
    N = lay.N
    if P == 1:
        fp = np.fft.fftn(us[0], axes=(0, 1, 2)).astype(ctype)
        for ax in (2, 1, 0):
            fp = trunc_axis_nofold(fp, int(N[ax]), ax)
        return [(fp / padsize ** 3).astype(ctype)]
    Mp0, Np1, Nf = int(padsize * lay.Np[0]), int(lay.Np[1]), lay.Nf
    a = []
    for u in us:
        x = np.fft.fft2(u, axes=(1, 2)).astype(ctype)
        x = trunc_axis(trunc_axis(x, int(N[2]), 2), int(N[1]), 1)     # copy_from_padded: folds in y and z
        a.append(slab_pack(x, P))
    recv = alltoall_world(a)
    out = []
    for r in recv:
        x = np.fft.fft(r.reshape(P * Mp0, Np1, Nf), axis=0).astype(ctype)
        out.append((trunc_axis_nofold(x, int(N[0]), 0) / padsize ** 3).astype(ctype))
    return out


# --------------------------------------------------------------------------
# pencil layouts (pencil.py:190-199, 248-287, 908-943)
# --------------------------------------------------------------------------

class PencilLayout:
    """alignment 'X' (R2CX) or 'Y' (R2CY); comm0_rank = rank % P1,
    comm1_rank = rank // P1 (pencil.py:192-195 with the legacy int colour)."""

    def __init__(self, N, P, P1=None, alignment="X", padsize=1.5):
        N = np.asarray(N, dtype=int)
        assert len(N) == 3
        assert P > 1                                  # pencil.py:176
        if P1 is None:
            P1, P2 = compute_dims(P)
        else:
            P2 = P // P1
        if not (P % 2 == 0 or P == 1):
            raise IOError("Number of cpus must be even")            # :201-202
        if (P1 % 2 != 0) or (P2 % 2 != 0):
            raise IOError("Number of cpus in each direction must be even")  # :204-205
        self.N, self.P, self.P1, self.P2 = N, P, P1, P2
        self.alignment, self.padsize = alignment, padsize
        self.N1 = N // P1
        self.N2 = N // P2
        self.Nf = int(N[2] // 2 + 1)

    def ranks(self, rank):
        return rank % self.P1, rank // self.P1        # (comm0_rank, comm1_rank)

    def N1f(self, c0):                                 # pencil.py:197
        h = int(self.N1[2] // 2)
        return h if c0 < self.P1 - 1 else h + 1

    def N2f(self, c1):                                 # pencil.py:908
        h = int(self.N2[2] // 2)
        return h if c1 < self.P2 - 1 else h + 1

    def real_shape(self):
        return (int(self.N1[0]), int(self.N2[1]), int(self.N[2]))

    def real_shape_padded(self):
        p = self.padsize
        return (int(p * self.N1[0]), int(p * self.N2[1]), int(p * self.N[2]))

    def complex_shape(self, rank):
        c0, c1 = self.ranks(rank)
        if self.alignment == "Y":
            return (int(self.N2[0]), int(self.N[1]), self.N1f(c0))
        return (int(self.N[0]), int(self.N1[1]), self.N2f(c1))

    def real_local_slice(self, rank, padsize=1):
        c0, c1 = self.ranks(rank)
        return (slice(int(padsize * c0 * self.N1[0]), int(padsize * (c0 + 1) * self.N1[0]), 1),
                slice(int(padsize * c1 * self.N2[1]), int(padsize * (c1 + 1) * self.N2[1]), 1),
                slice(0, int(padsize * self.N[2])))

    def complex_local_slice(self, rank):
        c0, c1 = self.ranks(rank)
        if self.alignment == "Y":
            z0 = int(c0 * self.N1[2] // 2)
            return (slice(int(c1 * self.N2[0]), int((c1 + 1) * self.N2[0]), 1),
                    slice(0, int(self.N[1])),
                    slice(z0, z0 + self.N1f(c0), 1))
        z0 = int(c1 * self.N2[2] // 2)
        return (slice(0, int(self.N[0])),
                slice(int(c0 * self.N1[1]), int((c0 + 1) * self.N1[1]), 1),
                slice(z0, z0 + self.N2f(c1), 1))

    def global_complex_shape(self, padsize=1.0):
        return (int(padsize * self.N[0]), int(padsize * self.N[1]),
                int(padsize * self.N[2] // 2 + 1))

    def comm0_members(self, rank):
        """world ranks sharing rank//P1 (consecutive), ordered by comm0_rank."""
        base = (rank // self.P1) * self.P1
        return [base + i for i in range(self.P1)]

    def comm1_members(self, rank):
        """world ranks sharing rank%P1 (stride P1), ordered by comm1_rank."""
        return [rank % self.P1 + i * self.P1 for i in range(self.P2)]


def _exchange_split_gather(bufs, groups, split_axis, gather_axis, split_chunks,
                           gather_chunks, out_shape_of):
    """Generic Alltoallw over sub-groups: every member splits `split_axis` of
    its buffer by split_chunks and the receiver concatenates what it gets along
    `gather_axis` in group order (the subarray types of pencil.py:218-246,
    971-999 describe exactly these boxes)."""
    out = [None] * len(bufs)
    for members in groups:
        for gi, r in enumerate(members):
            ln, st = split_chunks[gi]
            parts = []
            for gj, s in enumerate(members):
                sl = [slice(None)] * 3
                sl[split_axis] = slice(st, st + ln)
                parts.append(bufs[s][tuple(sl)])
            res = np.concatenate(parts, axis=gather_axis)
            assert res.shape == tuple(out_shape_of(r)), (res.shape, out_shape_of(r))
            out[r] = np.ascontiguousarray(res)
    return out


def _groups(lay, which):
    seen, out = set(), []
    for r in range(lay.P):
        m = tuple(lay.comm0_members(r) if which == 0 else lay.comm1_members(r))
        if m not in seen:
            seen.add(m)
            out.append(list(m))
    return out


# --------------------------------------------------------------------------
# pencil forward / inverse (un-padded)
# --------------------------------------------------------------------------

def pencil_r2c_forward(us, N, P1=None, alignment="X", precision="double"):
    P = len(us)
    lay = PencilLayout(N, P, P1, alignment)
    _, ctype = dtypes(precision)
    N = lay.N
    P1, P2, Nf = lay.P1, lay.P2, lay.Nf
    N1, N2 = lay.N1, lay.N2
    a = [np.fft.rfft(u, axis=2).astype(ctype) for u in us]       # (N1[0], N2[1], Nf)
    if alignment == "Y":
        # pencil.py:738-753: z -> P1 chunks over comm0, gather x; fft x;
        # x -> P2 chunks over comm1, gather y; fft y
        zc = pencil_chunks(Nf, P1)
        b = _exchange_split_gather(
            a, _groups(lay, 0), 2, 0, zc, None,
            lambda r: (int(N[0]), int(N2[1]), lay.N1f(lay.ranks(r)[0])))
        b = [np.fft.fft(x, axis=0).astype(ctype) for x in b]
        xc = pencil_chunks(int(N[0]), P2)
        c = _exchange_split_gather(
            b, _groups(lay, 1), 0, 1, xc, None,
            lambda r: (int(N2[0]), int(N[1]), lay.N1f(lay.ranks(r)[0])))
        return [np.fft.fft(x, axis=1).astype(ctype) for x in c]
    # alignment X, pencil.py:1321-1336: z -> P2 chunks over comm1, gather y;
    # fft y; y -> P1 chunks over comm0, gather x; fft x
    zc = pencil_chunks(Nf, P2)
    b = _exchange_split_gather(
        a, _groups(lay, 1), 2, 1, zc, None,
        lambda r: (int(N1[0]), int(N[1]), lay.N2f(lay.ranks(r)[1])))
    b = [np.fft.fft(x, axis=1).astype(ctype) for x in b]
    yc = pencil_chunks(int(N[1]), P1)
    c = _exchange_split_gather(
        b, _groups(lay, 0), 1, 0, yc, None,
        lambda r: (int(N[0]), int(N1[1]), lay.N2f(lay.ranks(r)[1])))
    return [np.fft.fft(x, axis=0).astype(ctype) for x in c]


def pencil_r2c_backward(fus, N, P1=None, alignment="X", precision="double"):
    P = len(fus)
    lay = PencilLayout(N, P, P1, alignment)
    rtype, ctype = dtypes(precision)
    N = lay.N
    P1, P2, Nf = lay.P1, lay.P2, lay.Nf
    N1, N2 = lay.N1, lay.N2
    zshape = lambda r: (int(N1[0]), int(N2[1]), Nf)
    if alignment == "Y":
        # pencil.py:487-507
        a = [np.fft.ifft(f, axis=1).astype(ctype) for f in fus]
        yc = pencil_chunks(int(N[1]), P2)
        b = _exchange_split_gather(
            a, _groups(lay, 1), 1, 0, yc, None,
            lambda r: (int(N[0]), int(N2[1]), lay.N1f(lay.ranks(r)[0])))
        b = [np.fft.ifft(x, axis=0).astype(ctype) for x in b]
        xc = pencil_chunks(int(N[0]), P1)
        c = _exchange_split_gather(b, _groups(lay, 0), 0, 2, xc, None, zshape)
    else:
        # pencil.py:1086-1105
        a = [np.fft.ifft(f, axis=0).astype(ctype) for f in fus]
        xc = pencil_chunks(int(N[0]), P1)
        b = _exchange_split_gather(
            a, _groups(lay, 0), 0, 1, xc, None,
            lambda r: (int(N1[0]), int(N[1]), lay.N2f(lay.ranks(r)[1])))
        b = [np.fft.ifft(x, axis=1).astype(ctype) for x in b]
        yc = pencil_chunks(int(N[1]), P2)
        c = _exchange_split_gather(b, _groups(lay, 1), 1, 2, yc, None, zshape)
    return [np.fft.irfft(x, n=int(N[2]), axis=2).astype(rtype) for x in c]


# --------------------------------------------------------------------------
# pencil 'AlltoallN' mode (pencil.py:410-432, 647-668, 1024-1047, 1237-1261): the z-Nyquist
# column is neglected so that every rank holds N2/(2*Pz) columns; the inverse sets it to zero.
# --------------------------------------------------------------------------

class PencilNLayout(PencilLayout):
    def N1f(self, c0):
        return int(self.N1[2] // 2)

    def N2f(self, c1):
        return int(self.N2[2] // 2)


def pencil_r2c_forward_n(us, N, P1=None, alignment="X", precision="double"):
    P = len(us)
    lay = PencilNLayout(N, P, P1, alignment)
    _, ctype = dtypes(precision)
    N = lay.N
    P1, P2 = lay.P1, lay.P2
    N1, N2 = lay.N1, lay.N2
    half = int(N[2] // 2)
    a = [np.fft.rfft(u, axis=2).astype(ctype)[:, :, :half] for u in us]
    if alignment == "Y":
        b = _exchange_split_gather(a, _groups(lay, 0), 2, 0, pencil_chunks(half, P1), None,
                                   lambda r: (int(N[0]), int(N2[1]), half // P1))
        b = [np.fft.fft(x, axis=0).astype(ctype) for x in b]
        c = _exchange_split_gather(b, _groups(lay, 1), 0, 1, pencil_chunks(int(N[0]), P2), None,
                                   lambda r: (int(N2[0]), int(N[1]), half // P1))
        return [np.fft.fft(x, axis=1).astype(ctype) for x in c]
    b = _exchange_split_gather(a, _groups(lay, 1), 2, 1, pencil_chunks(half, P2), None,
                               lambda r: (int(N1[0]), int(N[1]), half // P2))
    b = [np.fft.fft(x, axis=1).astype(ctype) for x in b]
    c = _exchange_split_gather(b, _groups(lay, 0), 1, 0, pencil_chunks(int(N[1]), P1), None,
                               lambda r: (int(N[0]), int(N1[1]), half // P2))
    return [np.fft.fft(x, axis=0).astype(ctype) for x in c]


def pencil_r2c_backward_n(fus, N, P1=None, alignment="X", precision="double"):
    P = len(fus)
    lay = PencilNLayout(N, P, P1, alignment)
    rtype, ctype = dtypes(precision)
    N = lay.N
    P1, P2 = lay.P1, lay.P2
    N1, N2 = lay.N1, lay.N2
    half = int(N[2] // 2)
    zshape = lambda r: (int(N1[0]), int(N2[1]), half)
    if alignment == "Y":
        a = [np.fft.ifft(f, axis=1).astype(ctype) for f in fus]
        b = _exchange_split_gather(a, _groups(lay, 1), 1, 0, pencil_chunks(int(N[1]), P2), None,
                                   lambda r: (int(N[0]), int(N2[1]), half // P1))
        b = [np.fft.ifft(x, axis=0).astype(ctype) for x in b]
        c = _exchange_split_gather(b, _groups(lay, 0), 0, 2, pencil_chunks(int(N[0]), P1), None, zshape)
    else:
        a = [np.fft.ifft(f, axis=0).astype(ctype) for f in fus]
        b = _exchange_split_gather(a, _groups(lay, 0), 0, 1, pencil_chunks(int(N[0]), P1), None,
                                   lambda r: (int(N1[0]), int(N[1]), half // P2))
        b = [np.fft.ifft(x, axis=1).astype(ctype) for x in b]
        c = _exchange_split_gather(b, _groups(lay, 1), 1, 2, pencil_chunks(int(N[1]), P2), None, zshape)
    out = []
    for x in c:
        z = np.zeros(x.shape[:2] + (half + 1,), dtype=ctype)
        z[:, :, :half] = x
        out.append(np.fft.irfft(z, n=int(N[2]), axis=2).astype(rtype))
    return out


# --------------------------------------------------------------------------
# EXTENSION, PARITY UNPINNED: pencil C2C.  The reference has no pencil C2C class
# (pencil.py defines only R2CY / R2CX), so there is nothing to pin this against
# except the DFT definition itself (numpy.fft.fftn of the gathered array).  Same
# stage order and exchanges as the R2C pencils with Nf = N2 split evenly.
# --------------------------------------------------------------------------

class PencilC2CLayout(PencilLayout):
    def __init__(self, N, P, P1=None, alignment="X"):
        PencilLayout.__init__(self, N, P, P1, alignment)
        self.Nf = int(self.N[2])

    def N1f(self, c0):
        return int(self.N1[2])

    def N2f(self, c1):
        return int(self.N2[2])

    def complex_local_slice(self, rank):
        c0, c1 = self.ranks(rank)
        if self.alignment == "Y":
            z0 = int(c0 * self.N1[2])
            return (slice(int(c1 * self.N2[0]), int((c1 + 1) * self.N2[0]), 1),
                    slice(0, int(self.N[1])), slice(z0, z0 + int(self.N1[2]), 1))
        z0 = int(c1 * self.N2[2])
        return (slice(0, int(self.N[0])),
                slice(int(c0 * self.N1[1]), int((c0 + 1) * self.N1[1]), 1),
                slice(z0, z0 + int(self.N2[2]), 1))

    def global_complex_shape(self, padsize=1.0):
        return tuple(int(padsize * n) for n in self.N)


def pencil_c2c_forward(us, N, P1=None, alignment="X", precision="double"):
    P = len(us)
    lay = PencilC2CLayout(N, P, P1, alignment)
    _, ctype = dtypes(precision)
    N = lay.N
    P1, P2 = lay.P1, lay.P2
    N1, N2 = lay.N1, lay.N2
    a = [np.fft.fft(u, axis=2).astype(ctype) for u in us]
    if alignment == "Y":
        b = _exchange_split_gather(a, _groups(lay, 0), 2, 0, pencil_chunks(int(N[2]), P1), None,
                                   lambda r: (int(N[0]), int(N2[1]), int(N1[2])))
        b = [np.fft.fft(x, axis=0).astype(ctype) for x in b]
        c = _exchange_split_gather(b, _groups(lay, 1), 0, 1, pencil_chunks(int(N[0]), P2), None,
                                   lambda r: (int(N2[0]), int(N[1]), int(N1[2])))
        return [np.fft.fft(x, axis=1).astype(ctype) for x in c]
    b = _exchange_split_gather(a, _groups(lay, 1), 2, 1, pencil_chunks(int(N[2]), P2), None,
                               lambda r: (int(N1[0]), int(N[1]), int(N2[2])))
    b = [np.fft.fft(x, axis=1).astype(ctype) for x in b]
    c = _exchange_split_gather(b, _groups(lay, 0), 1, 0, pencil_chunks(int(N[1]), P1), None,
                               lambda r: (int(N[0]), int(N1[1]), int(N2[2])))
    return [np.fft.fft(x, axis=0).astype(ctype) for x in c]


# --------------------------------------------------------------------------
# pencil, 3/2-rule (Alltoallw branches: pencil.py:604-632, 858-883, 1196-1224,
# 1440-1475).  Padding of a distributed axis happens right before the FFT
# along it, when that axis is locally complete.
# --------------------------------------------------------------------------

def pencil_r2c_backward_padded(fus, N, P1=None, alignment="X",
                               precision="double", padsize=1.5):
    P = len(fus)
    lay = PencilLayout(N, P, P1, alignment, padsize)
    rtype, ctype = dtypes(precision)
    N = lay.N
    P1, P2, Nf = lay.P1, lay.P2, lay.Nf
    M0, M1, M2 = (int(padsize * n) for n in N)
    Nfp = int(padsize * N[2] // 2) + 1
    if alignment == "Y":
        a = [np.fft.ifft(pad_axis(f * padsize ** 3, M1, int(N[1]), 1), axis=1).astype(ctype)
             for f in fus]
        yc = pencil_chunks(M1, P2)
        b = _exchange_split_gather(
            a, _groups(lay, 1), 1, 0, yc, None,
            lambda r: (int(N[0]), M1 // P2, lay.N1f(lay.ranks(r)[0])))
        b = [np.fft.ifft(pad_axis(x, M0, int(N[0]), 0), axis=0).astype(ctype) for x in b]
        xc = pencil_chunks(M0, P1)
        c = _exchange_split_gather(
            b, _groups(lay, 0), 0, 2, xc, None,
            lambda r: (M0 // P1, M1 // P2, Nf))
    else:
        a = [np.fft.ifft(pad_axis(f * padsize ** 3, M0, int(N[0]), 0), axis=0).astype(ctype)
             for f in fus]
        xc = pencil_chunks(M0, P1)
        b = _exchange_split_gather(
            a, _groups(lay, 0), 0, 1, xc, None,
            lambda r: (M0 // P1, int(N[1]), lay.N2f(lay.ranks(r)[1])))
        b = [np.fft.ifft(pad_axis(x, M1, int(N[1]), 1), axis=1).astype(ctype) for x in b]
        yc = pencil_chunks(M1, P2)
        c = _exchange_split_gather(
            b, _groups(lay, 1), 1, 2, yc, None,
            lambda r: (M0 // P1, M1 // P2, Nf))
    return [np.fft.irfft(pad_z(x, Nfp), n=M2, axis=2).astype(rtype) for x in c]


def pencil_r2c_forward_padded(us, N, P1=None, alignment="X",
                              precision="double", padsize=1.5):
    P = len(us)
    lay = PencilLayout(N, P, P1, alignment, padsize)
    _, ctype = dtypes(precision)
    N = lay.N
    P1, P2, Nf = lay.P1, lay.P2, lay.Nf
    M0, M1, M2 = (int(padsize * n) for n in N)
    a = [np.fft.rfft(u, axis=2).astype(ctype)[:, :, :Nf] for u in us]   # (M0/P1, M1/P2, Nf)
    if alignment == "Y":
        zc = pencil_chunks(Nf, P1)
        b = _exchange_split_gather(
            a, _groups(lay, 0), 2, 0, zc, None,
            lambda r: (M0, M1 // P2, lay.N1f(lay.ranks(r)[0])))
        b = [trunc_axis(np.fft.fft(x, axis=0).astype(ctype), int(N[0]), 0) for x in b]
        xc = pencil_chunks(int(N[0]), P2)
        c = _exchange_split_gather(
            b, _groups(lay, 1), 0, 1, xc, None,
            lambda r: (int(N[0]) // P2, M1, lay.N1f(lay.ranks(r)[0])))
        return [(trunc_axis(np.fft.fft(x, axis=1).astype(ctype), int(N[1]), 1)
                 / padsize ** 3).astype(ctype) for x in c]
    zc = pencil_chunks(Nf, P2)
    b = _exchange_split_gather(
        a, _groups(lay, 1), 2, 1, zc, None,
        lambda r: (M0 // P1, M1, lay.N2f(lay.ranks(r)[1])))
    b = [trunc_axis(np.fft.fft(x, axis=1).astype(ctype), int(N[1]), 1) for x in b]
    yc = pencil_chunks(int(N[1]), P1)
    c = _exchange_split_gather(
        b, _groups(lay, 0), 1, 0, yc, None,
        lambda r: (M0, int(N[1]) // P1, lay.N2f(lay.ranks(r)[1])))
    return [(trunc_axis(np.fft.fft(x, axis=0).astype(ctype), int(N[0]), 0)
             / padsize ** 3).astype(ctype) for x in c]


# --------------------------------------------------------------------------
# 2-D "line" transforms (line.py:41-340): real (N0/P, N1), complex (N0, Npf) with
# the ky axis split over the ranks and the Nyquist column on the last one.
# --------------------------------------------------------------------------

class LineLayout:
    """line.py:64-103."""
    def __init__(self, N, P, padsize=1.5):
        self.N = [int(n) for n in N]
        self.P = int(P)
        self.padsize = padsize
        self.Np = [self.N[0] // P, self.N[1] // P]
        self.Nf = self.N[1] // 2 + 1

    def Npf(self, r):
        return self.Np[1] // 2 + 1 if r + 1 == self.P else self.Np[1] // 2

    def real_shape(self):
        return (self.Np[0], self.N[1])

    def complex_shape(self, r):
        return (self.N[0], self.Npf(r))

    def real_shape_padded(self):
        return (int(self.padsize * self.Np[0]), int(self.padsize * self.N[1]))

    def real_slice(self, r, padsize=1):
        return (slice(int(padsize * r * self.Np[0]), int(padsize * (r + 1) * self.Np[0])),
                slice(0, int(padsize * self.N[1])))

    def complex_slice(self, r):
        s = r * self.Np[1] // 2
        return (slice(0, self.N[0]), slice(s, s + self.Npf(r)))


def _exact_ks(n):
    """Signed frequency indices 0..n/2-1, -n/2..-1.  The reference computes them as
    (fftfreq(n) * n).astype(int) (line.py:61, slab.py:575), which TRUNCATES values such as 4.999999999999999 to 4 for
    n = 24, 28, 36, 48, 68, ... and then reads / writes the wrong rows on one rank (P = 1 padded paths of line.R2C
    and slab.C2C).  That float artefact is not reproduced: exact integers here and in the kernels; for the sizes
    where the reference's own indices are exact (every power of two, 12, 16, 20, ...) the results are identical,
    which is what the harness and the golden fixtures pin."""
    return np.rint(np.fft.fftfreq(int(n)) * int(n)).astype(int)


def _line_gather_x(cols, P):
    """The exchange of line.py:196-207 / 225-236 without the Nyquist packing: every rank ends up with all
    x rows of its own ky chunk.  cols[r] has shape (rows_r, Nf)."""
    lay_q = (cols[0].shape[1] - 1) // P
    full = np.concatenate(cols, axis=0)                 # (sum rows, Nf): x is rank-major
    out = []
    for r in range(P):
        hi = (r + 1) * lay_q + (1 if r == P - 1 else 0)
        out.append(np.ascontiguousarray(full[:, r * lay_q:hi]))
    return out


def _line_scatter_x(cols, P, rows):
    """Inverse of _line_gather_x: cols[r] (P*rows, chunk_r) -> per rank (rows, Nf)."""
    full = np.concatenate(cols, axis=1)
    return [np.ascontiguousarray(full[r * rows:(r + 1) * rows]) for r in range(P)]


def line_r2c_forward(us, N, precision="double"):
    """line.py:177-215 (P = 1: rfft2; P > 1: rfft y -> exchange -> fft x; the Nyquist packing of
    line.py:194, 209-215 is exact for real rows, so the result is the plain 2-D transform)."""
    rt, ct = dtypes(precision)
    P = len(us)
    if P == 1:
        return [np.fft.rfft2(us[0].astype(rt), axes=(0, 1)).astype(ct)]
    y = [np.fft.rfft(u.astype(rt), axis=1).astype(ct) for u in us]
    for a in y:                                        # bins 0 and N1/2 of a real row are real
        a[:, 0] = a[:, 0].real
        a[:, -1] = a[:, -1].real
    g = _line_gather_x(y, P)
    return [np.fft.fft(a, axis=0).astype(ct) for a in g]


def line_r2c_backward(fus, N, precision="double"):
    """line.py:275-311."""
    rt, ct = dtypes(precision)
    P = len(fus)
    if P == 1:
        return [np.fft.irfft2(fus[0].astype(ct), s=(int(N[0]), int(N[1])), axes=(0, 1)).astype(rt)]
    x = [np.fft.ifft(f.astype(ct), axis=0).astype(ct) for f in fus]
    rows = int(N[0]) // P
    parts = _line_scatter_x(x, P, rows)
    return [np.fft.irfft(a, n=int(N[1]), axis=1).astype(rt) for a in parts]


def line_r2c_backward_padded(fus, N, precision="double", padsize=1.5):
    """line.py:283-287 (P = 1) and 313-338 (P > 1)."""
    rt, ct = dtypes(precision)
    P = len(fus)
    N0, N1 = int(N[0]), int(N[1])
    M0, M1 = int(padsize * N0), int(padsize * N1)
    Nf, Mf = N1 // 2 + 1, int(padsize * N1 / 2 + 1)
    if P == 1:
        ks = _exact_ks(N0)
        fp = np.zeros((M0, Mf), dtype=ct)
        fp[ks, :Nf] = fus[0]
        return [np.fft.irfft2(fp * padsize ** 2, s=(M0, M1), axes=(0, 1)).astype(rt)]
    x = []
    for f in fus:
        fp = np.zeros((M0, f.shape[1]), dtype=ct)
        fp[:N0 // 2] = f[:N0 // 2]
        fp[-(N0 // 2):] = f[N0 // 2:]
        x.append(np.fft.ifft(fp, axis=0).astype(ct))
    rows = M0 // P
    parts = _line_scatter_x(x, P, rows)
    out = []
    for a in parts:
        fp = np.zeros((rows, Mf), dtype=ct)
        fp[:, :Nf] = a
        out.append(np.fft.irfft(fp * padsize ** 2, n=M1, axis=1).astype(rt))
    return out


def line_r2c_forward_padded(us, N, precision="double", padsize=1.5):
    """line.py:182-185 (P = 1: plain truncation, no fold) and 217-248 (P > 1).  For P > 1 the reference packs
    column Nf-1 of the padded y spectrum into the imaginary part of column 0 (line.py:231) although that bin
    is not real there; after the separation by Hermitian symmetry (swap_Nq, line.py:27-39) column 0 is the
    transform of Re(c0) - Im(cN) and the last column that of Re(cN).  Reproduced as such."""
    rt, ct = dtypes(precision)
    P = len(us)
    N0, N1 = int(N[0]), int(N[1])
    M0 = int(padsize * N0)
    Nf = N1 // 2 + 1
    if P == 1:
        ks = _exact_ks(N0)
        fp = np.fft.rfft2(us[0].astype(rt) / padsize ** 2, axes=(0, 1)).astype(ct)
        return [np.ascontiguousarray(fp[ks, :Nf])]
    y = []
    for u in us:
        a = np.fft.rfft(u.astype(rt) / padsize, axis=1).astype(ct)[:, :Nf].copy()
        c0 = a[:, 0].real - a[:, -1].imag
        cN = a[:, -1].real.copy()
        a[:, 0] = c0
        a[:, -1] = cN
        y.append(a)
    g = _line_gather_x(y, P)
    out = []
    for a in g:
        U = np.fft.fft(a / padsize, axis=0).astype(ct)
        fu = np.zeros((N0, a.shape[1]), dtype=ct)
        fu[:N0 // 2 + 1] = U[:N0 // 2 + 1]
        fu[N0 // 2:] += U[M0 - N0 // 2:]
        out.append(fu)
    return out


def line_dealias_mask(N, L, lay, r):
    """line.py:129-134 (kmax from N//2+1 per axis, scaled wavenumbers as get_local_wavenumbermesh() defaults)."""
    N = np.asarray(N)
    L = np.asarray(L, dtype=float)
    kx = np.fft.fftfreq(int(N[0]), 1. / N[0]) * (2 * np.pi / L[0])
    ky = np.fft.rfftfreq(int(N[1]), 1. / N[1])[lay.complex_slice(r)[1]] * (2 * np.pi / L[1])
    kmax = 2. / 3. * (N // 2 + 1)
    return np.array((abs(kx[:, None]) < kmax[0]) * (abs(ky[None, :]) < kmax[1]), dtype=np.uint8)


# --------------------------------------------------------------------------
# 2/3-rule mask (slab.py:191-197, maths.pyx:9-19)
# --------------------------------------------------------------------------

def dealias_mask(N, kx, ky, kz):
    N = np.asarray(N, dtype=int)
    kmax = 2.0 / 3.0 * (N // 2 + 1)
    K = np.meshgrid(kx, ky, kz, indexing="ij", sparse=True)
    return np.array((abs(K[0]) < kmax[0]) * (abs(K[1]) < kmax[1]) *
                    (abs(K[2]) < kmax[2]), dtype=np.uint8)


def apply_mask(fu, mask):
    return fu * mask


# --------------------------------------------------------------------------
# helpers for tests: scatter / gather global arrays
# --------------------------------------------------------------------------

def scatter_real(A, lay, padsize=1):
    return [np.ascontiguousarray(A[lay.real_local_slice(r, padsize)])
            for r in range(lay.P)]


def scatter_complex(C, lay):
    return [np.ascontiguousarray(C[lay.complex_local_slice(r)])
            for r in range(lay.P)]


def gather_complex(parts, lay, dtype):
    C = np.zeros(lay.global_complex_shape(), dtype=dtype)
    for r, p in enumerate(parts):
        C[lay.complex_local_slice(r)] = p
    return C


def gather_real(parts, lay, dtype, padsize=1):
    shp = tuple(int(padsize * n) for n in lay.N)
    A = np.zeros(shp, dtype=dtype)
    for r, p in enumerate(parts):
        A[lay.real_local_slice(r, padsize)] = p
    return A


def rel_l2(x, ref):
    x = np.asarray(x)
    ref = np.asarray(ref)
    d = np.linalg.norm((x - ref).ravel())
    n = np.linalg.norm(ref.ravel())
    return float(d / n) if n > 0 else float(d)
