"""Randomised parity sweep (developer tool): random meshes / rank counts / decompositions / precisions / dealias
modes against the oracle.  python scripts/fuzz_parity.py [ncases] [seed]"""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import L, TOL, cdtype, rdtype, orc, run_ranks
from mpifft4py_amd import Line_R2C, Pencil_C2C, Pencil_R2C, Slab_R2C
from mpifft4py_amd.slab import C2C as Slab_C2C

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
NICE = [4, 6, 8, 10, 12, 16, 18, 20, 24, 32, 36, 40, 48, 50, 64, 72, 80, 96, 100, 128, 144, 160, 192, 200, 256]
ODD = [14, 22, 26, 28, 30, 34, 42, 44, 52, 54, 56, 60, 66, 70, 84, 90, 98, 110, 126, 130, 150, 170, 210, 250,
       120, 180, 240, 300,      # round 3: more of the lengths with 3 and 5 among their factors (plans.h group L)
       108, 216, 288, 432]      # round 6: 27 * 2^a (plans.h group T) and 9 * 2^a meshes whose 3/2-rule images they are


def pick(P, need_div, even_quot=False):
    for _ in range(200):
        n = int(rng.choice(NICE if rng.random() < 0.6 else ODD))
        if n % need_div == 0 and n >= 2 * P and (not even_quot or (n // need_div) % 2 == 0):
            return n
    return 64 * need_div


def line_case(prec, dealias):
    """2-D class: forward / inverse (plain, 2/3-rule) and the 3/2-rule pair against the oracle's restatement of line.py."""
    P = int(rng.choice([1, 2, 4]))
    N = [pick(P, P), pick(P, 2 * P, True)]
    if dealias == "3/2-rule":
        N = [max(n - n % (4 * P), 4 * P) for n in N]
    rt, ct = rdtype(prec), cdtype(prec)
    L2 = np.array([2 * np.pi, 3 * np.pi])
    lay = orc.LineLayout(N, P)
    A = rng.random(N).astype(rt)
    us = [np.ascontiguousarray(A[lay.real_slice(r)]) for r in range(P)]
    want_c = orc.line_r2c_forward(us, N, prec)
    if dealias == "3/2-rule":
        want_b = orc.line_r2c_backward_padded(want_c, N, prec)
        want_c2 = orc.line_r2c_forward_padded(want_b, N, prec)
    elif dealias == "2/3-rule":
        want_b = orc.line_r2c_backward([c * orc.line_dealias_mask(N, L2, lay, r) for r, c in enumerate(want_c)], N, prec)
    else:
        want_b = orc.line_r2c_backward(want_c, N, prec)

    def body(comm):
        F = Line_R2C(np.array(N), L2, comm, prec)
        r = comm.Get_rank()
        c = F.fft2(us[r], np.zeros(want_c[r].shape, dtype=ct))
        b = F.ifft2(c, np.zeros(want_b[r].shape, dtype=rt), dealias)
        c2 = F.fft2(b, np.zeros(want_c[r].shape, dtype=ct), "3/2-rule") if dealias == "3/2-rule" else None
        return c, b, c2
    worst = 0.0
    for r, (c, b, c2) in enumerate(run_ranks(P, body)):
        worst = max(worst, orc.rel_l2(c, want_c[r]), orc.rel_l2(b, want_b[r]))
        if c2 is not None:
            worst = max(worst, orc.rel_l2(c2, want_c2[r]))
    return "line N=%s P=%d %s dealias=%s" % (N, P, prec, dealias), worst


def pencil_n_case(prec):
    """communication='AlltoallN' (pencil.py:410-432): the z-Nyquist column is neither exchanged nor returned."""
    P = int(rng.choice([4, 8]))
    P1P2 = 4 if P == 4 else 8
    N = [pick(P, P1P2), pick(P, P1P2), pick(P, 2 * P1P2, True)]
    al = str(rng.choice(["X", "Y"]))
    rt, ct = rdtype(prec), cdtype(prec)
    A = rng.random(N).astype(rt)
    lay = orc.PencilNLayout(N, P, None, al)
    us = orc.scatter_real(A, lay)
    want_c = orc.pencil_r2c_forward_n(us, N, None, al, prec)
    want_b = orc.pencil_r2c_backward_n(want_c, N, None, al, prec)

    def body(comm):
        F = Pencil_R2C(np.array(N), L, comm, prec, communication="AlltoallN", alignment=al)
        r = comm.Get_rank()
        c = F.fftn(np.ascontiguousarray(us[r]), np.zeros(want_c[r].shape, dtype=ct))
        b = F.ifftn(c, np.zeros(want_b[r].shape, dtype=rt))
        return c, b
    worst = 0.0
    for r, (c, b) in enumerate(run_ranks(P, body)):
        worst = max(worst, orc.rel_l2(c, want_c[r]), orc.rel_l2(b, want_b[r]))
    return "pencilN%s N=%s P=%d %s" % (al, N, P, prec), worst


def pencil_c2c_case(prec):
    P = int(rng.choice([4, 8]))
    P1P2 = 4 if P == 4 else 8
    N = [pick(P, P1P2), pick(P, P1P2), pick(P, P1P2)]
    al = str(rng.choice(["X", "Y"]))
    ct = cdtype(prec)
    A = (rng.random(N) + 1j * rng.random(N)).astype(ct)
    B = np.fft.fftn(A.astype(np.complex128))

    def body(comm):
        F = Pencil_C2C(np.array(N), L, comm, prec, alignment=al)
        a = np.ascontiguousarray(A[F.original_local_slice()])
        c = F.fftn(a, np.zeros(F.transformed_shape(), dtype=ct))
        b = F.ifftn(c, np.zeros(F.original_shape(), dtype=ct))
        return orc.rel_l2(c, B[F.transformed_local_slice()]), orc.rel_l2(b, a)
    worst = max(max(r) for r in run_ranks(P, body))
    return "pencilc2c%s N=%s P=%d %s" % (al, N, P, prec), worst


fails = 0
t0 = time.time()
for case in range(ncases):
    kind = rng.choice(["slab", "slab", "pencilX", "pencilY", "slabc2c", "line", "pencilc2c", "pencilN"])
    if kind in ("line", "pencilc2c", "pencilN"):
        prec = str(rng.choice(["double", "single"]))
        dealias = rng.choice([None, "3/2-rule", "2/3-rule"])
        try:
            tag, worst = (line_case(prec, dealias) if kind == "line" else
                          pencil_c2c_case(prec) if kind == "pencilc2c" else pencil_n_case(prec))
            ok = worst < 4 * TOL[prec]
            print("%-70s %.2e %s" % (tag, worst, "ok" if ok else "FAIL"))
            fails += 0 if ok else 1
        except Exception as e:      # noqa: BLE001
            fails += 1
            print("%-70s EXCEPTION %s: %s" % (kind, type(e).__name__, str(e)[:300]))
            traceback.print_exc(limit=3)
        continue
    prec = str(rng.choice(["double", "single"]))
    dealias = rng.choice([None, None, "3/2-rule", "2/3-rule"])
    if kind == "slab":
        P = int(rng.choice([1, 1, 2, 4, 8]))
        N = [pick(P, P), pick(P, P), pick(1, 2)]
    elif kind == "slabc2c":
        P = int(rng.choice([1, 2, 4]))
        N = [pick(P, P), pick(P, P), pick(1, 1)]
        if dealias == "2/3-rule":
            dealias = None
    else:
        P = int(rng.choice([4, 8]))
        P1, P2 = (2, 2) if P == 4 else (4, 2)
        N = [pick(P, P1 * P2), pick(P, P1 * P2), pick(P, 2 * P1 * P2, True)]
    if dealias == "3/2-rule":       # the reference needs padded extents divisible as well: use multiples of 4P
        N = [n - n % (4 * P) or 4 * P for n in N]
        if kind.startswith("pencil"):
            N = [max(n - n % 16, 16) for n in N]
    rt, ct = rdtype(prec), cdtype(prec)
    tag = "%s N=%s P=%d %s dealias=%s" % (kind, N, P, prec, dealias)
    try:
        if kind == "slabc2c":
            A = (rng.random(N) + 1j * rng.random(N)).astype(ct)
            lay = orc.SlabLayout(N, P)
        else:
            A = rng.random(N).astype(rt)
        if kind == "slab":
            lay = orc.SlabLayout(N, P)
            depth = int(rng.choice([0, 1, 2, 4, 8, -2, -4, -5]))   # exchange pipeline (0 default, <0 row batches)
            tag += " pipeline=%d" % depth
            pitch = "auto" if rng.random() < 0.4 else None      # round 6: spectrum rows a whole number of cache lines apart
            tag += " pitch=%s" % pitch
            make = lambda comm: Slab_R2C(np.array(N), L, comm, prec, pipeline=depth, complex_pitch=pitch)
            fwd, bwd = orc.slab_r2c_forward, orc.slab_r2c_backward
            fwdp, bwdp = orc.slab_r2c_forward_padded, orc.slab_r2c_backward_padded
            extra = ()
        elif kind == "slabc2c":
            make = lambda comm: Slab_C2C(np.array(N), L, comm, prec)
            fwd, bwd = orc.slab_c2c_forward, orc.slab_c2c_backward
            fwdp, bwdp = orc.slab_c2c_forward_padded, orc.slab_c2c_backward_padded
            extra = ()
        else:
            al = kind[-1]
            lay = orc.PencilLayout(N, P, None, al)
            pdepth = int(rng.choice([0, 0, 2, 4, 7]))         # X alignment: opt-in exchange pipeline
            tag += " pipeline=%d" % pdepth
            make = lambda comm: Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=al,
                                           pipeline=pdepth)
            fwd = lambda us, N_, p: orc.pencil_r2c_forward(us, N_, None, al, p)
            bwd = lambda fs, N_, p: orc.pencil_r2c_backward(fs, N_, None, al, p)
            fwdp = lambda us, N_, p: orc.pencil_r2c_forward_padded(us, N_, None, al, p)
            bwdp = lambda fs, N_, p: orc.pencil_r2c_backward_padded(fs, N_, None, al, p)
        us = orc.scatter_real(A, lay)
        want_c = fwd(us, N, prec)
        if dealias == "3/2-rule":
            want_b = bwdp(want_c, N, prec)
            want_c2 = fwdp(want_b, N, prec)
        else:
            want_b = bwd(want_c, N, prec)

        def body(comm):
            F = make(comm)
            r = comm.Get_rank()
            c = F.fftn(np.ascontiguousarray(us[r]), np.zeros(want_c[r].shape, dtype=ct))
            if dealias == "3/2-rule":
                b = F.ifftn(c, np.zeros(want_b[r].shape, dtype=want_b[r].dtype), "3/2-rule")
                c2 = F.fftn(b, np.zeros(want_c[r].shape, dtype=ct), "3/2-rule")
                return c, b, c2
            if dealias == "2/3-rule":
                b = F.ifftn(c, np.zeros(want_b[r].shape, dtype=want_b[r].dtype), "2/3-rule")
                m = F.get_dealias_filter()
                return c, b, m
            b = F.ifftn(c, np.zeros(want_b[r].shape, dtype=want_b[r].dtype))
            return c, b, None
        res = run_ranks(P, body)
        if dealias == "2/3-rule":
            want_b = bwd([want_c[r] * res[r][2] for r in range(P)], N, prec)
        worst = 0.0
        for r, (c, b, x) in enumerate(res):
            worst = max(worst, orc.rel_l2(c, want_c[r]), orc.rel_l2(b, want_b[r]))
            if dealias == "3/2-rule":
                worst = max(worst, orc.rel_l2(x, want_c2[r]))
        ok = worst < 4 * TOL[prec]
        print("%-70s %.2e %s" % (tag, worst, "ok" if ok else "FAIL"))
        fails += 0 if ok else 1
    except Exception as e:      # noqa: BLE001
        fails += 1
        print("%-70s EXCEPTION %s: %s" % (tag, type(e).__name__, str(e)[:300]))
        traceback.print_exc(limit=3)
print("fuzz: %d cases, %d failures, %.0f s" % (ncases, fails, time.time() - t0))
sys.exit(1 if fails else 0)
