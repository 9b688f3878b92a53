"""Randomised sweep of the serialFFT function set (developer tool): random 1-D / 2-D / 3-D shapes with radix and
chirp-z lengths, every function, both precisions, against numpy.fft.  python scripts/fuzz_stages.py [ncases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mpifft4py_amd as m

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
LENS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 15, 16, 18, 20, 21, 24, 25, 27, 30, 32, 36, 40, 48, 49, 50, 60, 64, 72, 96, 100,
        121, 128, 144, 160, 200, 243, 256, 300]
def rel(x, r): return float(np.linalg.norm((x - r).ravel()) / max(np.linalg.norm(r.ravel()), 1e-300))
fails = 0
for case in range(ncases):
    prec = str(rng.choice(["double", "single"]))
    rt, ct = (np.float64, np.complex128) if prec == "double" else (np.float32, np.complex64)
    tol = 1e-10 if prec == "double" else 1e-4
    fn = str(rng.choice(["fft", "ifft", "fft2", "ifft2", "fftn", "ifftn", "rfft", "irfft", "rfft2", "irfft2", "rfftn", "irfftn"]))
    nd = {"fft": (1, 3), "ifft": (1, 3), "rfft": (1, 3), "irfft": (1, 3), "fft2": (2, 3), "ifft2": (2, 3), "rfft2": (2, 3),
          "irfft2": (2, 3), "fftn": (3, 3), "ifftn": (3, 3), "rfftn": (3, 3), "irfftn": (3, 3)}[fn]
    ndim = int(rng.integers(nd[0], nd[1] + 1))
    shape = [int(rng.choice(LENS)) for _ in range(ndim)]
    tag = "%s %s %s" % (fn, shape, prec)
    try:
        if fn in ("fft", "ifft"):
            ax = int(rng.integers(0, ndim))
            a = (rng.random(shape) + 1j * rng.random(shape)).astype(ct)
            got = getattr(m, fn)(a, axis=ax); ref = getattr(np.fft, fn)(a.astype(np.complex128), axis=ax)
            tag += " axis=%d" % ax
        elif fn in ("fft2", "ifft2"):
            axes = (0, 1) if ndim == 2 else tuple(sorted(rng.choice(3, 2, replace=False).tolist()))
            a = (rng.random(shape) + 1j * rng.random(shape)).astype(ct)
            got = getattr(m, fn)(a, axes=axes); ref = getattr(np.fft, fn)(a.astype(np.complex128), axes=axes)
            tag += " axes=%s" % (axes,)
        elif fn in ("fftn", "ifftn"):
            a = (rng.random(shape) + 1j * rng.random(shape)).astype(ct)
            got = getattr(m, fn)(a, axes=(0, 1, 2)); ref = getattr(np.fft, fn)(a.astype(np.complex128), axes=(0, 1, 2))
        else:
            real_axes = {"rfft": (ndim - 1,), "irfft": (ndim - 1,), "rfft2": (ndim - 2, ndim - 1), "irfft2": (ndim - 2, ndim - 1),
                         "rfftn": (0, 1, 2), "irfftn": (0, 1, 2)}[fn]
            if shape[-1] < 2:
                shape[-1] = 2
            a = rng.random(shape).astype(rt)
            if fn.startswith("r"):
                kw = {"axis": real_axes[0]} if fn == "rfft" else {"axes": real_axes}
                got = getattr(m, fn)(a, **kw); ref = getattr(np.fft, fn)(a.astype(np.float64), **kw)
            else:
                npf = {"irfft": np.fft.rfft, "irfft2": np.fft.rfft2, "irfftn": np.fft.rfftn}[fn]
                kw = {"axis": real_axes[0]} if fn == "irfft" else {"axes": real_axes}
                c = npf(a.astype(np.float64), **kw).astype(ct)
                out = np.zeros(shape, dtype=rt)
                got = getattr(m, fn)(c, out, **kw)
                ikw = dict(kw); ikw["n" if fn == "irfft" else "s"] = shape[-1] if fn == "irfft" else [shape[x] for x in real_axes]
                ref = getattr(np.fft, fn)(c.astype(np.complex128), **ikw)
        e = rel(np.asarray(got), ref)
        ok = e < tol and got.shape == ref.shape
        if not ok:
            fails += 1
        print("%-60s %.2e %s" % (tag, e, "ok" if ok else "FAIL"))
    except Exception as ex:      # noqa: BLE001
        fails += 1
        print("%-60s EXCEPTION %s: %s" % (tag, type(ex).__name__, str(ex)[:200]))
print("stage fuzz: %d cases, %d failures" % (ncases, fails))
sys.exit(1 if fails else 0)
