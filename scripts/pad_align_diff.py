"""Where do the one-rank 3/2-rule results of the compact and the line-aligned route differ? (developer tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpifft4py_amd import Slab_R2C, SelfComm
N = [int(x) for x in sys.argv[1:4]] if len(sys.argv) > 3 else [32, 64, 128]
prec = sys.argv[4] if len(sys.argv) > 4 else "single"
ct = np.complex64 if prec == "single" else np.complex128
rt = np.float32 if prec == "single" else np.float64
L = np.array([2 * np.pi] * 3)
A = np.random.default_rng(sum(N)).random(N)
C0 = np.fft.rfftn(A).astype(ct)
got = {}
for mode in ("1", "0", "1b", "0b"):
    os.environ["MFFT_PAD_ALIGN"] = mode[0]
    F = Slab_R2C(np.array(N), L, SelfComm(0), prec)
    up = F.ifftn(C0.copy(), np.zeros(F.real_shape_padded(), dtype=rt), "3/2-rule")
    fu = F.fftn(up.copy(), np.zeros(F.complex_shape(), dtype=ct), "3/2-rule")
    got[mode] = (up.copy(), fu.copy())
for pair in (("1", "0"), ("1", "1b"), ("0", "0b")):
  for i, name in enumerate(("up", "fu")):
    print(pair, end=" ")
    a, b = got[pair[0]][i], got[pair[1]][i]
    d = np.abs(a - b)
    print(name, "max abs diff", d.max(), "of max", np.abs(b).max(), "count", int((d > 0).sum()), "of", d.size)
    if d.max() > 0:
        idx = np.argwhere(d > 0)
        print("  axis-wise ranges of differing indices:", [(int(idx[:, k].min()), int(idx[:, k].max()), len(set(idx[:, k].tolist()))) for k in range(3)])
