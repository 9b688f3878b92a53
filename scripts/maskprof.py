"""Cost of the 2/3-rule on the inverse transform (developer tool): python scripts/maskprof.py n [precision] [slab|pitched|X|Y]
Times ifftn(fu, u) and ifftn(fu, u, dealias="2/3-rule") on one GPU and prints the stage times of the latter
(X / Y: the pencil class on a 1 x 1 grid; MFFT_NO_PRUNE=1 gives the mask-byte path for comparison)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Pencil_R2C, Slab_R2C, SelfComm, DeviceArray
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prec = sys.argv[2] if len(sys.argv) > 2 else "double"
N = np.array([n] * 3); L = np.array([2 * np.pi] * 3)
kind = sys.argv[3] if len(sys.argv) > 3 else "slab"
if kind == "pitched":          # slab, spectrum rows a whole number of cache lines apart (complex_pitch="auto")
    F = Slab_R2C(N, L, SelfComm(0), prec, complex_pitch="auto")
elif kind == "slab":
    F = Slab_R2C(N, L, SelfComm(0), prec)
else:
    F = Pencil_R2C(N, L, SelfComm(0), prec, communication="Alltoallw", alignment=kind, allow_single=True)
fu = F.empty_complex() if kind == "pitched" else DeviceArray.random(F.complex_shape(), F.complex, seed=1)
u = DeviceArray.empty(F.real_shape(), F.float)
def run(dealias, reps=8):
    for _ in range(2):
        F.ifftn(fu, u, dealias)
    F.sync()
    t = time.perf_counter()
    for _ in range(reps):
        F.ifftn(fu, u, dealias)
    F.sync()
    return (time.perf_counter() - t) / reps * 1e3
a = run(None); b = run("2/3-rule")
F.enable_timing(True)
run("2/3-rule", 4)
print("n=%d %s %s ifftn %.3f ms, with the 2/3-rule %.3f ms (%+.3f)%s" % (n, prec, kind, a, b, b - a, "  [MFFT_NO_PRUNE]" if os.environ.get("MFFT_NO_PRUNE") else ""))
print(" ".join("%s=%.3f" % (k, v[0] / max(v[1], 1)) for k, v in sorted(F.stage_times().items())))
