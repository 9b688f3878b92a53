"""Small meshes: where does a transform's time go -- Python wrapper, C launch path, GPU?  (developer tool, round 4)
python scripts/small_mesh_overhead.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray, _lib

L = np.array([2 * np.pi] * 3)
for n in (32, 48, 64, 96, 128, 256):
    F = Slab_R2C(np.array([n] * 3), L, SelfComm(0), "double")
    u = DeviceArray.random(F.real_shape(), F.float, seed=1)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    u2 = DeviceArray.empty(F.real_shape(), F.float)
    reps = 2000 if n <= 128 else 300

    def loop(fn):
        for _ in range(20):
            fn()
        F.sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        th = time.perf_counter() - t          # host time to enqueue
        F.sync()
        return th / reps * 1e6, (time.perf_counter() - t) / reps * 1e6

    def api():
        F.fftn(u, fu); F.ifftn(fu, u2)

    def raw():
        _lib.call("mfft_forward", F._plan, u.ptr, fu.ptr, 0); _lib.call("mfft_backward", F._plan, fu.ptr, u2.ptr, 0)

    def api23():
        F.fftn(u, fu); F.ifftn(fu, u2, "2/3-rule")

    a = loop(api); r = loop(raw); d = loop(api23)
    F.enable_timing(True)
    for _ in range(3):
        api()
    F.sync(); F.reset_timing()
    for _ in range(50):
        api()
    F.sync()
    st = F.stage_times()
    gpu = sum(v[0] / max(v[1], 1) for v in st.values()) * 1e3
    F.enable_timing(False)
    print("%4d^3 pair: python API enqueue %.1f us, done %.1f us | raw C ABI enqueue %.1f, done %.1f | with 2/3-rule %.1f, %.1f | sum of kernel times %.1f us"
          % (n, a[0], a[1], r[0], r[1], d[0], d[1], gpu), flush=True)
