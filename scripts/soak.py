"""Stability / leak check (developer tool): many plan create/destroy cycles and many transforms, watching free HBM."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import DeviceArray, LocalGroup, Pencil_R2C, SelfComm, Slab_R2C, _lib
hip = ctypes.CDLL("libamdhip64.so")

def free_gb():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 1e9

L = np.array([2 * np.pi] * 3)
comm = SelfComm(0)
f0 = free_gb()
for i in range(200):
    n = [64, 96, 100, 128, 36][i % 5]
    F = Slab_R2C(np.array([n] * 3), L, comm, "double" if i % 2 else "single")
    u = DeviceArray.random(F.real_shape(), F.float, seed=i)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    F.fftn(u, fu); F.ifftn(fu, u)
    if i % 3 == 0:
        up = DeviceArray.empty(F.real_shape_padded(), F.float)
        F.ifftn(fu, up, "3/2-rule"); F.fftn(up, fu, "3/2-rule")
        del up
    F.sync()
    del F, u, fu
f1 = free_gb()
print("200 plan cycles: free HBM %.3f -> %.3f GB (delta %.3f GB; twiddle/chirp caches are kept by design)" % (f0, f1, f0 - f1))
for rep in range(3):
    g = LocalGroup(4, devices=[0] * 4)
    def body(c):
        F = Pencil_R2C(np.array([64, 64, 64]), L, c, "double", communication="Alltoallw", alignment="X")
        u = DeviceArray.random(F.real_shape(), F.float, seed=1)
        fu = DeviceArray.empty(F.complex_shape(), F.complex)
        for _ in range(20):
            F.fftn(u, fu); F.ifftn(fu, u)
        F.sync()
        return 0
    g.run(body); g.free()
f2 = free_gb()
print("3 x 4-rank groups: free HBM %.3f GB (delta %.3f GB)" % (f2, f1 - f2))
F = Slab_R2C(np.array([1024] * 3), L, comm, "double")
u = DeviceArray.random(F.real_shape(), F.float, seed=5)
fu = DeviceArray.empty(F.complex_shape(), F.complex)
u2 = DeviceArray.empty(F.real_shape(), F.float)
F.fftn(u, fu); F.ifftn(fu, u2); F.sync()
ref = u2.leading(0, 2).get().copy()
f3 = free_gb()
t = time.perf_counter()
for i in range(400):
    F.fftn(u, fu); F.ifftn(fu, u2)
F.sync()
dt = time.perf_counter() - t
same = np.array_equal(u2.leading(0, 2).get(), ref)
print("400 pairs at 1024^3: %.2f ms/pair, results bit-identical to the first pair: %s, free HBM delta %.3f GB"
      % (dt / 400 * 1e3, same, f3 - free_gb()))
print("SOAK_OK" if same and (f1 - f2) < 0.05 and abs(f3 - free_gb()) < 0.05 else "SOAK_CHECK")
