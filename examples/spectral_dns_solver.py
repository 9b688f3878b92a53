"""Taylor-Green vortex with a pseudo-spectral Navier-Stokes solver (RK4) on top
of mpifft4py_amd -- the counterpart of the reference's
demo/spectral_dns_solver.py (same parameters, same update rule, same known
answer k = 0.124953117517 after 10 steps at 32^3).

    python examples/spectral_dns_solver.py                      # 1 GPU
    python examples/spectral_dns_solver.py --ranks 4            # 4 in-process ranks
    python -m torch.distributed.run --nproc-per-node 2 examples/spectral_dns_solver.py   # RCCL

Only the FFT class is swapped: everything between transforms is the reference
demo's host-side numpy arithmetic (the wavenumber mesh is an array here because
the upstream list-of-sparse-arrays form no longer multiplies under numpy 2).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mpifft4py_amd import work_arrays  # noqa: E402
from mpifft4py_amd.pencil import R2C as Pencil_R2C  # noqa: E402
from mpifft4py_amd.slab import R2C as Slab_R2C  # noqa: E402


def solve(comm, M=5, dealias='3/2-rule', decomposition='slab', precision="double", alignment="X",
          nu=0.000625, T=0.1, dt=0.01):
    N = np.array([2 ** M] * 3, dtype=int)
    L = np.array([2 * np.pi] * 3, dtype=float)
    if decomposition == 'slab':
        FFT = Slab_R2C(N, L, comm, precision)
    else:
        FFT = Pencil_R2C(N, L, comm, precision, communication="Alltoallw", alignment=alignment)
    float_, complex_ = FFT.float, FFT.complex

    U = np.empty((3,) + FFT.real_shape(), dtype=float_)
    U_hat = np.empty((3,) + FFT.complex_shape(), dtype=complex_)
    P_hat = np.empty(FFT.complex_shape(), dtype=complex_)
    U_hat0 = np.empty_like(U_hat)
    U_hat1 = np.empty_like(U_hat)
    dU = np.empty_like(U_hat)
    work = work_arrays()
    X = FFT.get_local_mesh()
    K = np.array(FFT.get_local_wavenumbermesh(scaled=True, broadcast=True), dtype=float_)
    K2 = np.sum(K * K, 0, dtype=float_)
    K_over_K2 = K.astype(float_) / np.where(K2 == 0, 1, K2).astype(float_)
    a = [1. / 6., 1. / 3., 1. / 3., 1. / 6.]
    b = [0.5, 0.5, 1.]

    def cross(x, y, z):
        z[0] = FFT.fftn(x[1] * y[2] - x[2] * y[1], z[0], dealias)
        z[1] = FFT.fftn(x[2] * y[0] - x[0] * y[2], z[1], dealias)
        z[2] = FFT.fftn(x[0] * y[1] - x[1] * y[0], z[2], dealias)
        return z

    def curl(x, z):
        z[2] = FFT.ifftn(1j * (K[0] * x[1] - K[1] * x[0]), z[2], dealias)
        z[1] = FFT.ifftn(1j * (K[2] * x[0] - K[0] * x[2]), z[1], dealias)
        z[0] = FFT.ifftn(1j * (K[1] * x[2] - K[2] * x[1]), z[0], dealias)
        return z

    def compute_rhs(rhs):
        U_dealiased = work[((3,) + FFT.work_shape(dealias), float_, 0)]
        curl_dealiased = work[((3,) + FFT.work_shape(dealias), float_, 1)]
        for i in range(3):
            U_dealiased[i] = FFT.ifftn(U_hat[i], U_dealiased[i], dealias)
        curl_dealiased = curl(U_hat, curl_dealiased)
        rhs = cross(U_dealiased, curl_dealiased, rhs)
        P_hat[:] = np.sum(rhs * K_over_K2, 0, out=P_hat)
        rhs -= P_hat * K
        rhs -= nu * K2 * U_hat
        return rhs

    U[0] = np.sin(X[0]) * np.cos(X[1]) * np.cos(X[2])
    U[1] = -np.cos(X[0]) * np.sin(X[1]) * np.cos(X[2])
    U[2] = 0
    for i in range(3):
        U_hat[i] = FFT.fftn(U[i], U_hat[i])

    t = 0.0
    while t < T - 1e-8:
        t += dt
        U_hat1[:] = U_hat0[:] = U_hat
        for rk in range(4):
            dU = compute_rhs(dU)
            if rk < 3:
                U_hat[:] = U_hat0 + b[rk] * dt * dU
            U_hat1[:] += a[rk] * dt * dU
        U_hat[:] = U_hat1[:]

    for i in range(3):
        U[i] = FFT.ifftn(U_hat[i], U[i])
    k = FFT.comm.reduce(float(np.sum(U.astype(np.float64) * U) / N[0] / N[1] / N[2] / 2))
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=1, help="in-process virtual ranks (ignored under torchrun/mpirun)")
    ap.add_argument("--dealias", default="3/2-rule", choices=["3/2-rule", "2/3-rule", "None"])
    ap.add_argument("--decomposition", default="slab", choices=["slab", "pencil"])
    args = ap.parse_args()
    dealias = None if args.dealias == "None" else args.dealias
    from mpifft4py_amd import LocalGroup, SelfComm, from_env
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        comm = from_env()
        k = solve(comm, dealias=dealias, decomposition=args.decomposition)
        ks = [k]
    elif args.ranks > 1:
        g = LocalGroup(args.ranks)
        ks = g.run(lambda c: solve(c, dealias=dealias, decomposition=args.decomposition))
    else:
        ks = [solve(SelfComm(), dealias=dealias, decomposition=args.decomposition)]
    if ks[0] is not None:
        print("k =", repr(ks[0]))
        assert round(ks[0] - 0.124953117517, 7) == 0
        print("matches the reference demo's known answer 0.124953117517")


if __name__ == "__main__":
    main()
