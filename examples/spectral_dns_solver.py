"""Taylor-Green vortex with a pseudo-spectral Navier-Stokes solver (RK4) on top
of mpifft4py_amd -- the counterpart of the reference's
demo/spectral_dns_solver.py (same parameters, same update rule, same known
answer k = 0.124953117517 after 10 steps at 32^3).

    python examples/spectral_dns_solver.py                      # 1 GPU
    python examples/spectral_dns_solver.py --ranks 4            # 4 in-process ranks
    python -m torch.distributed.run --nproc-per-node 2 examples/spectral_dns_solver.py   # RCCL

Only the transforms run on the GPU here: everything between them is host-side numpy
arithmetic, as in the reference demo (examples/spectral_dns_device.py keeps the whole
state in HBM instead).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mpifft4py_amd import work_arrays  # noqa: E402
from mpifft4py_amd.pencil import R2C as Pencil_R2C  # noqa: E402
from mpifft4py_amd.slab import R2C as Slab_R2C  # noqa: E402


class TaylorGreen(object):
    """Rotational-form Navier-Stokes in a periodic box: dU^/dt = P[(u x w)^] - nu k^2 U^, with w = curl u and
    P the projection onto divergence-free fields.  Transforms are `fft.fftn / ifftn`; the rest is numpy."""

    RK4_WEIGHTS = (1. / 6., 1. / 3., 1. / 3., 1. / 6.)
    RK4_NODES = (0.5, 0.5, 1.)

    def __init__(self, fft, viscosity, dealias):
        self.fft, self.nu, self.dealias = fft, viscosity, dealias
        rt, ct = fft.float, fft.complex
        self.kvec = np.array(fft.get_local_wavenumbermesh(scaled=True, broadcast=True), dtype=rt)
        self.ksq = np.sum(self.kvec * self.kvec, 0, dtype=rt)
        self.k_over_ksq = self.kvec / np.where(self.ksq == 0, 1, self.ksq).astype(rt)
        self.u = np.empty((3,) + fft.real_shape(), dtype=rt)
        self.u_hat = np.empty((3,) + fft.complex_shape(), dtype=ct)
        self.pressure = np.empty(fft.complex_shape(), dtype=ct)
        self.pool = work_arrays()

    def set_taylor_green(self):
        x, y, z = self.fft.get_local_mesh()
        self.u[0] = np.sin(x) * np.cos(y) * np.cos(z)
        self.u[1] = -np.cos(x) * np.sin(y) * np.cos(z)
        self.u[2] = 0
        for c in range(3):
            self.u_hat[c] = self.fft.fftn(self.u[c], self.u_hat[c])

    def tendency(self, out):
        fft, d, k, uh = self.fft, self.dealias, self.kvec, self.u_hat
        vel = self.pool[((3,) + fft.work_shape(d), fft.float, 0)]
        vort = self.pool[((3,) + fft.work_shape(d), fft.float, 1)]
        for c in range(3):
            vel[c] = fft.ifftn(uh[c], vel[c], d)
        for c in range(3):                       # vorticity: i k x U^, component by component
            p, q = (c + 1) % 3, (c + 2) % 3
            vort[c] = fft.ifftn(1j * (k[p] * uh[q] - k[q] * uh[p]), vort[c], d)
        for c in range(3):                       # (u x w)^
            p, q = (c + 1) % 3, (c + 2) % 3
            out[c] = fft.fftn(vel[p] * vort[q] - vel[q] * vort[p], out[c], d)
        self.pressure[:] = np.sum(out * self.k_over_ksq, 0, out=self.pressure)
        out -= self.pressure * k
        out -= self.nu * self.ksq * uh
        return out

    def advance(self, dt, nsteps):
        start = np.empty_like(self.u_hat)
        accum = np.empty_like(self.u_hat)
        slope = np.empty_like(self.u_hat)
        for _ in range(nsteps):
            start[:] = self.u_hat
            accum[:] = self.u_hat
            for stage, weight in enumerate(self.RK4_WEIGHTS):
                slope = self.tendency(slope)
                if stage < 3:
                    self.u_hat[:] = start + self.RK4_NODES[stage] * dt * slope
                accum += weight * dt * slope
            self.u_hat[:] = accum

    def kinetic_energy(self):
        for c in range(3):
            self.u[c] = self.fft.ifftn(self.u_hat[c], self.u[c])
        n = self.fft.N
        local = float(np.sum(self.u.astype(np.float64) * self.u) / n[0] / n[1] / n[2] / 2)
        return self.fft.comm.reduce(local)


def solve(comm, M=5, dealias='3/2-rule', decomposition='slab', precision="double", alignment="X",
          nu=0.000625, T=0.1, dt=0.01):
    N = np.array([2 ** M] * 3, dtype=int)
    L = np.array([2 * np.pi] * 3, dtype=float)
    if decomposition == 'slab':
        fft = Slab_R2C(N, L, comm, precision)
    else:
        fft = Pencil_R2C(N, L, comm, precision, communication="Alltoallw", alignment=alignment)
    flow = TaylorGreen(fft, nu, dealias)
    flow.set_taylor_green()
    flow.advance(dt, int(round(T / dt)))
    return flow.kinetic_energy()


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
