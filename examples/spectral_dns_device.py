"""Taylor-Green vortex, pseudo-spectral RK4 -- DEVICE-RESIDENT version: velocity,
spectra and all RK4 work arrays live in HBM, the transforms are mpifft4py_amd
plans and everything between them is a fused element-wise HIP kernel
(mpifft4py_amd.spectral).  Same equations, parameters and known answer as the
reference's demo/spectral_dns_solver.py (k = 0.124953117517 at 32^3, 10 steps).

    python examples/spectral_dns_device.py --M 5            # the reference demo's case
    python examples/spectral_dns_device.py --M 8 --steps 5  # 256^3, prints ms per RK4 step
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mpifft4py_amd import DeviceArray, spectral  # noqa: E402
from mpifft4py_amd.pencil import R2C as Pencil_R2C  # noqa: E402
from mpifft4py_amd.slab import R2C as Slab_R2C  # noqa: E402


def _zero(a):
    from mpifft4py_amd import _lib
    _lib.call("mfft_memset", a.ptr, 0, a.nbytes)
    return a


def solve(comm, M=5, dealias='3/2-rule', decomposition='slab', precision="double", nu=0.000625, dt=0.01, steps=10,
          report=None, fused=True, timing=False, complex_pitch="default", edge=None):
    """fused=True (round 6): the nonlinear term is ONE plan operation (spectral.cross_transform: no real-space work
    arrays, the z stages one kernel) and a Runge-Kutta stage's projection, viscous term, both updates and the next
    curl are ONE sweep (spectral.ns_rk_stage).  fused=False: the composition of rounds 3 - 5 (nine transforms, cross,
    curl, rhs and axpbz kernels per stage), kept for A/B timing and as the parity partner of the fused path."""
    if complex_pitch == "default":       # the fused loop on ONE rank keeps its spectra pitched (rows a whole number of cache lines
        # apart: every pass runs on them); several ranks and the composition of rounds 3 - 5 keep compact rows
        complex_pitch = "auto" if (fused and comm.Get_size() == 1) else None
    N = np.array([edge or 2 ** M] * 3, dtype=int)        # edge: a mesh that is not a power of two (576, 1152 ...)
    L = np.array([2 * np.pi] * 3, dtype=float)
    if decomposition == 'slab':
        FFT = Slab_R2C(N, L, comm, precision, complex_pitch=complex_pitch)
    else:
        FFT = Pencil_R2C(N, L, comm, precision, communication="Alltoallw", alignment="X", complex_pitch=complex_pitch)
    rs, cs, ws = FFT.real_shape(), FFT.complex_shape(), FFT.work_shape(dealias)
    fl, cx = FFT.float, FFT.complex
    K = spectral.Wavenumbers(FFT)

    # Taylor-Green initial condition (demo:81-84), built on the host in chunks of x planes (a 1024^3 field is 26 GB: the
    # chunks keep the host side at a few hundred MB) from the 1-D coordinates of this rank's block; everything else on the device
    sl = FFT.real_local_slice()
    x, y, z = (np.arange(s_.start, s_.stop, dtype=float) * (L[i] / N[i]) for i, s_ in enumerate(sl))
    U = DeviceArray.empty((3,) + rs, fl)
    cyz = np.cos(y)[:, None] * np.cos(z)[None, :]
    syz = np.sin(y)[:, None] * np.cos(z)[None, :]
    step = max(1, (64 << 20) // (8 * rs[1] * rs[2]))
    for i0 in range(0, rs[0], step):
        i1 = min(rs[0], i0 + step)
        U.component(0).leading(i0, i1).set((np.sin(x[i0:i1])[:, None, None] * cyz[None]).astype(fl))
        U.component(1).leading(i0, i1).set((-np.cos(x[i0:i1])[:, None, None] * syz[None]).astype(fl))
    _zero(U.component(2))
    U_hat, U_hat0, U_hat1, dU = (FFT.empty_complex(3) for _ in range(4))      # (3,) + cs, rows `complex_pitch` apart if asked for
    a = [1. / 6., 1. / 3., 1. / 3., 1. / 6.]
    b = [0.5, 0.5, 1.]
    if timing:
        FFT.enable_timing(True)

    if fused:
        for i in range(3):
            FFT.fftn(U.component(i), U_hat.component(i))
        spectral.axpbz(FFT, U_hat0, U_hat, U_hat, 1.0, 0.0)
        spectral.axpbz(FFT, U_hat1, U_hat, U_hat, 1.0, 0.0)
        spectral.curl_hat(FFT, K, U_hat, dU)               # dU doubles as the curl's spectrum between the stages
        # warm-up outside the timed loop: the plan allocates its buffers (25 GB at 512^3) at the first call; the state is restored
        spectral.cross_transform(FFT, U_hat, dU, dU, dealias)
        spectral.curl_hat(FFT, K, U_hat, dU)
        FFT.sync()
        if timing:
            FFT.reset_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            for rk in range(4):
                spectral.cross_transform(FFT, U_hat, dU, dU, dealias)          # dU = fftn(U x curl U)
                spectral.ns_rk_stage(FFT, K, dU, U_hat, U_hat0, U_hat1, nu, a[rk] * dt, b[rk] * dt if rk < 3 else 0.0, rk == 3)
        FFT.sync()
        wall = time.perf_counter() - t0
        if report is not None:
            report["ms_per_step"] = 1e3 * wall / steps
            report["fused_nonlinear"] = FFT.plan_info({"3/2-rule": "nonlinear_fused_3_2", "2/3-rule": "nonlinear_fused_2_3"}.get(
                dealias, "nonlinear_fused_none"))
            report["work_bytes"] = FFT.plan_info("nonlinear_bytes") + FFT.workspace_bytes()
            if timing:
                report["stages"] = {k: (v[0] / steps, v[1] // steps) for k, v in FFT.stage_times().items()}
        for i in range(3):
            FFT.ifftn(U_hat.component(i), U.component(i))
        return FFT.comm.reduce(spectral.sumsq(FFT, U) / float(N[0]) / float(N[1]) / float(N[2]) / 2)

    W_hat = FFT.empty_complex(3)
    Ud = DeviceArray.empty((3,) + ws, fl)
    Cd = DeviceArray.empty((3,) + ws, fl)
    Rd = DeviceArray.empty((3,) + ws, fl)

    def compute_rhs():
        for i in range(3):
            FFT.ifftn(U_hat.component(i), Ud.component(i), dealias)
        spectral.curl_hat(FFT, K, U_hat, W_hat)
        for i in range(3):
            FFT.ifftn(W_hat.component(i), Cd.component(i), dealias)
        spectral.cross(FFT, Ud, Cd, Rd)
        for i in range(3):
            FFT.fftn(Rd.component(i), dU.component(i), dealias)
        spectral.ns_rhs(FFT, K, dU, U_hat, nu)

    for i in range(3):
        FFT.fftn(U.component(i), U_hat.component(i))
    compute_rhs()                      # warm-up outside the timed loop (work buffers of the padded transforms); writes dU only
    FFT.sync()
    if timing:
        FFT.reset_timing()
    t0 = time.perf_counter()
    for _ in range(steps):
        spectral.axpbz(FFT, U_hat0, U_hat, U_hat, 1.0, 0.0)
        spectral.axpbz(FFT, U_hat1, U_hat, U_hat, 1.0, 0.0)
        for rk in range(4):
            compute_rhs()
            if rk < 3:
                spectral.axpbz(FFT, U_hat, U_hat0, dU, 1.0, b[rk] * dt)
            spectral.axpbz(FFT, U_hat1, U_hat1, dU, 1.0, a[rk] * dt)
        spectral.axpbz(FFT, U_hat, U_hat1, U_hat1, 1.0, 0.0)
    FFT.sync()
    wall = time.perf_counter() - t0
    for i in range(3):
        FFT.ifftn(U_hat.component(i), U.component(i))
    k = FFT.comm.reduce(spectral.sumsq(FFT, U) / float(N[0]) / float(N[1]) / float(N[2]) / 2)
    if report is not None:
        report["ms_per_step"] = 1e3 * wall / steps
        report["fused_nonlinear"] = 0
        report["work_bytes"] = FFT.workspace_bytes() + 3 * Ud.nbytes
        if timing:
            report["stages"] = {k: (v[0] / steps, v[1] // steps) for k, v in FFT.stage_times().items()}
    return k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=5)
    ap.add_argument("--N", type=int, default=0, help="mesh edge instead of 2**M (576, 1152 ...: any length with radix plans)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--ranks", type=int, default=1)
    ap.add_argument("--dealias", default="3/2-rule", choices=["3/2-rule", "2/3-rule", "None"])
    ap.add_argument("--precision", default="double")
    ap.add_argument("--composed", action="store_true", help="the nine-transform composition of rounds 3 - 5 instead of the fused operations")
    ap.add_argument("--stages", action="store_true", help="print the per-stage HIP-event times")
    ap.add_argument("--compact", action="store_true", help="compact spectra (rows of Nf bins) instead of the fused loop's default, rows a whole number of cache lines apart")
    args = ap.parse_args()
    dealias = None if args.dealias == "None" else args.dealias
    from mpifft4py_amd import LocalGroup, SelfComm
    rep = {}
    if args.ranks > 1:
        ks = LocalGroup(args.ranks).run(lambda c: solve(c, args.M, dealias, steps=args.steps, precision=args.precision,
                                                        report=rep if c.Get_rank() == 0 else None, fused=not args.composed,
                                                        timing=args.stages, complex_pitch=None if args.compact else "default",
                                                        edge=args.N or None))
    else:
        ks = [solve(SelfComm(), args.M, dealias, steps=args.steps, precision=args.precision, report=rep,
                    fused=not args.composed, timing=args.stages, complex_pitch=None if args.compact else "default",
                    edge=args.N or None)]
    print("N = %d^3, %d RK4 steps, %.3f ms per step (%s, device-resident; plan work buffers %.2f GB)"
          % (args.N or 2 ** args.M, args.steps, rep.get("ms_per_step", float("nan")),
             "composed: 36 transforms + element-wise kernels" if args.composed else
             ("fused nonlinear z stage" if rep.get("fused_nonlinear") else "one plan operation per nonlinear term, composed inside"),
             rep.get("work_bytes", 0) / 1e9))
    for name, (ms, calls) in sorted(rep.get("stages", {}).items()):
        print("  %-10s %8.3f ms per step  (%d launches)" % (name, ms, calls))
    print("k =", repr(ks[0]))
    if args.M == 5 and not args.N and args.steps == 10 and args.precision == "double":
        assert round(ks[0] - 0.124953117517, 7) == 0
        print("matches the reference demo's known answer 0.124953117517")


if __name__ == "__main__":
    main()
