#!/usr/bin/env python3
"""Compact against pitched device spectrum (complex_pitch='auto'), one rank: stage times of the plain pair and of the 3/2-rule
ifftn + fftn pair.   python scripts/pitchprof.py N [precision] [pitch ...]      (pitch: none | auto | <elements>)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpifft4py_amd import Slab_R2C, SelfComm, DeviceArray
n = int(sys.argv[1]); prec = sys.argv[2] if len(sys.argv) > 2 else "double"
pitches = sys.argv[3:] or ["none", "auto"]
N = np.array([n, n, n]); L = np.array([2 * np.pi] * 3)
for pitch in pitches:
    cp = None if pitch == "none" else ("auto" if pitch == "auto" else int(pitch))
    F = Slab_R2C(N, L, SelfComm(0), prec, complex_pitch=cp)
    u = DeviceArray.random(F.real_shape(), F.float, seed=1)
    fu = F.empty_complex()
    u2 = DeviceArray.empty(F.real_shape(), F.float)

    def run(fn, reps):
        for _ in range(3):
            fn()
        F.sync(); F.enable_timing(True); F.reset_timing()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        F.sync()
        dt = (time.perf_counter() - t) / reps * 1e3
        st = {k: round(v[0] / max(v[1], 1), 3) for k, v in sorted(F.stage_times().items()) if v[1]}
        F.enable_timing(False)
        return dt, st

    def pair():
        F.fftn(u, fu); F.ifftn(fu, u2)
    dt, st = run(pair, 10)
    print("%d^3 %s pitch %-5s (%s bins per row)  plain pair %.3f ms  %s" % (n, prec, pitch, F.complex_pitch or F.complex_shape()[2], dt, st), flush=True)
    del u2
    if 27 * n ** 3 * (8 if prec == "double" else 4) < 300e9:
        up = DeviceArray.empty(F.real_shape_padded(), F.float)
        fu2 = F.empty_complex()

        def ppair():
            F.ifftn(fu, up, "3/2-rule"); F.fftn(up, fu2, "3/2-rule")
        dt, st = run(ppair, 5)
        print("%d^3 %s pitch %-5s 3/2-rule pair %.3f ms  %s" % (n, prec, pitch, dt, st), flush=True)
        del up, fu2
    del F, u, fu
