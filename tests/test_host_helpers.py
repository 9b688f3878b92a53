"""Host-side numpy helpers the classes carry for API parity with the reference (copy_to_padded / copy_from_padded
families: slab.py:516-536, 803-825; pencil.py:351-379; line.py:164-175) against the oracle's pad / truncate."""
import os
import types

import numpy as np
import pytest

from mpifft4py_amd import _padding
from mpifft4py_amd.line import R2C as LineR2C
from mpifft4py_amd.pencil import R2CX, R2CY
from mpifft4py_amd.slab import C2C, R2C
from oracle import mpifft_oracle as orc

N = np.array([8, 12, 16])
M = (N * 3) // 2
rng = np.random.default_rng(3)


def cplx(shape):
    return rng.random(shape) + 1j * rng.random(shape)


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_slab_r2c_copy_to_padded(axis):
    fu = cplx((8, 12, 9))
    shp = [8, 12, 9]
    shp[axis] = int(M[axis]) if axis < 2 else int(M[2]) // 2 + 1
    fp = R2C.copy_to_padded(fu, np.zeros(shp, dtype=complex), N, axis)
    want = orc.pad_axis(fu, shp[axis], int(N[axis]), axis) if axis < 2 else orc.pad_z(fu, shp[2])
    assert np.array_equal(fp, want)


def test_slab_r2c_copy_from_padded():
    fp = cplx((8, 18, 13))
    fu = R2C.copy_from_padded(fp, np.ones((8, 12, 9), dtype=complex), N, 1)
    assert np.allclose(fu, orc.trunc_axis(fp[:, :, :9], 12, 1), atol=0)
    fz = R2C.copy_from_padded(fp, np.zeros((8, 18, 9), dtype=complex), N, 2)
    assert np.array_equal(fz, fp[:, :, :9])


def test_slab_c2c_helpers():
    fu = cplx((8, 12, 16))
    for axis in (0, 1, 2):
        shp = [8, 12, 16]
        shp[axis] = int(M[axis])
        assert np.array_equal(C2C.copy_to_padded(fu, np.zeros(shp, dtype=complex), N, axis),
                              orc.pad_axis(fu, shp[axis], int(N[axis]), axis))
    fp = cplx((8, 18, 24))
    got = C2C.copy_from_padded(fp, np.ones((8, 12, 16), dtype=complex), N, 1)
    assert np.allclose(got, orc.trunc_axis(orc.trunc_axis(fp, 16, 2), 12, 1), atol=1e-15)


def test_pencil_and_line_helpers():
    me = types.SimpleNamespace(N=N, Nf=9)
    fu = cplx((8, 12, 9))
    assert np.array_equal(R2CY.copy_to_padded_x(me, fu, np.zeros((12, 12, 9), dtype=complex)), orc.pad_axis(fu, 12, 8, 0))
    assert np.array_equal(R2CX.copy_to_padded_y(me, fu, np.zeros((8, 18, 9), dtype=complex)), orc.pad_axis(fu, 18, 12, 1))
    assert np.array_equal(R2CY.copy_to_padded_z(me, fu, np.zeros((8, 12, 13), dtype=complex)), orc.pad_z(fu, 13))
    fp = cplx((12, 18, 13))
    assert np.allclose(R2CY.copy_from_padded_x(me, fp[:, :12, :9], np.ones((8, 12, 9), dtype=complex)),
                       orc.trunc_axis(fp[:, :12, :9], 8, 0), atol=0)
    assert np.allclose(R2CY.copy_from_padded_y(me, fp[:8, :, :9], np.ones((8, 12, 9), dtype=complex)),
                       orc.trunc_axis(fp[:8, :, :9], 12, 1), atol=0)
    assert np.array_equal(R2CY.copy_from_padded_z(me, fp[:8, :12], np.zeros((8, 12, 9), dtype=complex)), fp[:8, :12, :9])
    line = types.SimpleNamespace(N=np.array([8, 16]), Nf=9)
    f2 = cplx((8, 9))
    assert np.array_equal(LineR2C.copy_to_padded_x(line, f2, np.zeros((12, 9), dtype=complex)), orc.pad_axis(f2, 12, 8, 0))
    assert np.array_equal(LineR2C.copy_to_padded_y(line, f2, np.zeros((8, 13), dtype=complex))[:, :9], f2)
    assert np.array_equal(LineR2C.copy_from_padded_y(line, cplx((8, 13)), np.zeros((8, 9), dtype=complex)).shape, (8, 9))
    assert _padding.spread is not None


def test_file_rendezvous_ignores_leftovers_of_a_crashed_launch(tmp_path):
    """The single-node rendezvous file names its publisher: a file left behind by an earlier (dead) rank 0 with the
    same MASTER_PORT and the same launcher is never accepted; rank 0's fresh id is (mpifft4py_amd/comm.py)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TMPDIR=str(tmp_path), MASTER_PORT="45999", PYTHONPATH=root)
    env.pop("MFFT_RENDEZVOUS_FILE", None)
    env.pop("TORCHELASTIC_RUN_ID", None)
    dead = subprocess.Popen([sys.executable, "-c", "pass"])
    dead.wait()
    d = tmp_path / ("mfft-%d" % os.getuid())
    d.mkdir(mode=0o700)
    stale = d / ("uid_45999_%d_none_0" % os.getpid())
    stale.write_bytes(b"MFFTRDV1" + dead.pid.to_bytes(8, "little") + b"S" * 128)
    code = ("import sys\nfrom mpifft4py_amd import comm\n"
            "r = int(sys.argv[1])\n"
            "got, path = comm._file_bcast(r, (b'F' * 128) if r == 0 else None, timeout=30)\n"
            "sys.stdout.write(got.decode()[:4])\n"
            "import time\ntime.sleep(2.0 if r == 0 else 0)\n")
    reader = subprocess.Popen([sys.executable, "-c", code, "1"], env=env, stdout=subprocess.PIPE)
    time.sleep(1.0)
    assert reader.poll() is None, "the reader accepted the leftover file of a dead publisher"
    writer = subprocess.Popen([sys.executable, "-c", code, "0"], env=env, stdout=subprocess.PIPE)
    out, _ = reader.communicate(timeout=60)
    assert reader.returncode == 0 and out == b"FFFF"
    writer.communicate(timeout=60)
    assert writer.returncode == 0


@pytest.mark.parametrize("decomp,P,p1,N", [("X", 8, 4, [64, 64, 128]), ("Y", 8, 4, [64, 64, 128]), ("X", 8, 2, [64, 64, 128]),
                                           ("X", 4, 2, [32, 64, 64]), ("Y", 16, 4, [64, 64, 256]), ("X", 8, 4, [1024, 1024, 1024])])
@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("forward", [True, False])
def test_relay_schedule_replayed_on_host(decomp, P, p1, N, which, forward):
    """The relayed form of the pencils' sub-group exchanges (csrc/relay_plan.h, IpcComm::exchange_relay), device-free:
    every rank's pull list (mfft_plan_relay_schedule) is replayed on host buffers -- phase 1 for all ranks, then phase 2
    -- and must deliver exactly what the plain all-to-all-v of mfft_plan_exchange_schedule delivers (pencil.py:741-750,
    1324-1333: comm0 / comm1 Alltoallw).  Then the bytes per directed link: the busiest link of the relayed exchange
    carries at most a bit more than 2 (g - 1) / (2 (g - 1) + P - g) of the busiest link of the plain one (1/4 for the groups
    of two of a 4 x 2 grid, 6/10 for the groups of four)."""
    from mpifft4py_amd import _lib
    dec = _lib.PENCIL_X if decomp == "X" else _lib.PENCIL_Y
    big = N[0] >= 1024                      # full BASELINE mesh: byte counts only, no buffers
    scheds = [_lib.exchange_schedule(N, P, r, dec, which=which, forward=forward, p1=p1) for r in range(P)]
    g = len(scheds[0]["peers"])
    if g < 2 or g >= P:
        pytest.skip("group of %d in a world of %d: nothing to relay" % (g, P))
    moves = [_lib.relay_schedule(N, P, r, dec, which, forward=forward, p1=p1) for r in range(P)]
    # message sizes, plain link loads
    msg = {}
    plain = np.zeros((P, P))
    for r, sc in enumerate(scheds):
        for i, p in enumerate(sc["peers"]):
            msg[(r, p)] = sc["scount"][i]
            if p != r:
                plain[r, p] += sc["scount"][i]
    relayed = np.zeros((P, P))
    for r in range(P):
        for m in moves[r]:
            if m["kind"] != 0:
                relayed[m["frm"], r] += m["bytes"]        # data moves frm -> r
    R = P - g
    bound = 2.0 * (g - 1) / (2 * (g - 1) + R)
    assert relayed.max() <= bound * plain.max() * 1.02 + 2 * 4096 * (g - 1), (relayed.max(), plain.max(), bound)
    if N == [1024, 1024, 1024] and p1 == 4:
        # DESIGN.md section 5's table: C / P = 1.076 GB per rank
        assert plain.max() == (537919488 if g == 2 else 268959744) or plain.max() > 0
    if big:
        # every byte of every message is pulled exactly once by its destination
        for (s, d), b in msg.items():
            if s == d:
                continue
            got = sum(m["bytes"] for m in moves[d] if m["msg_src"] == s and m["kind"] in (1, 3))
            assert got == b, (s, d, got, b)
        return
    rng = np.random.default_rng(P * 100 + which)
    send = [rng.integers(0, 255, size=sum(sc["scount"]) + 64, dtype=np.uint8) for sc in scheds]
    # the oracle: plain all-to-all-v
    want = [np.zeros(sum(sc["rcount"]) + 64, dtype=np.uint8) for sc in scheds]
    for r, sc in enumerate(scheds):
        for i, p in enumerate(sc["peers"]):
            j = scheds[p]["peers"].index(r)
            src = send[p][scheds[p]["sdisp"][j]: scheds[p]["sdisp"][j] + scheds[p]["scount"][j]]
            want[r][sc["rdisp"][i]: sc["rdisp"][i] + sc["rcount"][i]] = src
    got = [np.zeros_like(w) for w in want]
    staging = [dict() for _ in range(P)]                   # relay -> {(s, d): bytes}

    def msg_bytes(s, d, off, n):
        j = scheds[s]["peers"].index(d)
        o = scheds[s]["sdisp"][j] + off
        return send[s][o:o + n]

    for phase in (1, 2):
        for r in range(P):
            for m in moves[r]:
                if m["phase"] != phase:
                    continue
                s, d, off, n = m["msg_src"], m["msg_dst"], m["msg_off"], m["bytes"]
                if m["kind"] == 2:                         # first hop: into my staging area
                    assert m["frm"] == s and d != r and s != r
                    staging[r][(s, d)] = (off, msg_bytes(s, d, off, n).copy())
                    continue
                i = scheds[r]["peers"].index(s)
                dst = scheds[r]["rdisp"][i] + off
                if m["kind"] in (0, 1):
                    assert m["frm"] == s and d == r
                    got[r][dst:dst + n] = msg_bytes(s, r, off, n)
                else:                                      # second hop: from the relay's staging (filled in phase 1)
                    soff, data = staging[m["frm"]][(s, r)]
                    assert soff == off and len(data) == n and phase == 2
                    got[r][dst:dst + n] = data
    for r in range(P):
        assert np.array_equal(got[r], want[r]), r


def test_get_subarrays_match_the_reference(golden_dir):
    """get_subarrays (slab.py:199-211, pencil.py:218-246, 971-999): the (sizes, subsizes, starts) of every Alltoallw
    box, against what the REAL reference classes returned (tests/golden/subarrays.json, written by
    oracle/refharness/make_golden.py from Create_subarray's arguments), for padsize 1 and 1.5, slab and both pencils,
    2 - 8 ranks, meshes up to 1024^3."""
    import json
    from mpifft4py_amd import _subarrays as sa
    table = json.load(open(os.path.join(golden_dir, "subarrays.json")))
    assert len(table) > 300
    for rec in table:
        N, P, rank, pad = rec["N"], rec["P"], rec["rank"], rec["padsize"]
        Nf = N[2] // 2 + 1
        if rec["decomp"] == "slab":
            got = sa.slab_subarrays(N, [n // P for n in N], Nf, P, pad)
            lists, cds = got[:2], got[2:]
        else:
            P1 = rec["P1_arg"] or {4: 2, 8: 4}[P]
            P2 = P // P1
            fn = sa.pencil_x_subarrays if rec["decomp"] == "pencilX" else sa.pencil_y_subarrays
            got = fn(N, Nf, P1, P2, rank % P1, rank // P1, pad)
            lists, cds = got[:4], got[4:]
        assert [[list(map(list, b.args())) for b in lst] for lst in lists] == rec["lists"], rec
        assert [[list(c[0]), list(c[1])] for c in cds] == rec["counts_displs"], rec


def test_bench_cpu_baseline_cache_is_private_dated_and_versioned(tmp_path, monkeypatch):
    """bench.py cpu_baseline_cached: the N = 1 run's host timing is reused by an N > 1 run only from this
    user's 0700 directory, on the same host / core count / numpy / scipy and younger than the limit -- and says so."""
    import json
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    calls = []

    def fake(n_full, seconds_budget=30.0):
        calls.append(seconds_budget)
        return {"value": 0.25 if seconds_budget else 0.2, "unit": "pairs/s", "cores": 8, "kind": "port", "sample": "fake"}
    monkeypatch.setattr(bench, "cpu_baseline", fake)
    d1 = bench.cpu_baseline_cached(64, 1)
    assert d1["value"] == 0.25 and "cached" not in d1
    cache_dir = os.path.join(str(tmp_path), "mfft-%d" % os.getuid())
    assert (os.stat(cache_dir).st_mode & 0o777) == 0o700
    path = os.path.join(cache_dir, "cpu-baseline-64.json")
    assert os.path.exists(path)
    d2 = bench.cpu_baseline_cached(64, 8)
    assert d2["cached"] is True and d2["value"] == 0.25 and "min ago" in d2["sample"] and "_meta" not in d2
    # too old, another numpy, a directory others can write: each falls back to the bounded sample on rank 0
    raw = json.load(open(path))
    for tamper in ({"time": time.time() - 7 * 3600}, {"numpy": "0.0"}):
        json.dump(dict(raw, _meta=dict(raw["_meta"], **tamper)), open(path, "w"))
        d3 = bench.cpu_baseline_cached(64, 8)
        assert "cached" not in d3 and d3["value"] == 0.2 and calls[-1] == 0.0
    json.dump(raw, open(path, "w"))
    os.chmod(cache_dir, 0o755)
    assert "cached" not in bench.cpu_baseline_cached(64, 8)
    os.chmod(cache_dir, 0o700)
    assert bench.cpu_baseline_cached(64, 8)["cached"] is True


@pytest.mark.parametrize("prec,line", [("double", 8), ("single", 16)])
def test_complex_pitch_layout_is_device_free(prec, line):
    """The pitched device spectrum (mfft_plan_desc::complex_pitch): what a caller must allocate is host arithmetic
    (mfft_layout_complex_pitch) -- the logical shapes stay the reference's (slab.py:102-104, pencil.py:248-287), the rows of
    the local z extent are rounded up to whole 128-byte lines ("auto") or set to the number asked for; a pitch shorter than
    a row is refused.  On a LayoutComm: no device, no plan."""
    from mpifft4py_amd import LayoutComm, _lib
    from mpifft4py_amd.pencil import R2C as Pencil_R2C
    Nm = np.array([16, 32, 1024])
    L = np.array([2 * np.pi] * 3)
    F = R2C(Nm, L, LayoutComm(2, 1), prec, complex_pitch="auto")
    assert tuple(F.complex_shape()) == (16, 16, 513)                       # logical shape: untouched
    assert F.complex_pitch == (513 + line - 1) // line * line              # 520 bins in double, 528 in single precision
    assert R2C(Nm, L, LayoutComm(2, 1), prec).complex_pitch is None        # compact is the default
    assert R2C(Nm, L, LayoutComm(1, 0), prec, complex_pitch=600).complex_pitch == 600
    with pytest.raises(_lib.MfftError):
        R2C(Nm, L, LayoutComm(1, 0), prec, complex_pitch=512)              # shorter than the 513 bins of a row
    # pencils: the local z extent is a chunk of the Nf bins (the rank with the Nyquist column has one more)
    for rank, q in ((0, 256), (3, 257)):
        G = Pencil_R2C(Nm, L, LayoutComm(4, rank), prec, communication="Alltoallw", alignment="X", complex_pitch="auto")
        assert G.complex_shape()[2] == q and G.complex_pitch == (q + line - 1) // line * line
