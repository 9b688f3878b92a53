"""CPU: the N > 1 host logic (groups, peers, all-to-all-v counts/displacements of the
slab and pencil plans) driven through real multi-process exchanges over gloo."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4, 8])
def test_schedules_over_gloo(world):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
    assert "DIST_OK world=%d" % world in outs[0]


def test_schedule_matches_reference_message_sizes():
    """Per-peer chunk sizes at 1024^3 fp64 from SURVEY.md section 2.1 (derived from the reference's Alltoall / Alltoallw
    call sites).  The exchanges whose receiver runs a strided x pass out of the chunks carry one cache line (8 complex128)
    between x rows where the compact pitch reads slowly (plan.hip xplane_pad, round 4): 128 * 513 elements (slab over 8
    ranks, forward), 256 * 257 (x-aligned pencil, forward, the ranks that hold the Nyquist column), 512 * 128 / 512 * 129
    (y-aligned pencil, inverse); the forward z-splitting exchange of the y-aligned pencil carries the rows of the
    Nyquist-holding rank's chunk (129 columns) a whole number of cache lines apart (zrow_pitch: 136); the opposite
    directions have the reference's sizes exactly."""
    from mpifft4py_amd import _lib
    N = [1024] * 3
    s = _lib.exchange_schedule(N, 8, 0, _lib.SLAB, forward=False)
    assert set(s["scount"]) == {128 * 128 * 513 * 16}                     # slab.py:281
    s = _lib.exchange_schedule(N, 8, 0, _lib.SLAB, forward=True)
    assert set(s["scount"]) == {128 * (128 * 513 + 8) * 16}               # slab.py:406 + 128 B per x row
    # R2CY first exchange over comm0 (P1 = 4): 256*512*128*16, 129 columns on the last rank (forward: its rows 136 apart)
    s = _lib.exchange_schedule(N, 8, 0, _lib.PENCIL_Y, which=0, forward=False)
    assert s["peers"] == [0, 1, 2, 3]
    assert s["rcount"] == [256 * 512 * 128 * 16] * 3 + [256 * 512 * 129 * 16]
    s = _lib.exchange_schedule(N, 8, 0, _lib.PENCIL_Y, which=0)
    assert s["scount"] == [256 * 512 * 128 * 16] * 3 + [256 * 512 * 136 * 16]
    s = _lib.exchange_schedule(N, 8, 0, _lib.PENCIL_Y, which=1)            # comm1 (P2 = 2), stride P1
    assert s["peers"] == [0, 4] and set(s["scount"]) == {512 * 512 * 128 * 16}
    s = _lib.exchange_schedule(N, 8, 3, _lib.PENCIL_Y, which=1)            # rank 3: 129 columns, rows 136 apart all the way
    assert set(s["scount"]) == {512 * 512 * 136 * 16}
    s = _lib.exchange_schedule(N, 8, 0, _lib.PENCIL_Y, which=1, forward=False)
    assert set(s["scount"]) == {512 * (512 * 128 + 8) * 16}
    s = _lib.exchange_schedule(N, 8, 3, _lib.PENCIL_Y, which=1, forward=False)      # rank 3 holds kz 384..512: 129 columns
    assert set(s["scount"]) == {512 * (512 * 129 + 8) * 16}
    # R2CX: comm1 first (256 | 257 columns), then comm0
    s = _lib.exchange_schedule(N, 8, 5, _lib.PENCIL_X, which=0, forward=False)
    assert s["peers"] == [1, 5] and s["rcount"] == [256 * 512 * 256 * 16, 256 * 512 * 257 * 16]
    s = _lib.exchange_schedule(N, 8, 5, _lib.PENCIL_X, which=0)
    assert s["scount"] == [256 * 512 * 256 * 16, 256 * 512 * 257 * 16]    # x-aligned: compact (a pitch would only slow its y pass)
    s = _lib.exchange_schedule(N, 8, 5, _lib.PENCIL_X, which=1, forward=False)
    assert s["peers"] == [4, 5, 6, 7] and set(s["scount"]) == {256 * 256 * 257 * 16}
    s = _lib.exchange_schedule(N, 8, 5, _lib.PENCIL_X, which=1, forward=True)
    assert set(s["scount"]) == {256 * (256 * 257 + 8) * 16}
    # MFFT_NO_XPAD=1: the reference's sizes in both directions (checked in a child: the switch is read per plan)
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\nfrom mpifft4py_amd import _lib\n"
            "assert set(_lib.exchange_schedule([1024] * 3, 8, 0, _lib.SLAB)['scount']) == {128 * 128 * 513 * 16}\n"
            "assert set(_lib.exchange_schedule([1024] * 3, 8, 5, _lib.PENCIL_X, which=1)['scount']) == {256 * 256 * 257 * 16}\n" % ROOT)
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, MFFT_NO_XPAD="1"))
    code = ("import sys; sys.path.insert(0, %r)\nfrom mpifft4py_amd import _lib\n"
            "assert _lib.exchange_schedule([1024] * 3, 8, 0, _lib.PENCIL_Y, which=0)['scount'] == [256 * 512 * 128 * 16] * 3 + [256 * 512 * 129 * 16]\n" % ROOT)
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, MFFT_NO_ZPITCH="1"))


def test_file_rendezvous_two_processes(tmp_path):
    """comm._file_bcast: rank 0 publishes 128 bytes atomically, rank 1 polls (the bootstrap
    bench.py uses under torch.distributed.run on one node)."""
    code = (
        "import os,sys; sys.path.insert(0, %r)\n"
        "from mpifft4py_amd import comm\n"
        "r=int(os.environ['RANK'])\n"
        "payload = bytes(range(128)) if r == 0 else None\n"
        "data, path = comm._file_bcast(r, payload, timeout=60)\n"
        "assert data == bytes(range(128)), data\n"
        "print('OK', r)\n" % ROOT)
    path = str(tmp_path / "uid")
    procs = []
    for r in (1, 0):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_PORT="12345", MFFT_RENDEZVOUS_FILE=path)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0 and b"OK" in out, out.decode()
