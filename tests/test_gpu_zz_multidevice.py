"""GPU, boxes with MORE THAN ONE device only (skipped -- visibly -- on a one-GPU lease): every transport with each rank
on its own GPU, so that `pytest -m gpu` itself moves bytes over xGMI where the links exist.

  * LocalGroup: hipDeviceEnablePeerAccess + peer-to-peer copies and cross-device events (csrc/comm.hip)
  * IpcComm:    pull kernel / per-peer copy streams / sequential copies through IPC mappings of a PEER device's memory,
                relay striping (csrc/ipc_comm.hip, ipc_pull.h, relay_plan.h) -- tests/mp_worker.py, one process per GPU
  * RcclComm:   the real librccl (it refuses two ranks on one device, so this is the only place it runs with P > 1)

The file sorts last on purpose: the rest of the suite (whose virtual ranks are spread over the devices too once
gpu_util.peer_probe passes) reports first.  Mirrors the reference's own rank-count gating (tests/test_FFT.py:27-34).

Nothing here has run yet (the pool leases one GPU).  Extra time on an 8-GPU box, ESTIMATED from the same workers on one
shared device (tests/test_gpu_multiprocess.py: mp_worker.py takes 1 / 6 / 21 s at 2 / 4 / 8 processes over IPC, 1 / 2.5 /
9 s over the stand-in for RCCL; a dedicated device per rank can only be faster, real RCCL adds a few seconds of
communicator set-up per world size): test_one_process_per_device 2 x (1 + 6 + 21) = under 60 s, the two LocalGroup tests
(virtual ranks, no process start) under 15 s, relay striping under 30 s -- about 100 s, against a limit of 240 s per
subprocess and MFFT_LOCAL_TIMEOUT = 30 s per wait.  Only modes that have run clean everywhere are swept: the per-peer
streams pull mode (`ipc_pull = 2`, stalled twice in round 4) needs MP_WORKER_STREAMS=1."""
import os

import numpy as np
import pytest

from gpu_util import L, TOL, have_gpu, orc, peer_probe, rank_devices, run_ranks

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ndev():
    from mpifft4py_amd import _lib
    return _lib.device_count()


@pytest.fixture(scope="module", autouse=True)
def _need_two_devices():
    if not have_gpu():
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    if ndev() < 2:
        pytest.skip("one device visible: nothing here can cross a link (needs >= 2 GPUs)")


def worlds():
    n = ndev()
    return [w for w in (2, 4, 8) if w <= n]


def test_peer_copies_between_devices_work():
    ok, text = peer_probe()
    assert ok, text
    assert rank_devices(2) == [0, 1] or os.environ.get("MFFT_TEST_ONE_DEVICE", "0") not in ("", "0")


@pytest.mark.parametrize("prec", ["double", "single"])
def test_local_group_one_rank_per_device(prec):
    """slab + both pencils, every rank on its own GPU, against numpy.fft on the gathered array."""
    from mpifft4py_amd import Pencil_R2C, Slab_R2C
    N = [64, 128, 256]
    A = np.random.default_rng(9).random(N).astype(np.float64 if prec == "double" else np.float32)
    B = np.fft.rfftn(A.astype(np.float64))
    for P in worlds():
        devs = list(range(P))

        def body(comm):
            assert comm.device == devs[comm.Get_rank()]
            comm.selftest(1 << 20, 30000)
            out = []
            Fs = [Slab_R2C(np.array(N), L, comm, prec, pipeline=pl) for pl in (1, 4)]
            if P >= 4:
                Fs += [Pencil_R2C(np.array(N), L, comm, prec, communication="Alltoallw", alignment=al, pipeline=pl)
                       for al in ("X", "Y") for pl in (1, 4)]
            for F in Fs:
                a = np.ascontiguousarray(A[F.real_local_slice()])
                c = F.fftn(a, np.zeros(F.complex_shape(), dtype=F.complex))
                b = F.ifftn(c, np.zeros(F.real_shape(), dtype=F.float))
                out.append((orc.rel_l2(c, B[F.complex_local_slice()]), orc.rel_l2(b, a)))
            return out
        for r, res in enumerate(run_ranks(P, body, devices=devs)):
            for e_fwd, e_back in res:
                assert e_fwd < TOL[prec] and e_back < 4 * TOL[prec], (P, r, res)


@pytest.mark.parametrize("transport", ["ipc", "rccl"])
def test_one_process_per_device(transport):
    """tests/mp_worker.py (selftest, slab x the shipped IPC pull modes -- pull kernel, copy engines -- x CU masks, padded / masked paths, C2C, pencils with and
    without relay striping, all against the oracle and bit-identical between modes) with LOCAL_RANK = rank, i.e. every
    process on its own GPU (comm.from_env); "rccl" = the real library."""
    from test_gpu_multiprocess import _spawn
    for world in worlds():
        rc, out, err = _spawn(world, [os.path.join(ROOT, "tests", "mp_worker.py")], transport=transport, timeout=240)
        assert rc == 0, (transport, world, out[-2000:], err[-4000:])
        assert "MP_OK world=%d" % world in out
