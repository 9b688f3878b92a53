"""Worker of tests/test_gpu_multiprocess.py::test_ipc_survivor_of_a_killed_peer: two processes over the IPC transport, CU-masked
plan streams (comm_cus > 0: blocking streams, the case ADVICE r03 found a deadlock in), pull kernel.  After one good pair
rank 1 is SIGKILLed -- it cannot mark the group broken -- and rank 0 starts the next transform: it must get an error within
the transport's timeout (MFFT_LOCAL_TIMEOUT), and it must still be able to drop its plan and communicator and leave."""
import os
import signal
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mpifft4py_amd import DeviceArray, Slab_R2C, _lib, from_env  # noqa: E402


def main():
    import faulthandler
    faulthandler.dump_traceback_later(200, exit=True)
    comm = from_env()
    rank = comm.Get_rank()
    comm.set_option("ipc_pull", 1)
    N = np.array([64, 64, 128])
    F = Slab_R2C(N, np.array([2 * np.pi] * 3), comm, "double", pipeline=4, comm_cus=int(os.environ.get("PEER_DIES_CUS", "16")))
    u = DeviceArray.random(F.real_shape(), F.float, seed=rank)
    fu = DeviceArray.empty(F.complex_shape(), F.complex)
    F.fftn(u, fu)
    F.ifftn(fu, u)
    F.sync()
    comm.barrier()
    if rank == 1:
        os.kill(os.getpid(), signal.SIGKILL)
    time.sleep(1.0)                      # the peer is gone
    t0 = time.time()
    failed = False
    try:
        F.fftn(u, fu)
        F.sync()
    except _lib.MfftError as e:
        failed = True
        print("survivor got: %s" % str(e)[:160], flush=True)
    dt = time.time() - t0
    assert failed, "a transform with a dead peer returned without an error"
    del F, u, fu
    comm.free()
    print("SURVIVOR_OK error after %.1f s, left after %.1f s" % (dt, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
