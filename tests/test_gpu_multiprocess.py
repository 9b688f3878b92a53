"""GPU: the process-per-GPU product path with 2, 4 and 8 PROCESSES launched by
torch.distributed.run -- exactly as the benchmark driver launches bench.py -- on the single
GPU of the box, over two transports:
  * "ipc":  the shipped IpcComm (csrc/ipc_comm.hip): every rank pulls its chunks out of the peers' IPC-mapped work
            buffers (one pull kernel over all peers / per-peer copy streams / copies in sequence), stream memory
            operations as flags between the processes.  Nothing is mocked: this IS the product's wire, it just runs
            with all ranks on one device.
  * "mock": RcclComm.  Real RCCL refuses two ranks on one device, so librccl is replaced behind the same dlsym'd entry
            points by tests/mock_rccl (shared-memory mailboxes); everything above the wire (rendezvous, DistComm,
            RcclComm.alltoallv, plans, pipeline, bench.py's rank-0 JSON line) is the shipped code."""
import json
import os
import socket
import subprocess
import sys

import pytest

from gpu_util import have_gpu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK_DIR = os.path.join(ROOT, "tests", "mock_rccl")
MOCK = os.path.join(MOCK_DIR, "libmockrccl.so")


@pytest.fixture(scope="module", autouse=True)
def mock_lib():
    if not have_gpu():
        pytest.fail("no GPU visible")
    src = os.path.join(MOCK_DIR, "mock_rccl.cpp")
    if not os.path.exists(MOCK) or os.path.getmtime(MOCK) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-std=c++17", src, "-o", MOCK, "-lrt"])
    return MOCK


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(transport="mock"):
    env = {k: v for k, v in os.environ.items() if k not in ("MFFT_TRANSPORT", "MFFT_RCCL_LIB")}
    env["OMP_NUM_THREADS"] = "1"
    env.setdefault("MP_WORKER_VERBOSE", "1")
    # a rank that waits for a peer gives up after 30 s (the transports' default is 180 s) and every subprocess below is
    # cut off after at most 240 s: a stall is a named failure well inside the driver's step limit
    env.setdefault("MFFT_LOCAL_TIMEOUT", "30")
    if transport == "ipc":
        env["MFFT_TRANSPORT"] = "ipc"
    elif transport == "rccl":            # the real librccl: needs a device per rank (tests/test_gpu_zz_multidevice.py)
        env["MFFT_TRANSPORT"] = "rccl"
    else:
        env.update(MFFT_RCCL_LIB=MOCK, MOCK_RCCL_SLOT_KB="2048")
    return env


TRANSPORTS = ["ipc", "mock"]


def _torchrun(nproc, script_args, timeout=240, transport="mock"):
    """The driver's launch line (python -m torch.distributed.run ...)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + script_args
    p = subprocess.run(cmd, env=_env(transport), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    return p.returncode, p.stdout.decode(), p.stderr.decode()


def _spawn(nproc, script_args, timeout=240, transport="mock"):
    """Same environment contract (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) without the launcher's
    own start-up cost: one python process per rank started directly."""
    port = _free_port()
    procs = []
    for r in range(nproc):
        env = dict(_env(transport), RANK=str(r), WORLD_SIZE=str(nproc), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + script_args, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, cwd=ROOT))
    rc, outs, errs = 0, [], []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            tails = []
            for q in procs:
                try:
                    o2, e2 = q.communicate(timeout=10)
                    tails.append(e2.decode()[-1500:])
                except Exception:  # noqa: BLE001
                    pass
            raise AssertionError("ranks still running after %d s; stderr tails:\n%s" % (timeout, "\n----\n".join(tails)))
        rc = rc or p.returncode
        outs.append(o.decode())
        errs.append(e.decode())
    return rc, "".join(outs), "\n".join(errs)


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_process_per_rank_parity(world, transport):
    rc, out, err = _spawn(world, [os.path.join(ROOT, "tests", "mp_worker.py")], transport=transport)
    if rc != 0:         # the whole story where a failed run can be read afterwards (pytest shortens the message below)
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "mp_worker_failure_%d_%s.txt" % (world, transport)), "w") as f:
            f.write("==== stdout\n" + out + "\n==== stderr\n" + err)
    assert rc == 0, (out[-2000:], err[-4000:])
    assert "MP_OK world=%d" % world in out


@pytest.mark.parametrize("world,n,transport", [(4, 512, "ipc"), (8, 512, "ipc"), (2, 1024, "ipc"),
                                               pytest.param(4, 512, "mock", marks=pytest.mark.slow)])
def test_process_per_rank_full_size(world, n, transport):
    """BASELINE config sizes over real PROCESSES and the shipped wire (these ran by hand only): 512^3
    over 4 and 8 processes, 1024^3 over 2, every pipeline flavour of the slab plan and both pencils, against the host's
    pocketfft of the whole cube (tests/mp_worker_big.py)."""
    import subprocess as sp
    env_extra = {"MP_N": str(n), "MP_WORKERS": str(max(4, (os.cpu_count() or 8) // world))}
    old = {k: os.environ.get(k) for k in env_extra}
    os.environ.update(env_extra)
    try:
        rc, out, err = _spawn(world, [os.path.join(ROOT, "tests", "mp_worker_big.py")], transport=transport, timeout=240)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert rc == 0, (out[-2000:], err[-4000:])
    assert "BIG_OK world=%d n=%d" % (world, n) in out


# 8 processes x two transports (the second one in child processes) x every candidate x every pencil grid on ONE device
# takes 90 s: `slow`; the 8-rank line over the IPC transport alone (30 s) and both transports at 2 and 4 ranks stay.
@pytest.mark.parametrize("world,launcher,transport", [
    (2, "torchrun", "ipc"), (2, "torchrun", "mock"), (4, "spawn", "ipc"), (4, "spawn", "mock"), (8, "spawn", "ipc"),
    pytest.param(8, "spawn", "mock", marks=pytest.mark.slow)])
def test_bench_multi_rank_prints_one_json_line(world, launcher, transport):
    run = _torchrun if launcher == "torchrun" else _spawn
    # "mock": --transport auto, i.e. the (mocked) RCCL path first and the IPC transport as the second candidate, the
    # way the driver's multi-GPU run goes; "ipc": the IPC transport alone (real RCCL would refuse the shared GPU)
    rc, out, err = run(world, [os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--size", "128", "--steps", "3",
                               "--warmup", "1", "--cpu-baseline", "off", "--transport", "ipc" if transport == "ipc" else "auto"],
                       transport=transport)
    assert rc == 0, (out[-2000:], err[-4000:])
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                      # exactly one line on stdout, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "strong" and not d.get("degraded")
    assert d["config"]["roundtrip_rel_l2"] < 1e-10
    assert d["value"] > 0 and d["roofline"]["achieved"] > 0
    # transport x exchange-pipeline flavour are measured before the timed region, the best candidate is used
    tun = d["config"]["exchange_pipeline_tuning_ms_per_pair"]
    assert "rejected" not in tun and "error" not in tun, tun
    cus = tun.pop("comm_cus", None)                  # CU masks tried for a pipelined winner
    # the IPC transport's ways of pulling, one communicator ("ipc:streams" is measured only when every rank owns a device)
    ipc_names = ["ipc", "ipc:copy"] + (["ipc:streams"] if d["config"]["gpus_visible"] >= world else [])
    assert sorted(tun) == sorted(ipc_names if transport == "ipc" else ipc_names + ["rccl"]), tun
    best = None
    for name, per in tun.items():
        assert sorted(int(k) for k in per) == [-8, -4, -2, 1, 2, 4, 8] and all(v > 0 for v in per.values()), tun
        for k, v in per.items():
            if best is None or v < best[0]:
                best = (v, name, int(k))
    assert (d["config"]["exchange_transport"], d["config"]["exchange_pipeline_depth"]) == best[1:]
    if cus is not None:
        assert best[2] != 1 and cus["candidate"] == "%s:%d" % best[1:] and {"8", "16", "32"} <= set(cus), cus
    if world >= 4:
        # every process grid the C ABI accepts is timed, the reference's default among them
        pen = d["extras"]["pencil_R2CX"]
        grids = {k.split(":")[0] for k in pen["ms_per_pair_by_grid_and_depth"]}
        assert grids == ({"1x4", "2x2", "4x1"} if world == 4 else {"1x8", "2x4", "4x2", "8x1"}), pen
        assert pen["roundtrip_rel_l2"] < 1e-10 and pen["default_grid_ms_per_pair"] > 0


@pytest.mark.parametrize("world,transport", [(2, "ipc"), (2, "mock"), (4, "ipc"), pytest.param(4, "mock", marks=pytest.mark.slow)])
def test_bench_starts_its_own_ranks_without_a_launcher(world, transport):
    """`python bench.py --gpus N` as the driver's 1-GPU command line would look with N > 1: no torch.distributed.run,
    no RANK / WORLD_SIZE in the environment.  bench.py starts its N ranks itself (fresh child processes) and still
    prints exactly one JSON line."""
    env = {k: v for k, v in _env(transport).items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--size", "128", "--steps", "3",
                        "--warmup", "1", "--cpu-baseline", "off", "--pencil-extra", "off", "--transport",
                        "ipc" if transport == "ipc" else "auto"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240, cwd=ROOT)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["value"] > 0 and d["config"]["roundtrip_rel_l2"] < 1e-10 and not d.get("degraded")


def test_bench_falls_back_when_rccl_refuses():
    """`python bench.py --gpus 2` with the REAL librccl on a box with one GPU: RCCL refuses two ranks on one device, every
    rank gets the same error at communicator creation, and `--transport auto` continues over the IPC transport -- one
    JSON line, the refusal recorded in the tuning table, `gpus_visible` telling that ranks share a device."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                             "MFFT_TRANSPORT", "MFFT_RCCL_LIB")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "128", "--steps", "3",
                        "--warmup", "1", "--cpu-baseline", "off", "--pencil-extra", "off"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240, cwd=ROOT)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and cfg["roundtrip_rel_l2"] < 1e-10 and not d.get("degraded")
    if cfg["gpus_visible"] < 2:
        # ("ipc", "ipc:copy" ...: whichever pull mode of the IPC transport measured fastest in this run)
        assert cfg["exchange_transport"].split(":")[0] == "ipc" and "error" in cfg["exchange_pipeline_tuning_ms_per_pair"]["rccl"]


def test_bench_measures_relay_striping_in_children():
    """The pencil extra's relay-striped candidates (csrc/relay_plan.h) run in child processes over the IPC transport
    (bench.py tune_in_children, task pencil_relay); forced here although the ranks share the box's one GPU."""
    env = {k: v for k, v in _env("ipc").items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MFFT_BENCH_RELAY"] = "force"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--size", "128", "--steps", "2",
                        "--warmup", "1", "--cpu-baseline", "off", "--transport", "ipc", "--pipeline", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd=ROOT)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    d = json.loads([l for l in out.splitlines() if l.strip()][0])
    pen = d["extras"]["pencil_R2CX"]
    table = pen["ms_per_pair_by_grid_and_depth"]
    assert {"2x2:1", "2x2:4", "2x2:1:relay", "2x2:4:relay", "1x4:1", "4x1:1"} <= set(table), table
    assert "rejected" not in pen and pen["roundtrip_rel_l2"] < 1e-10 and "pencil_relay_striping" not in d["extras"]


def test_bench_survives_a_second_transport_that_faults():
    """`--transport auto` measures the second transport in CHILD processes (bench.py tune_in_children): a child that dies
    the way a GPU fault kills a process (abort) costs the run nothing -- one JSON line from the first transport, rc 0,
    the failure recorded in the tuning table."""
    env = {k: v for k, v in _env("mock").items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MFFT_BENCH_CHILD_FAULT"] = "1"               # rank 1's child aborts right after the communicator is built
    env["MFFT_LOCAL_TIMEOUT"] = "10"                  # ... and rank 0's child gives up waiting for it after this long
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--steps", "2",
                        "--warmup", "1", "--cpu-baseline", "off", "--pencil-extra", "off"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=240, cwd=ROOT)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    tun = d["config"]["exchange_pipeline_tuning_ms_per_pair"]
    assert d["config"]["exchange_transport"] == "rccl" and not d.get("degraded") and "error" in tun["ipc"], tun
    assert sorted(int(k) for k in tun["rccl"]) == [-8, -4, -2, 1, 2, 4, 8]


def test_bench_ranks_stay_together_when_rank0_cannot_make_an_id():
    """a rank 0 that fails BEFORE it publishes the unique id (here: a transport name the library rejects in
    mfft_get_unique_id) must not leave the other ranks polling for the id while it moves on to the next rendezvous:
    it publishes the failure, every rank raises at once, and `--transport auto` continues over the IPC transport."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                             "MFFT_RCCL_LIB")}
    env["MFFT_TRANSPORT"] = "no-such-transport"
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "64", "--steps", "2",
                        "--warmup", "1", "--cpu-baseline", "off", "--pencil-extra", "off", "--pipeline", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240, cwd=ROOT)
    out, err = p.stdout.decode(), p.stderr.decode()
    assert p.returncode == 0, (out[-2000:], err[-4000:])
    assert time.time() - t0 < 150, "the ranks waited for each other's time-outs"
    d = json.loads([l for l in out.splitlines() if l.strip()][0])
    assert d["n_gpus"] == 2 and d["config"]["exchange_transport"].split(":")[0] == "ipc" and not d.get("degraded")
    assert "first transport unavailable" in err


@pytest.mark.parametrize("cus", [pytest.param(16, marks=pytest.mark.slow), -1])      # (CU-masked streams: with MFFT_TEST_SLOW=1)
def test_ipc_survivor_of_a_killed_peer(cus):
    """with CU-masked (= blocking) plan streams the host-side release of a hung exchange queued behind
    the very wait kernel it must release, so a dead peer meant a permanent hang.  Two IPC processes, rank 1 SIGKILLed between
    two transforms: rank 0 gets an error within the transport's timeout and can still free its plan and communicator
    (tests/mp_worker_peer_dies.py); with CU masks and without."""
    import time
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(_env("ipc"), RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MFFT_LOCAL_TIMEOUT="10", PEER_DIES_CUS=str(cus))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_worker_peer_dies.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT))
    t0 = time.time()
    try:
        out0, err0 = procs[0].communicate(timeout=240)
    except subprocess.TimeoutExpired:
        for q in procs:
            q.kill()
        out0, err0 = procs[0].communicate(timeout=10)
        raise AssertionError("the survivor hung; stderr tail:\n" + err0.decode()[-3000:])
    procs[1].wait(timeout=30)
    assert procs[1].returncode != 0                      # it was killed
    assert procs[0].returncode == 0 and b"SURVIVOR_OK" in out0, (out0.decode()[-2000:], err0.decode()[-3000:])
    assert time.time() - t0 < 200


@pytest.mark.parametrize("world", [2, 4])
def test_mpi4py_like_communicator_is_wrapped(world):
    """INTEGRATION.md route A with the caller's own communicator object: the constructors accept anything with
    Get_rank / Get_size / bcast (an mpi4py communicator) and build the RCCL communicator through it."""
    rc, out, err = _spawn(world, [os.path.join(ROOT, "tests", "mp_worker_mpi4py_like.py")])
    assert rc == 0 and "MPI4PY_LIKE_OK %d" % world in out, (out[-2000:], err[-4000:])
