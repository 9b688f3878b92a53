#!/usr/bin/env python
"""bench.py -- 3-D R2C+C2R pairs/s of the slab transform on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one fftn + one ifftn of a 1024^3 fp64 cube (BASELINE.json metric),
inputs and outputs resident in HBM.  N ranks share ONE cube (strong scaling,
slab decomposition, RCCL all-to-all over xGMI).  Rank 0 prints one JSON line.
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", "--n", dest="n", type=int, default=1024,
                    help="cube edge (use --size under torch.distributed.run: its parser rejects the abbreviation --n)")
    ap.add_argument("--decomp", default="slab", choices=["slab", "pencil"])
    ap.add_argument("--precision", default="double", choices=["double", "single"])
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "off"])
    ap.add_argument("--pipeline", type=int, default=0)
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "ipc"],
                    help="multi-rank exchange: RCCL send/recv kernels, copy-engine pulls through IPC-mapped work buffers "
                         "(one node), or auto = measure both and use the faster one")
    ap.add_argument("--rendezvous", default="file", choices=["file", "torch"],
                    help="how rank 0's RCCL unique id reaches the other ranks (torch = gloo process group)")
    ap.add_argument("--pencil-extra", default="auto", choices=["auto", "on", "off"],
                    help="also time the pencil (R2CX) decomposition of the same cube and report it under 'extras'")
    ap.add_argument("--tune-child", default=None, choices=["rccl", "ipc"], help=argparse.SUPPRESS)   # internal: see tune_in_children
    ap.add_argument("--child-task", default="slab", choices=["slab", "pencil_relay"], help=argparse.SUPPRESS)
    ap.add_argument("--stage-timing", default="on", choices=["on", "off"],
                    help="HIP events around every stage inside the timed region (roofline numbers)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one process per GPU, exactly as
    `torch.distributed.run --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), relay rank 0's JSON
    line and leave with the worst exit code.  This process has not touched the GPU (nothing of the package is imported
    yet) and the ranks are fresh children, not an exec of this one."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [c for c in codes if c != 0]
    sys.exit(0 if not bad else (bad[0] if bad[0] > 0 else 1))


ARGS = parse_args() if __name__ == "__main__" else None
if ARGS is not None and ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
    self_launch(ARGS)

import numpy as np  # noqa: E402

# the product library is loaded before anything that could drag in another HIP runtime
from mpifft4py_amd import _lib, comm as mcomm  # noqa: E402
from mpifft4py_amd import DeviceArray, Pencil_R2C, Slab_R2C  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when a communicator is created; stdout is for the JSON line."""
    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:                                   # the banner sits in the C library's stdout buffer until someone flushes it
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def make_comm(world, rendezvous, transport=None):
    if world == 1:
        return mcomm.SelfComm(int(os.environ.get("LOCAL_RANK", "0")) % max(_lib.device_count(), 1)), None
    bcast, dist = None, None
    if rendezvous == "file":
        # single node: rank 0 publishes the unique id in a private directory under /tmp (keyed by MASTER_PORT and the
        # launcher's pid); no second HIP runtime / process group is brought into the workers
        return mcomm.from_env(None, transport=transport), None
    try:
        import torch.distributed as dist       # plumbing only: rendezvous for the RCCL id
        dist.init_process_group("gloo")

        def bcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]
    except Exception as e:                      # noqa: BLE001
        sys.stderr.write("torch.distributed unavailable (%s): file rendezvous\n" % e)
        bcast, dist = None, None
    return mcomm.from_env(bcast, transport=transport), dist


def _default_grid(p):
    """MPI.Compute_dims(p, 2): balanced, non-increasing (what pencil.py:1479 gets from mpi4py)."""
    a = max(d for d in range(1, int(p ** 0.5) + 1) if p % d == 0)
    return p // a, a


def launched_col_kernel(n, precision):
    """The strided-axis kernel the library runs for length n: (plan as "8x8x4x4", tile, full registry name)."""
    import ctypes
    import re
    buf = ctypes.create_string_buffer(256)
    try:
        _lib.call("mfft_kernel_name", 0, n, 1 if precision == "double" else 0, 0, 0, buf, 256)
    except Exception:  # noqa: BLE001
        return None
    name = buf.value.decode()
    m = re.search(r"n%d\(([\d, ]+)\).* tile=(\d+)" % n, name)
    if not m:
        return None
    return m.group(1).replace(", ", "x"), int(m.group(2)), name


def pmc_traffic(n, precision, decomp, world):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc_traffic.json, written by scripts/summarize_profiles.py from separate
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  Only a profile of EXACTLY the kernel that has just
    been timed counts -- same length, radix plan, precision and tile, as the registry reports it
    (mfft_kernel_name) -- so a profile cannot outlive a kernel change; None otherwise."""
    import glob
    if world != 1 or decomp != "slab":
        return None, None
    k = launched_col_kernel(n, precision)
    if k is None:
        return None, "no radix kernel for this length"
    want = "ColFft n=%dx%s %s tile=%d " % (n, k[0], "double" if precision == "double" else "float", k[1])
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            prof = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if ("%d^3" % n) not in prof.get("workload", ""):
            continue
        vals = [v["hbm_bytes_per_launch"] for key, v in prof["kernels"].items() if key.startswith(want) and " pad" not in key]
        if vals:
            return sum(vals) / len(vals), "%s (kernel %s)" % (os.path.basename(path), k[2])
    return None, "no committed PMC profile of '%s'" % want.strip()


def cpu_baseline(n_full, seconds_budget=30.0):
    """The oracle's P = 1 path (numpy.fft semantics: rfftn + irfftn, slab.py:369/249) timed on
    the host cores with scipy.fft's pocketfft, all cores.  A 512^3 pair is timed first; if the
    full cube fits the time and memory budget it is timed as well and reported directly,
    otherwise the sample is scaled by the N^3 log2 N^3 work ratio."""
    import scipy.fft as sfft
    cores = os.cpu_count() or 1

    def pair(n, reps):
        a = np.random.default_rng(1234).random((n, n, n))
        best, err = None, None
        for _ in range(reps):
            t0 = time.perf_counter()
            c = sfft.rfftn(a, workers=cores)
            b = sfft.irfftn(c, s=a.shape, workers=cores)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            if err is None:
                err = float(np.linalg.norm((b[:8] - a[:8]).ravel()) / np.linalg.norm(a[:8].ravel()))
        return best, err
    ns = min(n_full, 512)
    t_s, err = pair(ns, 2)
    work = lambda m: m ** 3 * 3 * np.log2(m)
    scale = work(ns) / work(n_full)
    est_full = t_s / scale
    try:
        avail_kb = int([l for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0].split()[1])
    except Exception:  # noqa: BLE001
        avail_kb = 0
    need_kb = 5 * 8 * n_full ** 3 // 1024
    if ns < n_full and est_full * 3 < seconds_budget and avail_kb > need_kb:
        t_f, err_f = pair(n_full, 2)
        return {"value": 1.0 / t_f, "unit": "pairs/s", "cores": cores, "kind": "port",
                "sample": "FULL %d^3 fp64 rfftn+irfftn pair via scipy.fft (pocketfft, workers=%d): best of 2 = %.3f s "
                          "(round-trip rel-L2 %.1e); 512^3 pair %.3f s" % (n_full, cores, t_f, err_f, t_s)}
    return {"value": (1.0 / t_s) * scale, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d^3 fp64 rfftn+irfftn pair via scipy.fft (pocketfft, workers=%d): %.3f s/pair "
                      "(round-trip rel-L2 %.1e); scaled to %d^3 by N^3*log2(N^3) (x%.4f)"
                      % (ns, cores, t_s, err, n_full, scale)}


def cpu_baseline_cached(n_full, world, max_age_s=6 * 3600):
    """The host timing does not depend on the number of GPUs: the N = 1 run measures it (the full cube when time and
    memory allow) and leaves it in this user's PRIVATE cache directory (mode 0700; the file is written under a random name
    and renamed); a run with N > 1 on the same box reports that measurement next to its own number -- marked `"cached":
    true`, with its age in `sample` -- when it was taken on this host, with this core count, numpy and scipy, less than
    `max_age_s` ago; otherwise it times the bounded 512^3 sample on rank 0 alone (about a second; the other ranks wait at
    the final barrier)."""
    import tempfile
    import time
    d_cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mfft-%d" % os.getuid())
    path = os.path.join(d_cache, "cpu-baseline-%d.json" % n_full)

    def stamp():
        import numpy
        try:
            import scipy
            sv = scipy.__version__
        except Exception:  # noqa: BLE001
            sv = None
        return {"host": os.uname().nodename, "ncpu": os.cpu_count() or 1, "numpy": numpy.__version__, "scipy": sv}
    if world > 1:
        try:
            st = os.lstat(d_cache)
            import stat as _stat
            if _stat.S_ISDIR(st.st_mode) and st.st_uid == os.getuid() and not (st.st_mode & 0o077):
                with open(path) as f:
                    d = json.load(f)
                meta = d.pop("_meta", {})
                age = time.time() - float(meta.get("time", 0))
                if {k: meta.get(k) for k in ("host", "ncpu", "numpy", "scipy")} == stamp() and 0 <= age < max_age_s:
                    d["cached"] = True
                    d["sample"] = "measured %.0f min ago by this box's --gpus 1 run: %s" % (age / 60.0, d["sample"])
                    return d
        except (OSError, ValueError, TypeError):
            pass
        return cpu_baseline(n_full, seconds_budget=0.0)
    d = cpu_baseline(n_full)
    try:
        os.makedirs(d_cache, mode=0o700, exist_ok=True)
        st = os.lstat(d_cache)
        if st.st_uid == os.getuid() and not (st.st_mode & 0o077):
            fd, tmp = tempfile.mkstemp(prefix="cpu-baseline-", dir=d_cache)
            with os.fdopen(fd, "w") as f:
                json.dump(dict(d, _meta=dict(stamp(), time=time.time())), f)
            os.replace(tmp, path)
    except OSError:
        pass
    return d


PULL_NAMES = {1: "", 2: ":streams", 0: ":copy"}      # the IPC transport's ways of pulling (include/mpifft4py_amd.h)
DEPTHS = (1, 2, 4, 8, -2, -4, -8)                    # 1: blocking; kz slices / (negative) batches of local x rows
NWARM, NTIMED = 2, 5                                 # pairs per candidate; the slowest rank counts


def transport_of(c):
    return "ipc" if c.get_option("ipc_pull") >= 0 else "rccl"


def cand_label(c, pull):
    return transport_of(c) + (PULL_NAMES.get(pull, "") if pull is not None else "")


def pencil_grids(n, world):
    """P1 x P2 grids the C ABI takes for an n^3 R2C cube over `world` ranks (x-aligned pencils)"""
    return [(a, world // a) for a in range(1, world + 1) if world % a == 0 and n % a == 0 and n % (world // a) == 0
            and (n // 2 + 1) % (world // a) <= 1 and (world // a == 1 or (n // (world // a)) % 2 == 0)]


def pencil_candidate(pcomm, n, precision, world, rank, grid, depth, ksteps):
    """one x-aligned pencil configuration: 2 untimed + ksteps timed pairs, round trip of the first plane"""
    N, L = np.array([n, n, n]), np.array([2 * np.pi] * 3)
    Fp = Pencil_R2C(N, L, pcomm, precision, P1=(grid[0] if grid else None), communication="Alltoallw",
                    alignment="X", allow_single=True, allow_odd_grid=True, pipeline=depth)
    up = DeviceArray.random(Fp.real_shape(), Fp.float, seed=99 + rank)
    fup = DeviceArray.empty(Fp.complex_shape(), Fp.complex)
    up2 = DeviceArray.empty(Fp.real_shape(), Fp.float)
    for _ in range(2):
        Fp.fftn(up, fup)
        Fp.ifftn(fup, up2)
    Fp.sync()
    pcomm.barrier()
    tp = time.perf_counter()
    for _ in range(ksteps):
        Fp.fftn(up, fup)
        Fp.ifftn(fup, up2)
    Fp.sync()
    pcomm.barrier()
    dtp = time.perf_counter() - tp
    dtp = pcomm.allreduce(dtp, op=mcomm.MAX) if world > 1 else dtp
    a0 = up.leading(0, 1).get()
    b0 = up2.leading(0, 1).get()
    rt = float(np.linalg.norm((a0 - b0).ravel()) / np.linalg.norm(a0.ravel()))
    rt = pcomm.allreduce(rt if rt == rt else 1e30, op=mcomm.MAX) if world > 1 else rt
    return {"grid": [int(Fp.P1), int(Fp.P2)], "pairs_per_s": ksteps / dtp, "ms_per_pair": 1e3 * dtp / ksteps,
            "steps": ksteps, "exchange_pipeline_depth": depth, "roundtrip_rel_l2": rt}


class Tuner:
    """The exchange candidates of ONE communicator on a cube of edge n: transforms a scratch array back into itself and
    checks that it stays what it was, so a candidate only counts if it still computes the right thing on this wire."""

    def __init__(self, c, n, precision, world, rank):
        self.c, self.n, self.precision, self.world, self.rank = c, n, precision, world, rank
        self.N, self.L = np.array([n, n, n]), np.array([2 * np.pi] * 3)
        self.dtype = np.float64 if precision == "double" else np.float32
        self.tol = 1e-9 if precision == "double" else 1e-3
        self.rejected = {}
        self._fresh()

    def _fresh(self):
        self.ut = DeviceArray.random((self.n // self.world, self.n, self.n), self.dtype, seed=7 + self.rank)
        self.k = max(1, min(self.n // self.world, 2))
        self.u_ref = self.ut.leading(0, self.k).get()

    def candidate(self, pull, depth, cus):
        """ms per pair (max over ranks) or None"""
        c = self.c
        if pull is not None:
            c.set_option("ipc_pull", pull)
        Ft = Slab_R2C(self.N, self.L, c, self.precision, pipeline=depth, comm_cus=cus)
        fut = DeviceArray.empty(Ft.complex_shape(), Ft.complex)
        for it in range(NWARM + NTIMED):
            if it == NWARM:
                Ft.sync()
                c.barrier()
                tt = time.perf_counter()
            Ft.fftn(self.ut, fut)
            Ft.ifftn(fut, self.ut)
        Ft.sync()
        c.barrier()
        ms = c.allreduce((time.perf_counter() - tt) / NTIMED, op=mcomm.MAX) * 1e3
        got = self.ut.leading(0, self.k).get()
        err = float(np.linalg.norm((got - self.u_ref).ravel()) / np.linalg.norm(self.u_ref.ravel()))
        err = c.allreduce(err if err == err else 1e30, op=mcomm.MAX)
        del Ft, fut
        if not err <= self.tol:
            self.rejected["%s:%d%s" % (cand_label(c, pull), depth, (":cus%d" % cus) if cus and cus > 0 else "")] = err
            self._fresh()
            return None
        return ms

    def sweep(self, tuning, skip=()):
        """every (pull mode, depth) of this communicator into tuning[label][depth]; returns the best (ms, depth, pull)"""
        # The per-peer copy streams are only worth a measurement when every rank owns a device: with ranks SHARING a
        # device (functional runs) they add seven more queues per process, stream memory operations are spinning
        # kernels, and the hardware scheduler's time slices are all that gets measured (seconds per pair,
        # profiles/r03_ipc_pull_modes.txt) -- and the streams, once created, slow every later candidate down.
        # Round 4: the streams mode is not swept by default at all any more (MFFT_BENCH_PULL_STREAMS=1 adds it when every rank
        # owns a device): in the closing suite of the round one 4-process run of it stalled beyond the transport's 180 s on
        # a shared device, and a candidate that stalls costs the whole line (watchdog, exit code 3) for a mode that has
        # never been the fastest anywhere it could be measured.
        own_device = int(_lib.device_count()) >= self.world
        streams = own_device and os.environ.get("MFFT_BENCH_PULL_STREAMS", "0") == "1"
        pulls = ((1, 2, 0) if streams else (1, 0)) if self.c.get_option("ipc_pull") >= 0 else (None,)
        best = None
        for pull in pulls:
            tc = tuning.setdefault(cand_label(self.c, pull), {})
            for depth in DEPTHS:
                if (cand_label(self.c, pull), depth) in skip:
                    continue
                ms = self.candidate(pull, depth, -1)             # no CU masks here; tried for the winner later
                if ms is None:
                    continue
                tc[depth] = ms
                if best is None or ms < best[0]:
                    best = (ms, depth, pull)
        return best


def tune_child(args):
    """A rank of a CHILD group (bench.py --tune-child T, started by tune_in_children): builds transport T among the
    children, verifies it, sweeps its candidates and lets rank 0 print the table.  Whatever goes wrong here -- a
    refusal, a hang, a GPU fault -- stays in the children."""
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    with stdout_to_stderr():
        c = mcomm.from_env(None, transport=args.tune_child)
        if os.environ.get("MFFT_BENCH_CHILD_FAULT") == str(rank):      # test hook: this child dies the way a GPU fault kills
            os.abort()
        c.selftest(1 << 20, 20000)          # a verified 1 MB-per-peer exchange, at most 20 s: never trust an untried wire
        tuning, rej = {}, {}
        if args.child_task == "pencil_relay":
            # the pencils' sub-group exchanges relay-striped over the links to the ranks outside the group
            # (csrc/relay_plan.h): every grid with two real groups x blocking / pipelined
            c.set_option("ipc_relay", 1)
            for grid in pencil_grids(args.n, world):
                if grid[0] == 1 or grid[1] == 1:
                    continue                  # one exchange over all ranks: nothing to relay
                for depth in (1, 4):
                    tuning["%dx%d:%d:relay" % (grid[0], grid[1], depth)] = pencil_candidate(c, args.n, args.precision, world, rank, grid, depth, 5)
        else:
            t = Tuner(c, args.n, args.precision, world, rank)
            t.sweep(tuning)
            rej = t.rejected
        c.barrier()
    if rank == 0:
        sys.stdout.write(json.dumps({"tuning": tuning, "rejected": rej}) + "\n")
        sys.stdout.flush()


def tune_in_children(comm, other, args, world, rank, timeout_s=300.0, task="slab"):
    """Measure the OTHER transport without letting it near this process: every rank starts a child (same GPU, fresh HIP
    context), the children build the transport among themselves and sweep its candidates (tune_child).  Returns
    (table, rejected) -- the same on every rank -- or raises with what went wrong.  A transport that refuses, hangs or
    faults on this machine costs a few seconds here and nothing else."""
    import tempfile
    path = None
    if rank == 0:
        d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mfft-%d" % os.getuid())
        os.makedirs(d, mode=0o700, exist_ok=True)
        fd, path = tempfile.mkstemp(prefix="tunechild_", dir=d)
        os.close(fd)
        os.unlink(path)
    path = comm.bcast(path, root=0)
    env = {k: v for k, v in os.environ.items() if k not in ("MFFT_TRANSPORT",)}
    # a child whose peer died waits this long for it (less if the caller's environment says so: the tests)
    env.update(MFFT_RENDEZVOUS_FILE=path, MFFT_LOCAL_TIMEOUT=str(min(60, int(os.environ.get("MFFT_LOCAL_TIMEOUT", "60") or 60))))
    cmd = [sys.executable, os.path.abspath(__file__), "--tune-child", other, "--child-task", task, "--gpus", str(world),
           "--size", str(args.n), "--precision", args.precision]
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        out, err = p.communicate(timeout=timeout_s)
        rc, why = p.returncode, (err.decode(errors="replace").strip().splitlines() or ["exit code %d" % p.returncode])[-1]
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        rc, why = -1, "no answer within %.0f s" % timeout_s
    bad = comm.allreduce(0.0 if rc == 0 else 1.0, op=mcomm.MAX)
    res = None
    if rank == 0 and rc == 0:
        try:
            res = json.loads([l for l in out.decode().splitlines() if l.strip()][-1])
        except Exception as e:      # noqa: BLE001
            res, why = None, "unreadable answer (%s)" % e
    res = comm.bcast(res if bad == 0 else None, root=0)
    if res is None:
        raise RuntimeError(comm.bcast(why if rank == 0 else None, root=0) or "a rank's child failed")
    return res["tuning"], res["rejected"]


def main():
    args = ARGS if ARGS is not None else parse_args()
    if args.tune_child:
        return tune_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py --gpus %d was started with WORLD_SIZE=%d: the launcher must start one process per "
                         "GPU (or none: `python bench.py --gpus N` starts its ranks itself)\n" % (args.gpus, world))
        sys.exit(2)
    with stdout_to_stderr():
        first = None if args.transport == "auto" else args.transport     # auto: $MFFT_TRANSPORT, else rccl
        try:
            comm, dist = make_comm(world, args.rendezvous, first)
        except Exception as e:      # noqa: BLE001
            # auto: a machine on which RCCL cannot build the communicator (it refuses, e.g., two ranks on one device)
            # still has the IPC transport; communicator creation is collective, so every rank lands here together
            if args.transport != "auto" or world == 1 or os.environ.get("MFFT_TRANSPORT", "rccl") == "ipc":
                raise
            sys.stderr.write("first transport unavailable (%s: %s): IPC transport\n" % (type(e).__name__, e))
            first = "ipc"
            comm, dist = make_comm(world, "file", first)
    if world > 1:
        comm.transport_name = "ipc" if comm.get_option("ipc_pull") >= 0 else "rccl"      # what was actually built
    n = args.n
    N = np.array([n, n, n])
    L = np.array([2 * np.pi] * 3)

    def set_pull(c, pull):
        if pull is not None:
            c.set_option("ipc_pull", pull)

    def measure(pipeline, comm=comm, pull=None, comm_cus=0):
        """W warm-up pairs, then exactly K timed pairs bracketed by stream sync + device sync + barrier."""
        set_pull(comm, pull)
        if args.decomp == "slab":
            F = Slab_R2C(N, L, comm, args.precision, pipeline=pipeline, comm_cus=comm_cus)
        else:
            F = Pencil_R2C(N, L, comm, args.precision, communication="Alltoallw", alignment="X",
                           allow_single=True, pipeline=pipeline, comm_cus=comm_cus)
        # placement probe (profiles/r05_alloc_shift_probe.txt): a dummy allocation of this many MiB ahead of the arrays
        shift_mb = int(os.environ.get("MFFT_BENCH_ALLOC_SHIFT_MB", "0") or 0)
        shift_buf = DeviceArray.empty((shift_mb << 20,), np.uint8) if shift_mb > 0 else None      # noqa: F841 (kept alive)
        u = DeviceArray.random(F.real_shape(), F.float, seed=1234 + rank)
        fu = DeviceArray.empty(F.complex_shape(), F.complex)
        u2 = DeviceArray.empty(F.real_shape(), F.float)

        def sync_all():
            F.sync()
            _lib.call("mfft_device_sync")
            comm.barrier()

        # stage timing is switched on before the warm-up so that its HIP events exist (and the
        # queue's timestamping is live) before the timed region starts
        F.enable_timing(args.stage_timing == "on")
        for _ in range(args.warmup):
            F.fftn(u, fu)
            F.ifftn(fu, u2)
        sync_all()
        F.reset_timing()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            F.fftn(u, fu)
            F.ifftn(fu, u2)
        F.sync()
        _lib.call("mfft_device_sync")
        comm.barrier()
        dt = time.perf_counter() - t0
        dt = comm.allreduce(dt, op=mcomm.MAX) if world > 1 else dt
        stages = F.stage_times()
        # correctness gate on the data of the timed loop (after it, so that no host copy sits
        # between warm-up and timed region): round trip of the first x-planes
        k = max(1, min(F.real_shape()[0], 2))
        a0 = u.leading(0, k).get()
        b0 = u2.leading(0, k).get()
        rt_err = float(np.linalg.norm((a0 - b0).ravel()) / np.linalg.norm(a0.ravel()))
        return {"dt": dt, "stages": stages, "rt_err": rt_err, "pipeline": pipeline, "comm_cus": comm_cus,
                "transport": cand_name(comm, pull), "ranks": ranks_report(comm)}

    def comm_name(c):
        return getattr(c, "transport_name", "rccl" if world > 1 else "none")

    def ranks_report(c):
        """Who ran: the physical GPU behind every rank (PCI bus id) and, when the communicator is RCCL's, what the library
        ITSELF reports for it (csrc/comm.hip RcclComm::get_option: ncclCommCount / ncclGetVersion / ncclCommUserRank /
        ncclCommCuDevice) -- so that a multi-GPU line proves which library built a communicator over how many ranks
        (the reference asks MPI: slab.py:77-81).  Collective."""
        buf = ctypes.create_string_buffer(64)
        _lib.call("mfft_device_pci_bus_id", int(c.device), buf, 64)
        dom, bus, devfn = buf.value.decode().split(":")
        dev_, fn_ = devfn.split(".")
        code = (int(dom, 16) << 16) | (int(bus, 16) << 8) | (int(dev_, 16) << 3) | int(fn_, 16)
        v = np.zeros(3 * world)
        v[3 * rank] = code
        if comm_name(c) == "rccl":
            v[3 * rank + 1] = c.get_option("rccl_rank") + 1          # 0 = the library has no such entry point
            v[3 * rank + 2] = c.get_option("rccl_device") + 1
        v = c.allreduce(v) if world > 1 else v
        rep = {"devices": ["%04x:%02x:%02x.%x" % (int(x) >> 16, (int(x) >> 8) & 255, (int(x) >> 3) & 31, int(x) & 7) for x in v[0::3]]}
        if comm_name(c) == "rccl":
            ver = c.get_option("rccl_version")
            rep.update(rccl_nranks=c.get_option("rccl_nranks"),
                       rccl_version=("%d.%d.%d" % (ver // 10000, ver // 100 % 100, ver % 100)) if ver > 0 else None,
                       rccl_user_ranks=[int(x) - 1 for x in v[1::3]], rccl_devices=[int(x) - 1 for x in v[2::3]])
        return rep

    def cand_name(c, pull):
        return comm_name(c) + (PULL_NAMES.get(pull, "") if pull is not None else "")

    def link_model(n_, esz_, P_, decomp_):
        """What the node's point-to-point links allow for ONE cube over P GPUs ("What to expect"): every
        transform sends (P - 1) / P of a rank's C / P bytes, 1 / P of it to each peer over that pair's own link -- the slab
        (or an 8 x 1 pencil grid) all P - 1 links at once, the reference's default 4 x 2 pencil grid inside groups of 4 and 2
        (one link carries half of a rank's spectrum).  With the assumed rate per link and direction and the transforms at
        1 / P of the measured one-GPU pair: the speed-up over one GPU with the exchange fully hidden behind the transforms,
        and with nothing hidden.  BASELINE's north star asks for >= 6 x at 8 GPUs on the pencil path: at 1024^3 these
        links hold about 5.6 x for the slab and less for the 4 x 2 grid, whatever the software does -- reported here so that
        a scaling curve is read against it."""
        if P_ <= 1:
            return None
        link_gbs, t1_ms = 77.0, {"f64": 19.6, "f32": 10.2}["f64" if esz_ == 8 else "f32"] * (n_ / 1024.0) ** 3
        Cb = 2.0 * esz_ * n_ * n_ * (n_ // 2 + 1)
        per_peer = Cb / P_ / P_
        if decomp_ == "slab":
            busiest = per_peer                                   # every link carries one peer's chunk
        else:
            p1 = {2: 1, 4: 2, 8: 4, 16: 4}.get(P_, 1)
            p2 = P_ // p1
            # two exchanges one after the other, each inside its group: one peer's share of each (no relay striping)
            busiest = (Cb / P_ / p1 if p1 > 1 else 0.0) + (Cb / P_ / p2 if p2 > 1 else 0.0)
        xchg_ms = 2.0 * busiest / (link_gbs * 1e9) * 1e3          # forward + inverse
        fft_ms = t1_ms / P_
        return {"assumed_GBps_per_link_and_direction": link_gbs, "one_gpu_pair_ms": t1_ms,
                "bytes_over_busiest_link_per_transform": busiest, "exchange_ms_per_pair": xchg_ms,
                "transform_ms_per_pair_per_rank": fft_ms,
                "speedup_ceiling_exchange_hidden": t1_ms / max(fft_ms, xchg_ms),
                "speedup_ceiling_nothing_hidden": t1_ms / (fft_ms + xchg_ms)}

    def headline(mres, tuning):
        dt, stages, rt_err = mres["dt"], mres["stages"], mres["rt_err"]
        esz = 8 if args.precision == "double" else 4
        R = esz * n ** 3
        C = 2 * esz * n * n * (n // 2 + 1)
        alg_pair = 2.0 * (R + 5.0 * C)
        ms = 1e3 * dt / args.steps
        # dominant kernel family: the strided-axis c2c (stages *_x, *_y)
        col = [(k_, v) for k_, v in stages.items() if k_.endswith("_x") or k_.endswith("_y")]
        col_ms = sum(v[0] for _, v in col)
        col_calls = sum(v[1] for _, v in col)
        col_bytes = col[0][1][2] if col else 0.0
        avg_ms = col_ms / max(col_calls, 1)
        achieved = (col_bytes / (avg_ms * 1e-3)) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(n, args.precision, args.decomp, world)
        return {
            "metric": "3D R2C+C2R pairs/sec, %d^3 %s %s" % (n, "fp64" if args.precision == "double" else "fp32", args.decomp),
            "value": args.steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if args.precision == "double" else "f32",
            "data": "synthetic",
            "config": {"workload": "%d^3 %s %s R2C forward+inverse, device-resident, %d rank(s)"
                                   % (n, "fp64" if args.precision == "double" else "fp32", args.decomp, world),
                       "roundtrip_rel_l2": rt_err,
                       # ranks > visible GPUs: several ranks share a device (a functional run, not a scaling point)
                       "gpus_visible": int(_lib.device_count()),
                       "exchange_pipeline_depth": mres["pipeline"] if world > 1 else None,
                       "exchange_comm_cus": mres.get("comm_cus") if world > 1 else None,
                       "exchange_transport": mres["transport"] if world > 1 else None,
                       # the GPUs behind the ranks and, over RCCL, what the library itself says about the communicator
                       **mres["ranks"],
                       "exchange_pipeline_tuning_ms_per_pair": tuning,
                       "alg_bytes_per_pair": alg_pair,
                       "whole_path_hbm_GBs_per_gpu": alg_pair / world / (ms * 1e-3) / 1e9,
                       "whole_path_frac_of_8TBs": alg_pair / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "whole_path_frac_of_6.29TBs_copy_ceiling": alg_pair / world / (ms * 1e-3) / 1e9 / 6290.0,
                       "stage_ms": {k_: v[0] / max(v[1], 1) for k_, v in sorted(stages.items())},
                       "xgmi_link_model": link_model(n, esz, world, args.decomp)},
            "roofline": {"bound": "hbm", "kernel": "col_fft (strided-axis c2c, stages *_x/*_y)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "alg_bytes_per_launch": col_bytes, "avg_launch_ms": avg_ms},
        }

    import threading

    def arm_watchdog(seconds, fallback_line, why):
        """If the work that follows does not finish, rank 0 prints `fallback_line` (a complete, valid measurement
        taken earlier, marked "degraded": true) and every rank leaves with exit code 3: the number survives a
        transport problem in an optional step, but the run does not look green."""
        def give_up():
            line = fallback_line() if callable(fallback_line) else fallback_line     # the latest complete measurement
            if rank == 0 and line is not None:
                line["extras"] = {"note": why}
                line["degraded"] = True
                line.setdefault("cpu_baseline", None)
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            os._exit(3)                    # a step that hangs is a failure, whatever was measured before it
        t = threading.Timer(seconds, give_up)
        t.daemon = True
        t.start()
        return t

    tuning = None
    best = (comm, args.pipeline, None, 0)
    if args.decomp == "slab" and world > 1 and args.pipeline == 0:
        # 1. the plain, blocking exchange: W + K pairs, a complete measurement that is also the fallback line
        base = measure(1)
        tuning = {comm.transport_name: {1: 1e3 * base["dt"] / args.steps}}
        held = {"line": headline(base, {"note": "blocking exchange; the other candidates did not finish"}) if rank == 0 else None,
                "mres": base, "key": None}
        dog = arm_watchdog(600.0, lambda: held["line"], "the exchange candidates (transport, pipeline) did not finish within 600 s")
        # 2. transport x exchange pipeline (flavour and depth) measured on this machine's links, like a planner's MEASURE
        #    mode: 2 untimed + 5 timed pairs per candidate, slowest rank counts
        ipc_first = comm.get_option("ipc_pull") >= 0
        base_pull = 1 if ipc_first else None          # the first measurement ran with the transport's default (pull kernel)
        base_key = (comm, 1, base_pull, 0)
        best, best_ms = base_key, tuning[comm.transport_name][1]
        rejected = {}
        try:
            # 2a. this process's own transport
            tuner = Tuner(comm, n, args.precision, world, rank)
            got = tuner.sweep(tuning, skip={(cand_label(comm, base_pull), 1)})
            rejected.update(tuner.rejected)
            if got is not None and got[0] < best_ms:
                best, best_ms = (comm, got[1], got[2], -1), got[0]
            if args.transport == "auto":
                if best != base_key:
                    # take the winner's complete measurement NOW: whatever the second transport does on this machine, this
                    # number is in hand (the watchdog prints it)
                    held["mres"] = measure(best[1], best[0], best[2], best[3])
                    held["key"] = best
                    if rank == 0:
                        held["line"] = headline(held["mres"], dict(tuning, note="the second transport's candidates did not finish"))
                # 2b. the other transport, in CHILD processes first: a transport that refuses, hangs or faults on this
                #     machine never gets near the process that holds the measurement
                other = "ipc" if comm.transport_name != "ipc" else "rccl"
                try:
                    table, rej = tune_in_children(comm, other, args, world, rank)
                    rejected.update(rej)
                    cand = None
                    for label, per in table.items():
                        tuning[label] = {int(k_): v for k_, v in per.items()}
                        for k_, v in per.items():
                            if cand is None or v < cand[0]:
                                cand = (v, int(k_), label)
                    if cand is not None and cand[0] < best_ms:
                        # it wins: build it here too (it has just worked between the children, on the same devices and links)
                        with stdout_to_stderr():
                            c2, _ = make_comm(world, "file", other)
                        c2.transport_name = other
                        pull2 = {v: k_ for k_, v in PULL_NAMES.items()}.get(cand[2][len(other):], None) if other == "ipc" else None
                        best, best_ms = (c2, cand[1], pull2, -1), cand[0]
                except Exception as e:      # noqa: BLE001  - the same on every rank (tune_in_children agrees on it)
                    sys.stderr.write("transport %s unavailable (%s: %s)\n" % (other, type(e).__name__, e))
                    tuning[other] = {"error": "%s: %s" % (type(e).__name__, e)}
            # 2c. CUs of its own for the communication stream (mfft_plan_desc.comm_cus) only matter for a pipelined winner
            if best[1] != 1:
                t2 = tuner if best[0] is comm else Tuner(best[0], n, args.precision, world, rank)
                tcu = tuning.setdefault("comm_cus", {"candidate": "%s:%d" % (cand_name(best[0], best[2]), best[1]), "none": best_ms})
                for cus in (8, 16, 32):
                    ms = t2.candidate(best[2], best[1], cus)
                    if ms is None:
                        continue
                    tcu[str(cus)] = ms
                    if ms < best_ms:
                        best, best_ms = (best[0], best[1], best[2], cus), ms
                rejected.update(t2.rejected)
                del t2
            del tuner
            if rejected:
                tuning["rejected"] = rejected
        except Exception as e:      # noqa: BLE001  - every rank takes the same path
            sys.stderr.write("exchange tuning failed (%s: %s); keeping the best candidate so far\n" % (type(e).__name__, e))
            tuning["error"] = "%s: %s" % (type(e).__name__, e)
        # 3. the timed region with the best candidate (the first measurement stands if nothing beats it)
        mres = base if best == base_key else held["mres"] if best == held["key"] else measure(best[1], best[0], best[2], best[3])
        dog.cancel()
    else:
        mres = measure(args.pipeline)
    out = headline(mres, tuning) if rank == 0 else None

    # ---- secondary measurement: the pencil path on the same cube (needs a P1 x P2 grid with
    # even factors, i.e. 4 or 8 ranks, or the degenerate 1 x 1 grid as the 1-GPU denominator)
    extras = {}
    want_pencil = args.pencil_extra == "on" or (args.pencil_extra == "auto" and args.decomp == "slab"
                                                 and world in (1, 4, 8, 16))
    watchdog = None
    if want_pencil and world > 1:
        watchdog = arm_watchdog(400.0, out, "the pencil measurement did not finish within 400 s")
    if want_pencil:
        try:
            pcomm = best[0] if (world > 1 and tuning is not None) else comm      # the transport that won the slab measurement
            if world > 1:
                set_pull(pcomm, best[2])
            # process grids: the reference's default (MPI.Compute_dims: 4x2 for 8 ranks) and the others the C ABI takes.
            # On a fully connected xGMI node an exchange inside a group of g ranks uses g - 1 of a GPU's seven links, so
            # the P x 1 and 1 x P grids (one exchange over all ranks, like the slab) are candidates, not curiosities.
            grids = pencil_grids(n, world) if world > 1 else [None]
            per_cand = {}
            for grid in grids:
                for depth in ((1, 4) if world > 1 else (1,)):        # blocking exchanges / the X pipeline (batches of local x rows)
                    ksteps = max(3, min(args.steps, 10)) if len(grids) == 1 else 5
                    r_ = pencil_candidate(pcomm, n, args.precision, world, rank, grid, depth, ksteps)
                    r_.update(relay_striping=False, exchange_transport=cand_name(pcomm, best[2]) if world > 1 else None)
                    per_cand[(r_["grid"][0], r_["grid"][1], depth, 0)] = r_
            # Relay striping of the sub-group exchanges (IPC transport, csrc/relay_plan.h) when every rank owns a device:
            # measured in CHILD processes, like every transport path that has never run on this machine's links
            relay_env = os.environ.get("MFFT_BENCH_RELAY", "auto")       # "0": never, "force": also with ranks sharing a device (tests)
            if world > 1 and any(g[0] > 1 and g[1] > 1 for g in grids) and relay_env != "0" \
                    and (int(_lib.device_count()) >= world or relay_env == "force"):
                try:
                    table_r, _ = tune_in_children(pcomm, "ipc", args, world, rank, task="pencil_relay")
                    for key, r_ in table_r.items():
                        a_, b_ = r_["grid"]
                        r_.update(relay_striping=True, exchange_transport="ipc", measured_in="child processes")
                        per_cand[(a_, b_, r_["exchange_pipeline_depth"], 1)] = r_
                except Exception as e:      # noqa: BLE001
                    sys.stderr.write("relay striping not measured (%s: %s)\n" % (type(e).__name__, e))
                    extras["pencil_relay_striping"] = {"error": "%s: %s" % (type(e).__name__, e)}
            # a candidate that does not reproduce its input on THIS machine's wire is rejected, not reported (and not fatal:
            # the line is about the configurations that work)
            tolp = 1e-10 if args.precision == "double" else 1e-4
            rejected_p = {"%dx%d:%d%s" % (k_[0], k_[1], k_[2], ":relay" if k_[3] else ""): v["roundtrip_rel_l2"]
                          for k_, v in per_cand.items() if not v["roundtrip_rel_l2"] <= tolp}
            ok_cand = {k_: v for k_, v in per_cand.items() if v["roundtrip_rel_l2"] <= tolp}
            if ok_cand:
                per_cand = ok_cand
            bestp = min(per_cand, key=lambda k_: per_cand[k_]["ms_per_pair"])
            table = {"%dx%d:%d%s" % (k_[0], k_[1], k_[2], ":relay" if k_[3] else ""): v["ms_per_pair"] for k_, v in per_cand.items()}
            extras["pencil_R2CX"] = dict(per_cand[bestp], ms_per_pair_by_grid_and_depth=table,
                                         roundtrip_rel_l2=max(v["roundtrip_rel_l2"] for v in per_cand.values()))
            if rejected_p:
                extras["pencil_R2CX"]["rejected"] = rejected_p
            # the reference's own default grid, whatever won
            dflt = [v for k_, v in per_cand.items() if grids[0] is None or list(k_[:2]) == list(_default_grid(world))]
            if dflt:
                extras["pencil_R2CX"]["default_grid_ms_per_pair"] = min(v["ms_per_pair"] for v in dflt)
        except Exception as e:      # noqa: BLE001  - the headline (slab) line must survive
            extras["pencil_R2CX"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if watchdog is not None:
        watchdog.cancel()

    # ---- secondary measurement (one GPU): the dealiased transforms of the same class on the same cube -- the inverse
    # with the 2/3-rule (slab.py:237-245; here pruned passes) and the 3/2-rule ifftn + fftn pair (slab.py:247-346, 453-485)
    if world == 1 and args.decomp == "slab" and args.pencil_extra != "off":
        try:
            Fd = Slab_R2C(N, L, comm, args.precision)
            fud = DeviceArray.random(Fd.complex_shape(), Fd.complex, seed=5)
            ud = DeviceArray.empty(Fd.real_shape(), Fd.float)

            def timed(fn, reps=5):
                for _ in range(2):
                    fn()
                Fd.sync()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                Fd.sync()
                return 1e3 * (time.perf_counter() - t0) / reps
            timed(lambda: Fd.ifftn(fud, ud), 3)          # first use of the plan: work buffers, kernel attributes
            d_extra = {"ifftn_ms": timed(lambda: Fd.ifftn(fud, ud)),
                       "ifftn_two_thirds_rule_ms": timed(lambda: Fd.ifftn(fud, ud, "2/3-rule"))}
            del ud
            if 3 * n % 2 == 0 and 27 * n ** 3 * (8 if args.precision == "double" else 4) < 0.3 * 288e9 * 8:
                upd = DeviceArray.empty(Fd.real_shape_padded(), Fd.float)
                fud2 = DeviceArray.empty(Fd.complex_shape(), Fd.complex)

                def padded_pair():
                    Fd.ifftn(fud, upd, "3/2-rule")
                    Fd.fftn(upd, fud2, "3/2-rule")
                d_extra["three_halves_rule_ifftn_fftn_pair_ms"] = timed(padded_pair, 3)
                del upd, fud2
            extras["dealias"] = d_extra
            del Fd, fud
            # the same two pairs on a PITCHED device spectrum (complex_pitch="auto": rows of 520 instead of 513 bins at 1024^3
            # fp64; the logical shapes stay the reference's, slab.py:102-104).  Opt-in: the headline above is the compact layout.
            Fp = Slab_R2C(N, L, comm, args.precision, complex_pitch="auto")
            up_ = DeviceArray.random(Fp.real_shape(), Fp.float, seed=6)
            fup = Fp.empty_complex()
            up2 = DeviceArray.empty(Fp.real_shape(), Fp.float)
            Fd = Fp                                   # timed() synchronises through Fd

            def plain_pair():
                Fp.fftn(up_, fup)
                Fp.ifftn(fup, up2)
            timed(plain_pair, 3)
            Fp.enable_timing(True)
            Fp.reset_timing()
            p_extra = {"bins_per_row": Fp.complex_pitch, "pair_ms": timed(plain_pair, 10)}
            p_extra["stage_ms"] = {a_: round(b_[0] / max(b_[1], 1), 3) for a_, b_ in sorted(Fp.stage_times().items()) if b_[1]}
            Fp.enable_timing(False)
            rt = float(np.linalg.norm((up_.leading(0, 1).get() - up2.leading(0, 1).get()).ravel()) / np.linalg.norm(up_.leading(0, 1).get().ravel()))
            p_extra["roundtrip_rel_l2"] = rt
            del up2
            if 3 * n % 2 == 0 and 27 * n ** 3 * (8 if args.precision == "double" else 4) < 0.3 * 288e9 * 8:
                upd = DeviceArray.empty(Fp.real_shape_padded(), Fp.float)
                fup2 = Fp.empty_complex()

                def padded_pair_p():
                    Fp.ifftn(fup, upd, "3/2-rule")
                    Fp.fftn(upd, fup2, "3/2-rule")
                p_extra["three_halves_rule_ifftn_fftn_pair_ms"] = timed(padded_pair_p, 3)
                del upd, fup2
            extras["pitched_spectrum"] = p_extra
            del Fp, Fd, up_, fup
        except Exception as e:      # noqa: BLE001  - the headline (slab) line must survive
            extras["dealias"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # ---- secondary measurement (one GPU): the APPLICATION the transforms are for -- the reference demo's Taylor-Green RK4 loop
    # (demo/spectral_dns_solver.py:83-98) with the state in HBM, 3/2-rule, double precision.  `fused`: the nonlinear term as one
    # plan operation (mfft_nonlinear_cross, csrc/fft_nlz.h) + one sweep per Runge-Kutta stage; `composed`: rounds 3 - 5 (36
    # transforms + element-wise kernels per step).  ms per RK4 step and the split by plan stage (HIP events).
    if world == 1 and args.decomp == "slab" and args.pencil_extra != "off" and os.environ.get("MFFT_BENCH_DNS", "1") != "0":
        try:
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "examples"))
            import spectral_dns_device as dns
            tg = {}
            for m_, modes in ((8, ("fused",)), (9, ("fused", "composed"))):
                for mode in modes:
                    rep = {}
                    # (fused: pitched spectra, the demo's default; composed: compact, as rounds 3 - 5 measured it)
                    k_ = dns.solve(comm, M=m_, dealias="3/2-rule", steps=3, report=rep, fused=mode == "fused", timing=True)
                    tg["%d^3_%s" % (2 ** m_, mode)] = {
                        "rk4_step_ms": round(rep["ms_per_step"], 3), "k_after_3_steps": k_,
                        "fused_nonlinear_z_stage": bool(rep.get("fused_nonlinear")), "plan_work_GB": round(rep["work_bytes"] / 1e9, 2),
                        "stage_ms_per_step": {a_: round(b_[0], 3) for a_, b_ in sorted(rep.get("stages", {}).items()) if b_[1]}}
            f_, c_ = tg["512^3_fused"], tg["512^3_composed"]
            f_["element_wise_ms_per_step"] = round(f_["rk4_step_ms"] - sum(f_["stage_ms_per_step"].values()), 3)
            c_["element_wise_ms_per_step"] = round(c_["rk4_step_ms"] - sum(c_["stage_ms_per_step"].values()), 3)
            tg["fused_over_composed_512^3"] = round(f_["rk4_step_ms"] / c_["rk4_step_ms"], 3)
            # ... and under the 2/3-rule (the reference's cheaper dealiasing, slab.py:237-245): pruned inverse passes inside the operation
            for mode in ("fused", "composed"):
                rep = {}
                k_ = dns.solve(comm, M=9, dealias="2/3-rule", steps=3, report=rep, fused=mode == "fused", timing=True)
                tg["512^3_two_thirds_rule_%s" % mode] = {
                    "rk4_step_ms": round(rep["ms_per_step"], 3), "k_after_3_steps": k_,
                    "stage_ms_per_step": {a_: round(b_[0], 3) for a_, b_ in sorted(rep.get("stages", {}).items()) if b_[1]}}
            tg["fused_over_composed_512^3_two_thirds_rule"] = round(
                tg["512^3_two_thirds_rule_fused"]["rk4_step_ms"] / tg["512^3_two_thirds_rule_composed"]["rk4_step_ms"], 3)
            extras["taylor_green_rk4"] = tg
        except Exception as e:      # noqa: BLE001  - the headline (slab) line must survive
            extras["taylor_green_rk4"] = {"error": "%s: %s" % (type(e).__name__, e)}

    # correctness gate: a wrong transform must not print a headline number with rc 0
    tol = 1e-10 if args.precision == "double" else 1e-4
    rt_all = [mres["rt_err"]] + [v["roundtrip_rel_l2"] for v in extras.values() if "roundtrip_rel_l2" in v]
    bad = max(comm.allreduce(max(rt_all), op=mcomm.MAX) if world > 1 else max(rt_all), 0.0)
    failed = not (bad <= tol)              # also catches NaN
    if rank == 0:
        out["extras"] = extras
        if args.cpu_baseline == "auto" and not failed:
            out["cpu_baseline"] = cpu_baseline_cached(n, world)
        else:
            out["cpu_baseline"] = None
        if failed:
            out["degraded"] = True
            out["error"] = "round trip rel-L2 %.3e exceeds %.0e: the transform is wrong, the timing means nothing" % (bad, tol)
        print(json.dumps(out))
    # the line is out: whatever a library still holds in its stdout buffer (RCCL's banner when stdout is a pipe) must
    # not follow it
    sys.stdout.flush()
    os.dup2(2, 1)
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
