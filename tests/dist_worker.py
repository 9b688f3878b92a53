"""Worker of tests/test_dist_gloo.py (CPU, torch.distributed/gloo, no GPU).

Each rank runs the per-rank distributed algorithm with numpy stages and moves
the data with gloo point-to-point messages whose peers, byte counts and
displacements come from the C library's host-only schedule query
(mfft_plan_exchange_schedule) -- the very schedule the HIP executor hands to
RCCL.  The result must equal the oracle's world formulation.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from mpifft4py_amd import _lib  # noqa: E402
from oracle import mpifft_oracle as orc  # noqa: E402

ES = 16   # complex128


def exchange(rank, sched, send, recv_bytes):
    send = np.ascontiguousarray(send).view(np.uint8).reshape(-1)
    recv = np.zeros(recv_bytes, dtype=np.uint8)
    reqs, keep = [], []
    for i, peer in enumerate(sched["peers"]):
        sc, sd, rc, rd = (sched[k][i] for k in ("scount", "sdisp", "rcount", "rdisp"))
        if peer == rank:
            assert sc == rc
            recv[rd:rd + rc] = send[sd:sd + sc]
            continue
        t_out = torch.from_numpy(send[sd:sd + sc].copy())
        t_in = torch.from_numpy(recv[rd:rd + rc])
        keep += [t_out, t_in]
        reqs.append(dist.isend(t_out, dst=peer))
        reqs.append(dist.irecv(t_in, src=peer))
    for r in reqs:
        r.wait()
    assert sum(sched["rcount"]) == recv_bytes
    return recv.view(np.complex128)


def exchange_relayed(rank, P, N, dec, which, forward, p1, send, recv_bytes):
    """The same all-to-all-v, relay-striped as the IPC transport does it (csrc/relay_plan.h, mfft_plan_relay_schedule):
    every rank PULLS what its move list says, phase 1 (direct first halves, first hops into a staging area) for all
    ranks, then phase 2 (direct second halves, second hops out of the relays' staging areas).  Over gloo a pull is a
    message from the rank that is read: every rank knows every rank's list (the query is device-free), so the owner
    of the bytes sends them, tagged with the puller's move number."""
    send = np.ascontiguousarray(send).view(np.uint8).reshape(-1)
    recv = np.zeros(recv_bytes, dtype=np.uint8)
    scheds = [_lib.exchange_schedule(N, P, r, dec, which, forward, p1=p1) for r in range(P)]
    moves = [_lib.relay_schedule(N, P, r, dec, which, forward=forward, p1=p1) for r in range(P)]
    staging = {}

    def src_slice(s_, d_, off, n):          # bytes [off, off + n) of the message s_ -> d_ inside s_'s send buffer
        j = scheds[s_]["peers"].index(d_)
        o = scheds[s_]["sdisp"][j] + off
        return o, o + n

    def dst_slice(s_, off, n):              # ... inside MY receive buffer
        i = scheds[rank]["peers"].index(s_)
        o = scheds[rank]["rdisp"][i] + off
        return o, o + n

    for phase in (1, 2):
        reqs, keep, staged_in = [], [], []
        # what I am read for: everybody else's moves of this phase whose data lies with me
        for x in range(P):
            if x == rank:
                continue
            for t, m in enumerate(moves[x]):
                if m["phase"] != phase or m["frm"] != rank or m["kind"] == 0:
                    continue
                if m["kind"] in (1, 2):
                    a, b = src_slice(m["msg_src"], m["msg_dst"], m["msg_off"], m["bytes"])
                    assert m["msg_src"] == rank
                    buf = send[a:b].copy()
                else:                       # second hop: out of my staging area
                    off, data = staging[(m["msg_src"], m["msg_dst"])]
                    assert off == m["msg_off"] and len(data) == m["bytes"]
                    buf = data
                tt = torch.from_numpy(buf)
                keep.append(tt)
                reqs.append(dist.isend(tt, dst=x, tag=t))
        # what I pull
        for t, m in enumerate(moves[rank]):
            if m["phase"] != phase:
                continue
            if m["kind"] == 0:
                a, b = src_slice(rank, rank, 0, m["bytes"])
                c, d = dst_slice(rank, 0, m["bytes"])
                recv[c:d] = send[a:b]
            elif m["kind"] == 2:
                buf = np.zeros(m["bytes"], dtype=np.uint8)
                tt = torch.from_numpy(buf)
                keep.append(tt)
                staged_in.append(((m["msg_src"], m["msg_dst"]), m["msg_off"], buf))
                reqs.append(dist.irecv(tt, src=m["frm"], tag=t))
            else:
                c, d = dst_slice(m["msg_src"], m["msg_off"], m["bytes"])
                tt = torch.from_numpy(recv[c:d])
                keep.append(tt)
                reqs.append(dist.irecv(tt, src=m["frm"], tag=t))
        for r in reqs:
            r.wait()
        for key, off, buf in staged_in:
            staging[key] = (off, buf)
        dist.barrier()                      # phase 2 reads what phase 1 staged (the transport: the k1done flags)
    return recv.view(np.complex128)


def exchange_pieces(rank, pieces, send, recv_bytes):
    """A pipelined exchange, piece by piece into ONE receive buffer (displacements are relative to the whole buffers)."""
    send = np.ascontiguousarray(send).view(np.uint8).reshape(-1)
    recv = np.zeros(recv_bytes, dtype=np.uint8)
    got = 0
    for sched in pieces:
        reqs, keep = [], []
        for i, peer in enumerate(sched["peers"]):
            sc, sd, rc, rd = (sched[k][i] for k in ("scount", "sdisp", "rcount", "rdisp"))
            got += rc
            if peer == rank:
                recv[rd:rd + rc] = send[sd:sd + sc]
                continue
            t_out = torch.from_numpy(send[sd:sd + sc].copy())
            t_in = torch.from_numpy(recv[rd:rd + rc])
            keep += [t_out, t_in]
            reqs.append(dist.isend(t_out, dst=peer))
            reqs.append(dist.irecv(t_in, src=peer))
        for r in reqs:
            r.wait()
    assert got == recv_bytes, (got, recv_bytes)          # the pieces tile the receive buffer exactly
    return recv.view(np.complex128)


def slab_pipelined(rank, P, N, A, pipeline):
    """The device path's DEFAULT multi-rank slab transform (kz slices) and its row-batch flavour, with the piece
    schedules of mfft_plan_exchange_pieces -- the ones the executor hands to RCCL piece by piece."""
    lay = orc.SlabLayout(N, P)
    want = orc.slab_r2c_forward(orc.scatter_real(A, lay), N)
    Np0, Np1, Nf = int(lay.Np[0]), int(lay.Np[1]), lay.Nf
    u = np.ascontiguousarray(A[lay.real_local_slice(rank)])
    a = np.fft.rfft2(u, axes=(1, 2))                                      # (Np0, N1, Nf)
    total = Np0 * N[1] * Nf * ES
    fwd = _lib.exchange_pieces(N, P, rank, _lib.SLAB, 0, True, pipeline)
    bwd = _lib.exchange_pieces(N, P, rank, _lib.SLAB, 0, False, pipeline)
    packed = orc.slab_pack(a, P)                                          # (P, Np0, Np1, Nf)
    if pipeline < 0:        # row batches: the packed layout itself, pieces = row ranges of every peer block
        assert len(fwd) == min(-pipeline, Np0)
        r = exchange_pieces(rank, fwd, packed, total)
        fu = np.fft.fft(r.reshape(N[0], Np1, Nf), axis=0)
        assert orc.rel_l2(fu, want[rank]) < 1e-13
        r = exchange_pieces(rank, bwd, np.fft.ifft(fu, axis=0), total)
        back = np.fft.irfft2(orc.slab_unpack(r.reshape(P, Np0, Np1, Nf)), s=(N[1], N[2]), axes=(1, 2))
        assert orc.rel_l2(back, u) < 1e-13
        return
    # kz slices: send / receive buffers hold slice after slice, each packed as (P, Np0, Np1, kz_s)
    kz = [p["scount"][0] // (Np0 * Np1 * ES) for p in fwd]
    k0 = np.concatenate([[0], np.cumsum(kz)[:-1]]).astype(int)
    assert sum(kz) == Nf and all(p["sdisp"][0] == P * Np0 * Np1 * int(k) * ES for p, k in zip(fwd, k0))
    send = np.concatenate([packed[:, :, :, k:k + w].ravel() for k, w in zip(k0, kz)])
    r = exchange_pieces(rank, fwd, send, total)
    fu = np.empty((N[0], Np1, Nf), dtype=complex)
    off = 0
    for k, w in zip(k0, kz):
        fu[:, :, k:k + w] = np.fft.fft(r[off:off + N[0] * Np1 * w].reshape(N[0], Np1, w), axis=0)
        off += N[0] * Np1 * w
    assert orc.rel_l2(fu, want[rank]) < 1e-13
    b = np.fft.ifft(fu, axis=0)
    send = np.concatenate([b[:, :, k:k + w].ravel() for k, w in zip(k0, kz)])
    r = exchange_pieces(rank, bwd, send, total)
    t = np.empty((Np0, N[1], Nf), dtype=complex)
    off = 0
    for k, w in zip(k0, kz):
        t[:, :, k:k + w] = orc.slab_unpack(r[off:off + N[0] * Np1 * w].reshape(P, Np0, Np1, w))
        off += N[0] * Np1 * w
    back = np.fft.irfft2(t, s=(N[1], N[2]), axes=(1, 2))
    assert orc.rel_l2(back, u) < 1e-13


def slab(rank, P, N, A):
    lay = orc.SlabLayout(N, P)
    want = orc.slab_r2c_forward(orc.scatter_real(A, lay), N)
    Np0, Np1, Nf = int(lay.Np[0]), int(lay.Np[1]), lay.Nf
    u = np.ascontiguousarray(A[lay.real_local_slice(rank)])
    a = np.fft.rfft2(u, axes=(1, 2))
    s = _lib.exchange_schedule(N, P, rank, _lib.SLAB, 0, True)
    assert s["peers"] == list(range(P))
    r = exchange(rank, s, orc.slab_pack(a, P), Np0 * N[1] * Nf * ES)
    fu = np.fft.fft(r.reshape(N[0], Np1, Nf), axis=0)
    assert orc.rel_l2(fu, want[rank]) < 1e-13
    # inverse
    b = np.fft.ifft(fu, axis=0)
    s = _lib.exchange_schedule(N, P, rank, _lib.SLAB, 0, False)
    r = exchange(rank, s, b, Np0 * N[1] * Nf * ES)
    back = np.fft.irfft2(orc.slab_unpack(r.reshape(P, Np0, Np1, Nf)), s=(N[1], N[2]), axes=(1, 2))
    assert orc.rel_l2(back, u) < 1e-13


def pencil(rank, P, N, A, align, P1=None, pipeline=1, relay=False):
    """pipeline != 1: every exchange runs piece by piece with the schedules of mfft_plan_exchange_pieces (X: batches
    of local x rows through both exchanges; Y: batches of local x rows in the z-splitting exchange, batches of the
    rows owned afterwards in the x-chunk exchange), into the same buffers as the un-pipelined exchange."""
    lay = orc.PencilLayout(N, P, P1, align)

    def exchange(rank_, sched, send, recv_bytes, which=None, forward=None):
        if relay and len(sched["peers"]) < P:
            return exchange_relayed(rank_, P, N, dec, which, forward, P1 or 0, send, recv_bytes)
        if pipeline == 1:
            return globals()["exchange"](rank_, sched, send, recv_bytes)
        pieces = _lib.exchange_pieces(N, P, rank, dec, which, forward, pipeline, p1=P1 or 0)
        depth = pipeline if pipeline > 0 else 4
        assert len(pieces) == (min(depth, int(lay.N1[0])) if align == "X" else min(depth, int(lay.N1[0]), int(lay.N2[0])))
        return exchange_pieces(rank_, pieces, send, recv_bytes)
    want = orc.pencil_r2c_forward(orc.scatter_real(A, lay), N, P1, align)
    c0, c1 = lay.ranks(rank)
    m, n, Nf = int(lay.N1[0]), int(lay.N2[1]), lay.Nf
    N1_1, N2_0 = int(lay.N1[1]), int(lay.N2[0])
    dec = _lib.PENCIL_X if align == "X" else _lib.PENCIL_Y
    u = np.ascontiguousarray(A[lay.real_local_slice(rank)])
    a = np.fft.rfft(u, axis=2)
    s0 = _lib.exchange_schedule(N, P, rank, dec, 0, True, p1=P1 or 0)
    assert s0["peers"] == (lay.comm1_members(rank) if align == "X" else lay.comm0_members(rank))
    # z chunks: the pencils' rule (N2 / Pz / 2 columns each, the Nyquist column on the last rank: pencil.py:197, 908); in
    # the FORWARD exchange the rows of a block of 64 columns and more lie a whole number of cache lines apart (plan.hip
    # zrow_pitch: y-aligned plans; the byte counts say how far), the rest of the row is unused
    Pz = len(s0["peers"])
    lens = [N[2] // Pz // 2] * Pz
    lens[-1] += 1
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    pitch = [c // (m * n * ES) for c in s0["scount"]]

    def zpitch(ln):       # plan.hip zrow_pitch: y-aligned: whole lines; x-aligned (round 5): one more line for rows of k * 8 KiB
        if ln < 64:
            return ln
        if align == "Y":
            return -(-ln // 8) * 8
        return ln + 8 if (ln * ES) % 8192 == 0 else ln
    for ln, pt in zip(lens, pitch):
        assert pt == zpitch(ln), (ln, pt)
    bufs = []
    for ln, st, pt in zip(lens, starts, pitch):
        blk = np.zeros((m, n, pt), dtype=complex)
        blk[:, :, :ln] = a[:, :, st:st + ln]
        bufs.append(blk.ravel())
    send = np.concatenate(bufs)
    q = lay.complex_shape(rank)[2]
    qp = s0["rcount"][0] // (m * n * ES)                       # row pitch of the received blocks
    assert qp == zpitch(q) and all(c == m * n * qp * ES for c in s0["rcount"])
    r = exchange(rank, s0, send, sum(s0["rcount"]), 0, True)
    blocks = r.reshape(len(lens), m, n, qp)
    s1 = _lib.exchange_schedule(N, P, rank, dec, 1, True, p1=P1 or 0)
    if align == "X":
        b = np.fft.fft(np.concatenate(list(blocks[..., :q]), axis=1), axis=1)     # (m, N1, q): the y pass reads the pitched rows
        assert s1["peers"] == lay.comm0_members(rank)
        SX = s1["scount"][0] // (m * ES)                      # x-row pitch of the blocks: N1_1 * q (+ a line: plan.hip xplane_pad)
        assert SX in (N1_1 * q, N1_1 * q + 8)
        blk = np.zeros((lay.P1, m, SX), dtype=complex)
        for l in range(lay.P1):
            blk[l, :, :N1_1 * q] = b[:, l * N1_1:(l + 1) * N1_1, :].reshape(m, N1_1 * q)
        r = exchange(rank, s1, blk.ravel(), sum(s1["rcount"]), 1, True)
        fu = np.fft.fft(r.reshape(N[0], SX)[:, :N1_1 * q].reshape(N[0], N1_1, q), axis=0)
    else:
        # the x pass runs in place on (N0, n, qp) -- the unused columns ride along -- and the pitch travels on
        b = np.fft.fft(np.concatenate(list(blocks), axis=0), axis=0)              # (N0, n, qp)
        assert s1["peers"] == lay.comm1_members(rank) and set(s1["scount"]) == {N2_0 * n * qp * ES}
        r = exchange(rank, s1, b, sum(s1["rcount"]), 1, True)
        blocks = r.reshape(lay.P2, N2_0, n, qp)
        fu = np.fft.fft(np.concatenate(list(blocks[..., :q]), axis=1), axis=1)    # (N2_0, N1, q)
    assert fu.shape == lay.complex_shape(rank)
    assert orc.rel_l2(fu, want[rank]) < 1e-13, (align, rank)
    # inverse: mirror
    s1b = _lib.exchange_schedule(N, P, rank, dec, 1, False, p1=P1 or 0)
    s0b = _lib.exchange_schedule(N, P, rank, dec, 0, False, p1=P1 or 0)
    if align == "X":
        b = np.fft.ifft(fu, axis=0)                                                  # x chunks contiguous
        r = exchange(rank, s1b, b, sum(s1b["rcount"]), 1, False)
        blocks = r.reshape(lay.P1, m, N1_1, q)
        b = np.fft.ifft(np.concatenate(list(blocks), axis=1), axis=1)               # (m, N1, q)
        send = np.concatenate([b[:, l * n:(l + 1) * n, :].ravel() for l in range(lay.P2)])
    else:
        b = np.fft.ifft(fu, axis=1)                                                  # (N2_0, N1, q)
        send = np.concatenate([b[:, l * n:(l + 1) * n, :].ravel() for l in range(lay.P2)])
        r = exchange(rank, s1b, send, sum(s1b["rcount"]), 1, False)
        send = np.fft.ifft(r.reshape(N[0], n, q), axis=0)                            # x chunks contiguous
    r = exchange(rank, s0b, send, sum(s0b["rcount"]), 0, False)
    z = np.zeros((m, n, Nf), dtype=complex)
    off = 0
    for ln, st in zip(lens, starts):
        z[:, :, st:st + ln] = r[off:off + m * n * ln].reshape(m, n, ln)
        off += m * n * ln
    back = np.fft.irfft(z, n=N[2], axis=2)
    assert orc.rel_l2(back, u) < 1e-13, (align, rank)


def c2c_plane_padded_exchanges(rank, P):
    """Round 4: complex data on power-of-two meshes.  The x rows of the exchange that feeds the strided x pass would lie
    a multiple of 64 KiB apart; the plan then leaves one cache line (8 complex128) between them INSIDE the exchanged
    chunks (plan.hip xplane_pad) and the schedule says so.  Here: the slab C2C forward exchange (slab.py:759-766), the
    second forward exchange of the x-aligned pencil and the second inverse exchange of the y-aligned one, each executed
    over gloo with the library's byte counts on buffers laid out the way the kernels write / read them; the opposite
    directions (whose x pass reads the caller's array) must stay compact."""
    N = [8, 128, 256]
    rng = np.random.default_rng(2029)
    A = rng.random(N) + 1j * rng.random(N)
    B = np.fft.fftn(A)
    line = 128 // ES
    # ---- slab C2C ----------------------------------------------------------------------------------------------
    Np0, Np1, Nz = N[0] // P, N[1] // P, N[2]
    s = _lib.exchange_schedule(N, P, rank, _lib.SLAB, 0, True, kind=_lib.C2C)
    S = s["scount"][0] // (Np0 * ES)
    want_pad = line if (Np1 * Nz * ES) % 65536 == 0 else 0
    assert S == Np1 * Nz + want_pad and s["sdisp"] == [i * Np0 * S * ES for i in range(P)], (S, s)
    a = np.fft.fft2(A[rank * Np0:(rank + 1) * Np0], axes=(1, 2))                     # (Np0, N1, N2)
    send = np.zeros((P, Np0, S), dtype=complex)
    send[:, :, :Np1 * Nz] = orc.slab_pack(a, P).reshape(P, Np0, Np1 * Nz)            # x rows S apart, the line behind them unused
    r = exchange(rank, s, send, N[0] * S * ES).reshape(N[0], S)
    fu = np.fft.fft(r[:, :Np1 * Nz].reshape(N[0], Np1, Nz), axis=0)
    assert orc.rel_l2(fu, B[:, rank * Np1:(rank + 1) * Np1]) < 1e-13
    sb = _lib.exchange_schedule(N, P, rank, _lib.SLAB, 0, False, kind=_lib.C2C)
    assert set(sb["scount"]) == {Np0 * Np1 * Nz * ES}                                 # inverse: compact
    if P < 4:
        assert want_pad == line
        return want_pad
    # ---- pencils ------------------------------------------------------------------------------------------------
    lay = orc.PencilC2CLayout(N, P, None, "X")
    P1, P2 = lay.P1, lay.P2
    c0, c1 = lay.ranks(rank)
    m, n, N1_1, N2_0 = N[0] // P1, N[1] // P2, N[1] // P1, N[0] // P2
    # x-aligned, forward: after the y transform a rank holds (m, N1, q) with q = N2 / P2, sent as P1 blocks (m, N1_1, q)
    q = N[2] // P2
    G = np.fft.fft(np.fft.fft(A, axis=2), axis=1)                                     # z and y done, globally
    mine = G[c0 * m:(c0 + 1) * m, :, c1 * q:(c1 + 1) * q]
    s1 = _lib.exchange_schedule(N, P, rank, _lib.PENCIL_X, 1, True, kind=_lib.C2C)
    SX = s1["scount"][0] // (m * ES)
    padx = line if (N1_1 * q * ES) % 65536 == 0 else 0
    assert SX == N1_1 * q + padx, (SX, N1_1 * q, padx)
    send = np.zeros((P1, m, SX), dtype=complex)
    for l in range(P1):
        send[l, :, :N1_1 * q] = mine[:, l * N1_1:(l + 1) * N1_1, :].reshape(m, N1_1 * q)
    r = exchange(rank, s1, send, N[0] * SX * ES).reshape(N[0], SX)
    fu = np.fft.fft(r[:, :N1_1 * q].reshape(N[0], N1_1, q), axis=0)
    assert orc.rel_l2(fu, B[lay.complex_local_slice(rank)]) < 1e-13
    assert set(_lib.exchange_schedule(N, P, rank, _lib.PENCIL_X, 1, False, kind=_lib.C2C)["scount"]) == {m * N1_1 * q * ES}
    # y-aligned, inverse: after the inverse y transform a rank holds (N2_0, N1, q), q = N2 / P1, sent as P2 blocks (N2_0, n, q)
    layy = orc.PencilC2CLayout(N, P, None, "Y")
    qy = N[2] // P1
    Hy = np.fft.ifft(B, axis=1)                                                       # inverse y done, globally
    mine = Hy[c1 * N2_0:(c1 + 1) * N2_0, :, c0 * qy:(c0 + 1) * qy]
    s1b = _lib.exchange_schedule(N, P, rank, _lib.PENCIL_Y, 1, False, kind=_lib.C2C)
    SY = s1b["scount"][0] // (N2_0 * ES)
    pady = line if (n * qy * ES) % 65536 == 0 else 0
    assert SY == n * qy + pady, (SY, n * qy, pady)
    send = np.zeros((P2, N2_0, SY), dtype=complex)
    for l in range(P2):
        send[l, :, :n * qy] = mine[:, l * n:(l + 1) * n, :].reshape(N2_0, n * qy)
    r = exchange(rank, s1b, send, N[0] * SY * ES).reshape(N[0], SY)
    got = np.fft.ifft(r[:, :n * qy].reshape(N[0], n, qy), axis=0)                     # inverse x: (N0, n, q) of this rank
    want = np.fft.ifft(np.fft.ifft(B, axis=1), axis=0)[:, c1 * n:(c1 + 1) * n, c0 * qy:(c0 + 1) * qy]
    assert orc.rel_l2(got, want) < 1e-13
    assert set(_lib.exchange_schedule(N, P, rank, _lib.PENCIL_Y, 1, True, kind=_lib.C2C)["scount"]) == {N2_0 * n * qy * ES}
    assert layy.complex_local_slice(rank)[0] == slice(c1 * N2_0, (c1 + 1) * N2_0, 1)
    assert want_pad == padx == pady == line
    # the same exchange of the x-aligned pencil piece by piece (batches of local x rows, mfft_plan_exchange_pieces): the
    # pieces address the PADDED buffers
    pieces = _lib.exchange_pieces(N, P, rank, _lib.PENCIL_X, 1, True, 0, kind=_lib.C2C)
    assert len(pieces) == min(4, m) and sum(sum(pc["scount"]) for pc in pieces) == N[0] * SX * ES
    mine = G[c0 * m:(c0 + 1) * m, :, c1 * q:(c1 + 1) * q]
    send = np.zeros((P1, m, SX), dtype=complex)
    for l in range(P1):
        send[l, :, :N1_1 * q] = mine[:, l * N1_1:(l + 1) * N1_1, :].reshape(m, N1_1 * q)
    r = exchange_pieces(rank, pieces, send, N[0] * SX * ES).reshape(N[0], SX)
    fu = np.fft.fft(r[:, :N1_1 * q].reshape(N[0], N1_1, q), axis=0)
    assert orc.rel_l2(fu, B[lay.complex_local_slice(rank)]) < 1e-13
    return line


def slab_c2c_kz_slices_padded(rank, P):
    """The slab's DEFAULT multi-rank path (4 kz slices) on complex data: every slice's x rows lie N1/P * kz elements apart --
    a power of two here -- so each slice is exchanged with one cache line between its x rows (plan.hip slice_pitch); the
    piece schedules say where."""
    N = [8, 256, 512]
    A = np.random.default_rng(2030).random(N) + 1j * np.random.default_rng(2031).random(N)
    B = np.fft.fftn(A)
    Np0, Np1, Nz = N[0] // P, N[1] // P, N[2]
    fwd = _lib.exchange_pieces(N, P, rank, _lib.SLAB, 0, True, 0, kind=_lib.C2C)
    bwd = _lib.exchange_pieces(N, P, rank, _lib.SLAB, 0, False, 0, kind=_lib.C2C)
    assert len(fwd) == 4 and len(bwd) == 4
    kz = Nz // 4
    pitch = [pc["scount"][0] // (Np0 * ES) for pc in fwd]
    assert all(S == Np1 * kz + 8 for S in pitch), (pitch, Np1 * kz)             # (Np1 * kz * 16) % 65536 == 0 at every P
    assert all(pc["scount"][0] == Np0 * Np1 * kz * ES for pc in bwd)              # the inverse reads the caller's array: compact
    a = np.fft.fft2(A[rank * Np0:(rank + 1) * Np0], axes=(1, 2))
    packed = orc.slab_pack(a, P)                                                  # (P, Np0, Np1, Nz)
    bufs = []
    for s_, S in enumerate(pitch):
        blk = np.zeros((P, Np0, S), dtype=complex)
        blk[:, :, :Np1 * kz] = packed[:, :, :, s_ * kz:(s_ + 1) * kz].reshape(P, Np0, Np1 * kz)
        bufs.append(blk.ravel())
        assert fwd[s_]["sdisp"][0] == sum(P * Np0 * t * ES for t in pitch[:s_])
    total = sum(P * Np0 * S * ES for S in pitch)
    r = exchange_pieces(rank, fwd, np.concatenate(bufs), total)
    fu = np.empty((N[0], Np1, Nz), dtype=complex)
    off = 0
    for s_, S in enumerate(pitch):
        rows = r[off:off + N[0] * S].reshape(N[0], S)[:, :Np1 * kz].reshape(N[0], Np1, kz)
        fu[:, :, s_ * kz:(s_ + 1) * kz] = np.fft.fft(rows, axis=0)
        off += N[0] * S
    assert orc.rel_l2(fu, B[:, rank * Np1:(rank + 1) * Np1]) < 1e-13


def main():
    dist.init_process_group("gloo")
    rank, P = dist.get_rank(), dist.get_world_size()
    N = [16, 32, 64]
    A = np.random.default_rng(2026).random(N)
    slab(rank, P, N, A)
    Nw = [16, 32, 256]                                     # Nf = 129: the default depth really is 4 kz slices
    Aw = np.random.default_rng(2027).random(Nw)
    assert len(_lib.exchange_pieces(Nw, P, rank, _lib.SLAB, 0, True, 0)) == 4
    for pipeline in (0, 2, 3, -2, -3, -8):                 # default kz slices, other depths, row batches
        slab_pipelined(rank, P, N, A, pipeline)
        slab_pipelined(rank, P, Nw, Aw, pipeline)
    if P >= 4:
        for align in ("X", "Y"):
            pencil(rank, P, N, A, align)
        for pipeline in (0, 2, 3):                         # the pencils' exchange pipelines, piece by piece
            pencil(rank, P, N, A, "X", pipeline=pipeline)
            pencil(rank, P, N, A, "Y", pipeline=pipeline)
        if P == 8:
            for align in ("X", "Y"):
                pencil(rank, P, N, A, align, P1=2)
        # chunks of 64 columns and more: the forward z exchange carries line-aligned rows (129 -> 136, 65 -> 72 columns)
        Nq = [16, 32, 512]
        Aq = np.random.default_rng(2032).random(Nq)
        for align in ("X", "Y"):
            for pipeline in (1, 0):
                pencil(rank, P, Nq, Aq, align, pipeline=pipeline)
        # x-aligned, chunks whose rows are 8 KiB (512 complex128 columns): one cache line between the rows (BASELINE config 5's
        # y pass reads rows a power of two apart otherwise); the Nyquist-holding rank's 513 columns stay compact
        Nz = [16, 32, 2048]
        Az = np.random.default_rng(2033).random(Nz)
        for pipeline in (1, 0):
            pencil(rank, P, Nz, Az, "X", pipeline=pipeline)
        # relay striping of the sub-group exchanges (IPC transport): two-hop schedule executed over gloo
        Nr = [32, 64, 128]                                 # messages large enough for 4 KiB stripes
        Ar = np.random.default_rng(2028).random(Nr)
        for align in ("X", "Y"):
            pencil(rank, P, Nr, Ar, align, relay=True)
            if P == 8:
                pencil(rank, P, Nr, Ar, align, P1=2, relay=True)
    pad_seen = c2c_plane_padded_exchanges(rank, P)
    assert pad_seen == 8, pad_seen                     # one 128-byte line of complex128
    slab_c2c_kz_slices_padded(rank, P)
    dist.barrier()
    if rank == 0:
        print("DIST_OK world=%d" % P)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
