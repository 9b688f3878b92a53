// fft_persist_experiment.h -- persistent, register double-buffered form of the strided-axis kernel.
//
// NOT part of the library: measured and rejected (kbench3.hip / membench.hip, profiles/r02_colfft_experiments.md).
// Kept so that the measurement can be repeated.  Findings on MI355X at 1024^3 fp64:
//   * with the next tile's 16 loads issued as one burst the issuing wave sits in the issue stage for 5.8k (x) to
//     8.9k (y) cycles -- a CU accepts only some tens of KB of outstanding misses -- so nothing overlaps;
//   * spread over the passes (this version) the loads do overlap, but a tile still takes 29-33k cycles, exactly what
//     the plain tile-pattern COPY takes (k_tile in membench.hip: 3.4-3.6 ms per 17.2 GB): the strided passes are
//     bound by what the memory system delivers for 128-byte row segments at a large pitch, not by the turnaround
//     of a workgroup.  Static tile ownership adds imbalance (workgroup lifetimes 2.9 .. 4.0 ms in the x pass).
//   -> 3.79 / 3.99 ms (y / x) against 3.52 / 3.57 ms for the per-tile kernel and 3.44 / 3.35 ms for the per-tile
//      kernel at two workgroups per CU (plans.h, MFFT_COLPLANS_F64_B), which is what the library uses.
#pragma once
#include "fft_kernels.h"

namespace mfft {

// ---------------------------------------------------------------------------
// strided-axis c2c, persistent form: one workgroup per CU walks over a sequence of tiles and keeps
// TWO tiles in registers -- while tile t goes through its passes (VALU + LDS only: the twiddles sit
// in LDS), the loads of tile t+1 are in flight, and the stores of tile t drain under the passes of
// tile t+1.  The non-persistent ColFft serialises load -> passes -> store inside every workgroup
// (stamped build, profiles/r02_membench.txt: 40 % / 49 % / 10 % of a workgroup's life at 1024^3).
//
// Addressing: a tile's rows are reached as  uniform base (SGPR pair) + 32-bit per-thread byte offset,
// so the 2*E row addresses cost no VGPRs.  That needs a row map that separates into
// row_off(j + k*TPT) = row_off(k*TPT) + j*lo  (no split, or split % TPT == 0) and per-thread offsets
// below 4 GiB; the launcher falls back to ColFft otherwise (colp_applicable).
// ---------------------------------------------------------------------------
template <typename T>
inline bool colp_map_ok(const RowMap& m, int tpt, int cols) {
  const bool plain = m.split == 0x7FFFFFFFu;
  if (!plain && (m.split % (unsigned)tpt) != 0) return false;
  if (m.lo < 0 || m.hi < 0) return false;
  const unsigned long long span = (unsigned long long)(tpt - 1) * (unsigned long long)m.lo + (unsigned long long)cols;
  return span * sizeof(cx<T>) < 0xFFFFFFFFull;
}

template <class S, typename T, int COLS, bool INV, int VEC = 1>
struct ColFftP {
  static_assert(COLS % VEC == 0, "VEC must divide COLS");
  static_assert(S::NP > 1, "single-pass lengths have nothing to overlap");
  static constexpr int E = S::E;
  static constexpr int CG = COLS / VEC;
  static constexpr int THREADS = S::TPT * CG;
  static constexpr int TW_BYTES = ((int)(S::TW * sizeof(cx<T>)) + 15) / 16 * 16;
  static constexpr int XCH_BYTES = (int)(S::N * COLS * sizeof(cx<T>));
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;
  typedef PackV<cx<T>, VEC> GPack;
  struct Slot {
    int c;
    MFFT_D int operator()(int pos) const { return pos * CG + c; }
  };
  struct Thread {
    int c, j;
    unsigned vin, vout;          // byte offsets of (row j, column c*VEC) inside a tile
  };

  // Tiles are numbered with the full ones first: t < nfull -> (outer, tc) = (t / nfc, t % nfc) with
  // nfc = ncols / COLS full tile columns; the ragged last tile column (ncols % COLS != 0) follows,
  // one tile per outer index.  The main loop only ever sees full tiles, so it is branch-free and
  // hipcc can count its vmcnt waits exactly (with a ragged/full branch inside it, it drained the
  // freshly issued prefetch with vmcnt(0)).
  static MFFT_D void tile_base(const ColParams<T>& P, int outer, int tc, const char*& ib, char*& ob) {
    ib = reinterpret_cast<const char*>(P.in + (i64)outer * P.in_outer + (i64)tc * COLS);
    ob = reinterpret_cast<char*>(P.out + (i64)outer * P.out_outer + (i64)tc * COLS);
  }

  template <bool RAGGED>
  static MFFT_D void load_tile(const ColParams<T>& P, const Thread& th, int outer, int tc, cx<T> (&v)[VEC][E]) {
    const char* ib; char* ob;
    tile_base(P, outer, tc, ib, ob);
    const int nact = RAGGED ? P.ncols - tc * COLS - th.c * VEC : VEC;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const char* row = ib + row_off(P.in_map, (unsigned)(k * S::TPT)) * (i64)sizeof(cx<T>);
      if constexpr (!RAGGED) {
        const GPack g = *reinterpret_cast<const GPack*>(row + th.vin);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i][k] = INV ? swapri(g.e[i]) : g.e[i];
      } else {
        const cx<T>* src = reinterpret_cast<const cx<T>*>(row + th.vin);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          cx<T> x = mk<T>((T)0, (T)0);
          if (i < nact) x = src[i];
          v[i][k] = INV ? swapri(x) : x;
        }
      }
    }
  }

  template <bool RAGGED>
  static MFFT_D void store_tile(const ColParams<T>& P, const Thread& th, int outer, int tc, cx<T> (&v)[VEC][E]) {
    const char* ib; char* ob;
    tile_base(P, outer, tc, ib, ob);
    const int nact = RAGGED ? P.ncols - tc * COLS - th.c * VEC : VEC;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      char* row = ob + row_off(P.out_map, (unsigned)(k * S::TPT)) * (i64)sizeof(cx<T>);
      if constexpr (!RAGGED) {
        GPack g;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const cx<T> x = scale(v[i][k], P.scale);
          g.e[i] = INV ? swapri(x) : x;
        }
        *reinterpret_cast<GPack*>(row + th.vout) = g;
      } else {
        cx<T>* dst = reinterpret_cast<cx<T>*>(row + th.vout);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (i < nact) {
            const cx<T> x = scale(v[i][k], P.scale);
            dst[i] = INV ? swapri(x) : x;
          }
        }
      }
    }
  }

  // The loads of the NEXT tile are issued in small groups at NHOOK points spread over the passes of the
  // current one.  Issued as one burst they do not overlap anything: a CU accepts only some tens of KB of
  // outstanding misses, so a wave that issues 16 KB-sized loads in a row sits in the issue stage until the
  // memory system has served most of them (stamped build: 8.9k cycles to ISSUE 16 loads, during which
  // the wave computes nothing).
  static constexpr int NHOOK = 3 * (S::NP - 1) + 1;
  struct Prefetch {
    const ColParams<T>& P;
    const Thread& th;
    const char* ib;
    cx<T> (&dst)[VEC][E];
    template <int I>
    MFFT_D void at() {
      constexpr int k0 = I * E / NHOOK, k1 = (I + 1) * E / NHOOK;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int k = k0; k < k1; ++k) {
        const char* row = ib + row_off(P.in_map, (unsigned)(k * S::TPT)) * (i64)sizeof(cx<T>);
        const GPack g = *reinterpret_cast<const GPack*>(row + th.vin);
#pragma unroll
        for (int i = 0; i < VEC; ++i) dst[i][k] = INV ? swapri(g.e[i]) : g.e[i];
      }
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
  };
  struct NoPrefetch {
    template <int I> MFFT_D void at() {}
  };

  template <int PASS, class Hook>
  static MFFT_D void passes(cx<T> (&v)[VEC][E], const Thread& th, const cx<T>* ltw, PackV<cx<T>, VEC>* buf, Hook& hook) {
    const Slot slot{th.c};
    hook.template at<3 * PASS>();
#pragma unroll
    for (int i = 0; i < VEC; ++i) pass_compute<S, PASS, T>(v[i], th.j, ltw);
    if constexpr (PASS + 1 < S::NP) {
      hook.template at<3 * PASS + 1>();
      MFFT_BARRIER();                    // everyone is done with the previous gather (also the previous tile's last one)
      pass_scatter<S, PASS>(th.j, [&](int pos, int reg) {
        PackV<cx<T>, VEC> p;
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.e[i] = v[i][reg];
        buf[slot(pos)] = p;
      });
      hook.template at<3 * PASS + 2>();
      MFFT_BARRIER();
      pass_gather<S>(th.j, [&](int pos, int reg) {
        const PackV<cx<T>, VEC> p = buf[slot(pos)];
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i][reg] = p.e[i];
      });
      passes<PASS + 1>(v, th, ltw, buf, hook);
    }
  }

  // this workgroup's share [first, first + step, ...) < end of n items: XCD x (= bid % 8 under round-robin
  // placement, speed only) owns a contiguous range, its workgroups walk through it side by side
  static MFFT_D void share(const ColParams<T>& P, int bid, int n, int& first, int& end, int& step) {
    if (P.remap && (P.nblocks & 7) == 0) {
      const int q = n >> 3, r = n & 7, x = bid & 7;
      const int lo = x * q + (x < r ? x : r);
      first = lo + (bid >> 3);
      end = lo + q + (x < r ? 1 : 0);
      step = P.nblocks >> 3;
    } else {
      first = bid;
      end = n;
      step = P.nblocks;
    }
  }

  static MFFT_D void body(const ColParams<T>& P, int bid, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    char* xbuf = lds + TW_BYTES;
    Thread th;
    th.c = tid % CG;
    th.j = tid / CG;
    th.vin = (unsigned)(((i64)th.j * P.in_map.lo + th.c * VEC) * (i64)sizeof(cx<T>));
    th.vout = (unsigned)(((i64)th.j * P.out_map.lo + th.c * VEC) * (i64)sizeof(cx<T>));
    const int nfc = P.ncols / COLS;                  // full tile columns
    const int nfull = nfc * P.nouter;
    int first, end, step;
    share(P, bid, nfull, first, end, step);
    int t = first;
    cx<T> a[VEC][E], b[VEC][E];
    PackV<cx<T>, VEC>* buf = reinterpret_cast<PackV<cx<T>, VEC>*>(xbuf);
    if (t < end) load_tile<false>(P, th, t / nfc, t % nfc, a);
    stage_twiddles<S, T>(ltw, P.tw, tid, THREADS);
#if defined(__HIP_DEVICE_COMPILE__)
    // Nothing may be pending when the loop is entered: hipcc merges the vmcnt bookkeeping of the loop's two
    // predecessors conservatively, and a pending first tile would make every iteration drain the previous
    // tile's stores before it touches its registers (seen in the ISA as vmcnt(2) instead of vmcnt(18)).
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0)
#endif
    MFFT_BARRIER();
    if (t < end) {
      for (;;) {
        {
          const int tn = t + step;
          const int tl = tn < end ? tn : t;          // past the end: a harmless reload keeps the loop branch-free
          const char* ib; char* ob;
          tile_base(P, tl / nfc, tl % nfc, ib, ob);
          Prefetch pf{P, th, ib, b};
          passes<0>(a, th, ltw, buf, pf);
          store_tile<false>(P, th, t / nfc, t % nfc, a);
          if (tn >= end) break;
          t = tn;
        }
        {
          const int tn = t + step;
          const int tl = tn < end ? tn : t;
          const char* ib; char* ob;
          tile_base(P, tl / nfc, tl % nfc, ib, ob);
          Prefetch pf{P, th, ib, a};
          passes<0>(b, th, ltw, buf, pf);
          store_tile<false>(P, th, t / nfc, t % nfc, b);
          if (tn >= end) break;
          t = tn;
        }
      }
    }
    if (P.ncols % COLS != 0) {                       // the ragged tile column, one tile per outer index
      share(P, bid, P.nouter, first, end, step);
      for (int o = first; o < end; o += step) {
        load_tile<true>(P, th, o, nfc, a);
        NoPrefetch np;
        passes<0>(a, th, ltw, buf, np);
        store_tile<true>(P, th, o, nfc, a);
      }
    }
  }
};

}  // namespace mfft
