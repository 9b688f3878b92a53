// fft_col3.h -- strided-axis transform of length N = 3 L as THREE length-L sub-transforms inside one workgroup (round 4).
//
// Why: every ColFft plan holds E = lcm(radices) values per thread and exchanges the WHOLE N-row tile through LDS between
// passes.  For N = 1536 / 3072 (the 3/2-rule images of 1024 / 2048: slab.py:310-344, 445-483 run them 4 times per
// dealiased pair) that is 24 values per thread and a 96 - 196 KB exchange buffer: ONE 512-thread workgroup per CU, 0.50 - 0.54
// of the HBM roofline where the 1024 kernel (two workgroups per CU) reaches 0.63.  Splitting N = 3 L by one radix-3 step
// leaves three INDEPENDENT length-L transforms; run one after the other they need an exchange buffer of L rows only
// (1536: 64 KB instead of 196 KB), so two workgroups fit a CU with the same 24 values per thread.
//
//   DIF (plain transforms, and the inverse with a zero band on input, PAD == 1):
//     X[3 q + r] = sum_{k1 < L} W_L^{q k1} { W_N^{r k1} sum_{k2 < 3} x[k1 + L k2] W_3^{r k2} }
//     a radix-3 butterfly across the three thirds of the input (rows p, p + L, p + 2L), the twiddles W_N^{r p}, then the
//     length-L transform of third r, whose natural-order result q is row 3 q + r of the output.  PAD == 1: the middle
//     third of the (logical) input is the zero band of the 3/2-rule spectrum and is never loaded.
//   DIT (the forward transform with a truncated output, PAD == 2):
//     X[k1 + L k2] = sum_r W_3^{r k2} W_N^{r k1} FFT_L(x[3 q + r])[k1]
//     the sub-transform of the rows 3 q + r, the twiddles, then the radix-3 butterfly across r; the middle third of the
//     output (k2 = 1) is the truncated band and is never stored (its Nyquist row L is folded into row 2 L: P.fold).
//   Inverse transforms run the same code on (im, re) (the swap identity of fft_core.h).
//
// Thread j (of TPT = L / E_L per column group) holds, for every third r, the E_L positions j + k TPT of that third -- in
// both forms the positions a sub-transform reads and the positions it leaves its natural-order result at -- so the
// addressing, the tiles, the two-level row maps (fused pack / unpack), the XCD remap and the NT variants are ColFft's.
// Twiddle table of an entry: [S_L::TW inter-pass twiddles of the sub-plan][W_N^p, p < L][W_N^{2p}, p < L].
#pragma once
#include <vector>
#include "fft_kernels.h"
#include "twiddle.h"

namespace mfft {

template <class SL, typename T, int COLS, bool INV, bool TWLDS, int SPLIT = 0, int VEC = 1, bool NT = false, int PAD = 0>
struct ColFft3 {
  static_assert(COLS % VEC == 0, "VEC must divide COLS");
  static_assert(PAD == 0 || (PAD == 1 && INV) || (PAD == 2 && !INV), "pad-on-load is an inverse mode, truncate-on-store a forward one");
  static_assert(SPLIT == 0 || SPLIT == 1, "whole-complex or split re / im exchange");
  static_assert(SL::NP > 1, "the barriers of the sub-transforms' exchanges are what makes the kernel safe in place");
  static constexpr int L = SL::N, N = 3 * SL::N, EL = SL::E;
  static constexpr int CG = COLS / VEC;
  static constexpr int THREADS = SL::TPT * CG;
  static constexpr int TW_BYTES = TWLDS ? (int)(SL::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = SL::NP > 1 ? (int)(L * COLS * (SPLIT ? sizeof(T) : sizeof(cx<T>))) : 0;
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;
  static constexpr int TW = SL::TW + 2 * L;                     // entries of the entry's twiddle table

  typedef ColFft<SL, T, COLS, INV, TWLDS, SPLIT, VEC, NT, 0> Base;       // its global load / store of VEC columns
  typedef typename Base::GPack GPack;
  typedef typename Base::Slot Slot;

  // load VEC columns of one row (zeros beyond the array), swapped for inverse transforms.  FULL: all VEC columns of this
  // thread exist -- one unconditional 16-byte load.  The callers branch ONCE per thread on that, around their whole load
  // phase: with the test inside (once per row) hipcc waits for every row's load inside its branch, i.e. the rows of a tile
  // come in one after the other (single precision, VEC = 2: 12 of 24 loads serialised, found in round 4 in the ISA).
  // (wrapped input columns, ColParams::in_wrap: a lane whose VEC columns straddle the end of an input row takes the ragged
  // path with `first` = its columns before the wrap and `gap` = the distance the others lie further)
  template <bool FULL>
  static MFFT_D void load_row(const cx<T>* src, int nact, cx<T> (&dst)[VEC], int first = VEC, int gap = 0) {
    if constexpr (FULL) {
      const GPack g = Base::load_pack(src);
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = INV ? swapri(g.e[i]) : g.e[i];
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        cx<T> x = mk<T>((T)0, (T)0);
        if (i < nact) x = src[i + (i >= first ? gap : 0)];
        dst[i] = INV ? swapri(x) : x;
      }
    }
  }
  struct FullLanes { static constexpr bool value = true; };
  struct RaggedLanes { static constexpr bool value = false; };
  static MFFT_D void store_row(cx<T>* dst, int nact, const cx<T> (&val)[VEC], T s) {
    if (nact >= VEC) {
      GPack g;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const cx<T> x = scale(val[i], s);
        g.e[i] = INV ? swapri(x) : x;
      }
      Base::store_pack(dst, g);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i)
        if (i < nact) {
          const cx<T> x = scale(val[i], s);
          dst[i] = INV ? swapri(x) : x;
        }
    }
  }

  template <class TwPtr>
  static MFFT_D void sub_transform(cx<T> (&w)[VEC][EL], int j, TwPtr tw, char* xbuf, int c) {
    if constexpr (SPLIT) {
      XchSplitV<T, VEC, Slot> xch{reinterpret_cast<PackV<T, VEC>*>(xbuf), Slot{c}};
      run_passes_v<SL, 0, T, VEC, TwPtr, XchSplitV<T, VEC, Slot>, true>(w, j, tw, xch);
    } else {
      XchFullV<T, VEC, Slot> xch{reinterpret_cast<PackV<cx<T>, VEC>*>(xbuf), Slot{c}};
      run_passes_v<SL, 0, T, VEC, TwPtr, XchFullV<T, VEC, Slot>, true>(w, j, tw, xch);
    }
  }

  // third R of the DIF form: its sub-transform, then its rows 3 q + R of the output (R a template parameter: the register
  // array is only ever indexed with compile-time constants)
  template <int R>
  static MFFT_D void dif_third(const ColParams<T>& P, cx<T> (&w)[3][VEC][EL], int j, int c, int nact, cx<T>* op,
                               const cx<T>* ltw, char* xbuf) {
    if constexpr (TWLDS) sub_transform(w[R], j, ltw, xbuf, c);
    else sub_transform(w[R], j, P.tw, xbuf, c);
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const unsigned row = 3u * (unsigned)(j + k * SL::TPT) + (unsigned)R;
      cx<T> val[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) val[i] = w[R][i][k];
      store_row(op + row_off(P.out_map, row), nact, val, P.scale);
    }
  }
  // third R of the DIT form: its rows 3 q + R of the input, then its sub-transform
  template <int R>
  static MFFT_D void dit_third(const ColParams<T>& P, cx<T> (&w)[3][VEC][EL], int j, int c, int nact, const cx<T>* ip,
                               const cx<T>* ltw, char* xbuf, int first) {
    auto loads = [&](auto lanes) {
#pragma unroll
      for (int k = 0; k < EL; ++k) {
        const unsigned row = 3u * (unsigned)(j + k * SL::TPT) + (unsigned)R;
        cx<T> x[VEC];
        load_row<decltype(lanes)::value>(ip + row_off(P.in_map, row), nact, x, first, P.in_wrap_gap);
#pragma unroll
        for (int i = 0; i < VEC; ++i) w[R][i][k] = x[i];
      }
    };
    if (nact >= VEC && first == VEC) loads(FullLanes{});
    else loads(RaggedLanes{});
    if constexpr (TWLDS) sub_transform(w[R], j, ltw, xbuf, c);
    else sub_transform(w[R], j, P.tw, xbuf, c);
  }

  static MFFT_D void body(const ColParams<T>& P, int bid_raw, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    const int bid = P.remap == 2 ? xcd_remap_skew(bid_raw, P.ntile_c * P.nouter)
                  : P.remap    ? xcd_remap(bid_raw, P.ntile_c * P.nouter) : bid_raw;
    const int outer = bid / P.ntile_c;
    const int tc = bid - outer * P.ntile_c;
    const int c = tid % CG;
    const int j = tid / CG;
    const int col = tc * COLS + c * VEC;
    const int nact = P.ncols - col;
    i64 icol = col;
    int first = VEC;                                 // wrapped input columns (ColParams::in_wrap), as in ColFft
    if (P.in_wrap > 0) {
      const int q = col / P.in_wrap;
      icol += (i64)q * P.in_wrap_gap;
      if (VEC > 1 && col - q * P.in_wrap + VEC > P.in_wrap) first = P.in_wrap - (col - q * P.in_wrap);
    }
    const cx<T>* ip = P.in + (i64)outer * P.in_outer + icol;
    cx<T>* op = P.out + (i64)outer * P.out_outer + col;
    const cx<T>* tw1 = P.tw + SL::TW;               // W_N^p
    const cx<T>* tw2 = tw1 + L;                      // W_N^{2p}
    char* xbuf = lds + TW_BYTES;
    if constexpr (TWLDS && SL::NP > 1) stage_twiddles<SL, T>(ltw, P.tw, tid, THREADS);      // (the first exchange's barrier publishes it)

    cx<T> w[3][VEC][EL];
    if constexpr (PAD != 2) {
      // ---- DIF: rows p, p + L, p + 2L -> radix-3 butterfly -> twiddles -> three sub-transforms -> rows 3 q + r --------
      // all loads first, straight into the registers the butterflies work on (as ColFft does: no second copy of the tile)
      auto loads = [&](auto lanes) {
        constexpr bool FULL = decltype(lanes)::value;
#pragma unroll
        for (int k = 0; k < EL; ++k) {
          const unsigned p = (unsigned)(j + k * SL::TPT);
          cx<T> x[VEC];
          load_row<FULL>(ip + row_off(P.in_map, p), nact, x, first, P.in_wrap_gap);
#pragma unroll
          for (int i = 0; i < VEC; ++i) w[0][i][k] = x[i];
          if constexpr (PAD != 1) {                   // PAD == 1: logical rows [L, 2L) are the zero band, never loaded
            load_row<FULL>(ip + row_off(P.in_map, p + (unsigned)L), nact, x, first, P.in_wrap_gap);
#pragma unroll
            for (int i = 0; i < VEC; ++i) w[1][i][k] = x[i];
          }
          // (PAD == 1: logical row 2L + p is physical row L + p)
          load_row<FULL>(ip + row_off(P.in_map, p + (PAD == 1 ? 1u : 2u) * (unsigned)L), nact, x, first, P.in_wrap_gap);
#pragma unroll
          for (int i = 0; i < VEC; ++i) w[2][i][k] = x[i];
        }
      };
      if (nact >= VEC && first == VEC) loads(FullLanes{});
      else loads(RaggedLanes{});
#pragma unroll
      for (int k = 0; k < EL; ++k) {
        const unsigned p = (unsigned)(j + k * SL::TPT);
        const cx<T> t1 = tw1[p], t2 = tw2[p];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          cx<T> b[3] = {w[0][i][k], PAD == 1 ? mk<T>((T)0, (T)0) : w[1][i][k], w[2][i][k]};
          Bfly<3>::run(b);
          w[0][i][k] = b[0];
          w[1][i][k] = b[1] * t1;
          w[2][i][k] = b[2] * t2;
        }
      }
      // (every sub-transform synchronises before its first scatter: that barrier also publishes the staged twiddles, which
      // pass 0 does not read, and makes the stores below safe in place: all loads of the workgroup are behind it)
      dif_third<0>(P, w, j, c, nact, op, ltw, xbuf);
      dif_third<1>(P, w, j, c, nact, op, ltw, xbuf);
      dif_third<2>(P, w, j, c, nact, op, ltw, xbuf);
    } else {
      // ---- DIT: rows 3 q + r -> three sub-transforms -> twiddles -> radix-3 butterfly -> rows k1 and (2L + k1 -> L + k1) ---
      dit_third<0>(P, w, j, c, nact, ip, ltw, xbuf, first);
      dit_third<1>(P, w, j, c, nact, ip, ltw, xbuf, first);
      dit_third<2>(P, w, j, c, nact, ip, ltw, xbuf, first);
#pragma unroll
      for (int k = 0; k < EL; ++k) {
        const unsigned p = (unsigned)(j + k * SL::TPT);
        const cx<T> t1 = tw1[p], t2 = tw2[p];
        cx<T> lo[VEC], hi[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          cx<T> b[3] = {w[0][i][k], w[1][i][k] * t1, w[2][i][k] * t2};
          Bfly<3>::run(b);
          lo[i] = b[0];
          hi[i] = b[2];
          if (k == 0 && j == 0 && P.fold) hi[i] = hi[i] + b[1];      // the Nyquist row L folded into row 2L (slab.py:529-533)
        }
        store_row(op + row_off(P.out_map, p), nact, lo, P.scale);
        store_row(op + row_off(P.out_map, p + (unsigned)L), nact, hi, P.scale);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// ColFft3S: the pad-on-load inverse (PAD == 1) with its three sub-transforms in THREE WORKGROUPS (round 4).
//
// The zero band makes the radix-3 step of the DIF form a two-term sum, X[3 q + r] = FFT_L{ (x[p] + w3^(2r) x[2L + p]) W_N^(r p) }[q],
// so third r needs nothing of the other two.  One workgroup per (tile, r): it reads the 2 L physical rows of its tile --
// the three workgroups of a tile are neighbours in the XCD-aware order, so two of the three reads are L2 hits --, holds
// E_L values per thread instead of 3 E_L (1536 in double precision: 4 instead of the 24 of the ColFft plan), exchanges L
// rows through LDS, and runs at the occupancy of the length-L kernel (two 1024-thread workgroups per CU at L = 512)
// where the ColFft plan of 1536 -- and ColFft3 with its 12 values per thread -- stay at one.  Output rows 3 q + r: every
// workgroup writes its own rows, so the kernel is for OUT-OF-PLACE passes only (which the pad-on-load passes are: their
// output has more rows than their input).  Launch: 3 x the tiles (KernelEntry::grid_mult).  Same twiddle table as ColFft3.
template <class SL, typename T, int COLS, bool INV, bool TWLDS, int SPLIT = 0, int VEC = 1, bool NT = false, int PAD = 1>
struct ColFft3S {
  static_assert(PAD == 1 && INV, "one third per workgroup: the pad-on-load inverse");
  typedef ColFft3<SL, T, COLS, INV, TWLDS, SPLIT, VEC, NT, PAD> Sib;      // its row load / store and its sub-transform
  static constexpr int L = SL::N, N = 3 * SL::N, EL = SL::E;
  static constexpr int CG = COLS / VEC;
  static constexpr int THREADS = Sib::THREADS;
  static constexpr int LDS_BYTES = Sib::LDS_BYTES;
  static constexpr int TW_BYTES = Sib::TW_BYTES;
  static constexpr int TW = Sib::TW;
  static constexpr int GRID_MULT = 3;

  static MFFT_D void body(const ColParams<T>& P, int bid_raw, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    const int nt3 = 3 * P.ntile_c * P.nouter;
    const int v = P.remap ? xcd_remap(bid_raw, nt3) : bid_raw;      // consecutive v on one XCD: the three thirds of a tile
    const int bid = v / 3;
    const int r = v - 3 * bid;
    const int outer = bid / P.ntile_c;
    const int tc = bid - outer * P.ntile_c;
    const int c = tid % CG;
    const int j = tid / CG;
    const int col = tc * COLS + c * VEC;
    const int nact = P.ncols - col;
    const cx<T>* ip = P.in + (i64)outer * P.in_outer + col;
    cx<T>* op = P.out + (i64)outer * P.out_outer + col;
    const cx<T>* twr = P.tw + SL::TW + (r == 2 ? L : 0);          // W_N^(r p), r = 1, 2
    char* xbuf = lds + TW_BYTES;
    if constexpr (TWLDS && SL::NP > 1) stage_twiddles<SL, T>(ltw, P.tw, tid, THREADS);      // (the first exchange's barrier publishes it)
    // w3^(2r) with w3 = exp(-2 pi i / 3): 1, (-1/2, +sqrt(3)/2), (-1/2, -sqrt(3)/2)
    const T h = (T)0.86602540378443864676372317075294L;
    const cx<T> coef = r == 0 ? mk<T>((T)1, (T)0) : mk<T>((T)-0.5, r == 1 ? h : -h);

    cx<T> w[VEC][EL];
    auto loads = [&](auto lanes) {
      constexpr bool FULL = decltype(lanes)::value;
      cx<T> a[EL][VEC], b[EL][VEC];
#pragma unroll
      for (int k = 0; k < EL; ++k) {
        const unsigned p = (unsigned)(j + k * SL::TPT);
        Sib::template load_row<FULL>(ip + row_off(P.in_map, p), nact, a[k]);
        Sib::template load_row<FULL>(ip + row_off(P.in_map, p + (unsigned)L), nact, b[k]);      // logical row 2L + p is physical row L + p
      }
#pragma unroll
      for (int k = 0; k < EL; ++k) {
        const unsigned p = (unsigned)(j + k * SL::TPT);
        const cx<T> t = keep_bits(twr[p], r != 0) + mk<T>(r == 0 ? (T)1 : (T)0, (T)0);           // W_N^(r p); 1 for r = 0
#pragma unroll
        for (int i = 0; i < VEC; ++i) w[i][k] = (a[k][i] + coef * b[k][i]) * t;
      }
    };
    if (nact >= VEC) loads(typename Sib::FullLanes{});
    else loads(typename Sib::RaggedLanes{});
    if constexpr (TWLDS) Sib::sub_transform(w, j, (const cx<T>*)ltw, xbuf, c);
    else Sib::sub_transform(w, j, P.tw, xbuf, c);
#pragma unroll
    for (int k = 0; k < EL; ++k) {
      const unsigned row = 3u * (unsigned)(j + k * SL::TPT) + (unsigned)r;
      cx<T> val[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) val[i] = w[i][k];
      Sib::store_row(op + row_off(P.out_map, row), nact, val, P.scale);
    }
  }
};

// twiddle table of a ColFft3 entry
template <class SL, typename T>
std::vector<cx<T>> build_col3_twiddles() {
  std::vector<cx<T>> tw = build_pass_twiddles<SL, T>();
  tw.resize((size_t)SL::TW);
  const int L = SL::N, N = 3 * SL::N;
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int m = 1; m <= 2; ++m)
    for (int p = 0; p < L; ++p) {
      const long long num = ((long long)m * p) % N;
      const long double a = two_pi * (long double)num / (long double)N;
      tw.push_back(mk<T>((T)cosl(a), (T)(-sinl(a))));
    }
  return tw;
}

}  // namespace mfft
