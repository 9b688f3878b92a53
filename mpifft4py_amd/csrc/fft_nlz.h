// fft_nlz.h -- the fused NONLINEAR z stage of a pseudo-spectral step (round 6).
//
// What the reference's demo does per Runge-Kutta stage (demo/spectral_dns_solver.py:53-71):
//     for i in 0..2: U[i]    = ifftn(U_hat[i])          three inverse transforms
//     for i in 0..2: curl[i] = ifftn(i K x U_hat)       three more
//     U x curl                                          in real space
//     for i in 0..2: dU[i]   = fftn((U x curl)[i])      three forward transforms
// In a slab / pencil transform the z axis is the last inverse stage and the first forward stage and it is local to a
// rank (slab.py:214-346, 349-485: irfft / rfft along axis 2), so the nine real work arrays exist only between the two
// z stages.  This kernel takes one (x, y) row of the SIX half-spectra (x and y already transformed back), produces the six
// real rows in registers, forms the cross product there and transforms the three result rows forward again: the real
// arrays -- at 1024^3 with the 3/2-rule 9 x 1536^3 x 8 B = 261 GB -- never exist, and the z stages move 9 complex rows per
// (x, y) instead of 9 complex + 27 real ones.
//
// Arithmetic.  Two real transforms ride on ONE complex transform of the full length M (no split pre- / post-pass with its
// twiddles): with A, B the Hermitian extensions of two half-spectra, the inverse DFT of Z = A + iB is a + ib; forward, the
// DFT of ra + i rb splits into Ra[k] = (Z[k] + conj Z[M-k]) / 2, Rb[k] = -i (Z[k] - conj Z[M-k]) / 2.  Per row: three
// inverse transforms (a_f + i b_f, f = 0..2) and one and a half forward ones -- r0 + i r1 of the row, and r2 of TWO
// consecutive rows together, which is why a thread group works through its rows in pairs.  4.5 transforms of length M per
// row, the minimum for nine real ones.
//
// Registers.  Thread j of a row holds positions j + k TPT (fft_core.h), the same for every field, so the cross product
// needs no exchange at all: two of the three inverse results are parked in 2 E complex registers while the third is
// computed, r2 of the first row of a pair in E more reals.  S is the COMPLEX plan of length M; `valid` (runtime) is the
// number of bins a row holds in memory (M/2 + 1, or N/2 + 1 of the un-padded mesh for the 3/2-rule: the rest reads as zero
// and is not stored -- what C2RFft / R2CFft LIMIT do).
#pragma once
#include <type_traits>
#include "fft_kernels.h"

namespace mfft {

template <typename T>
struct NlzParams {
  const cx<T>* a[3];           // half-spectra rows of the first vector field (x and y already in real space)
  const cx<T>* b[3];           // ... of the second
  cx<T>* out[3];               // half-spectra rows of (a x b); may alias a[] / b[] row for row
  const cx<T>* tw;             // inter-pass twiddles of S
  i64 in_stride, out_stride;   // complex elements between consecutive rows
  i64 nrows;
  int valid;                   // bins per OUTPUT row that are stored (and exist in memory)
  int valid_in;                // bins per INPUT row that exist (<= valid: the pruned 2/3-rule reads the kept kz only)
  T scale;                     // applied to a x b (both inverse transforms are un-normalised: 1 / M^2 gives numpy's irfft)
  const cx<T>* rt3;            // Nlz3Fft: exp(+2 pi i k / M), k = 0..L, then exp(+2 pi i 2k / M), k = 0..L   (M = 3 L)
};

// WAVE: a row's threads sit inside ONE wave (TPT divides 64), so the exchanges of its transforms need no workgroup barrier at
// all: LDS operations of a wave execute in issue order, what a lane wrote is there for every lane's later read, and the only
// thing to stop is the compiler moving a read above a write (a wave barrier: no instruction).  The waves of a workgroup then
// drift apart -- one loads while another computes -- instead of meeting 60 times per pair of rows (SQ_WAIT_ANY was 55 % of
// the wave cycles with workgroup barriers, profiles/r06_nlz_pmc_counters.txt).
#if defined(__HIP_DEVICE_COMPILE__)
#define MFFT_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define MFFT_WAVE_SYNC() MFFT_BARRIER()
#endif
template <bool WAVE> MFFT_D void nlz_sync() {
  if constexpr (WAVE) MFFT_WAVE_SYNC();
  else MFFT_BARRIER();
}
template <typename T, class Slot, bool WAVE>
struct XchFullW {
  cx<T>* buf;
  Slot slot;
  template <class S, int P>
  MFFT_D void exchange(cx<T> (&v)[S::E], int j, bool pre_barrier) {
    if (pre_barrier) nlz_sync<WAVE>();
    pass_scatter<S, P>(j, [&](int pos, int reg) { buf[slot(pos)] = v[reg]; });
    nlz_sync<WAVE>();
    pass_gather<S>(j, [&](int pos, int reg) { v[reg] = buf[slot(pos)]; });
  }
};

template <class S, typename T, int ROWS, bool TWLDS, bool SPLIT, bool WAVE = false>
struct NlzFft {
  static_assert(!WAVE || (!SPLIT && S::TPT <= 64 && 64 % S::TPT == 0), "wave-synchronous rows: whole rows inside a wave, whole-complex exchange");
  typedef typename RowXch<SPLIT, T, PadSlot<S::R(0)>>::elem XE;
  static constexpr int M = S::N;
  static constexpr int E = S::E;
  static constexpr int THREADS = S::TPT * ROWS;
  static constexpr int PD = S::R(0);       // (pad periods of 16 / 24 / 32 elements: no gain, 16 loses 50 % at 512: profiles/r06_nlz_variants.txt)
  static constexpr int PLEN = padded_len<M, PD>();
  static constexpr int TW_BYTES = (TWLDS && S::NP > 1) ? (int)(S::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = (int)(PLEN * ROWS * sizeof(XE));      // also the mirror exchange of the forward split
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;
  typedef typename std::conditional<WAVE, XchFullW<T, PadSlot<PD>, true>, typename RowXch<SPLIT, T, PadSlot<PD>>::type>::type Xch;

  // Z = A + iB at the positions of thread j, ready for the inverse passes (swap identity)
  static MFFT_D void load_pair(cx<T> (&v)[E], const cx<T>* ra, const cx<T>* rb, int j, int valid) {
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int p = j + k * S::TPT;
      const bool mir = p > M / 2;                  // upper half: conj of bin M - p
      const int q = mir ? M - p : p;
      const bool ok = q < valid;
      // unconditional loads of a clamped position (fft_core.h keep_bits): all 2 E of them in flight together
      cx<T> xa = keep_bits(ra[ok ? q : 0], ok), xb = keep_bits(rb[ok ? q : 0], ok);
      if (q == 0 || (M % 2 == 0 && q == M / 2)) {  // imaginary parts of the k = 0 and k = M/2 bins are ignored (as c2r does)
        xa.y = (T)0;
        xb.y = (T)0;
      }
      if (mir) { xa.y = -xa.y; xb.y = -xb.y; }
      v[k] = mk<T>(xa.y + xb.x, xa.x - xb.y);      // swapri(A + iB)
    }
  }

  template <class TwPtr>
  static MFFT_D void inverse_pair(cx<T> (&v)[E], const cx<T>* ra, const cx<T>* rb, int j, int valid, TwPtr tw, Xch& xc) {
    // (Hiding j here and in forward_pair, as Nlz3Fft does, makes the address arithmetic local to each of the five transforms:
    // 216 -> 160 VGPRs for the 8-values plans -- and 5 % SLOWER, 512: 0.555 -> 0.583 ms, 1024: 0.879 -> 0.923; the 12-values plans
    // spill more, not less, 768: 1.36 -> 1.66 ms: profiles/r06_nlz_variants.txt.  Not done.)
    load_pair(v, ra, rb, j, valid);
    nlz_sync<WAVE>();
    run_passes<S, 0, T>(v, j, tw, xc);
  }

  // forward transform of v = ra + i rb and its split into the two half-spectra, bins [0, valid) stored
  template <class TwPtr>
  static MFFT_D void forward_pair(cx<T> (&v)[E], int j, TwPtr tw, Xch& xc, XE* xb, cx<T>* oa, cx<T>* ob, bool sa, bool sb,
                                  int valid) {
    nlz_sync<WAVE>();                                // the buffer is free: everybody has left the previous exchange
    run_passes<S, 0, T>(v, j, tw, xc);
    constexpr int KMAX = (M / 2) / S::TPT;         // registers beyond it hold positions > M/2 only: nothing to store
    const T half = (T)0.5;
    auto emit = [&](int p, cx<T> z, cx<T> m) {
      const cx<T> zm = conj(m);
      if (p < valid) {
        if (sa) oa[p] = scale(z + zm, half);
        if (sb) ob[p] = mul_mi(scale(z - zm, half));
      }
    };
    if constexpr (SPLIT) {
      T mx[KMAX + 1];
      if constexpr (S::NP > 1) nlz_sync<WAVE>();
#pragma unroll
      for (int k = 0; k < E; ++k) xb[padpos<PD>(j + k * S::TPT)] = v[k].x;
      nlz_sync<WAVE>();
#pragma unroll
      for (int k = 0; k <= KMAX; ++k) {
        const int p = j + k * S::TPT;
        mx[k] = xb[padpos<PD>(p == 0 ? 0 : M - p)];
      }
      nlz_sync<WAVE>();
#pragma unroll
      for (int k = 0; k < E; ++k) xb[padpos<PD>(j + k * S::TPT)] = v[k].y;
      nlz_sync<WAVE>();
#pragma unroll
      for (int k = 0; k <= KMAX; ++k) {
        const int p = j + k * S::TPT;
        emit(p, v[k], mk<T>(mx[k], xb[padpos<PD>(p == 0 ? 0 : M - p)]));
      }
    } else {
      if constexpr (S::NP > 1) nlz_sync<WAVE>();
#pragma unroll
      for (int k = 0; k < E; ++k) xb[padpos<PD>(j + k * S::TPT)] = v[k];
      nlz_sync<WAVE>();
#pragma unroll
      for (int k = 0; k <= KMAX; ++k) {
        const int p = j + k * S::TPT;
        emit(p, v[k], xb[padpos<PD>(p == 0 ? 0 : M - p)]);
      }
    }
  }

  static MFFT_D void body(const NlzParams<T>& P, int bid, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    const int rl = tid / S::TPT;
    const int j = row_thread_index<S>(tid);
    XE* xb = reinterpret_cast<XE*>(lds + TW_BYTES) + rl * PLEN;
    if constexpr (TWLDS && S::NP > 1) {
      stage_twiddles<S, T>(ltw, P.tw, tid, THREADS);
      if constexpr (WAVE) MFFT_BARRIER();          // the table is shared by the workgroup's waves: the one real barrier
    }                                              // (otherwise the first barrier below covers it)
    const cx<T>* tw = (TWLDS && S::NP > 1) ? (const cx<T>*)ltw : P.tw;
    Xch xc{xb, PadSlot<PD>{}};
    const i64 unit = (i64)bid * ROWS + rl;         // a pair of rows
    T r2a[E];
#pragma unroll
    for (int k = 0; k < E; ++k) r2a[k] = (T)0;
    bool sa = false;
    i64 rowa = 0;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const i64 row = 2 * unit + h;
      const bool active = row < P.nrows;           // rows past the end re-read the last row and store nothing
      const i64 lrow = active ? row : P.nrows - 1;
      const i64 io = lrow * P.in_stride;
      cx<T> pk[2][E];
      cx<T> v[E];
      inverse_pair(v, P.a[0] + io, P.b[0] + io, j, P.valid_in, tw, xc);
#pragma unroll
      for (int k = 0; k < E; ++k) pk[0][k] = v[k];
      inverse_pair(v, P.a[1] + io, P.b[1] + io, j, P.valid_in, tw, xc);
#pragma unroll
      for (int k = 0; k < E; ++k) pk[1][k] = v[k];
      inverse_pair(v, P.a[2] + io, P.b[2] + io, j, P.valid_in, tw, xc);
      // the results are swapped (inverse through the swap identity): .y = a_f, .x = b_f at position j + k TPT
      T r2[E];
#pragma unroll
      for (int k = 0; k < E; ++k) {
        const T a0 = pk[0][k].y, a1 = pk[1][k].y, a2 = v[k].y;
        const T b0 = pk[0][k].x, b1 = pk[1][k].x, b2 = v[k].x;
        v[k] = mk<T>((a1 * b2 - a2 * b1) * P.scale, (a2 * b0 - a0 * b2) * P.scale);
        r2[k] = (a0 * b1 - a1 * b0) * P.scale;
      }
      const i64 oo = row * P.out_stride;
      forward_pair(v, j, tw, xc, xb, P.out[0] + oo, P.out[1] + oo, active, active, P.valid);
      if (h == 0) {
#pragma unroll
        for (int k = 0; k < E; ++k) r2a[k] = r2[k];
        sa = active;
        rowa = row;
      } else {
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = mk<T>(r2a[k], r2[k]);
        forward_pair(v, j, tw, xc, xb, P.out[2] + rowa * P.out_stride, P.out[2] + oo, sa, active, P.valid);
      }
    }
  }
};

// ---------------------------------------------------------------------------
// The same stage for the 3/2-rule, M = 3 L with the L + 1 bins of the un-padded mesh per row (N = 2 L): PRUNED transforms.
// With n = 3 m + s the inverse transform of a spectrum that is zero outside |k| <= L falls apart into three transforms of
// length L,
//     z[3m + s] = sum_{kappa < L} e^{2 pi i kappa m / L} Y_s[kappa],   Y_s[kappa] = W^{kappa s} (Z[kappa] + w^{-s} Z[kappa - L])
// (W = e^{2 pi i / M}, w = W^L = e^{2 pi i / 3}; kappa = 0 also takes Z[L] w^s), and forward the bins 0..L are
//     Z'[k] = sum_s conj(W^{sk}) F_s[k mod L],   F_s = DFT_L of r[3m + s].
// The radix-3 pass over a spectrum that is one third zeros is never run, and -- what decides -- the three sub-transforms go
// to three thread groups of SL::TPT threads with SL::E values each: a row of 768 points is 192 threads with 4 - 8 values
// instead of 64 threads with 12, the parked fields (2 E complex + E real per thread) shrink with it and the kernel runs at
// 3 - 5 waves per SIMD where NlzFft<Spec<768, 12, ..>> sits at 256 registers with scratch (1.7 - 2.0 TB/s against 4.1 - 4.5 for the
// 8-values plans of the powers of two: profiles/r06_nlz_variants.txt).  The rows of the six fields are staged through LDS
// (every bin is read from memory once; a thread needs the bins kappa and L - kappa of both fields of a pair), the forward
// combination reads the three F_s back out of LDS.  Same arithmetic contract as NlzFft with valid = L + 1.
// (A prefetching build -- the rows of the NEXT pair of fields loaded into three more registers per thread while the passes of the
// current pair run; hipcc's workgroup barrier waits for LDS only, so the loads do stay in flight -- was exact and SLOWER: 768
// 1.39 -> 1.64 ms, 1536 1.92 -> 2.10, profiles/r06_nlz_variants.txt section 4.  Removed.)
template <class SL, typename T, int ROWS, bool TWLDS>
struct Nlz3Fft {
  static constexpr int L = SL::N, M = 3 * SL::N, E = SL::E, TPT = SL::TPT, G = 3 * SL::TPT;
  static constexpr int THREADS = G * ROWS;
  static constexpr int PD = SL::R(0);
  static constexpr int PLEN = padded_len<L, PD>();                      // >= L + 1
  static constexpr int TW_BYTES = (TWLDS && SL::NP > 1) ? (int)(SL::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = (int)(3 * PLEN * ROWS * sizeof(cx<T>));
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;
  static constexpr int NSTAGE = (2 * (L + 1) + G - 1) / G;              // staging rounds of a pair of rows of L + 1 bins
  static constexpr int NOUT = (L + 1 + G - 1) / G;                      // output bins per thread
  typedef XchFull<T, PadSlot<PD>> Xch;

  // Y_s of this thread's positions out of the staged rows (see the header), ready for the inverse passes
  static MFFT_D void combine(cx<T> (&v)[E], int s, int j, const cx<T>* B, const cx<T>* rt) {
    const T h3 = (T)0.866025403784438646764L;
    const T wr = s == 0 ? (T)1 : (T)-0.5, wi = s == 0 ? (T)0 : (s == 1 ? -h3 : h3);      // w^{-s}: 1, conj(w), w
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const int kap = j + k * TPT;
      cx<T> a = B[kap], b = B[L + 1 + kap];
      const cx<T> am = B[L - kap], bm = B[2 * L + 1 - kap];
      if (kap == 0) { a.y = (T)0; b.y = (T)0; }    // Im of the k = 0 bins is ignored (as c2r does)
      const cx<T> u = mk<T>(a.x - b.y, a.y + b.x);                      // Z[kappa]     = a + i b
      const cx<T> vm = mk<T>(am.x + bm.y, bm.x - am.y);                 // Z[kappa - L] = conj(am) + i conj(bm)
      cx<T> y = mk<T>(u.x + wr * vm.x - wi * vm.y, u.y + wr * vm.y + wi * vm.x);
      if (kap == 0) {                              // ... and Z[L] w^s = (am + i bm) conj(w^{-s})
        const cx<T> zl = mk<T>(am.x - bm.y, am.y + bm.x);
        y = mk<T>(y.x + wr * zl.x + wi * zl.y, y.y + wr * zl.y - wi * zl.x);
      }
      if (s != 0) y = y * rt[(s - 1) * (L + 1) + kap];                  // W^{kappa s}
      v[k] = swapri(y);
    }
  }
  template <class TwPtr>
  static MFFT_D void inverse_pair(cx<T> (&v)[E], const cx<T>* ra, const cx<T>* rb, int t, int s, int j, cx<T>* B,
                                  const cx<T>* rt, TwPtr tw, Xch& xc) {
    // The kernel body holds five transforms; left alone, hipcc shares every LDS address between them and keeps all of them
    // live from the first to the last (E = 8 in double precision: 256 VGPRs where ONE transform needs ~80).  Hiding the
    // thread indices at the head of each transform makes their address arithmetic local to it again.
    MFFT_HIDE_RANGE(t);
    MFFT_HIDE_RANGE(j);
    MFFT_BARRIER();                                // B is free
#pragma unroll
    for (int ii = 0; ii < NSTAGE; ++ii) {          // the L + 1 bins of both rows, each read from memory once
      const int i = t + ii * G;
      const int ic = i < 2 * (L + 1) ? i : 2 * (L + 1) - 1;
      const cx<T>* src = ic <= L ? ra + ic : rb + (ic - (L + 1));       // one unconditional load of a selected address
      const cx<T> x = *src;
      if (i < 2 * (L + 1)) B[i] = x;
    }
    MFFT_BARRIER();
    combine(v, s, j, B, rt);
    MFFT_BARRIER();                                // the staged rows are consumed: B becomes the exchange buffer
    run_passes<SL, 0, T>(v, j, tw, xc);
  }

  template <class TwPtr>
  static MFFT_D void forward_pair(cx<T> (&v)[E], int t, int s, int j, cx<T>* B, const cx<T>* rt, TwPtr tw, Xch& xc,
                                  cx<T>* oa, cx<T>* ob, bool sa, bool sb) {
    MFFT_HIDE_RANGE(t);
    MFFT_HIDE_RANGE(j);
    MFFT_BARRIER();
    run_passes<SL, 0, T>(v, j, tw, xc);
    if constexpr (SL::NP > 1) MFFT_BARRIER();      // everybody's last gather: the padded regions are dead
#pragma unroll
    for (int k = 0; k < E; ++k) B[s * L + j + k * TPT] = v[k];          // F_s, plain layout
    MFFT_BARRIER();
    const T half = (T)0.5;
#pragma unroll
    for (int ii = 0; ii < NOUT; ++ii) {
      const int k = t + ii * G;
      if (k <= L) {
        const int k1 = k == L ? 0 : k, k2 = k == 0 ? 0 : L - k;
        const cx<T> w1 = rt[k], w2 = rt[L + 1 + k];
        const cx<T> zk = B[k1] + conj(w1) * B[L + k1] + conj(w2) * B[2 * L + k1];      // Z'[k]
        const cx<T> zm = conj(B[k2] + w1 * B[L + k2] + w2 * B[2 * L + k2]);            // conj Z'[M - k]
        if (sa) oa[k] = scale(zk + zm, half);
        if (sb) ob[k] = mul_mi(scale(zk - zm, half));
      }
    }
  }

  static MFFT_D void body(const NlzParams<T>& P, int bid, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    const int rl = tid / G, t = tid - rl * G;
    const int s = t / TPT, j = t - s * TPT;
    cx<T>* B = reinterpret_cast<cx<T>*>(lds + TW_BYTES) + rl * 3 * PLEN;
    if constexpr (TWLDS && SL::NP > 1) stage_twiddles<SL, T>(ltw, P.tw, tid, THREADS);     // (the first barrier covers it)
    const cx<T>* tw = (TWLDS && SL::NP > 1) ? (const cx<T>*)ltw : P.tw;
    Xch xc{B + s * PLEN, PadSlot<PD>{}};
    const i64 unit = (i64)bid * ROWS + rl;         // a pair of rows
    T r2a[E];
#pragma unroll
    for (int k = 0; k < E; ++k) r2a[k] = (T)0;
    bool sa = false;
    i64 rowa = 0;
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const i64 row = 2 * unit + h;
      const bool active = row < P.nrows;
      const i64 io = (active ? row : P.nrows - 1) * P.in_stride;
      cx<T> pk[2][E];
      cx<T> v[E];
      inverse_pair(v, P.a[0] + io, P.b[0] + io, t, s, j, B, P.rt3, tw, xc);
#pragma unroll
      for (int k = 0; k < E; ++k) pk[0][k] = v[k];
      inverse_pair(v, P.a[1] + io, P.b[1] + io, t, s, j, B, P.rt3, tw, xc);
#pragma unroll
      for (int k = 0; k < E; ++k) pk[1][k] = v[k];
      inverse_pair(v, P.a[2] + io, P.b[2] + io, t, s, j, B, P.rt3, tw, xc);
      T r2[E];
#pragma unroll
      for (int k = 0; k < E; ++k) {                // swapped results: .y = a_f, .x = b_f at n = 3 (j + k TPT) + s
        const T a0 = pk[0][k].y, a1 = pk[1][k].y, a2 = v[k].y;
        const T b0 = pk[0][k].x, b1 = pk[1][k].x, b2 = v[k].x;
        v[k] = mk<T>((a1 * b2 - a2 * b1) * P.scale, (a2 * b0 - a0 * b2) * P.scale);
        r2[k] = (a0 * b1 - a1 * b0) * P.scale;
      }
      const i64 oo = row * P.out_stride;
      forward_pair(v, t, s, j, B, P.rt3, tw, xc, P.out[0] + oo, P.out[1] + oo, active, active);
      if (h == 0) {
#pragma unroll
        for (int k = 0; k < E; ++k) r2a[k] = r2[k];
        sa = active;
        rowa = row;
      } else {
#pragma unroll
        for (int k = 0; k < E; ++k) v[k] = mk<T>(r2a[k], r2[k]);
        forward_pair(v, t, s, j, B, P.rt3, tw, xc, P.out[2] + rowa * P.out_stride, P.out[2] + oo, sa, active);
      }
    }
  }
};

}  // namespace mfft
