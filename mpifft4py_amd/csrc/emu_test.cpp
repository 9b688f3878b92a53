// emu_test.cpp -- host-side workgroup emulator for the kernel bodies in
// fft_kernels.h (TEST INFRASTRUCTURE, built with g++, never part of the product
// library).  Each GPU thread of a workgroup is a ucontext fibre; MFFT_BARRIER()
// yields to a round-robin scheduler, so the very same body code that hipcc
// compiles for gfx950 runs here with real barrier semantics and a real shared
// "LDS" buffer.  It checks every kernel family and radix plan against an
// O(N^2) long-double DFT, so index/twiddle mistakes are found without a GPU.
//
//   make -j6 emu && for b in build/emu_test_?; do $b; done
#include <ucontext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <functional>
#include <random>
#include <vector>

#include "fft_kernels.h"
#include "fft_chirpz.h"
#include "fft_col3.h"
#include "fft_nlz.h"
#include "twiddle.h"
#include "plans.h"

namespace mfft {

struct EmuState {
  std::vector<ucontext_t> ctx;
  std::vector<char> done;
  std::vector<int> nbar;
  ucontext_t sched;
  int cur = -1;
  std::function<void(int)> fn;
};
static EmuState* g_emu = nullptr;

void emu_barrier() {
  EmuState* e = g_emu;
  e->nbar[e->cur]++;
  swapcontext(&e->ctx[e->cur], &e->sched);
}

// wave shuffle of the kernels' host build (fft_kernels.h wave_shfl): every thread parks its value, all threads meet,
// every thread reads the slot of lane `src` of its own wave, all threads meet again (the slots are reused)
static std::vector<double> g_shfl_slot;
double emu_shfl(double x, int src) {
  EmuState* e = g_emu;
  const int tid = e->cur, n = (int)e->ctx.size();
  if ((int)g_shfl_slot.size() < n) g_shfl_slot.resize(n);
  g_shfl_slot[tid] = x;
  emu_barrier();
  const int from = (tid & ~63) + src;
  if (src < 0 || src > 63) { fprintf(stderr, "EMU: shuffle from lane %d\n", src); abort(); }
  const double r = from < n ? g_shfl_slot[from] : 0.0;      // a lane beyond the workgroup: undefined on the device
  emu_barrier();
  return r;
}

static void fibre_entry() {
  EmuState* e = g_emu;
  int tid = e->cur;
  e->fn(tid);
  e->done[tid] = 1;
  swapcontext(&e->ctx[tid], &e->sched);
}

// run one workgroup of `block` threads
static void emu_run_block(int block, const std::function<void(int)>& fn) {
  static std::vector<char> stacks;
  const size_t STK = 256 * 1024;
  if (stacks.size() < STK * block) stacks.resize(STK * block);
  EmuState e;
  e.ctx.resize(block);
  e.done.assign(block, 0);
  e.nbar.assign(block, 0);
  e.fn = fn;
  g_emu = &e;
  for (int t = 0; t < block; ++t) {
    getcontext(&e.ctx[t]);
    e.ctx[t].uc_stack.ss_sp = stacks.data() + STK * t;
    e.ctx[t].uc_stack.ss_size = STK;
    e.ctx[t].uc_link = nullptr;
    makecontext(&e.ctx[t], fibre_entry, 0);
  }
  bool all_done = false;
  while (!all_done) {
    all_done = true;
    for (int t = 0; t < block; ++t) {
      if (e.done[t]) continue;
      e.cur = t;
      swapcontext(&e.sched, &e.ctx[t]);
      if (!e.done[t]) all_done = false;
    }
    // every thread must have passed the same number of barriers
    for (int t = 1; t < block; ++t)
      if (e.nbar[t] != e.nbar[0]) {
        fprintf(stderr, "EMU: divergent barrier count (thread %d: %d vs %d)\n", t, e.nbar[t], e.nbar[0]);
        abort();
      }
  }
  g_emu = nullptr;
}

template <class Body>
static void emu_launch(int grid, int block, size_t lds_bytes, Body body) {
  std::vector<char> lds(lds_bytes + 64);
  for (int b = 0; b < grid; ++b) {
    memset(lds.data(), 0xCD, lds.size());
    emu_run_block(block, [&](int tid) { body(b, tid, lds.data()); });
  }
}

}  // namespace mfft

using namespace mfft;

typedef std::vector<cx<long double>> lvec;

static lvec naive_dft_slow(const lvec& x, int sign) {
  int n = (int)x.size();
  lvec X(n);
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int k = 0; k < n; ++k) {
    long double sr = 0, si = 0;
    for (int t = 0; t < n; ++t) {
      long long m = ((long long)k * t) % n;
      long double a = sign * two_pi * (long double)m / n;
      long double c = cosl(a), s = sinl(a);
      sr += x[t].x * c - x[t].y * s;
      si += x[t].x * s + x[t].y * c;
    }
    X[k].x = sr;
    X[k].y = si;
  }
  return X;
}

// O(N log N) long-double reference (recursive decimation in time over the
// smallest prime factor), validated against the O(N^2) sum in main().
static lvec naive_dft(const lvec& x, int sign) {
  int n = (int)x.size();
  int p = 0;
  for (int q : {2, 3, 5}) if (n % q == 0) { p = q; break; }
  if (n <= 5 || p == 0) return naive_dft_slow(x, sign);
  int m = n / p;
  std::vector<lvec> sub(p);
  for (int q = 0; q < p; ++q) {
    lvec s(m);
    for (int t = 0; t < m; ++t) s[t] = x[(size_t)t * p + q];
    sub[q] = naive_dft(s, sign);
  }
  lvec X(n);
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int k = 0; k < n; ++k) {
    long double sr = 0, si = 0;
    for (int q = 0; q < p; ++q) {
      long long mm = ((long long)k * q) % n;
      long double a = sign * two_pi * (long double)mm / n;
      long double c = cosl(a), s = sinl(a);
      const cx<long double>& z = sub[q][k % m];
      sr += z.x * c - z.y * s;
      si += z.x * s + z.y * c;
    }
    X[k].x = sr;
    X[k].y = si;
  }
  return X;
}

static int g_fail = 0;
static void report(const char* what, int n, const char* prec, double err, double tol) {
  bool ok = err < tol && err == err;
  printf("%-28s N=%-5d %-6s rel-L2 err %.3e  %s\n", what, n, prec, err, ok ? "ok" : "FAIL");
  if (!ok) g_fail++;
}

template <typename T> static double tol_of() { return sizeof(T) == 8 ? 1e-14 : 5e-6; }
template <typename T> static const char* pname() { return sizeof(T) == 8 ? "double" : "single"; }

// ---------------------------------------------------------------------------
template <class S, typename T, int COLS, bool INV, bool TWLDS, int SPLIT = 0, int VEC = 1>
static void test_col(bool two_level) {
  typedef ColFft<S, T, COLS, INV, TWLDS, SPLIT, VEC> K;
  const int N = S::N;
  const int ncols = COLS * 2 + 3;          // ragged last tile
  const int nouter = 2;
  std::mt19937_64 rng(1234 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  // input layout [outer][row][col] with row pitch pin; output pitch pout
  const int pin = ncols + 2, pout = ncols + 5;
  const int split = two_level ? (N % 4 == 0 ? N / 4 : N) : N;
  // two-level output: row r -> (r/split)*hi + (r%split)*lo with hi = "chunk" stride
  const i64 out_lo = pout, out_hi = (i64)split * pout * nouter;   // chunks interleave the outer index
  std::vector<cx<T>> in((size_t)nouter * N * pin), out((size_t)nouter * N * pout * 2, mk<T>((T)777, (T)777));
  for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
  auto tw = build_pass_twiddles<S, T>();
  ColParams<T> P;
  P.in = in.data();
  P.out = out.data();
  P.tw = tw.data();
  P.in_outer = (i64)N * pin;
  P.in_map = make_rowmap(0, pin, N, N);
  if (two_level) {
    P.out_outer = (i64)split * pout;
    P.out_map = make_rowmap(out_hi, out_lo, split, N);
  } else {
    P.out_outer = (i64)N * pout;
    P.out_map = make_rowmap(0, pout, N, N);
  }
  P.ncols = ncols;
  P.ntile_c = (ncols + COLS - 1) / COLS;
  P.nouter = nouter;
  P.remap = two_level ? 1 : 0;
  P.scale = INV ? (T)(1.0 / N) : (T)1;
  emu_launch(P.ntile_c * nouter, K::THREADS, K::LDS_BYTES,
             [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int o = 0; o < nouter; ++o)
    for (int c = 0; c < ncols; ++c) {
      lvec x(N);
      for (int r = 0; r < N; ++r) {
        cx<T> z = in[(size_t)o * N * pin + (size_t)r * pin + c];
        x[r].x = z.x;
        x[r].y = z.y;
      }
      lvec X = naive_dft(x, INV ? +1 : -1);
      for (int r = 0; r < N; ++r) {
        i64 off = (i64)o * P.out_outer + row_off(P.out_map, r) + c;
        cx<T> g = out[off];
        long double ex = X[r].x * (INV ? 1.0L / N : 1.0L), ey = X[r].y * (INV ? 1.0L / N : 1.0L);
        num += (g.x - ex) * (g.x - ex) + (g.y - ey) * (g.y - ey);
        den += ex * ex + ey * ey;
      }
    }
  char name[64];
  snprintf(name, sizeof name, "col c%d v%d %s%s%s%s", COLS, VEC, INV ? "inv" : "fwd", TWLDS ? " twlds" : "", two_level ? " 2lvl" : "", SPLIT == 2 ? " quarter" : SPLIT ? " split" : "");
  report(name, N, pname<T>(), (double)sqrtl(num / den), tol_of<T>());
}

// Wrapped input columns (ColParams::in_wrap, round 5): the same truncating forward kernel on a compact input and on one whose
// columns sit in rows of W columns at a pitch of W + 3 (W odd, so that a lane's VEC columns straddle the end of a row
// somewhere) must store the same bits.  K: any strided kernel type whose output has `nout` rows.
template <class K, typename T>
static void test_wrapped_columns(const char* what, int N, int nout, const cx<T>* tw, int cols_per_tile) {
  const int W = 5, G = 2 * cols_per_tile / W + 3, ncols = W * G, Wp = W + 3;
  const int pin_a = ncols + 1, pin_b = G * Wp + 2, pout = ncols + 2;
  std::mt19937_64 rng(4321 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<cx<T>> a((size_t)N * pin_a), b((size_t)N * pin_b, mk<T>((T)99, (T)99));
  for (int r = 0; r < N; ++r)
    for (int c = 0; c < ncols; ++c) {
      const cx<T> z = mk<T>((T)U(rng), (T)U(rng));
      a[(size_t)r * pin_a + c] = z;
      b[(size_t)r * pin_b + (size_t)(c / W) * Wp + c % W] = z;
    }
  std::vector<cx<T>> out_a((size_t)nout * pout, mk<T>((T)5, (T)5)), out_b = out_a;
  for (int wrapped = 0; wrapped < 2; ++wrapped) {
    ColParams<T> P;
    memset(&P, 0, sizeof P);
    P.in = wrapped ? b.data() : a.data(); P.out = wrapped ? out_b.data() : out_a.data(); P.tw = tw;
    P.in_map = make_rowmap(0, wrapped ? pin_b : pin_a, 0, N); P.out_map = make_rowmap(0, pout, 0, nout);
    P.ncols = ncols; P.ntile_c = (ncols + cols_per_tile - 1) / cols_per_tile; P.nouter = 1; P.remap = wrapped; P.fold = 1;
    P.scale = (T)0.5;
    if (wrapped) { P.in_wrap = W; P.in_wrap_gap = Wp - W; }
    emu_launch(P.ntile_c, K::THREADS, K::LDS_BYTES, [&](int bb, int t, char* lds) { K::body(P, bb, t, lds); });
  }
  size_t bad = 0;
  for (size_t i = 0; i < out_a.size(); ++i) bad += memcmp(&out_a[i], &out_b[i], sizeof(cx<T>)) != 0;
  bool any = false;
  for (int c = 0; c < ncols; ++c) any = any || out_a[c].x != (T)5;
  report(what, N, pname<T>(), (bad == 0 && any) ? 0.0 : 1.0, tol_of<T>());
}

// 3/2-rule fusion: PAD == 1 (inverse, zero band on load) and PAD == 2 (forward, truncate on store,
// with and without the Nyquist fold) against explicit pad / truncate around a plain DFT
template <class S, typename T, int COLS, int VEC>
static void test_col_pad() {
  const int N = S::N, n = 2 * N / 3, h = n / 2;
  const int ncols = COLS + 3, nouter = 2;
  std::mt19937_64 rng(31 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  auto tw = build_pass_twiddles<S, T>();
  {  // PAD 1, inverse: in (nouter, n, pin) -> out (nouter, N, pout)
    typedef ColFft<S, T, COLS, true, false, false, VEC, false, 1> K;
    const int pin = ncols + 1, pout = ncols + 2;
    std::vector<cx<T>> in((size_t)nouter * n * pin), out((size_t)nouter * N * pout);
    for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
    ColParams<T> P;
    P.in = in.data(); P.out = out.data(); P.tw = tw.data();
    P.in_outer = (i64)n * pin; P.out_outer = (i64)N * pout;
    P.in_map = make_rowmap(0, pin, n, n); P.out_map = make_rowmap(0, pout, N, N);
    P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = nouter; P.remap = 1; P.fold = 0;
    P.scale = (T)(1.0 / N);
    emu_launch(P.ntile_c * nouter, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    long double num = 0, den = 0;
    for (int o = 0; o < nouter; ++o)
      for (int c = 0; c < ncols; ++c) {
        lvec x(N);
        for (int r = 0; r < N; ++r) { x[r].x = 0; x[r].y = 0; }
        for (int r = 0; r < h; ++r) { cx<T> z = in[(size_t)o * n * pin + (size_t)r * pin + c]; x[r].x = z.x; x[r].y = z.y; }
        for (int r = h; r < n; ++r) { cx<T> z = in[(size_t)o * n * pin + (size_t)r * pin + c]; x[N - n + r].x = z.x; x[N - n + r].y = z.y; }
        lvec X = naive_dft(x, +1);
        for (int r = 0; r < N; ++r) {
          cx<T> g = out[(size_t)o * N * pout + (size_t)r * pout + c];
          long double ex = X[r].x / N, ey = X[r].y / N;
          num += (g.x - ex) * (g.x - ex) + (g.y - ey) * (g.y - ey);
          den += ex * ex + ey * ey;
        }
      }
    char name[64];
    snprintf(name, sizeof name, "col c%d v%d pad-on-load inv", COLS, VEC);
    report(name, N, pname<T>(), (double)sqrtl(num / den), tol_of<T>());
  }
  for (int fold = 0; fold < 2; ++fold) {  // PAD 2, forward: in (nouter, N, pin) -> out (nouter, n, pout)
    typedef ColFft<S, T, COLS, false, false, false, VEC, false, 2> K;
    const int pin = ncols + 1, pout = ncols + 2;
    std::vector<cx<T>> in((size_t)nouter * N * pin), out((size_t)nouter * n * pout, mk<T>((T)55, (T)55));
    for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
    ColParams<T> P;
    P.in = in.data(); P.out = out.data(); P.tw = tw.data();
    P.in_outer = (i64)N * pin; P.out_outer = (i64)n * pout;
    P.in_map = make_rowmap(0, pin, N, N); P.out_map = make_rowmap(0, pout, n, n);
    P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = nouter; P.remap = 0; P.fold = fold;
    P.scale = (T)0.5;
    emu_launch(P.ntile_c * nouter, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    long double num = 0, den = 0;
    for (int o = 0; o < nouter; ++o)
      for (int c = 0; c < ncols; ++c) {
        lvec x(N);
        for (int r = 0; r < N; ++r) { cx<T> z = in[(size_t)o * N * pin + (size_t)r * pin + c]; x[r].x = z.x; x[r].y = z.y; }
        lvec X = naive_dft(x, -1);
        lvec e(n);
        for (int r = 0; r < n; ++r) { e[r].x = 0; e[r].y = 0; }
        if (fold) {
          for (int r = 0; r <= h; ++r) e[r] = X[r];
          for (int r = 0; r < h; ++r) { e[h + r].x += X[N - h + r].x; e[h + r].y += X[N - h + r].y; }
        } else {
          for (int r = 0; r < h; ++r) e[r] = X[r];
          for (int r = h; r < n; ++r) e[r] = X[N - n + r];
        }
        for (int r = 0; r < n; ++r) {
          cx<T> g = out[(size_t)o * n * pout + (size_t)r * pout + c];
          long double ex = e[r].x * 0.5L, ey = e[r].y * 0.5L;
          num += (g.x - ex) * (g.x - ex) + (g.y - ey) * (g.y - ey);
          den += ex * ex + ey * ey;
        }
      }
    char name[64];
    snprintf(name, sizeof name, "col c%d v%d trunc-on-store%s", COLS, VEC, fold ? " fold" : "");
    report(name, N, pname<T>(), (double)sqrtl(num / den), tol_of<T>());
  }
  {
    char name[64];
    snprintf(name, sizeof name, "col c%d v%d trunc-on-store, wrapped input columns", COLS, VEC);
    test_wrapped_columns<ColFft<S, T, COLS, false, false, false, VEC, false, 2>, T>(name, N, n, tw.data(), COLS);
    snprintf(name, sizeof name, "col c%d v%d plain fwd, wrapped input columns", COLS, VEC);
    test_wrapped_columns<ColFft<S, T, COLS, false, false, false, VEC, false, 0>, T>(name, N, N, tw.data(), COLS);
  }
}

// 2/3-rule: the masked-load build (PAD == 3) against the plain inverse kernel on a pre-masked copy of the input
template <class S, typename T, int COLS, int VEC>
static void test_col_mask() {
  typedef ColFft<S, T, COLS, true, false, false, VEC, false, 3> KM;
  typedef ColFft<S, T, COLS, true, false, false, VEC> K0;
  const int N = S::N, ncols = COLS * 2 + 3, nouter = 2, pin = ncols + 1;
  std::mt19937_64 rng(77 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<cx<T>> in((size_t)nouter * N * pin), pre(in.size()), out1(in.size(), mk<T>((T)5, (T)5)), out0(in.size(), mk<T>((T)5, (T)5));
  std::vector<unsigned char> mask(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    in[i] = mk<T>((T)U(rng), (T)U(rng));
    mask[i] = (unsigned char)((rng() % 3) != 0);
    pre[i] = mask[i] ? in[i] : mk<T>((T)0, (T)0);
  }
  auto tw = build_pass_twiddles<S, T>();
  ColParams<T> P;
  memset(&P, 0, sizeof P);
  P.tw = tw.data();
  P.in_outer = P.out_outer = (i64)N * pin;
  P.in_map = P.out_map = make_rowmap(0, pin, N, N);
  P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = nouter; P.remap = 1; P.scale = (T)(1.0 / N);
  P.in = in.data(); P.out = out1.data(); P.mask = mask.data();
  emu_launch(P.ntile_c * nouter, KM::THREADS, KM::LDS_BYTES, [&](int b, int t, char* lds) { KM::body(P, b, t, lds); });
  P.in = pre.data(); P.out = out0.data(); P.mask = nullptr;
  emu_launch(P.ntile_c * nouter, K0::THREADS, K0::LDS_BYTES, [&](int b, int t, char* lds) { K0::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int o = 0; o < nouter; ++o)
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < ncols; ++c) {
        const size_t i = (size_t)o * N * pin + (size_t)r * pin + c;
        num += (out1[i].x - out0[i].x) * (out1[i].x - out0[i].x) + (out1[i].y - out0[i].y) * (out1[i].y - out0[i].y);
        den += out0[i].x * out0[i].x + out0[i].y * out0[i].y;
      }
  char name[64];
  snprintf(name, sizeof name, "col c%d v%d inv masked load", COLS, VEC);
  report(name, N, pname<T>(), (double)sqrtl(num / den), 1e-30);     // same arithmetic on the same values: identical
}

// pruned 2/3-rule pass (PAD == 4): removed rows read as zero, tiles without a kept column untouched, kept columns
// identical to the plain inverse kernel on an input whose removed rows were zeroed
template <class S, typename T, int COLS, int VEC>
static void test_col_band() {
  typedef ColFft<S, T, COLS, true, false, false, VEC, false, 4> KB;
  typedef ColFft<S, T, COLS, true, false, false, VEC> K0;
  const int N = S::N, ncols = COLS * 3 + 1, nouter = 2, pin = ncols + 1;
  if (N < 3) return;
  std::mt19937_64 rng(177 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  const cx<T> sentinel = mk<T>((T)5, (T)5);
  std::vector<cx<T>> in((size_t)nouter * N * pin), pre(in.size()), out1(in.size(), sentinel), out0(in.size(), sentinel);
  const int row_lo = N / 3 > 0 ? N / 3 : 1, row_hi = (2 * N) / 3 > row_lo ? (2 * N) / 3 : row_lo;
  for (int o = 0; o < nouter; ++o)
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < pin; ++c) {
        const size_t i = (size_t)o * N * pin + (size_t)r * pin + c;
        in[i] = mk<T>((T)U(rng), (T)U(rng));
        pre[i] = (r >= row_lo && r < row_hi) ? mk<T>((T)0, (T)0) : in[i];
      }
  auto tw = build_pass_twiddles<S, T>();
  ColParams<T> P;
  memset(&P, 0, sizeof P);
  P.tw = tw.data();
  P.in_outer = P.out_outer = (i64)N * pin;
  P.in_map = P.out_map = make_rowmap(0, pin, N, N);
  P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = nouter; P.remap = 1; P.scale = (T)(1.0 / N);
  P.b_row_lo = row_lo; P.b_row_hi = row_hi; P.b_coff = 2; P.b_cper = 5; P.b_clim = 3; P.b_goff = 1; P.b_gstep = 2; P.b_glo = 3; P.b_ghi = 5;
  P.in = in.data(); P.out = out1.data();
  emu_launch(P.ntile_c * nouter, KB::THREADS, KB::LDS_BYTES, [&](int b, int t, char* lds) { KB::body(P, b, t, lds); });
  P.in = pre.data(); P.out = out0.data();
  emu_launch(P.ntile_c * nouter, K0::THREADS, K0::LDS_BYTES, [&](int b, int t, char* lds) { K0::body(P, b, t, lds); });
  long double num = 0, den = 0;
  int bad = 0, skipped = 0, kept = 0;
  for (int o = 0; o < nouter; ++o)
    for (int c = 0; c < ncols; ++c) {
      const int t = 2 + c, z = t % 5, y = 1 + t / 5 + o * 2;
      const bool keep = z < 3 && (y < 3 || y >= 5);
      kept += keep;
      for (int r = 0; r < N; ++r) {
        const size_t i = (size_t)o * N * pin + (size_t)r * pin + c;
        const bool untouched = out1[i].x == sentinel.x && out1[i].y == sentinel.y;
        if (untouched) {
          if (keep) ++bad;               // a kept column must have been transformed
          ++skipped;
          continue;
        }
        num += (out1[i].x - out0[i].x) * (out1[i].x - out0[i].x) + (out1[i].y - out0[i].y) * (out1[i].y - out0[i].y);
        den += out0[i].x * out0[i].x + out0[i].y * out0[i].y;
      }
    }
  char name[64];
  snprintf(name, sizeof name, "col c%d v%d inv pruned (kept %d, skipped %d)", COLS, VEC, kept, skipped / N);
  report(name, N, pname<T>(), bad || kept == 0 ? 1.0 : (double)sqrtl(num / (den > 0 ? den : 1)), 1e-30);
  // complete-output mode (b_gzero = 2, the pencils' first inverse pass): EVERY column is written -- kept ones as the plain
  // kernel on the row-zeroed input, columns of a removed y or z as exact zeros (per column, also inside a lane's VEC pair)
  std::vector<cx<T>> out2(in.size(), sentinel);
  P.b_gzero = 2;
  P.in = in.data(); P.out = out2.data();
  emu_launch(P.ntile_c * nouter, KB::THREADS, KB::LDS_BYTES, [&](int b, int t, char* lds) { KB::body(P, b, t, lds); });
  num = den = 0;
  bad = 0;
  for (int o = 0; o < nouter; ++o)
    for (int c = 0; c < ncols; ++c) {
      const int t = 2 + c, z = t % 5, y = 1 + t / 5 + o * 2;
      const bool keep = z < 3 && (y < 3 || y >= 5);
      for (int r = 0; r < N; ++r) {
        const size_t i = (size_t)o * N * pin + (size_t)r * pin + c;
        if (out2[i].x == sentinel.x && out2[i].y == sentinel.y) { ++bad; continue; }       // nothing may stay unwritten
        const cx<T> want = keep ? out0[i] : mk<T>((T)0, (T)0);
        if (!keep && (out2[i].x != (T)0 || out2[i].y != (T)0)) ++bad;
        num += (out2[i].x - want.x) * (out2[i].x - want.x) + (out2[i].y - want.y) * (out2[i].y - want.y);
        den += want.x * want.x + want.y * want.y;
      }
    }
  snprintf(name, sizeof name, "col c%d v%d inv band, complete output", COLS, VEC);
  report(name, N, pname<T>(), bad ? 1.0 : (double)sqrtl(num / (den > 0 ? den : 1)), 1e-30);
}

template <class S, bool HAS3 = (S::E % 3 == 0 && S::N >= 6)> struct PadTests {
  static void run() {}
};
template <class S> struct PadTests<S, true> {
  static void run() {
    test_col_pad<S, double, 4, 1>();
    test_col_pad<S, float, 8, 2>();
  }
};

template <class S, typename T, int ROWS, bool INV, bool TWLDS, bool SPLIT = false>
static void test_row() {
  typedef RowFft<S, T, ROWS, INV, TWLDS, false, SPLIT> K;
  const int N = S::N;
  const int nrows = ROWS * 2 + 1;
  std::mt19937_64 rng(99 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  const int pin = N + 3, pout = N + 1;
  std::vector<cx<T>> in((size_t)nrows * pin), out((size_t)nrows * pout);
  for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
  auto tw = build_pass_twiddles<S, T>();
  RowParams<T> P{in.data(), out.data(), tw.data(), pin, pout, nrows, INV ? (T)(1.0 / N) : (T)1};
  emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES,
             [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    lvec x(N);
    for (int i = 0; i < N; ++i) { x[i].x = in[(size_t)r * pin + i].x; x[i].y = in[(size_t)r * pin + i].y; }
    lvec X = naive_dft(x, INV ? +1 : -1);
    for (int i = 0; i < N; ++i) {
      long double s = INV ? 1.0L / N : 1.0L;
      cx<T> g = out[(size_t)r * pout + i];
      num += (g.x - X[i].x * s) * (g.x - X[i].x * s) + (g.y - X[i].y * s) * (g.y - X[i].y * s);
      den += X[i].x * s * X[i].x * s + X[i].y * s * X[i].y * s;
    }
  }
  char name[64];
  snprintf(name, sizeof name, "row r%d %s%s%s", ROWS, INV ? "inv" : "fwd", TWLDS ? " twlds" : "", SPLIT ? " split" : "");
  report(name, N, pname<T>(), (double)sqrtl(num / den), tol_of<T>());
  // z-chunked side (pencil C2C): forward stores into / inverse loads out of nch equal blocks (rows_total, q)
  if (N % 4 == 0 && N >= 8) {
    const int nch = 4, q = N / nch, rows_total = nrows + 2, row0 = 1;
    const ZSplit zs = make_zsplit(q, nch, q, rows_total);
    std::vector<cx<T>> blocks((size_t)rows_total * N, mk<T>((T)3, (T)3)), out2((size_t)nrows * pout);
    typedef RowFft<S, T, ROWS, INV, TWLDS, true, SPLIT> K;
    RowParams<T> Pc{in.data(), out2.data(), tw.data(), pin, pout, nrows, INV ? (T)(1.0 / N) : (T)1, zs, row0};
    if (INV) {       // scatter the plain input rows into the blocks, transform out of them
      for (int r = 0; r < nrows; ++r)
        for (int k = 0; k < N; ++k) blocks[(size_t)(k / q) * rows_total * q + (size_t)(row0 + r) * q + k % q] = in[(size_t)r * pin + k];
      Pc.in = blocks.data();
    } else {
      Pc.out = blocks.data();
    }
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(Pc, b, t, lds); });
    long double nn = 0, dd = 0;
    for (int r = 0; r < nrows; ++r)
      for (int k = 0; k < N; ++k) {
        const cx<T> e = out[(size_t)r * pout + k];
        const cx<T> g = INV ? out2[(size_t)r * pout + k] : blocks[(size_t)(k / q) * rows_total * q + (size_t)(row0 + r) * q + k % q];
        nn += (g.x - e.x) * (g.x - e.x) + (g.y - e.y) * (g.y - e.y);
        dd += e.x * e.x + e.y * e.y;
      }
    snprintf(name, sizeof name, "row r%d %s z-chunked", ROWS, INV ? "inv" : "fwd");
    report(name, N, pname<T>(), (double)sqrtl(nn / dd), 1e-30 + (double)0);   // same arithmetic, other addresses: identical
  }
}

template <class S, typename T, int ROWS, bool TWLDS, bool SPLIT = false, bool MLDS = false>
static void test_real() {
  const int M = S::N, N = 2 * M;
  const int nrows = ROWS + 2;
  std::mt19937_64 rng(7 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  const int pin = N + 2, pout = M + 1 + 2;
  std::vector<T> in((size_t)nrows * pin);
  std::vector<cx<T>> out((size_t)nrows * pout);
  for (auto& z : in) z = (T)U(rng);
  auto tw = build_pass_twiddles<S, T>();
  auto rtw = build_real_twiddles<T>(N);
  {
    typedef R2CFft<S, T, ROWS, TWLDS, false, false, SPLIT> K;
    RealParams<T> P{in.data(), out.data(), tw.data(), rtw.data(), pin, pout, nrows, M + 1, (T)1};
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES,
               [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  }
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    lvec x(N);
    for (int i = 0; i < N; ++i) { x[i].x = in[(size_t)r * pin + i]; x[i].y = 0; }
    lvec X = naive_dft(x, -1);
    for (int k = 0; k <= M; ++k) {
      cx<T> g = out[(size_t)r * pout + k];
      num += (g.x - X[k].x) * (g.x - X[k].x) + (g.y - X[k].y) * (g.y - X[k].y);
      den += X[k].x * X[k].x + X[k].y * X[k].y;
    }
  }
  char name[64];
  snprintf(name, sizeof name, "r2c r%d%s%s", ROWS, TWLDS ? " twlds" : "", SPLIT ? " split" : "");
  report(name, N, pname<T>(), (double)sqrtl(num / den), tol_of<T>());
  // c2r of (the r2c result with garbage imaginary parts in bins 0 and M) must return the input
  std::vector<T> back((size_t)nrows * pin, (T)0);
  for (int r = 0; r < nrows; ++r) {
    out[(size_t)r * pout + 0].y = (T)3.5;
    out[(size_t)r * pout + M].y = (T)-2.25;
  }
  {
    typedef C2RFft<S, T, ROWS, TWLDS, false, false, SPLIT, false, MLDS> K;
    RealParams<T> P{out.data(), back.data(), tw.data(), rtw.data(), pout, pin, nrows, M + 1, (T)(1.0 / N)};
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES,
               [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  }
  num = den = 0;
  for (int r = 0; r < nrows; ++r)
    for (int i = 0; i < N; ++i) {
      long double d = (long double)back[(size_t)r * pin + i] - in[(size_t)r * pin + i];
      num += d * d;
      den += (long double)in[(size_t)r * pin + i] * in[(size_t)r * pin + i];
    }
  snprintf(name, sizeof name, "c2r(r2c) r%d%s%s%s", ROWS, TWLDS ? " twlds" : "", SPLIT ? " split" : "", MLDS ? " mlds" : "");
  report(name, N, pname<T>(), (double)sqrtl(num / den), 4 * tol_of<T>());
  // 3/2-rule column handling: r2c keeps only the first `valid` bins, c2r treats the missing ones as zero
  if (M >= 4) {
    const int valid = M / 2 + 1;
    std::vector<cx<T>> part((size_t)nrows * valid, mk<T>((T)9, (T)9));
    {
      typedef R2CFft<S, T, ROWS, TWLDS, true, false, SPLIT> K;
      RealParams<T> P{in.data(), part.data(), tw.data(), rtw.data(), pin, valid, nrows, valid, (T)1};
      emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES,
                 [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    long double nn = 0, dd = 0;
    for (int r = 0; r < nrows; ++r)
      for (int k = 0; k < valid; ++k) {
        cx<T> g = part[(size_t)r * valid + k], e = out[(size_t)r * pout + k];
        if (k == 0) e.y = 0;        // out[] had its bin-0 imaginary part poisoned above
        nn += (g.x - e.x) * (g.x - e.x) + (g.y - e.y) * (g.y - e.y);
        dd += e.x * e.x + e.y * e.y;
      }
    snprintf(name, sizeof name, "r2c valid<M+1 r%d", ROWS);
    report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
    std::vector<T> b2((size_t)nrows * pin, (T)0);
    {
      typedef C2RFft<S, T, ROWS, TWLDS, true, false, SPLIT, false, MLDS> K;
      RealParams<T> P{part.data(), b2.data(), tw.data(), rtw.data(), valid, pin, nrows, valid, (T)(1.0 / N)};
      emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES,
                 [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    nn = dd = 0;
    for (int r = 0; r < nrows; ++r) {
      lvec X(N);
      for (int k = 0; k < N; ++k) { X[k].x = 0; X[k].y = 0; }
      for (int k = 0; k < valid; ++k) {
        cx<T> z = part[(size_t)r * valid + k];
        X[k].x = z.x; X[k].y = (k == 0) ? 0 : z.y;
        if (k > 0) { X[N - k].x = z.x; X[N - k].y = -z.y; }
      }
      lvec x = naive_dft(X, +1);
      for (int i = 0; i < N; ++i) {
        long double e = x[i].x / N, g = b2[(size_t)r * pin + i];
        nn += (g - e) * (g - e);
        dd += e * e;
      }
    }
    snprintf(name, sizeof name, "c2r valid<M+1 r%d", ROWS);
    report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
  }
  // z-chunked complex side (fused pencil pack / unpack): the M + 1 bins of a row go to nch blocks (rows_total, len_l);
  // `drop`: the last block has no room for the Nyquist bin (the 'AlltoallN' mode), which then reads back as zero
  if (M >= 4 && M % 2 == 0) {
    for (int dp = 0; dp < 3; ++dp) {        // dp == 2: rows of the blocks further apart than their length (ZSplit pitch)
      const int drop = dp == 1;
      const int nch = 2, q = M / nch, last = q + (drop ? 0 : 1), rows_total = nrows + 3, row0 = 2;
      const int pq = dp == 2 ? q + 3 : q, plast = dp == 2 ? last + 5 : last;
      const ZSplit zs = dp == 2 ? make_zsplit(q, nch, last, rows_total, pq, plast) : make_zsplit(q, nch, last, rows_total);
      std::vector<cx<T>> blocks((size_t)rows_total * (pq * (nch - 1) + plast), mk<T>((T)7, (T)7));
      {
        typedef R2CFft<S, T, ROWS, TWLDS, false, true, SPLIT> K;
        RealParams<T> P{in.data(), blocks.data(), tw.data(), rtw.data(), pin, 0, nrows, M + 1, (T)1, zs, row0};
        emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      }
      long double nn = 0, dd = 0;
      for (int r = 0; r < nrows; ++r)
        for (int k = 0; k <= M - drop; ++k) {
          const int l = std::min(k / q, nch - 1), len = l == nch - 1 ? plast : pq;
          cx<T> g = blocks[(size_t)l * rows_total * pq + (size_t)(row0 + r) * len + (k - l * q)], e = out[(size_t)r * pout + k];
          if (k == 0 || k == M) e.y = 0;
          nn += (g.x - e.x) * (g.x - e.x) + (g.y - e.y) * (g.y - e.y);
          dd += e.x * e.x + e.y * e.y;
        }
      snprintf(name, sizeof name, "r2c z-chunked%s r%d", drop ? " drop" : dp == 2 ? " pitched" : "", ROWS);
      report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
      std::vector<T> b3((size_t)nrows * pin, (T)0);
      {
        typedef C2RFft<S, T, ROWS, TWLDS, false, true, SPLIT, false, MLDS> K;
        RealParams<T> P{blocks.data(), b3.data(), tw.data(), rtw.data(), 0, pin, nrows, M + 1, (T)(1.0 / N), zs, row0};
        emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      }
      nn = dd = 0;
      for (int r = 0; r < nrows; ++r) {
        lvec X(N);
        for (int k = 0; k < N; ++k) { X[k].x = 0; X[k].y = 0; }
        for (int k = 0; k <= M - drop; ++k) {
          const cx<T> z = out[(size_t)r * pout + k];
          X[k].x = z.x; X[k].y = (k == 0 || k == M) ? 0 : z.y;
          if (k > 0 && k < M) { X[N - k].x = z.x; X[N - k].y = -z.y; }
        }
        lvec x = naive_dft(X, +1);
        for (int i = 0; i < N; ++i) {
          long double e = x[i].x / N, g = b3[(size_t)r * pin + i];
          nn += (g - e) * (g - e);
          dd += e * e;
        }
      }
      snprintf(name, sizeof name, "c2r z-chunked%s r%d", drop ? " drop" : dp == 2 ? " pitched" : "", ROWS);
      report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
    }
  }
  // column-limited AND z-chunked (the 3/2-rule pencil transforms): only the first `valid` bins exist, split into
  // nch blocks with an uneven last one (valid = 2 * q + 1: the un-padded mesh's Nyquist column on the last rank)
  if (M >= 12 && M % 3 == 0) {
    const int valid = M / 3 * 2 / 2 * 2 + 1 <= M ? (M / 3) / 2 * 2 + 1 : 3;      // odd, about a third of the bins
    const int nch = 2, q = (valid - 1) / nch, last = q + 1, rows_total = nrows + 2, row0 = 1;
    const ZSplit zs = make_zsplit(q, nch, last, rows_total);
    std::vector<cx<T>> blocks((size_t)rows_total * (q * (nch - 1) + last), mk<T>((T)7, (T)7));
    {
      typedef R2CFft<S, T, ROWS, TWLDS, true, true, SPLIT> K;
      RealParams<T> P{in.data(), blocks.data(), tw.data(), rtw.data(), pin, 0, nrows, valid, (T)1, zs, row0};
      emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    long double nn = 0, dd = 0;
    for (int r = 0; r < nrows; ++r)
      for (int k = 0; k < valid; ++k) {
        const int l = std::min(k / q, nch - 1), len = l == nch - 1 ? last : q;
        cx<T> g = blocks[(size_t)l * rows_total * q + (size_t)(row0 + r) * len + (k - l * q)], e = out[(size_t)r * pout + k];
        if (k == 0) e.y = 0;
        nn += (g.x - e.x) * (g.x - e.x) + (g.y - e.y) * (g.y - e.y);
        dd += e.x * e.x + e.y * e.y;
      }
    snprintf(name, sizeof name, "r2c valid<M+1 z-chunked r%d", ROWS);
    report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
    std::vector<T> b4((size_t)nrows * pin, (T)0);
    {
      typedef C2RFft<S, T, ROWS, TWLDS, true, true, SPLIT, false, MLDS> K;
      RealParams<T> P{blocks.data(), b4.data(), tw.data(), rtw.data(), 0, pin, nrows, valid, (T)(1.0 / N), zs, row0};
      emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    nn = dd = 0;
    for (int r = 0; r < nrows; ++r) {
      lvec X(N);
      for (int k = 0; k < N; ++k) { X[k].x = 0; X[k].y = 0; }
      for (int k = 0; k < valid; ++k) {
        const cx<T> z = out[(size_t)r * pout + k];
        X[k].x = z.x; X[k].y = (k == 0) ? 0 : z.y;
        if (k > 0) { X[N - k].x = z.x; X[N - k].y = -z.y; }
      }
      lvec x = naive_dft(X, +1);
      for (int i = 0; i < N; ++i) {
        long double e = x[i].x / N, g = b4[(size_t)r * pin + i];
        nn += (g - e) * (g - e);
        dd += e * e;
      }
    }
    snprintf(name, sizeof name, "c2r valid<M+1 z-chunked r%d", ROWS);
    report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
  }
}

// ---------------------------------------------------------------------------
// chirp-z (Bluestein) kernels: runtime length n on the compiled plan S of length M >= 2n-1
// ---------------------------------------------------------------------------
template <class S, typename T, int COLS, bool INV, bool SPLIT, int VEC>
static void test_col_z(int n) {
  typedef ColFftZ<S, T, COLS, INV, SPLIT, VEC> K;
  const int ncols = COLS + 3, nouter = 2;
  std::mt19937_64 rng(4321 + n);
  std::uniform_real_distribution<double> U(-1, 1);
  const int pin = ncols + 2, pout = ncols + 5;
  std::vector<cx<T>> in((size_t)nouter * n * pin), out((size_t)nouter * n * pout, mk<T>((T)777, (T)777));
  for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
  auto tw = build_pass_twiddles<S, T>();
  auto ch = build_chirp<T>(n);
  auto bh = build_chirp_filter<T>(n, S::N);
  ColParamsZ<T> P;
  P.in = in.data(); P.out = out.data(); P.tw = tw.data();
  P.in_outer = (i64)n * pin; P.out_outer = (i64)n * pout;
  P.in_map = make_rowmap(0, pin, n, n); P.out_map = make_rowmap(0, pout, n, n);
  P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = nouter; P.remap = 1; P.fold = 0;
  P.scale = INV ? (T)(1.0 / n) : (T)1;
  P.chirp = ch.data(); P.bhat = bh.data(); P.n = n;
  emu_launch(P.ntile_c * nouter, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int o = 0; o < nouter; ++o)
    for (int c = 0; c < ncols; ++c) {
      lvec x(n);
      for (int r = 0; r < n; ++r) { cx<T> z = in[(size_t)o * n * pin + (size_t)r * pin + c]; x[r].x = z.x; x[r].y = z.y; }
      lvec X = naive_dft(x, INV ? +1 : -1);
      for (int r = 0; r < n; ++r) {
        cx<T> g = out[(size_t)o * n * pout + (size_t)r * pout + c];
        long double sc = INV ? 1.0L / n : 1.0L, ex = X[r].x * sc, ey = X[r].y * sc;
        num += (g.x - ex) * (g.x - ex) + (g.y - ey) * (g.y - ey);
        den += ex * ex + ey * ey;
      }
    }
  char name[64];
  snprintf(name, sizeof name, "chirpz col M=%d c%d v%d %s%s", S::N, COLS, VEC, INV ? "inv" : "fwd", SPLIT ? " split" : "");
  report(name, n, pname<T>(), (double)sqrtl(num / den), 8 * tol_of<T>());
}

template <class S, typename T, int ROWS, bool INV>
static void test_row_z(int n) {
  typedef RowFftZ<S, T, ROWS, 0, INV> K;
  const int nrows = ROWS * 2 + 1;
  std::mt19937_64 rng(990 + n);
  std::uniform_real_distribution<double> U(-1, 1);
  const int pin = n + 3, pout = n + 1;
  std::vector<cx<T>> in((size_t)nrows * pin), out((size_t)nrows * pout);
  for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
  auto tw = build_pass_twiddles<S, T>();
  auto ch = build_chirp<T>(n);
  auto bh = build_chirp_filter<T>(n, S::N);
  RowParamsZ<T> P;
  P.in = in.data(); P.out = out.data(); P.tw = tw.data(); P.in_stride = pin; P.out_stride = pout; P.nrows = nrows;
  P.scale = INV ? (T)(1.0 / n) : (T)1; P.chirp = ch.data(); P.bhat = bh.data(); P.n = n;
  emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    lvec x(n);
    for (int i = 0; i < n; ++i) { x[i].x = in[(size_t)r * pin + i].x; x[i].y = in[(size_t)r * pin + i].y; }
    lvec X = naive_dft(x, INV ? +1 : -1);
    for (int i = 0; i < n; ++i) {
      long double s = INV ? 1.0L / n : 1.0L;
      cx<T> g = out[(size_t)r * pout + i];
      num += (g.x - X[i].x * s) * (g.x - X[i].x * s) + (g.y - X[i].y * s) * (g.y - X[i].y * s);
      den += X[i].x * s * X[i].x * s + X[i].y * s * X[i].y * s;
    }
  }
  char name[64];
  snprintf(name, sizeof name, "chirpz row M=%d r%d %s", S::N, ROWS, INV ? "inv" : "fwd");
  report(name, n, pname<T>(), (double)sqrtl(num / den), 8 * tol_of<T>());
}

template <class S, typename T, int ROWS>
static void test_real_z(int n) {
  const int nh = n / 2, nrows = ROWS + 2;
  std::mt19937_64 rng(70 + n);
  std::uniform_real_distribution<double> U(-1, 1);
  const int pin = n + 1, pout = nh + 1 + 2;         // odd real pitch on purpose
  std::vector<T> in((size_t)nrows * pin), back((size_t)nrows * pin, (T)0);
  std::vector<cx<T>> out((size_t)nrows * pout);
  for (auto& z : in) z = (T)U(rng);
  auto tw = build_pass_twiddles<S, T>();
  auto ch = build_chirp<T>(n);
  auto bh = build_chirp_filter<T>(n, S::N);
  RealParamsZ<T> P;
  P.in = in.data(); P.out = out.data(); P.tw = tw.data(); P.rtw = nullptr; P.in_stride = pin; P.out_stride = pout;
  P.nrows = nrows; P.valid = nh + 1; P.scale = (T)1; P.chirp = ch.data(); P.bhat = bh.data(); P.n = n;
  {
    typedef RowFftZ<S, T, ROWS, 1, false> K;
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  }
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    lvec x(n);
    for (int i = 0; i < n; ++i) { x[i].x = in[(size_t)r * pin + i]; x[i].y = 0; }
    lvec X = naive_dft(x, -1);
    for (int k = 0; k <= nh; ++k) {
      cx<T> g = out[(size_t)r * pout + k];
      num += (g.x - X[k].x) * (g.x - X[k].x) + (g.y - X[k].y) * (g.y - X[k].y);
      den += X[k].x * X[k].x + X[k].y * X[k].y;
    }
  }
  char name[64];
  snprintf(name, sizeof name, "chirpz r2c M=%d r%d", S::N, ROWS);
  report(name, n, pname<T>(), (double)sqrtl(num / den), 8 * tol_of<T>());
  for (int r = 0; r < nrows; ++r) {                 // garbage where c2r must not look
    out[(size_t)r * pout + 0].y = (T)3.5;
    if (n % 2 == 0) out[(size_t)r * pout + nh].y = (T)-2.25;
  }
  P.in = out.data(); P.out = back.data(); P.in_stride = pout; P.out_stride = pin; P.scale = (T)(1.0 / n);
  {
    typedef RowFftZ<S, T, ROWS, 2, true> K;
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  }
  num = den = 0;
  for (int r = 0; r < nrows; ++r)
    for (int i = 0; i < n; ++i) {
      long double d = (long double)back[(size_t)r * pin + i] - in[(size_t)r * pin + i];
      num += d * d;
      den += (long double)in[(size_t)r * pin + i] * in[(size_t)r * pin + i];
    }
  snprintf(name, sizeof name, "chirpz c2r(r2c) M=%d r%d", S::N, ROWS);
  report(name, n, pname<T>(), (double)sqrtl(num / den), 16 * tol_of<T>());
}

// half-length real chirp-z (KIND 3 / 4): even real length n = 2m, chirp-z of length m on the plan S (M >= 2m-1)
template <class S, typename T, int ROWS>
static void test_real_zh(int m) {
  const int n = 2 * m, nrows = ROWS + 2;
  std::mt19937_64 rng(170 + n);
  std::uniform_real_distribution<double> U(-1, 1);
  const int pin = n + 2, pout = m + 1 + 2;
  std::vector<T> in((size_t)nrows * pin), back((size_t)nrows * pin, (T)0);
  std::vector<cx<T>> out((size_t)nrows * pout);
  for (auto& z : in) z = (T)U(rng);
  auto tw = build_pass_twiddles<S, T>();
  auto ch = build_chirp<T>(m);
  auto bh = build_chirp_filter<T>(m, S::N);
  auto rtw = build_real_twiddles<T>(n);
  RealParamsZ<T> P;
  P.in = in.data(); P.out = out.data(); P.tw = tw.data(); P.rtw = rtw.data(); P.in_stride = pin; P.out_stride = pout;
  P.nrows = nrows; P.valid = m + 1; P.scale = (T)1; P.chirp = ch.data(); P.bhat = bh.data(); P.n = n;
  {
    typedef RowFftZ<S, T, ROWS, 3, false> K;
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  }
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    lvec x(n);
    for (int i = 0; i < n; ++i) { x[i].x = in[(size_t)r * pin + i]; x[i].y = 0; }
    lvec X = naive_dft(x, -1);
    for (int k = 0; k <= m; ++k) {
      cx<T> g = out[(size_t)r * pout + k];
      num += (g.x - X[k].x) * (g.x - X[k].x) + (g.y - X[k].y) * (g.y - X[k].y);
      den += X[k].x * X[k].x + X[k].y * X[k].y;
    }
  }
  char name[64];
  snprintf(name, sizeof name, "chirpz/2 r2c M=%d r%d", S::N, ROWS);
  report(name, n, pname<T>(), (double)sqrtl(num / den), 8 * tol_of<T>());
  for (int r = 0; r < nrows; ++r) {
    out[(size_t)r * pout + 0].y = (T)3.5;
    out[(size_t)r * pout + m].y = (T)-2.25;
  }
  P.in = out.data(); P.out = back.data(); P.in_stride = pout; P.out_stride = pin; P.scale = (T)(1.0 / n);
  {
    typedef RowFftZ<S, T, ROWS, 4, true> K;
    emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  }
  num = den = 0;
  for (int r = 0; r < nrows; ++r)
    for (int i = 0; i < n; ++i) {
      long double d = (long double)back[(size_t)r * pin + i] - in[(size_t)r * pin + i];
      num += d * d;
      den += (long double)in[(size_t)r * pin + i] * in[(size_t)r * pin + i];
    }
  snprintf(name, sizeof name, "chirpz/2 c2r(r2c) M=%d r%d", S::N, ROWS);
  report(name, n, pname<T>(), (double)sqrtl(num / den), 16 * tol_of<T>());
}

// wave-packed c2r (fft_kernels.h C2RFft WP): plain, column-limited and z-chunked, against the input of the r2c kernels
template <class S, typename T, int WAVES, bool SPLIT>
static void test_c2r_wave_packed() {
  if constexpr (S::TPT < 64 && 64 % S::TPT != 0 && S::E % 2 == 0) {      // (register pairing: even E -- registry.h wave_packable)
    constexpr int RPW = 64 / S::TPT, ROWS = WAVES * RPW;
    const int M = S::N, N = 2 * M, nrows = 2 * ROWS + 1;       // the last workgroup is ragged, its last wave too
    std::mt19937_64 rng(99 + N);
    std::uniform_real_distribution<double> U(-1, 1);
    const int pin = N + 2, pout = M + 1 + 2;
    std::vector<T> in((size_t)nrows * pin), back((size_t)nrows * pin, (T)0);
    std::vector<cx<T>> out((size_t)nrows * pout);
    for (auto& z : in) z = (T)U(rng);
    auto tw = build_pass_twiddles<S, T>();
    auto rtw = build_real_twiddles<T>(N);
    {
      typedef R2CFft<S, T, 2, false, false, false, false> K;
      RealParams<T> P{in.data(), out.data(), tw.data(), rtw.data(), pin, pout, nrows, M + 1, (T)1};
      emu_launch((nrows + 1) / 2, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    char name[64];
    {     // r2c, wave-packed, against the dense kernel's output (and, column-limited, its first bins)
      std::vector<cx<T>> o2((size_t)nrows * pout, mk<T>((T)7, (T)7)), o3((size_t)nrows * (M / 2 + 1), mk<T>((T)7, (T)7));
      {
        typedef R2CFft<S, T, ROWS, false, false, false, SPLIT, true> K;
        static_assert(K::THREADS == WAVES * 64, "whole waves");
        RealParams<T> P{in.data(), o2.data(), tw.data(), rtw.data(), pin, pout, nrows, M + 1, (T)1};
        emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      }
      {
        typedef R2CFft<S, T, ROWS, false, true, false, SPLIT, true> K;
        RealParams<T> P{in.data(), o3.data(), tw.data(), rtw.data(), pin, M / 2 + 1, nrows, M / 2 + 1, (T)1};
        emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      }
      long double nn = 0, dd = 0, n3 = 0;
      for (int r = 0; r < nrows; ++r)
        for (int k = 0; k <= M; ++k) {
          const cx<T> a = o2[(size_t)r * pout + k], e = out[(size_t)r * pout + k];
          nn += (long double)(a.x - e.x) * (a.x - e.x) + (long double)(a.y - e.y) * (a.y - e.y);
          dd += (long double)e.x * e.x + (long double)e.y * e.y;
          if (k <= M / 2) {
            const cx<T> c = o3[(size_t)r * (M / 2 + 1) + k];
            n3 += (long double)(c.x - e.x) * (c.x - e.x) + (long double)(c.y - e.y) * (c.y - e.y);
          }
        }
      snprintf(name, sizeof name, "r2c wave-packed w%d%s", WAVES, SPLIT ? " split" : "");
      report(name, N, pname<T>(), (double)sqrtl(nn / dd), 4 * tol_of<T>());
      snprintf(name, sizeof name, "r2c wave-packed valid<M+1 w%d", WAVES);
      report(name, N, pname<T>(), (double)sqrtl(n3 / dd), 4 * tol_of<T>());
    }
    for (int r = 0; r < nrows; ++r) { out[(size_t)r * pout].y = (T)3.5; out[(size_t)r * pout + M].y = (T)-2.25; }
    {
      typedef C2RFft<S, T, ROWS, false, false, false, SPLIT, true> K;
      static_assert(K::THREADS == WAVES * 64, "whole waves");
      RealParams<T> P{out.data(), back.data(), tw.data(), rtw.data(), pout, pin, nrows, M + 1, (T)(1.0 / N)};
      emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    long double num = 0, den = 0;
    for (int r = 0; r < nrows; ++r)
      for (int i = 0; i < N; ++i) {
        long double d = (long double)back[(size_t)r * pin + i] - in[(size_t)r * pin + i];
        num += d * d;
        den += (long double)in[(size_t)r * pin + i] * in[(size_t)r * pin + i];
      }
    snprintf(name, sizeof name, "c2r wave-packed w%d%s", WAVES, SPLIT ? " split" : "");
    report(name, N, pname<T>(), (double)sqrtl(num / den), 4 * tol_of<T>());
    // column-limited: bins >= valid read as zero
    const int valid = M / 2 + 1;
    std::vector<cx<T>> part((size_t)nrows * valid);
    for (int r = 0; r < nrows; ++r)
      for (int k = 0; k < valid; ++k) part[(size_t)r * valid + k] = out[(size_t)r * pout + k];
    std::vector<T> b1((size_t)nrows * pin, (T)0), b2((size_t)nrows * pin, (T)0);
    {
      typedef C2RFft<S, T, ROWS, false, true, false, SPLIT, true> K;
      RealParams<T> P{part.data(), b1.data(), tw.data(), rtw.data(), valid, pin, nrows, valid, (T)(1.0 / N)};
      emu_launch((nrows + ROWS - 1) / ROWS, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    {
      typedef C2RFft<S, T, 2, false, true, false, false, false> K;      // the unpacked kernel as the reference
      RealParams<T> P{part.data(), b2.data(), tw.data(), rtw.data(), valid, pin, nrows, valid, (T)(1.0 / N)};
      emu_launch((nrows + 1) / 2, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    num = den = 0;
    for (size_t i = 0; i < b1.size(); ++i) { long double d = (long double)b1[i] - b2[i]; num += d * d; den += (long double)b2[i] * b2[i]; }
    snprintf(name, sizeof name, "c2r wave-packed valid<M+1 w%d", WAVES);
    report(name, N, pname<T>(), (double)sqrtl(num / den), 4 * tol_of<T>());
  }
}

// N = 3 L as three sub-transforms (fft_col3.h): plain forward / inverse (in place and out of place, two-level row maps on
// both sides, ragged last column tile), pad-on-load inverse and truncate-on-store forward with the Nyquist fold, against the
// long-double DFT of the logical data.
template <class SL, typename T, int COLS, int SPLIT, int VEC>
static void test_col3() {
  constexpr int L = SL::N, N = 3 * L;
  const int ncols = COLS + COLS / 2 + 1, nouter = 2;         // one full and one ragged tile per outer batch
  const int pitch = ncols + 3;
  std::mt19937_64 rng(4242 + N);
  std::uniform_real_distribution<double> U(-1, 1);
  auto tw = build_col3_twiddles<SL, T>();
  char name[96];
  for (int inv = 0; inv < 2; ++inv) {
    for (int inplace = 0; inplace < 2; ++inplace) {
      // rows through a two-level map: row r -> (r / split) * hi + (r % split) * lo
      const int split = N / 4, lo = pitch, hi = split * pitch + 5 * pitch;
      const size_t outer_stride = (size_t)4 * hi + 7;
      std::vector<cx<T>> in(outer_stride * nouter), out(outer_stride * nouter, mk<T>((T)7, (T)7));
      for (auto& z : in) z = mk<T>((T)U(rng), (T)U(rng));
      std::vector<cx<T>> src = in;
      ColParams<T> P;
      memset(&P, 0, sizeof P);
      P.in = in.data(); P.out = inplace ? in.data() : out.data(); P.tw = tw.data();
      P.in_outer = (i64)outer_stride; P.out_outer = (i64)outer_stride;
      P.in_map = make_rowmap(hi, lo, split, N); P.out_map = make_rowmap(hi, lo, split, N);
      P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = nouter; P.remap = 1; P.scale = (T)(inv ? 1.0 / N : 1.0);
      auto run = [&](auto k) {
        typedef decltype(k) K;
        emu_launch(P.ntile_c * nouter, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      };
      if (inv) run(ColFft3<SL, T, COLS, true, true, SPLIT, VEC, false, 0>{});
      else run(ColFft3<SL, T, COLS, false, false, SPLIT, VEC, false, 0>{});
      const std::vector<cx<T>>& res = inplace ? in : out;
      long double num = 0, den = 0;
      for (int o = 0; o < nouter; ++o)
        for (int cc = 0; cc < ncols; cc += 3) {
          lvec x(N);
          for (int r = 0; r < N; ++r) {
            const cx<T> z = src[o * outer_stride + (size_t)(r / split) * hi + (size_t)(r % split) * lo + cc];
            x[r].x = z.x; x[r].y = z.y;
          }
          lvec X = naive_dft(x, inv ? +1 : -1);
          for (int r = 0; r < N; ++r) {
            const cx<T> g = res[o * outer_stride + (size_t)(r / split) * hi + (size_t)(r % split) * lo + cc];
            const long double s = inv ? 1.0L / N : 1.0L;
            num += (g.x - X[r].x * s) * (g.x - X[r].x * s) + (g.y - X[r].y * s) * (g.y - X[r].y * s);
            den += X[r].x * s * X[r].x * s + X[r].y * s * X[r].y * s;
          }
        }
      snprintf(name, sizeof name, "col3 c%d v%d %s%s%s", COLS, VEC, inv ? "inv" : "fwd", inplace ? " in place" : "", SPLIT ? " split" : "");
      report(name, N, pname<T>(), (double)sqrtl(num / den), 2 * tol_of<T>());
    }
  }
  // 3/2-rule: inverse with the zero band on input (2L physical rows), forward with the truncated output + Nyquist fold
  {
    const int n = 2 * L;
    std::vector<cx<T>> phys((size_t)n * pitch), big((size_t)N * pitch, mk<T>((T)7, (T)7)), back((size_t)n * pitch, mk<T>((T)7, (T)7));
    for (auto& z : phys) z = mk<T>((T)U(rng), (T)U(rng));
    ColParams<T> P;
    memset(&P, 0, sizeof P);
    P.in = phys.data(); P.out = big.data(); P.tw = tw.data();
    P.in_map = make_rowmap(0, pitch, 0, n); P.out_map = make_rowmap(0, pitch, 0, N);
    P.ncols = ncols; P.ntile_c = (ncols + COLS - 1) / COLS; P.nouter = 1; P.remap = 0; P.scale = (T)1;
    {
      typedef ColFft3<SL, T, COLS, true, true, SPLIT, VEC, false, 1> K;
      emu_launch(P.ntile_c, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
    }
    long double num = 0, den = 0;
    for (int cc = 0; cc < ncols; cc += 2) {
      lvec x(N);
      for (int r = 0; r < N; ++r) { x[r].x = 0; x[r].y = 0; }
      for (int r = 0; r < L; ++r) {
        x[r].x = phys[(size_t)r * pitch + cc].x; x[r].y = phys[(size_t)r * pitch + cc].y;
        x[2 * L + r].x = phys[(size_t)(L + r) * pitch + cc].x; x[2 * L + r].y = phys[(size_t)(L + r) * pitch + cc].y;
      }
      lvec X = naive_dft(x, +1);
      for (int r = 0; r < N; ++r) {
        const cx<T> g = big[(size_t)r * pitch + cc];
        num += (g.x - X[r].x) * (g.x - X[r].x) + (g.y - X[r].y) * (g.y - X[r].y);
        den += X[r].x * X[r].x + X[r].y * X[r].y;
      }
    }
    snprintf(name, sizeof name, "col3 c%d v%d pad-on-load inv%s", COLS, VEC, SPLIT ? " split" : "");
    report(name, N, pname<T>(), (double)sqrtl(num / den), 2 * tol_of<T>());
    // the same with one third of a tile's transform per workgroup (ColFft3S): three times the workgroups, XCD-aware order
    // and plain order, against the kernel above (same arithmetic up to the order of the radix-3 step's sums)
    for (int remap = 0; remap < 2; ++remap) {
      std::vector<cx<T>> big2((size_t)N * pitch, mk<T>((T)7, (T)7));
      P.out = big2.data(); P.remap = remap;
      typedef ColFft3S<SL, T, COLS, true, true, SPLIT, VEC, false, 1> K;
      emu_launch(3 * P.ntile_c, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      num = den = 0;
      for (int r = 0; r < N; ++r)
        for (int cc = 0; cc < pitch; ++cc) {
          const cx<T> g = big2[(size_t)r * pitch + cc], w0 = big[(size_t)r * pitch + cc];
          num += (long double)(g.x - w0.x) * (g.x - w0.x) + (long double)(g.y - w0.y) * (g.y - w0.y);
          den += (long double)w0.x * w0.x + (long double)w0.y * w0.y;
        }
      snprintf(name, sizeof name, "col3s c%d v%d pad-on-load inv, a third per workgroup%s%s", COLS, VEC, remap ? " remap" : "", SPLIT ? " split" : "");
      report(name, N, pname<T>(), (double)sqrtl(num / den), 2 * tol_of<T>());
    }
    P.out = big.data(); P.remap = 0;
    for (int fold = 0; fold < 2; ++fold) {
      std::vector<cx<T>> full((size_t)N * pitch);
      for (auto& z : full) z = mk<T>((T)U(rng), (T)U(rng));
      P.in = full.data(); P.out = back.data(); P.in_map = make_rowmap(0, pitch, 0, N); P.out_map = make_rowmap(0, pitch, 0, n);
      P.fold = fold; P.scale = (T)0.5;
      {
        typedef ColFft3<SL, T, COLS, false, false, SPLIT, VEC, false, 2> K;
        emu_launch(P.ntile_c, K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
      }
      num = den = 0;
      for (int cc = 0; cc < ncols; cc += 2) {
        lvec x(N);
        for (int r = 0; r < N; ++r) { x[r].x = full[(size_t)r * pitch + cc].x; x[r].y = full[(size_t)r * pitch + cc].y; }
        lvec X = naive_dft(x, -1);
        for (int r = 0; r < n; ++r) {
          cx<long double> want = r < L ? X[r] : X[L + r];                  // physical row L + i = logical row 2L + i
          if (fold && r == L) { want.x += X[L].x; want.y += X[L].y; }
          const cx<T> g = back[(size_t)r * pitch + cc];
          num += (g.x - 0.5L * want.x) * (g.x - 0.5L * want.x) + (g.y - 0.5L * want.y) * (g.y - 0.5L * want.y);
          den += 0.25L * (want.x * want.x + want.y * want.y);
        }
      }
      snprintf(name, sizeof name, "col3 c%d v%d truncate-on-store fwd%s%s", COLS, VEC, fold ? " fold" : "", SPLIT ? " split" : "");
      report(name, N, pname<T>(), (double)sqrtl(num / den), 2 * tol_of<T>());
    }
    snprintf(name, sizeof name, "col3 c%d v%d truncate-on-store fwd, wrapped input columns%s", COLS, VEC, SPLIT ? " split" : "");
    test_wrapped_columns<ColFft3<SL, T, COLS, false, false, SPLIT, VEC, false, 2>, T>(name, N, n, tw.data(), COLS);
    snprintf(name, sizeof name, "col3 c%d v%d plain fwd, wrapped input columns%s", COLS, VEC, SPLIT ? " split" : "");
    test_wrapped_columns<ColFft3<SL, T, COLS, false, false, SPLIT, VEC, false, 0>, T>(name, N, N, tw.data(), COLS);
  }
}

// ---------------------------------------------------------------------------
// fused nonlinear z stage (fft_nlz.h): out_f = rfft((irfft(a) x irfft(b))_f), rows of `valid` bins, against long-double DFTs
template <class S, typename T, int ROWS, bool TWLDS, bool SPLIT, bool WAVE = false>
static void test_nlz(int valid, bool inplace, int valid_in = 0) {      // valid_in: fewer input bins than stored ones (pruned 2/3-rule)
  typedef NlzFft<S, T, ROWS, TWLDS, SPLIT, WAVE> K;
  const int vin = valid_in > 0 ? valid_in : valid;
  const int M = S::N;
  const int nrows = 2 * ROWS + 3;                 // an odd count: the last pair has one row
  const int pin = valid + 2, pout = inplace ? pin : valid + 1;
  std::mt19937_64 rng(4242 + M + valid);
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<cx<T>> in[6], out[3];
  for (auto& f : in) {
    f.resize((size_t)nrows * pin);
    for (auto& z : f) z = mk<T>((T)U(rng), (T)U(rng));
  }
  std::vector<cx<T>> keep[6];
  for (int f = 0; f < 6; ++f) keep[f] = in[f];
  for (auto& f : out) f.assign((size_t)nrows * pout, mk<T>((T)7, (T)7));
  auto tw = build_pass_twiddles<S, T>();
  NlzParams<T> P;
  for (int f = 0; f < 3; ++f) { P.a[f] = in[f].data(); P.b[f] = in[3 + f].data(); P.out[f] = inplace ? in[f].data() : out[f].data(); }
  P.tw = tw.data(); P.in_stride = pin; P.out_stride = pout; P.nrows = nrows; P.valid = valid; P.valid_in = vin;
  P.scale = (T)(1.0 / ((double)M * (double)M));
  emu_launch((nrows + 2 * ROWS - 1) / (2 * ROWS), K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    std::vector<std::vector<long double>> re(6, std::vector<long double>(M));
    for (int f = 0; f < 6; ++f) {
      lvec X(M);
      for (int p = 0; p < M; ++p) { X[p].x = 0; X[p].y = 0; }
      for (int q = 0; q < vin; ++q) {
        cx<T> z = keep[f][(size_t)r * pin + q];
        long double zr = z.x, zi = z.y;
        if (q == 0 || (M % 2 == 0 && q == M / 2)) zi = 0;
        X[q].x = zr; X[q].y = zi;
        if (q != 0 && q != M - q) { X[M - q].x = zr; X[M - q].y = -zi; }
      }
      lvec x = naive_dft(X, +1);
      for (int p = 0; p < M; ++p) re[f][p] = x[p].x / M;
    }
    for (int c = 0; c < 3; ++c) {
      const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      lvec x(M);
      for (int p = 0; p < M; ++p) { x[p].x = re[c1][p] * re[3 + c2][p] - re[c2][p] * re[3 + c1][p]; x[p].y = 0; }
      lvec X = naive_dft(x, -1);
      const cx<T>* g = (inplace ? in[c].data() : out[c].data()) + (size_t)r * pout;
      for (int q = 0; q < valid; ++q) {
        num += (g[q].x - X[q].x) * (g[q].x - X[q].x) + (g[q].y - X[q].y) * (g[q].y - X[q].y);
        den += X[q].x * X[q].x + X[q].y * X[q].y;
      }
      if (!inplace)
        for (int q = valid; q < pout; ++q)
          if (g[q].x != (T)7 || g[q].y != (T)7) { num += 1; }      // nothing is stored beyond the valid bins
    }
  }
  char name[64];
  snprintf(name, sizeof name, "nlz r%d v%d/%d%s%s%s%s", ROWS, vin, valid, TWLDS ? " twlds" : "", SPLIT ? " split" : "", inplace ? " inpl" : "", WAVE ? " wave" : "");
  report(name, M, pname<T>(), (double)sqrtl(num / den), sizeof(T) == 8 ? 4e-14 : 2e-5);
}
// the pruned 3/2-rule flavour (Nlz3Fft): M = 3 L, L + 1 bins per row, three sub-transforms per row
template <class SL, typename T, int ROWS, bool TWLDS>
static void test_nlz3(bool inplace) {
  typedef Nlz3Fft<SL, T, ROWS, TWLDS> K;
  const int L = SL::N, M = 3 * L, valid = L + 1;
  const int nrows = 2 * ROWS + 3;
  const int pin = valid + 2, pout = inplace ? pin : valid + 1;
  std::mt19937_64 rng(777 + M);
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<cx<T>> in[6], out[3], keep[6];
  for (auto& f : in) {
    f.resize((size_t)nrows * pin);
    for (auto& z : f) z = mk<T>((T)U(rng), (T)U(rng));
  }
  for (int f = 0; f < 6; ++f) keep[f] = in[f];
  for (auto& f : out) f.assign((size_t)nrows * pout, mk<T>((T)7, (T)7));
  auto tw = build_pass_twiddles<SL, T>();
  auto rt = build_nlz3_twiddles<T>(L);
  NlzParams<T> P;
  for (int f = 0; f < 3; ++f) { P.a[f] = in[f].data(); P.b[f] = in[3 + f].data(); P.out[f] = inplace ? in[f].data() : out[f].data(); }
  P.tw = tw.data(); P.rt3 = rt.data(); P.in_stride = pin; P.out_stride = pout; P.nrows = nrows; P.valid = valid;
  P.scale = (T)(1.0 / ((double)M * (double)M));
  emu_launch((nrows + 2 * ROWS - 1) / (2 * ROWS), K::THREADS, K::LDS_BYTES, [&](int b, int t, char* lds) { K::body(P, b, t, lds); });
  long double num = 0, den = 0;
  for (int r = 0; r < nrows; ++r) {
    std::vector<std::vector<long double>> re(6, std::vector<long double>(M));
    for (int f = 0; f < 6; ++f) {
      lvec X(M);
      for (int p = 0; p < M; ++p) { X[p].x = 0; X[p].y = 0; }
      for (int q = 0; q < valid; ++q) {
        cx<T> z = keep[f][(size_t)r * pin + q];
        long double zr = z.x, zi = z.y;
        if (q == 0) zi = 0;
        X[q].x = zr; X[q].y = zi;
        if (q != 0) { X[M - q].x = zr; X[M - q].y = -zi; }
      }
      lvec x = naive_dft(X, +1);
      for (int p = 0; p < M; ++p) re[f][p] = x[p].x / M;
    }
    for (int c = 0; c < 3; ++c) {
      const int c1 = (c + 1) % 3, c2 = (c + 2) % 3;
      lvec x(M);
      for (int p = 0; p < M; ++p) { x[p].x = re[c1][p] * re[3 + c2][p] - re[c2][p] * re[3 + c1][p]; x[p].y = 0; }
      lvec X = naive_dft(x, -1);
      const cx<T>* g = (inplace ? in[c].data() : out[c].data()) + (size_t)r * pout;
      for (int q = 0; q < valid; ++q) {
        num += (g[q].x - X[q].x) * (g[q].x - X[q].x) + (g[q].y - X[q].y) * (g[q].y - X[q].y);
        den += X[q].x * X[q].x + X[q].y * X[q].y;
      }
      if (!inplace)
        for (int q = valid; q < pout; ++q)
          if (g[q].x != (T)7 || g[q].y != (T)7) { num += 1; }
    }
  }
  char name[64];
  snprintf(name, sizeof name, "nlz3 r%d%s%s", ROWS, TWLDS ? " twlds" : "", inplace ? " inpl" : "");
  report(name, M, pname<T>(), (double)sqrtl(num / den), sizeof(T) == 8 ? 4e-14 : 2e-5);
}
template <class SL> static void test_nlz3_all() {
  test_nlz3<SL, double, 2, true>(false);
  test_nlz3<SL, double, 1, false>(true);
  test_nlz3<SL, float, 3, false>(false);
  test_nlz3<SL, float, 2, true>(true);
}
template <class S> static void test_nlz_all() {
  const int M = S::N;
  const int full = M / 2 + 1, lim = M / 3 + 1;     // every bin / the bins of the un-padded mesh under the 3/2-rule
  if constexpr (S::TPT <= 64 && 64 % S::TPT == 0) {      // the wave-synchronous build (in the emulator its wave barriers are workgroup barriers)
    test_nlz<S, double, 2, true, false, true>(lim, true);
    test_nlz<S, float, 3, false, false, true>(full, false);
  }
  test_nlz<S, double, 2, true, false>(full, false);
  test_nlz<S, double, 1, false, true>(lim, true);
  test_nlz<S, float, 3, false, false>(lim, false);
  test_nlz<S, float, 2, true, true>(full, true);
  if (lim < full) {                                 // pruned 2/3-rule: the kept kz bins in, every bin out
    test_nlz<S, double, 2, true, false>(full, true, lim);
    test_nlz<S, float, 3, false, false>(full, false, (2 * full) / 3);
  }
}

template <class S> static void test_chirpz_all() {
  const int nmax = (S::N + 1) / 2;
  for (int n : {nmax, nmax - 1, (S::N / 4) + 1, 7}) {
    if (n < 2 || 2 * n - 1 > S::N) continue;
    test_col_z<S, double, 4, false, false, 1>(n);
    test_col_z<S, double, 4, true, true, 1>(n);
    test_col_z<S, float, 8, false, false, 2>(n);
    test_col_z<S, float, 8, true, true, 2>(n);
    test_row_z<S, double, 2, false>(n);
    test_row_z<S, float, 3, true>(n);
    test_real_z<S, double, 2>(n);
    test_real_z<S, float, 3>(n);
    test_real_zh<S, double, 2>(n);
    test_real_zh<S, float, 3>(n);
  }
}

template <class S> static void test_spec_all() {
  test_col<S, double, 4, false, true>(false);
  test_col<S, double, 4, true, false>(true);
  test_col<S, float, 8, false, false>(true);
  test_col<S, float, 8, true, true>(false);
  test_col<S, double, 8, false, true, true>(true);
  test_col<S, float, 4, true, false, true>(false);
  test_col<S, float, 8, false, true, false, 2>(true);
  test_col<S, float, 8, true, false, true, 2>(false);
  test_col<S, float, 16, false, false, false, 4>(true);
  if constexpr (S::E % 2 == 0 && S::NP > 1) {        // the four-round exchange (fft_kernels.h XchQuarterV)
    test_col<S, double, 8, false, false, 2>(true);
    test_col<S, double, 4, true, true, 2>(false);
    test_col<S, float, 8, true, false, 2, 2>(true);
  }
  test_row<S, double, 2, false, true>();
  test_row<S, double, 2, true, false>();
  test_row<S, float, 3, false, false>();
  test_real<S, double, 2, true>();
  test_real<S, float, 3, false>();
  test_col_mask<S, double, 4, 1>();
  test_col_mask<S, float, 8, 2>();
  test_col_band<S, double, 4, 1>();
  test_col_band<S, float, 8, 2>();
  if constexpr (S::NP > 1 && (S::E >= 12 || S::N == 64)) {   // split re/im exchange of the contiguous-axis kernels (registry.h row_split)
    test_row<S, double, 2, false, false, true>();
    test_row<S, double, 2, true, false, true>();
    test_real<S, double, 2, false, true>();
  }
  if constexpr (S::NP > 1 && !(S::TPT <= 64 && 64 % S::TPT == 0)) {      // mirrors through LDS (C2RFft MLDS): the plans no shuffle serves
    test_real<S, double, 2, false, false, true>();
    test_real<S, float, 3, true, false, true>();
    if constexpr (S::E >= 12) test_real<S, double, 2, false, true, true>();
  }
  PadTests<S>::run();
  test_c2r_wave_packed<S, double, 1, false>();
  test_c2r_wave_packed<S, float, 2, false>();
  if constexpr (S::NP > 1 && S::E >= 12) test_c2r_wave_packed<S, double, 2, true>();
}

// The plan list is split over EMU_PART = 0..5 so that the parts compile (and run) in parallel;
// without EMU_PART everything goes into one binary.
#ifndef EMU_PART
#define EMU_PART -1
#endif
#define EMU_HAS(p) (EMU_PART < 0 || EMU_PART == (p))

int main() {
#if EMU_HAS(0)
  {  // validate the fast reference against the plain O(N^2) sum
    for (int n : {12, 40, 96, 250}) {
      lvec x(n);
      for (int i = 0; i < n; ++i) { x[i].x = sinl(1.0L + i * i); x[i].y = cosl(3.0L * i + 0.5L); }
      lvec a = naive_dft(x, -1), b = naive_dft_slow(x, -1);
      long double d = 0, r = 0;
      for (int i = 0; i < n; ++i) { d += (a[i].x - b[i].x) * (a[i].x - b[i].x) + (a[i].y - b[i].y) * (a[i].y - b[i].y); r += b[i].x * b[i].x + b[i].y * b[i].y; }
      report("reference self-check", n, "ldbl", (double)sqrtl(d / r), 1e-17);
    }
  }
#endif
  // every plan in plans.h is exercised
#define MFFT_PLAN(N, ...) test_spec_all<Spec<N, __VA_ARGS__>>();
#if EMU_HAS(0)
  MFFT_PLANS_A(MFFT_PLAN)
  test_chirpz_all<Spec<16, 16>>();
  test_chirpz_all<Spec<64, 8, 8>>();
  test_chirpz_all<Spec<96, 8, 4, 3>>();
  test_chirpz_all<Spec<512, 8, 8, 8>>();
  test_chirpz_all<Spec<160, 4, 4, 5, 2>>();
#endif
#if EMU_HAS(1)
  MFFT_PLANS_B(MFFT_PLAN) MFFT_PLANS_C(MFFT_PLAN) MFFT_COLPLANS_F32_C(MFFT_PLAN) MFFT_COLPLANS_F64_B(MFFT_PLAN)
#endif
#if EMU_HAS(2)
  MFFT_PLANS_D(MFFT_PLAN) MFFT_PLANS_E(MFFT_PLAN) MFFT_COLPLANS_F64_E(MFFT_PLAN)
  // three sub-transforms per workgroup: small stand-ins for every code path, then the shipped plans
  test_col3<Spec<16, 4, 4>, double, 4, 0, 1>();
  test_col3<Spec<16, 4, 4>, float, 8, 1, 2>();
  test_col3<Spec<32, 8, 4>, double, 2, 1, 1>();
#define MFFT_COL3(N, L, ...) test_col3<Spec<L, __VA_ARGS__>, double, 4, 0, 1>(); test_col3<Spec<L, __VA_ARGS__>, float, 4, 1, 2>();
  MFFT_COL3PLANS_E(MFFT_COL3)
#undef MFFT_COL3
#endif
#if EMU_HAS(3)
  MFFT_PLANS_F(MFFT_PLAN) MFFT_PLANS_G(MFFT_PLAN)
#endif
#if EMU_HAS(4)
  MFFT_PLANS_H(MFFT_PLAN) MFFT_PLANS_I(MFFT_PLAN) MFFT_PLANS_J(MFFT_PLAN) MFFT_COLPLANS_F64_I(MFFT_PLAN)
#endif
#if EMU_HAS(6)
  MFFT_PLANS_K(MFFT_PLAN)
#endif
#if EMU_HAS(7)
  MFFT_PLANS_L(MFFT_PLAN)
#endif
#if EMU_HAS(8)
  MFFT_PLANS_M(MFFT_PLAN)
  MFFT_COLPLANS_F64_M(MFFT_PLAN)
#endif
#if EMU_HAS(9)
  MFFT_PLANS_N(MFFT_PLAN)
  MFFT_PLANS_Q(MFFT_PLAN)      // round 5: 4608 ... 7168
#endif
#if EMU_HAS(10)
  MFFT_PLANS_O(MFFT_PLAN)
  MFFT_PLANS_R(MFFT_PLAN)      // round 5: 21 * 2^a, radix 42
#endif
#if EMU_HAS(13)
  MFFT_PLANS_S(MFFT_PLAN)      // round 6: 35 * 2^a, radix 70 (shipped in single precision)
#endif
#if EMU_HAS(14)
  MFFT_PLANS_T(MFFT_PLAN)      // round 6: 27 * 2^a, the 3/2-rule images of the 9 * 2^a meshes
#endif
#if EMU_HAS(15)
  MFFT_PLANS_U(MFFT_PLAN)      // round 6: 135 * 2^a, 1350 / 2700 / 2250, 675 / 1125
#endif
#if EMU_HAS(16)
  MFFT_PLANS_V(MFFT_PLAN)      // round 6: 81 * 2^a
#endif
#if EMU_HAS(17)
  MFFT_PLANS_W(MFFT_PLAN)      // round 6: 63 * 2^a
#endif
#if EMU_HAS(11)
  MFFT_PLANS_P(MFFT_PLAN)
  test_chirpz_all<Spec<8192, 32, 16, 16>>();
#endif
#if EMU_HAS(5)
  MFFT_FOR_EACH_ROWPLAN(MFFT_PLAN) MFFT_ROWPLANS_F64_K(MFFT_PLAN)
#endif
#if EMU_HAS(12)
#define MFFT_NLZ(N, ...) test_nlz_all<Spec<N, __VA_ARGS__>>();
  MFFT_NLZPLANS_P2(MFFT_NLZ) MFFT_NLZPLANS_3(MFFT_NLZ) MFFT_NLZPLANS_9(MFFT_NLZ)     // round 6: the fused nonlinear z stage
#undef MFFT_NLZ
#define MFFT_NLZ3(N, ...) test_nlz3_all<Spec<N, __VA_ARGS__>>();
  MFFT_NLZ3PLANS(MFFT_NLZ3)
  test_nlz3_all<Spec<256, 8, 8, 4>>();
#undef MFFT_NLZ3
#endif
#undef MFFT_PLAN
  printf("%s (%d failures)\n", g_fail ? "EMU TESTS FAILED" : "EMU TESTS PASSED", g_fail);
  return g_fail ? 1 : 0;
}
