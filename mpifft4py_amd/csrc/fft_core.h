// fft_core.h -- register-resident Stockham FFT building blocks for gfx950.
//
// Everything here is plain C++17 that compiles both as HIP device code (hipcc,
// --offload-arch=gfx950) and as host code (g++), so that the exact same index
// math can be exercised by the workgroup emulator in emu_test.cpp before a
// kernel ever reaches the GPU.  No CUDA-compat headers, no library FFT.
//
// Replaces (reference): the third-party per-rank FFT backends reached through
// mpiFFT4py/serialFFT/pyfftw_fft.py:26-203 and numpy_fft.py:25-107.
//
// Design (MI355X-first):
//  * A length-N transform is owned by TPT = N/E threads, each holding E complex
//    values in VGPRs.  A pass of radix R performs E/R butterflies per thread on
//    the registers v[m + r*E/R].  With that register layout EVERY pass reads the
//    positions  j + k*TPT (k = 0..E-1), so the first pass can be fed straight
//    from global memory, the last pass leaves the natural-order result in the
//    same registers (=> same global addresses: in-place safe), and LDS is only
//    used for the autosort exchange between passes.
//  * Inverse transforms reuse the forward butterflies through the re<->im swap
//    identity  ifft(x) = swap(fft(swap(x))), so there is one code path.
//  * Inter-pass twiddles come from a small per-(N,radix-sequence) table
//    (sum over passes of Ns*(R-1) entries, < N) that the kernels stage in LDS.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define MFFT_HD __host__ __device__ __forceinline__
#define MFFT_HDC __host__ __device__ constexpr
#else
#define MFFT_HD inline __attribute__((always_inline))
#define MFFT_HDC constexpr
#endif

namespace mfft {

template <typename T>
struct cx {
  T x, y;
};

template <typename T> MFFT_HD cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <typename T> MFFT_HD cx<T> operator+(cx<T> a, cx<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <typename T> MFFT_HD cx<T> operator-(cx<T> a, cx<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <typename T> MFFT_HD cx<T> operator*(cx<T> a, cx<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <typename T> MFFT_HD cx<T> scale(cx<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
template <typename T> MFFT_HD cx<T> conj(cx<T> a) { return mk<T>(a.x, -a.y); }
template <typename T> MFFT_HD cx<T> mul_mi(cx<T> a) { return mk<T>(a.y, -a.x); }   // a * (-i)
template <typename T> MFFT_HD cx<T> mul_pi(cx<T> a) { return mk<T>(-a.y, a.x); }   // a * (+i)
template <typename T> MFFT_HD cx<T> swapri(cx<T> a) { return mk<T>(a.y, a.x); }

// `a` if keep, +0 otherwise -- as an AND on the bit pattern, not a select.  For values that come out of an UNCONDITIONAL load of
// a clamped address ("this position does not exist: read position 0 instead and drop it"): hipcc turns `keep ? load : 0` into
// a branch around the load and then waits for every such load on its own inside its branch -- the loads of a thread, meant
// to be in flight together, become round trips to HBM one after the other (found in round 4 in the column-limited and the
// z-chunked c2r kernels: 12 - 32 serialised loads per thread, single precision 1024^3 3/2-rule c2r 7.7 ms -> see DESIGN.md).
// An AND cannot be turned into control flow, and unlike a multiplication by 0 it also drops a NaN / Inf at the clamped position.
MFFT_HD float keep_bits(float a, bool keep) {
  uint32_t u;
  memcpy(&u, &a, sizeof u);
  u &= keep ? 0xFFFFFFFFu : 0u;
  memcpy(&a, &u, sizeof u);
  return a;
}
MFFT_HD double keep_bits(double a, bool keep) {
  uint64_t u;
  memcpy(&u, &a, sizeof u);
  u &= keep ? 0xFFFFFFFFFFFFFFFFull : 0ull;
  memcpy(&a, &u, sizeof u);
  return a;
}
template <typename T> MFFT_HD cx<T> keep_bits(cx<T> a, bool keep) { return mk<T>(keep_bits(a.x, keep), keep_bits(a.y, keep)); }

// ---------------------------------------------------------------------------
// compile-time constants: cos/sin(2*pi*k/32), k = 0..8 (first octant+), the
// rest by symmetry.  Literals carry 21 significant digits.
// ---------------------------------------------------------------------------
MFFT_HDC long double c32_oct(int k) {
  return k == 0 ? 1.0L
       : k == 1 ? 0.980785280403230449126L
       : k == 2 ? 0.923879532511286756128L
       : k == 3 ? 0.831469612302545237079L
       : k == 4 ? 0.707106781186547524401L
       : k == 5 ? 0.555570233019602224743L
       : k == 6 ? 0.382683432365089771728L
       : k == 7 ? 0.195090322016128267848L
       : 0.0L;
}
// cos(2 pi k / 32), any k
MFFT_HDC long double cos32(int k) {
  k = ((k % 32) + 32) % 32;
  return k <= 8 ? c32_oct(k) : k <= 16 ? -c32_oct(16 - k) : k <= 24 ? -c32_oct(k - 16) : c32_oct(32 - k);
}
MFFT_HDC long double sin32(int k) { return cos32(k - 8); }

MFFT_HDC int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
MFFT_HDC int clcm(int a, int b) { return a / cgcd(a, b) * b; }

// ---------------------------------------------------------------------------
// In-register forward DFTs (sign -), natural order in / natural order out.
// ---------------------------------------------------------------------------
template <int R> struct Bfly;

template <> struct Bfly<1> {
  template <typename T> static MFFT_HD void run(cx<T> (&)[1]) {}
};

template <> struct Bfly<2> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[2]) {
    cx<T> a = v[0], b = v[1];
    v[0] = a + b;
    v[1] = a - b;
  }
};

template <> struct Bfly<3> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[3]) {
    const T s = (T)0.866025403784438646764L;   // sqrt(3)/2
    cx<T> t1 = v[1] + v[2];
    cx<T> t2 = mk<T>(v[0].x - (T)0.5 * t1.x, v[0].y - (T)0.5 * t1.y);
    cx<T> t3 = scale(v[1] - v[2], s);
    v[0] = v[0] + t1;
    v[1] = t2 + mul_mi(t3);
    v[2] = t2 + mul_pi(t3);
  }
};

template <> struct Bfly<4> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[4]) {
    cx<T> t0 = v[0] + v[2], t1 = v[0] - v[2];
    cx<T> t2 = v[1] + v[3], t3 = mul_mi(v[1] - v[3]);
    v[0] = t0 + t2;
    v[1] = t1 + t3;
    v[2] = t0 - t2;
    v[3] = t1 - t3;
  }
};

template <> struct Bfly<5> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[5]) {
    const T c1 = (T)0.309016994374947424102L;    // cos(2pi/5)
    const T c2 = (T)-0.809016994374947424102L;   // cos(4pi/5)
    const T s1 = (T)0.951056516295153572116L;    // sin(2pi/5)
    const T s2 = (T)0.587785252292473129169L;    // sin(4pi/5)
    cx<T> a1 = v[1] + v[4], a2 = v[2] + v[3];
    cx<T> b1 = v[1] - v[4], b2 = v[2] - v[3];
    cx<T> p1 = mk<T>(v[0].x + c1 * a1.x + c2 * a2.x, v[0].y + c1 * a1.y + c2 * a2.y);
    cx<T> p2 = mk<T>(v[0].x + c2 * a1.x + c1 * a2.x, v[0].y + c2 * a1.y + c1 * a2.y);
    cx<T> q1 = mk<T>(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y);
    cx<T> q2 = mk<T>(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y);
    v[0] = v[0] + a1 + a2;
    v[1] = p1 + mul_mi(q1);
    v[4] = p1 + mul_pi(q1);
    v[2] = p2 + mul_mi(q2);
    v[3] = p2 + mul_pi(q2);
  }
};

template <> struct Bfly<7> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[7]) {
    const T c1 = (T)0.623489801858733530525L;    // cos(2pi/7)
    const T c2 = (T)-0.222520933956314404289L;   // cos(4pi/7)
    const T c3 = (T)-0.900968867902419126236L;   // cos(6pi/7)
    const T s1 = (T)0.781831482468029808708L;    // sin(2pi/7)
    const T s2 = (T)0.974927912181823607018L;    // sin(4pi/7)
    const T s3 = (T)0.433883739117558120476L;    // sin(6pi/7)
    cx<T> a1 = v[1] + v[6], a2 = v[2] + v[5], a3 = v[3] + v[4];
    cx<T> b1 = v[1] - v[6], b2 = v[2] - v[5], b3 = v[3] - v[4];
    // X[k] = x0 + sum_n a_n cos(2 pi n k / 7) - i sum_n b_n sin(2 pi n k / 7); X[7 - k] its mirror
    cx<T> p1 = mk<T>(v[0].x + c1 * a1.x + c2 * a2.x + c3 * a3.x, v[0].y + c1 * a1.y + c2 * a2.y + c3 * a3.y);
    cx<T> p2 = mk<T>(v[0].x + c2 * a1.x + c3 * a2.x + c1 * a3.x, v[0].y + c2 * a1.y + c3 * a2.y + c1 * a3.y);
    cx<T> p3 = mk<T>(v[0].x + c3 * a1.x + c1 * a2.x + c2 * a3.x, v[0].y + c3 * a1.y + c1 * a2.y + c2 * a3.y);
    cx<T> q1 = mk<T>(s1 * b1.x + s2 * b2.x + s3 * b3.x, s1 * b1.y + s2 * b2.y + s3 * b3.y);
    cx<T> q2 = mk<T>(s2 * b1.x - s3 * b2.x - s1 * b3.x, s2 * b1.y - s3 * b2.y - s1 * b3.y);
    cx<T> q3 = mk<T>(s3 * b1.x - s1 * b2.x + s2 * b3.x, s3 * b1.y - s1 * b2.y + s2 * b3.y);
    v[0] = v[0] + a1 + a2 + a3;
    v[1] = p1 + mul_mi(q1);
    v[6] = p1 + mul_pi(q1);
    v[2] = p2 + mul_mi(q2);
    v[5] = p2 + mul_pi(q2);
    v[3] = p3 + mul_mi(q3);
    v[4] = p3 + mul_pi(q3);
  }
};

// multiply by W_32^K = exp(-2 pi i K / 32), K compile-time
template <int K, typename T> MFFT_HD cx<T> mul_w32(cx<T> a) {
  constexpr int k = ((K % 32) + 32) % 32;
  if constexpr (k == 0) {
    return a;
  } else if constexpr (k == 8) {
    return mul_mi(a);
  } else if constexpr (k == 16) {
    return mk<T>(-a.x, -a.y);
  } else if constexpr (k == 24) {
    return mul_pi(a);
  } else if constexpr (k == 4) {
    const T h = (T)0.707106781186547524401L;
    return mk<T>((a.x + a.y) * h, (a.y - a.x) * h);
  } else if constexpr (k == 12) {
    const T h = (T)0.707106781186547524401L;
    return mk<T>((a.y - a.x) * h, -(a.x + a.y) * h);
  } else if constexpr (k == 20) {
    const T h = (T)0.707106781186547524401L;
    return mk<T>(-(a.x + a.y) * h, (a.x - a.y) * h);
  } else if constexpr (k == 28) {
    const T h = (T)0.707106781186547524401L;
    return mk<T>((a.x - a.y) * h, (a.x + a.y) * h);
  } else {
    constexpr T c = (T)cos32(k);
    constexpr T s = (T)(-sin32(k));
    return mk<T>(a.x * c - a.y * s, a.x * s + a.y * c);
  }
}

// Cooley-Tukey composition R = RA * RB for the power-of-two radices 8/16/32:
//   X[a' + RA*b'] = sum_b W_RB^{b b'} W_R^{b a'} sum_a x[RB*a + b] W_RA^{a a'}
template <int R, int RA> struct BflyCT {
  static constexpr int RB = R / RA;
  template <int B, typename T> static MFFT_HD void stage1(cx<T> (&v)[R], cx<T> (&s)[R]) {
    if constexpr (B < RB) {
      cx<T> t[RA];
#pragma unroll
      for (int a = 0; a < RA; ++a) t[a] = v[RB * a + B];
      Bfly<RA>::run(t);
      tw_row<B, 0>(t, s);
      stage1<B + 1>(v, s);
    }
  }
  template <int B, int A, typename T> static MFFT_HD void tw_row(cx<T> (&t)[RA], cx<T> (&s)[R]) {
    if constexpr (A < RA) {
      s[B * RA + A] = mul_w32<(B * A) * (32 / R)>(t[A]);
      tw_row<B, A + 1>(t, s);
    }
  }
  template <typename T> static MFFT_HD void run(cx<T> (&v)[R]) {
    cx<T> s[R];
    stage1<0>(v, s);
#pragma unroll
    for (int a = 0; a < RA; ++a) {
      cx<T> t[RB];
#pragma unroll
      for (int b = 0; b < RB; ++b) t[b] = s[b * RA + a];
      Bfly<RB>::run(t);
#pragma unroll
      for (int b = 0; b < RB; ++b) v[a + RA * b] = t[b];
    }
  }
};

template <> struct Bfly<8> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[8]) { BflyCT<8, 4>::run(v); }
};
template <> struct Bfly<16> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[16]) { BflyCT<16, 4>::run(v); }
};
template <> struct Bfly<32> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[32]) { BflyCT<32, 4>::run(v); }
};
// Good-Thomas (prime-factor) composition R = RA * RB for COPRIME factors -- radices 6, 10, 15, 30: with the input index
// n = (RB n1 + RA n2) mod R and the output index k = (RB qb k1 + RA qa k2) mod R, qb = RB^-1 mod RA, qa = RA^-1 mod RB,
// the product n k mod R is RB n1 k1 * (RB qb) + RA n2 k2 * (RA qa), i.e. X[k1, k2] = sum_n1 W_RA^{n1 k1} sum_n2
// W_RB^{n2 k2} x[n1, n2]: two rounds of small butterflies and NO twiddles between them; both index maps are
// compile-time register renamings.  The plans with 3 and 5 among their factors (plans.h groups L, M) must keep 30 values
// per thread, which rules radix 4 out; these composites are what lets them run 3 - 6 passes instead of 6 - 8.
MFFT_HDC int cinv_mod(int a, int m) {   // a^-1 mod m (m small, gcd = 1)
  for (int x = 1; x < m; ++x)
    if ((a * x) % m == 1) return x;
  return 1;
}
template <int RA, int RB> struct BflyPFA {
  static constexpr int R = RA * RB;
  static constexpr int QB = cinv_mod(RB % RA, RA), QA = cinv_mod(RA % RB, RB);
  static_assert(cgcd(RA, RB) == 1, "prime-factor butterfly needs coprime factors");
  template <typename T> static MFFT_HD void run(cx<T> (&v)[R]) {
    cx<T> s[RA][RB];
#pragma unroll
    for (int n1 = 0; n1 < RA; ++n1) {
#pragma unroll
      for (int n2 = 0; n2 < RB; ++n2) s[n1][n2] = v[(RB * n1 + RA * n2) % R];
      Bfly<RB>::run(s[n1]);
    }
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) {
      cx<T> t[RA];
#pragma unroll
      for (int n1 = 0; n1 < RA; ++n1) t[n1] = s[n1][k2];
      Bfly<RA>::run(t);
#pragma unroll
      for (int k1 = 0; k1 < RA; ++k1) v[(RB * QB * k1 + RA * QA * k2) % R] = t[k1];
    }
  }
};
template <> struct Bfly<6> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[6]) { BflyPFA<2, 3>::run(v); }
};
template <> struct Bfly<10> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[10]) { BflyPFA<2, 5>::run(v); }
};
template <> struct Bfly<15> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[15]) { BflyPFA<3, 5>::run(v); }
};
template <> struct Bfly<12> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[12]) { BflyPFA<3, 4>::run(v); }
};
template <> struct Bfly<20> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[20]) { BflyPFA<5, 4>::run(v); }
};
template <> struct Bfly<24> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[24]) { BflyPFA<3, 8>::run(v); }
};
template <> struct Bfly<40> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[40]) { BflyPFA<5, 8>::run(v); }
};
template <> struct Bfly<30> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[30]) { BflyPFA<2, 15>::run(v); }
};
// round 4: the 7 * 2^a lengths (plans.h group O) hold 28 values per thread: radix 28 = 7 x 4 first, radix-4 / 2 passes after
template <> struct Bfly<14> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[14]) { BflyPFA<2, 7>::run(v); }
};
template <> struct Bfly<28> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[28]) { BflyPFA<7, 4>::run(v); }
};
// round 5: 21 * 2^a (672, 1344, 2688 ...: plans.h group R) hold 42 values per thread: radix 42 = 6 x 7 first, radix-2 passes after
template <> struct Bfly<42> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[42]) { BflyPFA<6, 7>::run(v); }
};
// round 6: 35 * 2^a in SINGLE precision (560, 1120, 2240: plans.h group S) hold 70 values per thread (140 VGPRs of data): radix 70 = 7 x 10
template <> struct Bfly<70> {
  template <typename T> static MFFT_HD void run(cx<T> (&v)[70]) { BflyPFA<7, 10>::run(v); }
};


// ---------------------------------------------------------------------------
// Transform specification: length N and its radix sequence (first pass first).
// ---------------------------------------------------------------------------
template <int... Rs> MFFT_HDC int rs_get(int p) {
  constexpr int r[sizeof...(Rs)] = {Rs...};
  return r[p];
}
template <int... Rs> MFFT_HDC int rs_lcm() {
  int e = 1;
  for (int p = 0; p < (int)sizeof...(Rs); ++p) e = clcm(e, rs_get<Rs...>(p));
  return e;
}
template <int... Rs> MFFT_HDC int rs_ns(int p) {   // product of the radices before pass p
  int s = 1;
  for (int q = 0; q < p; ++q) s *= rs_get<Rs...>(q);
  return s;
}
template <int... Rs> MFFT_HDC int rs_twoff(int p) {   // offset of pass p's twiddles (p >= 1)
  int o = 0;
  for (int q = 1; q < p; ++q) o += rs_ns<Rs...>(q) * (rs_get<Rs...>(q) - 1);
  return o;
}

template <int N_, int... Rs>
struct Spec {
  static constexpr int N = N_;
  static constexpr int NP = sizeof...(Rs);
  static constexpr int E = rs_lcm<Rs...>();   // complex values per thread
  static constexpr int TPT = N / E;           // threads per transform
  static constexpr int TW = rs_twoff<Rs...>(NP) > 0 ? rs_twoff<Rs...>(NP) : 1;   // table entries
  static MFFT_HDC int R(int p) { return rs_get<Rs...>(p); }
  static MFFT_HDC int Ns(int p) { return rs_ns<Rs...>(p); }
  static MFFT_HDC int tw_off(int p) { return rs_twoff<Rs...>(p); }
  static_assert(rs_ns<Rs...>(NP) == N, "radices must multiply to N");
  static_assert(E * TPT == N, "E must divide N");
};

// ---------------------------------------------------------------------------
// One pass on the registers of thread j (0 <= j < TPT).
//   tw  : pointer to the twiddle table (LDS or global), layout see Spec::tw_off
// ---------------------------------------------------------------------------
// hipcc (ROCm 7.2) miscompiles the contiguous-axis kernels of the plans with 3 and another odd prime among their radices (30 and
// 42 values per thread) when the thread's index j inside its transform has a known power-of-two range (j = tid % 8, % 16: 240,
// 480, 336, 672: a tenth to a third of the bins wrong on the device, exact in the emulator; tools/rowcheck.hip, round 3).  Round 5
// found WHERE (tools/rowcheck2.hip, profiles/r05_miscompile_cure_modes.txt): in the twiddle index `(j + m * TPT) % Ns` below and
// nowhere else -- hiding j's range only for this call (MFFT_LAUNDER_MODE 4: fft_kernels.h run_passes, the contiguous-axis kernels' driver) is as
// exact as hiding it at its origin (mode 1, rounds 3 - 4: fft_kernels.h row_thread_index) and leaves loads, stores, LDS slots
// and the wave shuffles of the real kernels their range information: r2c of 360 / 600 / 720 points 0.33 / 0.47 / 0.65 -> 0.20 /
// 0.31 / 0.54 ms per 2^16 rows, of 672 in single precision 0.45 -> 0.26.  (The c2r kernels keep the cure at j's origin: they
// compile to fewer registers that way, fft_kernels.h row_thread_index.)  Modes 0 (none), 2 (here and in pass_scatter, every
// caller), 3 (pass_scatter only: still wrong) exist for that tool.
#ifndef MFFT_LAUNDER_MODE
#define MFFT_LAUNDER_MODE 4
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define MFFT_HIDE_RANGE(x) asm volatile("" : "+v"(x))
#else
#define MFFT_HIDE_RANGE(x) ((void)0)
#endif
// (round 6: the 70-values plans of 35 * 2^a show it too -- 560: j = tid % 8, wrong bins along the contiguous axis in single precision)
template <class S> MFFT_HDC bool launder_plan() { return S::E % 15 == 0 || S::E % 21 == 0 || S::E % 35 == 0; }
template <class S, int P, typename T, class TwPtr>
MFFT_HD void pass_compute(cx<T> (&v)[S::E], int j, TwPtr tw) {
  constexpr int R = S::R(P), Ns = S::Ns(P), G = S::E / R, OFF = S::tw_off(P);
#if MFFT_LAUNDER_MODE == 2      /* tools/rowcheck2.hip only; mode 4 hides j in the caller (fft_kernels.h run_passes) */
  if constexpr (P > 0 && launder_plan<S>()) MFFT_HIDE_RANGE(j);
#endif
#pragma unroll
  for (int m = 0; m < G; ++m) {
    cx<T> t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) t[r] = v[m + r * G];
    if constexpr (P > 0) {
      const int w = (j + m * S::TPT) % Ns;
#pragma unroll
      for (int r = 1; r < R; ++r) t[r] = t[r] * tw[OFF + (r - 1) * Ns + w];
    }
    Bfly<R>::run(t);
#pragma unroll
    for (int r = 0; r < R; ++r) v[m + r * G] = t[r];
  }
}

// Autosort scatter of pass P's outputs: value r of butterfly jb goes to
// position (jb/Ns)*Ns*R + jb%Ns + r*Ns.  `put(pos, reg)` receives the
// (compile-time after unrolling) register index so that callers can move the
// whole complex value or only one component.
template <class S, int P, class Put>
MFFT_HD void pass_scatter(int j, Put put) {
  constexpr int R = S::R(P), Ns = S::Ns(P), G = S::E / R;
#if MFFT_LAUNDER_MODE == 2 || MFFT_LAUNDER_MODE == 3      /* tools/rowcheck2.hip only */
  if constexpr (launder_plan<S>()) MFFT_HIDE_RANGE(j);
#endif
#pragma unroll
  for (int m = 0; m < G; ++m) {
    const int jb = j + m * S::TPT;
    const int base = (jb / Ns) * (Ns * R) + (jb % Ns);
#pragma unroll
    for (int r = 0; r < R; ++r) put(base + r * Ns, m + r * G);
  }
}

// Gather for the next pass: register k receives position j + k*TPT.
template <class S, class Get>
MFFT_HD void pass_gather(int j, Get get) {
#pragma unroll
  for (int k = 0; k < S::E; ++k) get(j + k * S::TPT, k);
}

}  // namespace mfft
