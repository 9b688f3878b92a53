// twiddle.h -- host-side construction of the twiddle tables the kernels read.
// Values are evaluated in long double and rounded once to the target precision.
#pragma once
#include <cmath>
#include <vector>
#include "fft_core.h"

namespace mfft {

// inter-pass twiddles: for pass p >= 1 with radix R and accumulated length Ns,
//   tw[off(p) + (r-1)*Ns + w] = exp(-2 pi i r w / (Ns R)),  r = 1..R-1, w = 0..Ns-1
template <class S, typename T>
std::vector<cx<T>> build_pass_twiddles() {
  std::vector<cx<T>> tw(S::TW);
  tw[0] = mk<T>((T)1, (T)0);
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int p = 1; p < S::NP; ++p) {
    const int R = S::R(p), Ns = S::Ns(p), off = S::tw_off(p);
    for (int r = 1; r < R; ++r)
      for (int w = 0; w < Ns; ++w) {
        // reduce the angle exactly before calling the libm
        const long long num = ((long long)r * w) % ((long long)Ns * R);
        const long double a = two_pi * (long double)num / (long double)((long long)Ns * R);
        tw[off + (r - 1) * Ns + w] = mk<T>((T)cosl(a), (T)(-sinl(a)));
      }
  }
  return tw;
}

// real-transform twiddles exp(-2 pi i k / N), k = 0..N/2-1
template <typename T>
std::vector<cx<T>> build_real_twiddles(int N) {
  std::vector<cx<T>> tw(N / 2 > 0 ? N / 2 : 1);
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int k = 0; k < N / 2; ++k) {
    const long double a = two_pi * (long double)k / (long double)N;
    tw[k] = mk<T>((T)cosl(a), (T)(-sinl(a)));
  }
  return tw;
}

// pruned 3/2-rule z stage (fft_nlz.h Nlz3Fft), M = 3 L: exp(+2 pi i k / M), k = 0..L, then exp(+2 pi i 2k / M), k = 0..L
template <typename T>
std::vector<cx<T>> build_nlz3_twiddles(int L) {
  std::vector<cx<T>> tw(2 * (L + 1));
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int q = 1; q <= 2; ++q)
    for (int k = 0; k <= L; ++k) {
      const long double a = two_pi * (long double)((long long)q * k % (3LL * L)) / (long double)(3LL * L);
      tw[(q - 1) * (L + 1) + k] = mk<T>((T)cosl(a), (T)sinl(a));
    }
  return tw;
}

// ---- chirp-z (Bluestein) tables, fft_chirpz.h ------------------------------------
// chirp w[r] = exp(-i pi r^2 / n), r = 0..n-1; the angle is reduced exactly: r^2 mod 2n
template <typename T>
std::vector<cx<T>> build_chirp(int n) {
  std::vector<cx<T>> w(n);
  const long double pi = 3.141592653589793238462643383279503L;
  for (long long r = 0; r < n; ++r) {
    const long long q = (r * r) % (2LL * n);
    const long double a = pi * (long double)q / (long double)n;
    w[r] = mk<T>((T)cosl(a), (T)(-sinl(a)));
  }
  return w;
}

// bhat = FFT_M(b) / M for the wrapped filter b[m] = b[M-m] = conj(w[m]) (|m| < n), 0 elsewhere.
// b is even, so bhat[k] = (b[0] + 2 sum_{m=1}^{n-1} b[m] cos(2 pi k m / M)) / M: a direct O(M n)
// long-double sum (<= 8.4M terms at M = 4096), done once per (n, M).
template <typename T>
std::vector<cx<T>> build_chirp_filter(int n, int M) {
  const long double pi = 3.141592653589793238462643383279503L;
  std::vector<long double> br(n), bi(n), cs(M);
  for (long long m = 0; m < n; ++m) {
    const long long q = (m * m) % (2LL * n);
    const long double a = pi * (long double)q / (long double)n;
    br[m] = cosl(a);
    bi[m] = sinl(a);
  }
  for (int t = 0; t < M; ++t) cs[t] = cosl(2.0L * pi * (long double)t / (long double)M);
  std::vector<cx<T>> out(M);
  for (int k = 0; k < M; ++k) {
    long double sr = 0, si = 0;
    int t = 0;                         // k*m mod M, advanced incrementally
    for (int m = 1; m < n; ++m) {
      t += k;
      if (t >= M) t -= M;
      sr += br[m] * cs[t];
      si += bi[m] * cs[t];
    }
    out[k] = mk<T>((T)((br[0] + 2 * sr) / M), (T)((bi[0] + 2 * si) / M));
  }
  return out;
}

}  // namespace mfft
