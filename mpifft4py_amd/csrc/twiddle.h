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

}  // namespace mfft
