// registry_nlz.h -- registration of the fused nonlinear z-stage kernels (fft_nlz.h); included by kernels_nlz_*.hip only, so
// that an edit of those kernels does not rebuild every translation unit.
#pragma once
#include "registry.h"
#include "fft_nlz.h"

namespace mfft {

// Round 6: the fused nonlinear z stage (fft_nlz.h NlzFft), one kernel per (length, precision).  256 threads per workgroup
// where a row has fewer; whole complex values through LDS while two workgroups (with their twiddle tables) fit a CU,
// real and imaginary parts one after the other beyond; the twiddles in LDS while the table stays under 48 KB.  The register
// cap asks for two waves per SIMD: the kernel parks 5 E real values per thread next to a transform's working set.
template <class S, typename T> constexpr int nlz_rows() { return 256 / S::TPT > 0 ? 256 / S::TPT : 1; }
template <class S, typename T> constexpr bool nlz_twlds() { return S::NP > 1 && (long long)S::TW * (int)sizeof(cx<T>) <= 49152; }
template <class S, typename T> constexpr bool nlz_split() {
  constexpr long long tw = nlz_twlds<S, T>() ? (long long)S::TW * (int)sizeof(cx<T>) : 0;
  return tw + (long long)padded_len<S::N, S::R(0)>() * nlz_rows<S, T>() * (int)sizeof(cx<T>) > 81920;
}
#ifndef MFFT_NLZ_OCC
#define MFFT_NLZ_OCC 2
#endif
// wave-synchronous build (fft_nlz.h NlzFft WAVE): rows inside one wave, no workgroup barriers in the transforms
#ifndef MFFT_NLZ_WAVE
#define MFFT_NLZ_WAVE 1
#endif
template <class S, typename T> constexpr bool nlz_wave() {
  return MFFT_NLZ_WAVE && S::TPT <= 64 && 64 % S::TPT == 0 && !nlz_split<S, T>();
}
template <class S, typename T>
void register_nlz(const char* name) {
  auto& reg = kernel_registry();
  constexpr int R = nlz_rows<S, T>();
  constexpr int W = MFFT_NLZ_OCC > 1 ? 16 + MFFT_NLZ_OCC : 0;        // waves per SIMD, said directly (registry.h mfft_kern_occ)
  reg.push_back(make_entry<NlzFft<S, T, R, nlz_twlds<S, T>(), nlz_split<S, T>(), nlz_wave<S, T>()>, NlzParams<T>, S, T, W>(FAM_NLZ, S::N, 0, R, name));
}

// ... and its pruned 3/2-rule flavour (Nlz3Fft: pad code 3, entry.n = M = 3 L): three thread groups of SL::TPT threads per row
template <class SL, typename T> constexpr int nlz3_rows() { return 256 / (3 * SL::TPT) > 0 ? 256 / (3 * SL::TPT) : 1; }
template <class SL, typename T>
void register_nlz3(const char* name) {
  auto& reg = kernel_registry();
  constexpr int R = nlz3_rows<SL, T>();
  reg.push_back(make_entry<Nlz3Fft<SL, T, R, true>, NlzParams<T>, SL, T>(FAM_NLZ, 3 * SL::N, 0, R, name));
  reg.back().pad = 3;
}

}  // namespace mfft
