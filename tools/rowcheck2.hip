// rowcheck2.hip -- round 5: WHERE does the hipcc miscompile of the 30- / 42-values contiguous-axis kernels have to be kept out, and
// what does each cure cost?  Builds with -DMFFT_LAUNDER_MODE=0 (none: wrong bins) / 1 (shipped: j's range hidden at its origin) /
// 2 (hidden only where pass_compute / pass_scatter divide by Ns) / 3 (pass_scatter only) / 4 (pass_compute only); checks the c2c row
// kernel of six plans against a host DFT and times the c2c and the r2c kernel over 2^16 rows.
//   for m in 0 1 2 3 4; do hipcc ... -DMFFT_LAUNDER_MODE=$m tools/rowcheck2.hip -o rowcheck2_$m; done
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include "fft_kernels.h"
#include "twiddle.h"
using namespace mfft;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class K, class P>
__global__ __launch_bounds__(K::THREADS) void kern(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  K::body(p, (int)blockIdx.x, (int)threadIdx.x, lds);
}

template <class S, typename T, int ROWS>
void check(const char* plan) {
  typedef RowFft<S, T, ROWS, false, false, false, true> K;
  typedef R2CFft<S, T, ROWS, false, false, false, true, false> KR;
  const int N = S::N, nrows = 1 << 16;
  std::vector<cx<T>> in((size_t)nrows * N), out((size_t)4 * ROWS * N);
  for (size_t i = 0; i < in.size(); ++i) in[i] = mk<T>((T)std::sin(0.37 * (double)(i % 100003) + 1.0), (T)std::cos(0.11 * (double)(i % 99991)));
  auto tw = build_pass_twiddles<S, T>();
  auto rtw = build_real_twiddles<T>(2 * N);
  cx<T>*din, *dout, *dtw, *drtw;
  CK(hipMalloc(&din, in.size() * sizeof(cx<T>))); CK(hipMalloc(&dout, (in.size() + (size_t)nrows) * sizeof(cx<T>)));
  CK(hipMalloc(&dtw, tw.size() * sizeof(cx<T>))); CK(hipMalloc(&drtw, rtw.size() * sizeof(rtw[0])));
  CK(hipMemcpy(din, in.data(), in.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  CK(hipMemcpy(drtw, rtw.data(), rtw.size() * sizeof(rtw[0]), hipMemcpyHostToDevice));
  RowParams<T> P;
  memset(&P, 0, sizeof P);
  P.in = din; P.out = dout; P.tw = dtw; P.in_stride = N; P.out_stride = N; P.nrows = nrows; P.scale = (T)1;
  P.zs = ZSplit{1, 1, 0, 0, 0, 1, 0};
  RealParams<T> Q;
  memset(&Q, 0, sizeof Q);
  Q.in = din; Q.out = dout; Q.tw = dtw; Q.rtw = reinterpret_cast<const cx<T>*>(drtw); Q.in_stride = 2 * N; Q.out_stride = N + 1; Q.nrows = nrows;
  Q.valid = N + 1; Q.scale = (T)1; Q.zs = ZSplit{1, 1, 0, 0, 0, 1, 0};
  if (K::LDS_BYTES > 65536) CK(hipFuncSetAttribute((const void*)kern<K, RowParams<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
  if (KR::LDS_BYTES > 65536) CK(hipFuncSetAttribute((const void*)kern<KR, RealParams<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, KR::LDS_BYTES));
  const int grid = (nrows + ROWS - 1) / ROWS;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_c = 0, ms_r = 0;
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kern<K, RowParams<T>>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, P);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((kern<K, RowParams<T>>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, P);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_c, e0, e1));
  CK(hipMemcpy(out.data(), dout, out.size() * sizeof(cx<T>), hipMemcpyDeviceToHost));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((kern<KR, RealParams<T>>), dim3(grid), dim3(KR::THREADS), KR::LDS_BYTES, 0, Q);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((kern<KR, RealParams<T>>), dim3(grid), dim3(KR::THREADS), KR::LDS_BYTES, 0, Q);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_r, e0, e1));
  long double num = 0, den = 0;
  for (int r = 0; r < 4 * ROWS; r += ROWS + 1) {
    for (int k = 0; k < N; ++k) {
      long double sx = 0, sy = 0;
      for (int n = 0; n < N; ++n) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)((long long)k * n % N) / N;
        const long double c = cosl(a), s = sinl(a), x = in[(size_t)r * N + n].x, y = in[(size_t)r * N + n].y;
        sx += x * c - y * s; sy += x * s + y * c;
      }
      const long double dx = out[(size_t)r * N + k].x - sx, dy = out[(size_t)r * N + k].y - sy;
      num += dx * dx + dy * dy; den += sx * sx + sy * sy;
    }
  }
  const double gb = 2.0 * (double)nrows * N * sizeof(cx<T>) / 1e9;
  printf("mode %d  %-20s %s rows=%d  c2c %.3f ms (%.0f GB/s)  r2c %.3f ms (%.0f GB/s)  c2c rel-L2 %.2e %s\n", MFFT_LAUNDER_MODE, plan,
         sizeof(T) == 8 ? "fp64" : "fp32", ROWS, ms_c / 10, gb / (ms_c / 10) * 1e3, ms_r / 10, gb / (ms_r / 10) * 1e3, (double)sqrtl(num / den),
         sqrtl(num / den) < (sizeof(T) == 8 ? 1e-12 : 1e-5) ? "ok" : "WRONG");
  CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dtw)); CK(hipFree(drtw));
}

int main() {
  check<Spec<360, 10, 6, 6>, double, 8>("360 10x6x6");
  check<Spec<480, 10, 6, 2, 2, 2>, double, 4>("480 10x6x2x2x2");
  check<Spec<600, 10, 10, 6>, double, 6>("600 10x10x6");
  check<Spec<720, 10, 6, 6, 2>, double, 4>("720 10x6x6x2");
  check<Spec<336, 42, 2, 2, 2>, double, 8>("336 42x2x2x2");
  check<Spec<672, 42, 2, 2, 2, 2>, double, 4>("672 42x2x2x2x2");
  check<Spec<480, 10, 6, 2, 2, 2>, float, 8>("480 10x6x2x2x2");
  check<Spec<720, 10, 6, 6, 2>, float, 4>("720 10x6x6x2");
  check<Spec<672, 42, 2, 2, 2, 2>, float, 4>("672 42x2x2x2x2");
  check<Spec<560, 70, 2, 2, 2>, float, 8>("560 70x2x2x2");        // round 6: the 70-values plans of 35 * 2^a (single precision only)
  return 0;
}
