// rowcheck.hip -- contiguous-axis kernels of chosen radix plans / rows per workgroup against a host DFT (developer tool,
// round 3: the row kernels of 5x3x2x2x2x2 (240) and 5x3x2x2x2x2x2 (480) came out wrong on the GPU while the emulator and the
// strided kernels of the same plans were right).   make -C tools rowcheck && tools/build/rowcheck
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

template <class S, typename T, int ROWS, bool TWLDS, bool SPLIT>
void check(const char* plan) {
  typedef RowFft<S, T, ROWS, false, TWLDS, false, SPLIT> K;
  const int N = S::N, nrows = 2 * ROWS + 1;
  std::vector<cx<T>> in((size_t)nrows * N), out(in.size());
  for (size_t i = 0; i < in.size(); ++i) in[i] = mk<T>((T)std::sin(0.37 * i + 1.0), (T)std::cos(0.11 * i));
  auto tw = build_pass_twiddles<S, T>();
  cx<T>*din, *dout, *dtw;
  CK(hipMalloc(&din, in.size() * sizeof(cx<T>))); CK(hipMalloc(&dout, in.size() * sizeof(cx<T>))); CK(hipMalloc(&dtw, tw.size() * sizeof(cx<T>)));
  CK(hipMemcpy(din, in.data(), in.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  CK(hipMemcpy(dtw, tw.data(), tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  RowParams<T> P;
  memset(&P, 0, sizeof P);
  P.in = din; P.out = dout; P.tw = dtw; P.in_stride = N; P.out_stride = N; P.nrows = nrows; P.scale = (T)1;
  P.zs = ZSplit{1, 1, 0, 0, 0};
  if (K::LDS_BYTES > 65536) CK(hipFuncSetAttribute((const void*)kern<K, RowParams<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
  hipLaunchKernelGGL((kern<K, RowParams<T>>), dim3((nrows + ROWS - 1) / ROWS), dim3(K::THREADS), K::LDS_BYTES, 0, P);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(out.data(), dout, out.size() * sizeof(cx<T>), hipMemcpyDeviceToHost));
  long double num = 0, den = 0;
  int bad = 0;
  for (int r = 0; r < nrows; r += ROWS) {          // one row per workgroup is enough for the reference
    for (int k = 0; k < N; ++k) {
      long double sx = 0, sy = 0;
      for (int n = 0; n < N; ++n) {
        const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)((long long)k * n % N) / N;
        const long double c = cosl(a), s = sinl(a), x = in[(size_t)r * N + n].x, y = in[(size_t)r * N + n].y;
        sx += x * c - y * s; sy += x * s + y * c;
      }
      const long double dx = out[(size_t)r * N + k].x - sx, dy = out[(size_t)r * N + k].y - sy;
      num += dx * dx + dy * dy; den += sx * sx + sy * sy;
      if (sqrtl(dx * dx + dy * dy) > 1e-4 * sqrtl(den / (k + 1) + 1e-30)) ++bad;
    }
  }
  printf("%-22s %s rows=%-2d threads=%-4d %s%s  rel-L2 %.2e  %s\n", plan, sizeof(T) == 8 ? "fp64" : "fp32", ROWS, K::THREADS,
         TWLDS ? "twlds " : "", SPLIT ? "split" : "", (double)sqrtl(num / den), sqrtl(num / den) < (sizeof(T) == 8 ? 1e-12 : 1e-5) ? "ok" : "WRONG");
  CK(hipFree(din)); CK(hipFree(dout)); CK(hipFree(dtw));
}

int main() {
  check<Spec<240, 5, 3, 2, 2, 2, 2>, double, 8, true, false>("240 5x3x2x2x2x2");
  check<Spec<240, 5, 3, 2, 2, 2, 2>, double, 8, false, false>("240 5x3x2x2x2x2");
  check<Spec<240, 5, 3, 2, 2, 2, 2>, double, 16, true, false>("240 5x3x2x2x2x2");
  check<Spec<240, 5, 3, 2, 2, 2, 2>, double, 4, true, false>("240 5x3x2x2x2x2");
  check<Spec<240, 5, 3, 2, 2, 2, 2>, double, 2, true, false>("240 5x3x2x2x2x2");
  check<Spec<240, 5, 3, 2, 2, 2, 2>, float, 16, true, false>("240 5x3x2x2x2x2");
  check<Spec<240, 5, 3, 2, 2, 2, 2>, float, 8, true, false>("240 5x3x2x2x2x2");
  check<Spec<240, 3, 5, 2, 2, 2, 2>, double, 8, true, false>("240 3x5x2x2x2x2");
  check<Spec<240, 2, 5, 3, 2, 2, 2>, double, 8, true, false>("240 2x5x3x2x2x2");
  check<Spec<240, 5, 3, 4, 4>, double, 4, true, false>("240 5x3x4x4 (E=60)");
  check<Spec<480, 5, 3, 2, 2, 2, 2, 2>, double, 4, false, true>("480 5x3x2^5");
  check<Spec<480, 5, 3, 2, 2, 2, 2, 2>, double, 8, false, true>("480 5x3x2^5");
  check<Spec<480, 5, 3, 2, 2, 2, 2, 2>, double, 2, false, true>("480 5x3x2^5");
  check<Spec<480, 5, 3, 2, 2, 2, 2, 2>, double, 4, false, false>("480 5x3x2^5");
  check<Spec<480, 3, 5, 2, 2, 2, 2, 2>, double, 4, false, true>("480 3x5x2^5");
  check<Spec<480, 2, 5, 3, 2, 2, 2, 2>, double, 4, false, true>("480 2x5x3x2^4");
  check<Spec<480, 5, 3, 2, 2, 2, 2, 2>, float, 8, false, false>("480 5x3x2^5");
  check<Spec<120, 5, 3, 2, 2, 2>, double, 16, true, false>("120 5x3x2x2x2");
  check<Spec<960, 5, 3, 2, 2, 2, 2, 2, 2>, double, 2, false, true>("960 5x3x2^6");
  // the composite-radix plans that replaced them (plans.h groups L, M as shipped)
  check<Spec<240, 10, 6, 2, 2>, double, 8, false, true>("240 10x6x2x2");
  check<Spec<240, 10, 6, 2, 2>, double, 8, true, false>("240 10x6x2x2");
  check<Spec<240, 10, 6, 2, 2>, float, 8, true, false>("240 10x6x2x2");
  check<Spec<480, 10, 6, 2, 2, 2>, double, 4, false, true>("480 10x6x2x2x2");
  check<Spec<480, 10, 6, 2, 2, 2>, double, 8, false, true>("480 10x6x2x2x2");
  check<Spec<480, 10, 6, 2, 2, 2>, float, 8, false, false>("480 10x6x2x2x2");
  check<Spec<960, 10, 6, 2, 2, 2, 2>, double, 2, false, true>("960 10x6x2^4");
  return 0;
}
