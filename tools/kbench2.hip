// kbench2.hip -- second kernel-variant explorer (developer tool, not part of the library): strided-axis c2c
// variants for the LONG lengths (2048, 4096) in both precisions, where the 1024-thread / 128-VGPR
// configurations of the default heuristics spill.
//   make kbench2 && build/kbench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "fft_kernels.h"
#include "twiddle.h"

using namespace mfft;

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <class K, class P>
__global__ __launch_bounds__(K::THREADS) void kern(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  K::body(p, (int)blockIdx.x, (int)threadIdx.x, lds);
}

static void* g_buf = nullptr;
static size_t g_bytes = 0;

template <class S, typename T, int COLS, bool TWLDS, bool SPLIT, int VEC>
void run(const char* plan, int nouter, int pitch) {
  typedef ColFft<S, T, COLS, false, TWLDS, SPLIT, VEC> K;
  const int N = S::N;
  const size_t need = (size_t)nouter * N * pitch * sizeof(cx<T>);
  if (need > g_bytes) { printf("skip %s: buffer too small\n", plan); return; }
  auto twh = build_pass_twiddles<S, T>();
  cx<T>* tw = nullptr;
  CK(hipMalloc(&tw, twh.size() * sizeof(cx<T>)));
  CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  if (K::LDS_BYTES > 65536)
    CK(hipFuncSetAttribute((const void*)kern<K, ColParams<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("N=%d %s %-10s c%-2d v%d%s%s thr %4d lds %5.1fK |", N, sizeof(T) == 8 ? "f64" : "f32", plan, COLS, VEC,
         TWLDS ? " twlds" : "", SPLIT ? " split" : "", K::THREADS, K::LDS_BYTES / 1024.0);
  for (int xdir = 0; xdir < 2; ++xdir) {
    ColParams<T> P;
    P.in = P.out = static_cast<cx<T>*>(g_buf);
    P.tw = tw;
    P.remap = 1;
    P.fold = 0;
    P.scale = (T)1;
    if (!xdir) {
      P.in_outer = P.out_outer = (i64)N * pitch;
      P.in_map = P.out_map = make_rowmap(0, pitch, N, N);
      P.ncols = pitch;
      P.nouter = nouter;
    } else {
      P.in_outer = P.out_outer = 0;
      P.in_map = P.out_map = make_rowmap(0, (i64)nouter * pitch, N, N);
      P.ncols = nouter * pitch;
      P.nouter = 1;
    }
    P.ntile_c = (P.ncols + COLS - 1) / COLS;
    const int grid = P.ntile_c * P.nouter;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((kern<K, ColParams<T>>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, P);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((kern<K, ColParams<T>>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, P);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf(" %s p%d: %.2f ms (%4.0f GB/s)", xdir ? "x" : "y", pitch, ms, 2.0 * need / (ms * 1e-3) / 1e9);
  }
  printf("\n");
  fflush(stdout);
  CK(hipFree(tw));
}

int main() {
  g_bytes = (size_t)6 << 30;
  CK(hipMalloc(&g_buf, g_bytes));
  CK(hipMemset(g_buf, 0, g_bytes));
  for (int pitch : {1025}) {
    // 3- and 5-smooth lengths in fp32: VEC = 2 (two columns per lane) against VEC = 1
    run<Spec<1536, 8, 8, 8, 3>, float, 16, false, true, 2>("8x8x8x3", 256, pitch);
    run<Spec<1536, 8, 8, 8, 3>, float, 16, false, true, 1>("8x8x8x3", 256, pitch);
    run<Spec<1536, 16, 8, 4, 3>, float, 16, false, true, 1>("16x8x4x3", 256, pitch);
    run<Spec<768, 8, 8, 4, 3>, float, 16, false, false, 2>("8x8x4x3", 512, pitch);
    run<Spec<768, 8, 8, 4, 3>, float, 16, false, false, 1>("8x8x4x3", 512, pitch);
    run<Spec<1280, 8, 8, 4, 5>, float, 16, false, true, 2>("8x8x4x5", 256, pitch);
    run<Spec<1280, 8, 8, 4, 5>, float, 16, false, true, 1>("8x8x4x5", 256, pitch);
    run<Spec<1280, 8, 8, 4, 5>, float, 16, false, false, 1>("8x8x4x5", 256, pitch);
    run<Spec<640, 8, 4, 4, 5>, float, 16, false, false, 2>("8x4x4x5", 512, pitch);
    run<Spec<640, 8, 4, 4, 5>, float, 16, false, false, 1>("8x4x4x5", 512, pitch);
    run<Spec<512, 8, 8, 8>, float, 16, false, false, 2>("8x8x8", 1024, pitch);
    run<Spec<512, 8, 8, 8>, float, 16, false, false, 1>("8x8x8", 1024, pitch);
    run<Spec<3072, 8, 8, 4, 4, 3>, float, 16, false, true, 2>("8x8x4x4x3", 128, pitch);
    run<Spec<3072, 8, 8, 4, 4, 3>, float, 8, false, true, 1>("8x8x4x4x3", 128, pitch);
    run<Spec<2560, 8, 8, 8, 5>, float, 16, false, true, 2>("8x8x8x5", 128, pitch);
    run<Spec<2560, 8, 8, 8, 5>, float, 16, false, true, 1>("8x8x8x5", 128, pitch);
    // fp64 references for the same lengths
    run<Spec<1536, 8, 8, 8, 3>, double, 8, false, true, 1>("8x8x8x3", 128, pitch);
    run<Spec<1280, 8, 8, 4, 5>, double, 8, false, true, 1>("8x8x4x5", 128, pitch);
  }
  return 0;
}
