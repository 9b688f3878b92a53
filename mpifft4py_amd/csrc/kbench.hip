// kbench.hip -- kernel-variant explorer (developer tool, not part of the library).
// Times strided-axis c2c variants (tile width, twiddle placement, split exchange,
// radix plan, XCD remap, row pitch) at the 1024^3 shapes of the slab path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 kbench.hip -o build/kbench && build/kbench [filter]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

struct Variant {
  std::string name;
  int cols, threads, lds;
  void (*launch)(const ColParams<double>&, int grid);
  std::vector<cx<double>> (*tw)();
};

template <class K>
void launch_k(const ColParams<double>& p, int grid) {
  static bool attr = false;
  if (!attr) {
    if (K::LDS_BYTES > 65536)
      CK(hipFuncSetAttribute((const void*)kern<K, ColParams<double>>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL((kern<K, ColParams<double>>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, p);
}

template <class S, int COLS, bool TWLDS, bool SPLIT>
Variant make(const char* plan) {
  typedef ColFft<S, double, COLS, false, TWLDS, SPLIT> K;
  char nm[128];
  snprintf(nm, sizeof nm, "%s c%d%s%s", plan, COLS, TWLDS ? " twlds" : "", SPLIT ? " split" : "");
  return Variant{nm, COLS, K::THREADS, K::LDS_BYTES, &launch_k<K>, &build_pass_twiddles<S, double>};
}

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  const int N = 1024, NF = 513;
  typedef Spec<1024, 16, 8, 8> SA;
  typedef Spec<1024, 32, 32> SB;
  typedef Spec<1024, 16, 16, 4> SC;
  typedef Spec<1024, 8, 8, 4, 4> SD;
  typedef Spec<1024, 32, 8, 4> SE;
  std::vector<Variant> vs = {
      make<SA, 4, false, false>("16x8x8"),  make<SA, 4, true, false>("16x8x8"),   make<SA, 4, false, true>("16x8x8"),
      make<SA, 4, true, true>("16x8x8"),    make<SA, 8, false, false>("16x8x8"),  make<SA, 8, true, false>("16x8x8"),
      make<SA, 8, false, true>("16x8x8"),   make<SA, 8, true, true>("16x8x8"),    make<SA, 2, true, false>("16x8x8"),
      make<SA, 16, false, true>("16x8x8"),
      make<SB, 8, false, false>("32x32"),   make<SB, 8, true, false>("32x32"),    make<SB, 8, false, true>("32x32"),
      make<SB, 8, true, true>("32x32"),     make<SB, 4, true, false>("32x32"),    make<SB, 16, true, true>("32x32"),
      make<SC, 8, true, true>("16x16x4"),   make<SC, 4, true, false>("16x16x4"),
      make<SD, 8, true, true>("8x8x4x4"),   make<SD, 4, true, false>("8x8x4x4"),  make<SD, 8, true, false>("8x8x4x4"),
      make<SE, 8, true, true>("32x8x4"),    make<SE, 4, true, false>("32x8x4"),
  };
  const size_t elems = (size_t)N * N * 520;
  cx<double>* buf = nullptr;
  CK(hipMalloc(&buf, elems * sizeof(cx<double>)));
  {
    std::vector<cx<double>> h((size_t)4 * N * 520);
    for (size_t i = 0; i < h.size(); ++i) h[i] = mk<double>((double)((i * 2654435761u) % 1000) / 1000.0 - 0.5, (double)((i * 40503u) % 977) / 977.0 - 0.5);
    for (size_t off = 0; off < elems; off += h.size())
      CK(hipMemcpy(buf + off, h.data(), std::min(h.size(), elems - off) * sizeof(cx<double>), hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double alg_bytes = 2.0 * N * N * NF * 16.0;
  printf("%-28s %5s %6s | %-34s\n", "variant", "thr", "ldsKB", "layout: ms (alg GB/s)");
  struct Layout { const char* name; int pitch; bool xdir; int remap; };
  const Layout layouts[] = {
      {"y p513", 513, false, 0}, {"y p513 rm", 513, false, 1}, {"y p520", 520, false, 0}, {"y p520 rm", 520, false, 1},
      {"x p513", 513, true, 0},  {"x p513 rm", 513, true, 1},  {"x p520 rm", 520, true, 1},
  };
  for (const Variant& v : vs) {
    if (filter[0] && !strstr(v.name.c_str(), filter)) continue;
    auto twh = v.tw();
    cx<double>* tw = nullptr;
    CK(hipMalloc(&tw, twh.size() * sizeof(cx<double>)));
    CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<double>), hipMemcpyHostToDevice));
    printf("%-28s %5d %6.1f |", v.name.c_str(), v.threads, v.lds / 1024.0);
    for (const Layout& L : layouts) {
      ColParams<double> P;
      P.in = buf;
      P.out = buf;
      P.tw = tw;
      P.remap = L.remap;
      P.scale = 1.0;
      if (!L.xdir) {
        P.in_outer = P.out_outer = (i64)N * L.pitch;
        P.in_map = P.out_map = make_rowmap(0, L.pitch, N, N);
        P.ncols = NF;
        P.nouter = N;
      } else {
        // x direction over the flattened (y, kz) index; with pitch 520 the padding columns are transformed too
        P.in_outer = P.out_outer = 0;
        P.in_map = P.out_map = make_rowmap(0, (i64)N * L.pitch, N, N);
        P.ncols = N * L.pitch;
        P.nouter = 1;
      }
      P.ntile_c = (P.ncols + v.cols - 1) / v.cols;
      const int grid = P.ntile_c * P.nouter;
      for (int i = 0; i < 2; ++i) v.launch(P, grid);
      CK(hipDeviceSynchronize());
      const int reps = 6;
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) v.launch(P, grid);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      const double bytes = L.xdir ? alg_bytes * L.pitch / 513.0 : alg_bytes;
      printf(" %s: %.2f (%4.0f)", L.name, ms, bytes / (ms * 1e-3) / 1e9);
      fflush(stdout);
    }
    printf("\n");
    CK(hipFree(tw));
  }
  return 0;
}
