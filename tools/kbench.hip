// kbench.hip -- kernel-variant explorer (developer tool, not part of the library).
// Times strided-axis c2c variants (tile width, twiddle placement, split exchange,
// radix plan, XCD remap, row pitch) at the 1024^3 shapes of the slab path.
//   make -C tools kbench && tools/build/kbench [filter]
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

// ---- calibration kernels -------------------------------------------------------
struct alignas(16) v16 { double a, b; };
__global__ __launch_bounds__(256) void copy_stream(const v16* __restrict__ src, v16* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// same addressing as ColFft (COLS columns x 1024 rows per workgroup, E values per thread), no FFT
typedef double d2 __attribute__((ext_vector_type(2)));
template <int COLS, int E, int NT>
__global__ __launch_bounds__(1024 / E * COLS) void copy_tile_nt(ColParams<double> P) {
  constexpr int TPT = 1024 / E;
  const int bid = P.remap ? xcd_remap((int)blockIdx.x, P.ntile_c * P.nouter) : (int)blockIdx.x;
  const int outer = bid / P.ntile_c, tc = bid - outer * P.ntile_c;
  const int c = threadIdx.x % COLS, j = threadIdx.x / COLS, col = tc * COLS + c;
  if (col >= P.ncols) return;
  const d2* ip = reinterpret_cast<const d2*>(P.in + (i64)outer * P.in_outer + col);
  d2* op = reinterpret_cast<d2*>(P.out + (i64)outer * P.out_outer + col);
  d2 v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    const d2* a = ip + row_off(P.in_map, (unsigned)(j + k * TPT));
    v[k] = (NT & 1) ? __builtin_nontemporal_load(a) : *a;
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    d2* a = op + row_off(P.out_map, (unsigned)(j + k * TPT));
    d2 x = v[k];
    x.x += 1.0;
    if (NT & 2) __builtin_nontemporal_store(x, a); else *a = x;
  }
}
template <int COLS, int E, int NT>
void launch_copy_tile_nt(const ColParams<double>& p, int grid) {
  hipLaunchKernelGGL((copy_tile_nt<COLS, E, NT>), dim3(grid), dim3(1024 / E * COLS), 0, 0, p);
}

template <int COLS, int E>
__global__ __launch_bounds__(1024 / E * COLS) void copy_tile(ColParams<double> P) {
  constexpr int TPT = 1024 / E;
  const int bid = P.remap ? xcd_remap((int)blockIdx.x, P.ntile_c * P.nouter) : (int)blockIdx.x;
  const int outer = bid / P.ntile_c, tc = bid - outer * P.ntile_c;
  const int c = threadIdx.x % COLS, j = threadIdx.x / COLS, col = tc * COLS + c;
  if (col >= P.ncols) return;
  const cx<double>* ip = P.in + (i64)outer * P.in_outer + col;
  cx<double>* op = P.out + (i64)outer * P.out_outer + col;
  cx<double> v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = ip[row_off(P.in_map, (unsigned)(j + k * TPT))];
#pragma unroll
  for (int k = 0; k < E; ++k) op[row_off(P.out_map, (unsigned)(j + k * TPT))] = mk<double>(v[k].x + 1.0, v[k].y);
}

// persistent, register double-buffered variant of ColFft: the loads of the next
// tile are in flight while the current tile is transformed and stored.
template <class S, int COLS, bool SPLIT, bool LATE = false>
__global__ __launch_bounds__(S::TPT * COLS) void col_persist(ColParams<double> P) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef double T;
  const int ntiles = P.ntile_c * P.nouter;
  const int G = (int)gridDim.x;                 // multiple of 8
  const int b = (int)blockIdx.x;
  const int lane_tile = (b & 7) * (G >> 3) + (b >> 3);   // XCD x owns G/8 consecutive tiles of every step
  const int c = threadIdx.x % COLS, j = threadIdx.x / COLS;
  struct Slot { int c; __device__ int operator()(int pos) const { return pos * COLS + c; } };
  cx<T> cur[S::E], nxt[S::E];
  auto tile_ptrs = [&](int t, const cx<T>*& ip, cx<T>*& op, bool& active) {
    const int outer = t / P.ntile_c, tc = t - outer * P.ntile_c;
    const int col = tc * COLS + c;
    active = col < P.ncols;
    ip = P.in + (i64)outer * P.in_outer + col;
    op = P.out + (i64)outer * P.out_outer + col;
  };
  int t = lane_tile;
  {
    const cx<T>* ip; cx<T>* op; bool act;
    if (t < ntiles) {
      tile_ptrs(t, ip, op, act);
#pragma unroll
      for (int k = 0; k < S::E; ++k) cur[k] = act ? ip[row_off(P.in_map, (unsigned)(j + k * S::TPT))] : mk<T>(0, 0);
    }
  }
  for (; t < ntiles; t += G) {
    const int tn = t + G;
    auto prefetch = [&] {
      if (tn < ntiles) {
        const cx<T>* ip; cx<T>* op; bool act;
        tile_ptrs(tn, ip, op, act);
#pragma unroll
        for (int k = 0; k < S::E; ++k) nxt[k] = act ? ip[row_off(P.in_map, (unsigned)(j + k * S::TPT))] : mk<T>(0, 0);
      }
    };
    if constexpr (!LATE) prefetch();
    if constexpr (SPLIT) {
      XchSplit<T, Slot> xch{reinterpret_cast<T*>(lds), Slot{c}};
      run_passes<S, 0, T>(cur, j, P.tw, xch);
    } else {
      XchFull<T, Slot> xch{reinterpret_cast<cx<T>*>(lds), Slot{c}};
      run_passes<S, 0, T>(cur, j, P.tw, xch);
    }
    if constexpr (LATE) prefetch();      // next tile's loads in flight while this tile's stores drain
    {
      const cx<T>* ip; cx<T>* op; bool act;
      tile_ptrs(t, ip, op, act);
      if (act) {
#pragma unroll
        for (int k = 0; k < S::E; ++k) op[row_off(P.out_map, (unsigned)(j + k * S::TPT))] = scale(cur[k], P.scale);
      }
    }
    __syncthreads();      // the next tile's first scatter must not overtake this tile's last gather
#pragma unroll
    for (int k = 0; k < S::E; ++k) cur[k] = nxt[k];
  }
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

static int g_persist_grid = 256;
template <class S, int COLS, bool SPLIT, bool LATE = false>
void launch_persist(const ColParams<double>& p, int) {
  constexpr int LDS = S::N * COLS * (SPLIT ? 8 : 16);
  static bool attr = false;
  if (!attr) {
    if (LDS > 65536) CK(hipFuncSetAttribute((const void*)col_persist<S, COLS, SPLIT, LATE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr = true;
  }
  hipLaunchKernelGGL((col_persist<S, COLS, SPLIT, LATE>), dim3(g_persist_grid), dim3(S::TPT * COLS), LDS, 0, p);
}
template <class S, int COLS, bool SPLIT, bool LATE = false>
Variant make_persist(const char* plan) {
  char nm[128];
  snprintf(nm, sizeof nm, "persist%s %s c%d%s", LATE ? "-late" : "", plan, COLS, SPLIT ? " split" : "");
  return Variant{nm, COLS, S::TPT * COLS, S::N * COLS * (SPLIT ? 8 : 16), &launch_persist<S, COLS, SPLIT, LATE>, &build_pass_twiddles<S, double>};
}
template <int COLS, int E>
void launch_copy_tile(const ColParams<double>& p, int grid) {
  hipLaunchKernelGGL((copy_tile<COLS, E>), dim3(grid), dim3(1024 / E * COLS), 0, 0, p);
}
template <int COLS, int E>
Variant make_copy() {
  char nm[128];
  snprintf(nm, sizeof nm, "copytile c%d e%d", COLS, E);
  return Variant{nm, COLS, 1024 / E * COLS, 0, &launch_copy_tile<COLS, E>, &build_pass_twiddles<Spec<1024, 16, 8, 8>, double>};
}

template <int COLS, int E, int NT>
Variant make_copy_nt() {
  char nm[128];
  snprintf(nm, sizeof nm, "copytile c%d e%d nt%d", COLS, E, NT);
  return Variant{nm, COLS, 1024 / E * COLS, 0, &launch_copy_tile_nt<COLS, E, NT>, &build_pass_twiddles<Spec<1024, 16, 8, 8>, double>};
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
  if (getenv("KB_PGRID")) g_persist_grid = atoi(getenv("KB_PGRID"));
  std::vector<Variant> vs = {
      make_copy<8, 16>(), make_copy<8, 32>(), make_copy<4, 16>(), make_copy<16, 16>(),
      make_copy_nt<8, 16, 0>(), make_copy_nt<8, 16, 1>(), make_copy_nt<8, 16, 2>(), make_copy_nt<8, 16, 3>(),
      make_persist<SA, 8, false>("16x8x8"), make_persist<SA, 8, true>("16x8x8"), make_persist<SD, 8, true>("8x8x4x4"),
      make_persist<SB, 8, false>("32x32"),
      make_persist<SA, 8, false, true>("16x8x8"), make_persist<SA, 8, true, true>("16x8x8"), make_persist<SB, 8, false, true>("32x32"),
      make_persist<SD, 8, true, true>("8x8x4x4"),
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
  {  // streaming-copy ceiling on this device: 8.6 GB -> 8.6 GB
    const size_t n = (size_t)N * N * 256;       // v16 elements = 4.29 GB each way, two halves of buf
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(copy_stream, dim3(16384), dim3(256), 0, 0, (const v16*)buf, (v16*)(buf + n), n);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) printf("streaming copy: %.3f ms for %.2f GB moved -> %.0f GB/s\n", ms, 2.0 * n * 16 / 1e9, 2.0 * n * 16 / (ms * 1e-3) / 1e9);
    }
  }
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
