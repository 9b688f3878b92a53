// kbench3.hip -- persistent (register double-buffered) strided-axis kernel against the per-tile one
// (developer tool, not part of the library).  Interleaved rounds in one process, results compared.
//   make -C tools kbench3 && tools/build/kbench3 [filter] [rounds]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "fft_kernels.h"
#include "fft_persist_experiment.h"
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

// the same with a register cap that lets WGS workgroups share a CU (waves per SIMD = WGS * THREADS / 256)
template <class K, class P, int WGS>
__global__ __launch_bounds__(K::THREADS, (WGS * K::THREADS + 255) / 256) void kern_occ(P p) {      // (rounded up, as registry.h does)
  extern __shared__ __attribute__((aligned(16))) char lds[];
  K::body(p, (int)blockIdx.x, (int)threadIdx.x, lds);
}

template <typename T>
struct Variant {
  std::string name;
  bool persistent;
  int cols;
  std::function<void(const ColParams<T>&, int)> launch;   // (params, grid)
  std::vector<cx<T>> tw;
};

template <class K, typename T>
void launch_k(const ColParams<T>& p, int grid) {
  static bool attr = false;
  if (!attr) {
    if (K::LDS_BYTES > 65536)
      CK(hipFuncSetAttribute((const void*)kern<K, ColParams<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL((kern<K, ColParams<T>>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, p);
}

template <class K, typename T, int WGS>
void launch_k_occ(const ColParams<T>& p, int grid) {
  static bool attr = false;
  if (!attr) {
    if (K::LDS_BYTES > 65536)
      CK(hipFuncSetAttribute((const void*)kern_occ<K, ColParams<T>, WGS>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern_occ<K, ColParams<T>, WGS>, K::THREADS, K::LDS_BYTES));
    printf("   (occupancy of the %d-workgroup build: %d per CU)\n", WGS, occ);
    attr = true;
  }
  hipLaunchKernelGGL((kern_occ<K, ColParams<T>, WGS>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, p);
}
template <class S, typename T, int COLS, bool TWLDS, int SPLIT, int VEC, bool NT, int WGS>
Variant<T> make_tile_occ(const char* plan) {
  typedef ColFft<S, T, COLS, false, TWLDS, SPLIT, VEC, NT> K;
  char nm[128];
  snprintf(nm, sizeof nm, "tile    %s c%d v%d%s%s%s thr%d lds%dK wgs%d", plan, COLS, VEC, TWLDS ? " twlds" : "", SPLIT == 2 ? " quarter" : SPLIT ? " split" : "", NT ? " nt" : "", K::THREADS,
           K::LDS_BYTES / 1024, WGS);
  return Variant<T>{nm, false, COLS, &launch_k_occ<K, T, WGS>, build_pass_twiddles<S, T>()};
}

template <class S, typename T, int COLS, bool TWLDS, int SPLIT, int VEC, bool NT = false>
Variant<T> make_tile(const char* plan) {
  typedef ColFft<S, T, COLS, false, TWLDS, SPLIT, VEC, NT> K;
  char nm[128];
  snprintf(nm, sizeof nm, "tile    %s c%d v%d%s%s%s thr%d lds%dK", plan, COLS, VEC, TWLDS ? " twlds" : "", SPLIT == 2 ? " quarter" : SPLIT ? " split" : "", NT ? " nt" : "", K::THREADS,
           K::LDS_BYTES / 1024);
  return Variant<T>{nm, false, COLS, &launch_k<K, T>, build_pass_twiddles<S, T>()};
}
template <class S, typename T, int COLS, int VEC>
Variant<T> make_persist(const char* plan) {
  typedef ColFftP<S, T, COLS, false, VEC> K;
  char nm[128];
  snprintf(nm, sizeof nm, "persist %s c%d v%d thr%d lds%dK", plan, COLS, VEC, K::THREADS, K::LDS_BYTES / 1024);
  return Variant<T>{nm, true, COLS, &launch_k<K, T>, build_pass_twiddles<S, T>()};
}

static int g_pgrid = 256;
static int g_nf_override = 0, g_remap = 1;

template <typename T>
void run_all(std::vector<Variant<T>>& vs, int N, const char* filter, int rounds) {
  const int NF = g_nf_override ? g_nf_override : N / 2 + 1;
  const size_t elems = (size_t)N * N * NF;
  cx<T>*src = nullptr, *buf = nullptr, *ref = nullptr;
  CK(hipMalloc(&src, elems * sizeof(cx<T>)));
  CK(hipMalloc(&buf, elems * sizeof(cx<T>)));
  CK(hipMalloc(&ref, elems * sizeof(cx<T>)));
  {
    std::vector<cx<T>> h((size_t)4 * N * NF);
    for (size_t i = 0; i < h.size(); ++i)
      h[i] = mk<T>((T)((double)((i * 2654435761u) % 1000) / 1000.0 - 0.5), (T)((double)((i * 40503u) % 977) / 977.0 - 0.5));
    for (size_t off = 0; off < elems; off += h.size())
      CK(hipMemcpy(src + off, h.data(), std::min(h.size(), elems - off) * sizeof(cx<T>), hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double alg_bytes = 2.0 * elems * sizeof(cx<T>);
  struct Layout { const char* name; bool xdir; bool inplace; };
  const Layout layouts[] = {{"y inplace", false, true}, {"x inplace", true, true}, {"x outofplace", true, false}};
  std::vector<cx<T>*> twd(vs.size());
  for (size_t i = 0; i < vs.size(); ++i) {
    CK(hipMalloc(&twd[i], vs[i].tw.size() * sizeof(cx<T>)));
    CK(hipMemcpy(twd[i], vs[i].tw.data(), vs[i].tw.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  }
  std::vector<cx<T>> href(1 << 16), hout(1 << 16);
  for (const Layout& L : layouts) {
    printf("== N=%d %s %s\n", N, sizeof(T) == 8 ? "fp64" : "fp32", L.name);
    std::vector<std::vector<double>> times(vs.size());
    bool have_ref = false;
    for (int round = 0; round < rounds; ++round) {
      for (size_t i = 0; i < vs.size(); ++i) {
        Variant<T>& v = vs[i];
        if (filter[0] && !strstr(v.name.c_str(), filter) && i != 0) continue;
        ColParams<T> P;
        memset(&P, 0, sizeof P);
        P.tw = twd[i];
        P.remap = g_remap;
        P.scale = (T)1;
        if (!L.xdir) {
          P.in_outer = P.out_outer = (i64)N * NF;
          P.in_map = P.out_map = make_rowmap(0, NF, N, N);
          P.ncols = NF;
          P.nouter = N;
        } else {
          P.in_outer = P.out_outer = 0;
          P.in_map = P.out_map = make_rowmap(0, (i64)N * NF, N, N);
          P.ncols = N * NF;
          P.nouter = 1;
        }
        P.ntile_c = (P.ncols + v.cols - 1) / v.cols;
        const int ntiles = P.ntile_c * P.nouter;
        const int grid = v.persistent ? std::min(g_pgrid, ntiles / 8 * 8 > 0 ? ntiles / 8 * 8 : ntiles) : ntiles;
        P.nblocks = grid;
        if (round == 0) {   // correctness: src -> buf (or in place on a copy of src), compare with variant 0
          CK(hipMemcpy(buf, src, elems * sizeof(cx<T>), hipMemcpyDeviceToDevice));
          P.in = buf; P.out = buf;
          v.launch(P, grid);
          CK(hipDeviceSynchronize());
          if (!have_ref) {
            CK(hipMemcpy(ref, buf, elems * sizeof(cx<T>), hipMemcpyDeviceToDevice));
            have_ref = true;
          } else {
            double maxd = 0, maxv = 0;
            for (size_t off : {(size_t)0, elems / 3, elems / 2, elems - href.size()}) {
              CK(hipMemcpy(href.data(), ref + off, href.size() * sizeof(cx<T>), hipMemcpyDeviceToHost));
              CK(hipMemcpy(hout.data(), buf + off, hout.size() * sizeof(cx<T>), hipMemcpyDeviceToHost));
              for (size_t q = 0; q < href.size(); ++q) {
                maxd = std::max(maxd, (double)std::fabs(href[q].x - hout[q].x));
                maxd = std::max(maxd, (double)std::fabs(href[q].y - hout[q].y));
                maxv = std::max(maxv, (double)std::fabs(href[q].x));
              }
            }
            printf("   check %-44s max|diff| %.3e (max|ref| %.3e)%s\n", v.name.c_str(), maxd, maxv, maxd > 1e-9 * maxv * (sizeof(T) == 8 ? 1 : 1e7) ? "  MISMATCH" : "");
          }
        }
        P.in = L.inplace ? buf : src;
        P.out = buf;
        v.launch(P, grid);
        CK(hipDeviceSynchronize());
        const int reps = 5;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) v.launch(P, grid);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        times[i].push_back(ms / reps);
      }
    }
    for (size_t i = 0; i < vs.size(); ++i) {
      if (times[i].empty()) continue;
      std::vector<double> t = times[i];
      std::sort(t.begin(), t.end());
      printf("   %-48s min %.3f med %.3f max %.3f ms  (%.0f GB/s at median)\n", vs[i].name.c_str(), t.front(), t[t.size() / 2], t.back(),
             alg_bytes / (t[t.size() / 2] * 1e-3) / 1e9);
    }
    fflush(stdout);
  }
  for (auto p : twd) CK(hipFree(p));
  CK(hipFree(src));
  CK(hipFree(buf));
  CK(hipFree(ref));
}


// Infinity Cache experiment: the same launch geometry (a section of `np` x-planes for the y pass, or a kz-chunk of
// `nf` columns for the x pass) either always on the same section (resident in the 256 MiB cache when it fits) or on
// a different section of a 8.6 GB array every time (served from HBM).
template <class K, typename T>
void mall_probe(const char* name, int cols, bool persistent, const std::vector<cx<T>>& twh) {
  const int N = 1024, NF = 513;
  const size_t elems = (size_t)N * N * NF;
  cx<T>* buf = nullptr;
  CK(hipMalloc(&buf, elems * sizeof(cx<T>)));
  CK(hipMemset(buf, 0, elems * sizeof(cx<T>)));
  cx<T>* tw = nullptr;
  CK(hipMalloc(&tw, twh.size() * sizeof(cx<T>)));
  CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("== Infinity Cache probe, %s\n", name);
  for (int np : {4, 8, 16, 32, 64}) {            // y pass over np planes (np * 8.4 MB)
    ColParams<T> P;
    memset(&P, 0, sizeof P);
    P.tw = tw; P.remap = 1; P.scale = (T)1;
    P.in_outer = P.out_outer = (i64)N * NF;
    P.in_map = P.out_map = make_rowmap(0, NF, N, N);
    P.ncols = NF; P.nouter = np;
    P.ntile_c = (P.ncols + cols - 1) / cols;
    const int ntiles = P.ntile_c * P.nouter;
    const int grid = persistent ? std::min(g_pgrid, ntiles / 8 * 8) : ntiles;
    P.nblocks = grid;
    const int nsec = N / np, reps = 64;
    double ms_hot = 0, ms_cold = 0;
    for (int mode = 0; mode < 2; ++mode) {
      for (int r = 0; r < 4; ++r) { P.in = P.out = buf; launch_k<K, T>(P, grid); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        cx<T>* sec = buf + (mode == 0 ? 0 : (size_t)((r * 37) % nsec) * np * N * NF);
        P.in = P.out = sec;
        launch_k<K, T>(P, grid);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      (mode == 0 ? ms_hot : ms_cold) = ms / reps;
    }
    const double bytes = 2.0 * np * N * NF * sizeof(cx<T>);
    printf("   y pass, %2d planes (%6.1f MB): same section %.4f ms (%.0f GB/s)   rotating sections %.4f ms (%.0f GB/s)\n", np, bytes / 2e6, ms_hot,
           bytes / ms_hot / 1e6, ms_cold, bytes / ms_cold / 1e6);
    fflush(stdout);
  }
  for (int nf : {8, 16, 32}) {                   // x pass over a (1024, 1024, nf) array, in place, rows of nf*1024 columns
    ColParams<T> P;
    memset(&P, 0, sizeof P);
    P.tw = tw; P.remap = 1; P.scale = (T)1;
    P.in_outer = P.out_outer = 0;
    P.in_map = P.out_map = make_rowmap(0, (i64)N * nf, N, N);
    P.ncols = N * nf; P.nouter = 1;
    P.ntile_c = (P.ncols + cols - 1) / cols;
    const int ntiles = P.ntile_c;
    const int grid = persistent ? std::min(g_pgrid, ntiles / 8 * 8) : ntiles;
    P.nblocks = grid;
    const size_t sec_elems = (size_t)N * N * nf;
    const int nsec = (int)(elems / sec_elems), reps = 64;
    double ms_hot = 0, ms_cold = 0;
    for (int mode = 0; mode < 2; ++mode) {
      for (int r = 0; r < 4; ++r) { P.in = P.out = buf; launch_k<K, T>(P, grid); }
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        cx<T>* sec = buf + (mode == 0 ? 0 : (size_t)((r * 37) % nsec) * sec_elems);
        P.in = P.out = sec;
        launch_k<K, T>(P, grid);
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      (mode == 0 ? ms_hot : ms_cold) = ms / reps;
    }
    const double bytes = 2.0 * sec_elems * sizeof(cx<T>);
    printf("   x pass, nf = %2d (%6.1f MB): same section %.4f ms (%.0f GB/s)   rotating sections %.4f ms (%.0f GB/s)\n", nf, bytes / 2e6, ms_hot,
           bytes / ms_hot / 1e6, ms_cold, bytes / ms_cold / 1e6);
    fflush(stdout);
  }
  CK(hipFree(tw));
  CK(hipFree(buf));
}

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : "";
  const int rounds = argc > 2 ? atoi(argv[2]) : 3;
  if (getenv("KB_PGRID")) g_pgrid = atoi(getenv("KB_PGRID"));
  typedef Spec<1024, 16, 8, 8> SA;
  typedef Spec<1024, 16, 16, 4> SC;
  typedef Spec<512, 8, 8, 8> S512;
  typedef Spec<512, 16, 8, 4> S512b;
  if (!filter[0] || strstr("mall", filter)) {
    mall_probe<ColFft<SA, double, 8, false, false, false, 1>, double>("tile 16x8x8 c8", 8, false, build_pass_twiddles<SA, double>());
    mall_probe<ColFftP<SA, double, 8, false, 1>, double>("persist 16x8x8 c8", 8, true, build_pass_twiddles<SA, double>());
    if (filter[0]) return 0;
  }
  typedef Spec<1024, 8, 8, 4, 4> SD;
  typedef Spec<1024, 8, 8, 8, 2> SF;
  if (filter[0] && strstr("pitch520", filter)) {   // aligned intermediates for the y pass: what do the three passes of an inverse cost then?
    typedef double T;
    const int N = 1024, NF = 513, PF = 520;
    cx<T>*fu = nullptr, *W = nullptr;
    CK(hipMalloc(&fu, (size_t)N * N * NF * sizeof(cx<T>)));
    CK(hipMalloc(&W, (size_t)N * (N * PF + 8) * sizeof(cx<T>)));
    CK(hipMemset(fu, 0, (size_t)N * N * NF * sizeof(cx<T>)));
    CK(hipMemset(W, 0, (size_t)N * (N * PF + 8) * sizeof(cx<T>)));
    auto twh = build_pass_twiddles<SD, T>();
    cx<T>* tw = nullptr;
    CK(hipMalloc(&tw, twh.size() * sizeof(cx<T>)));
    CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, std::function<void()> f) {
      for (int i = 0; i < 3; ++i) f();
      CK(hipDeviceSynchronize());
      std::vector<double> t;
      for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 5);
      }
      std::sort(t.begin(), t.end());
      printf("   %-70s min %.3f med %.3f ms\n", name, t.front(), t[2]);
    };
    auto params = [&](const cx<T>* in, cx<T>* out, i64 in_outer, i64 out_outer, i64 in_lo, i64 out_lo, int ncols, int nouter) {
      ColParams<T> P;
      memset(&P, 0, sizeof P);
      P.in = in; P.out = out; P.tw = tw; P.remap = 1; P.scale = 1;
      P.in_outer = in_outer; P.out_outer = out_outer;
      P.in_map = make_rowmap(0, in_lo, N, N); P.out_map = make_rowmap(0, out_lo, N, N);
      P.ncols = ncols; P.nouter = nouter; P.ntile_c = (ncols + 7) / 8;
      return P;
    };
    typedef ColFft<SD, T, 8, false, true, true, 1, false> K;
    typedef ColFft<SD, T, 8, false, true, true, 1, true> KNT;
    {
      ColParams<T> P = params(fu, fu, (i64)N * NF, (i64)N * NF, NF, NF, NF, N);
      timeit("y in place, pitch 513 (today)", [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(W, W, (i64)N * PF, (i64)N * PF, PF, PF, NF, N);
      timeit("y in place, pitch 520, nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      timeit("y in place, pitch 520", [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(fu, W, 0, 0, (i64)N * NF, (i64)N * NF, N * NF, 1);
      timeit("x out of place, flattened, both pitch 513, nt (today's inverse x)", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(fu, W, NF, PF, (i64)N * NF, (i64)N * PF, NF, N);      // per (y, kz tile): in pitch 513 (misaligned), out pitch 520 (aligned)
      timeit("x out of place, fu(513) -> W(520), tiles per y row", [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(W, fu, PF, NF, (i64)N * PF, (i64)N * NF, NF, N);      // the forward direction with the same tiling: aligned loads, misaligned stores
      timeit("x out of place, W(520) -> fu(513), tiles per y row (partial-line stores)", [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(fu, fu, 0, 0, (i64)N * NF, (i64)N * NF, N * NF, 1);
      timeit("x in place, flattened, pitch 513, nt (today's forward x)", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      // the same with planes of the padded array one cache line further apart: a plane of 1024 x 520 x 16 B is a multiple
      // of 2^17 B, the worst row stride for the x pass (profiles/r02_power_of_two_stride.txt)
      const i64 PLW = (i64)N * PF + 8;
      P = params(W, W, PLW, PLW, PF, PF, NF, N);
      timeit("y in place, pitch 520, planes +128 B, nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(fu, W, NF, PF, (i64)N * NF, PLW, NF, N);
      timeit("x out of place, fu(513) -> W(520, planes +128 B), tiles per y row", [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(W, fu, PF, NF, PLW, (i64)N * NF, NF, N);
      timeit("x out of place, W(520, planes +128 B) -> fu(513), tiles per y row", [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(W, W, 0, 0, PLW, PLW, N * PF, 1);
      timeit("x in place in W(520, planes +128 B), flattened (pad columns transformed too), nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
    }
    return 0;
  }
  if (filter[0] && strstr("padplane", filter)) {   // C2C 1024^3: what a padded plane stride / row pitch of the intermediate buys
    typedef double T;
    const int N = 1024;
    const i64 PL0 = (i64)N * N;
    std::vector<i64> pads = {0, 8, 64, 264, 520, 8 * 1024 + 8};
    if (getenv("KB_PADSWEEP") || getenv("KB_STRIDESWEEP")) pads.clear();
    const i64 PITCH2 = N + 8;
    cx<T>*A = nullptr, *W = nullptr;
    const size_t wel = (size_t)N * (size_t)(N * PITCH2 + 16384) + 4096;
    CK(hipMalloc(&A, wel * sizeof(cx<T>)));
    CK(hipMalloc(&W, wel * sizeof(cx<T>)));
    CK(hipMemset(A, 0, wel * sizeof(cx<T>)));
    CK(hipMemset(W, 0, wel * sizeof(cx<T>)));
    auto twh = build_pass_twiddles<SD, T>();
    cx<T>* tw = nullptr;
    CK(hipMalloc(&tw, twh.size() * sizeof(cx<T>)));
    CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, std::function<void()> f) {
      for (int i = 0; i < 3; ++i) f();
      CK(hipDeviceSynchronize());
      std::vector<double> t;
      for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) f();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms / 5);
      }
      std::sort(t.begin(), t.end());
      printf("   %-84s min %.3f med %.3f ms  (%.0f GB/s)\n", name, t.front(), t[2], 2.0 * N * N * N * 16 / t[2] / 1e6);
      fflush(stdout);
    };
    auto params = [&](const cx<T>* in, cx<T>* out, i64 in_outer, i64 out_outer, i64 in_lo, i64 out_lo, i64 ncols, int nouter) {
      ColParams<T> P;
      memset(&P, 0, sizeof P);
      P.in = in; P.out = out; P.tw = tw; P.remap = 1; P.scale = 1;
      P.in_outer = in_outer; P.out_outer = out_outer;
      P.in_map = make_rowmap(0, in_lo, N, N); P.out_map = make_rowmap(0, out_lo, N, N);
      P.ncols = ncols; P.nouter = nouter; P.ntile_c = (int)((ncols + 7) / 8);
      return P;
    };
    typedef ColFft<SD, T, 8, false, true, true, 1, false> K;
    typedef ColFft<SD, T, 8, false, true, true, 1, true> KNT;
    char nm[160];
    for (i64 pad : pads) {
      const i64 PL = PL0 + pad;
      ColParams<T> P = params(A, A, 0, 0, PL, PL, PL0, 1);
      snprintf(nm, sizeof nm, "x in place, flattened columns, plane stride N*N+%lld, nt", (long long)pad);
      timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      snprintf(nm, sizeof nm, "x in place, flattened columns, plane stride N*N+%lld", (long long)pad);
      timeit(nm, [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      P = params(A, W, 0, 0, PL, PL0, PL0, 1);
      snprintf(nm, sizeof nm, "x out of place, padded (+%lld) -> power of two, nt", (long long)pad);
      timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(A, W, 0, 0, PL0, PL, PL0, 1);
      snprintf(nm, sizeof nm, "x out of place, power of two -> padded (+%lld), nt", (long long)pad);
      timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(A, A, PL, PL, N, N, N, N);
      snprintf(nm, sizeof nm, "y in place, pitch N, plane stride N*N+%lld, nt", (long long)pad);
      timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
    }
    {
      const i64 PL = (i64)N * PITCH2;
      ColParams<T> P = params(A, A, PL, PL, PITCH2, PITCH2, N, N);
      timeit("y in place, pitch N+8, nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(A, A, PITCH2, PITCH2, PL, PL, N, N);
      timeit("x in place, tiles per y row, pitch N+8, nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(A, W, PITCH2, N, PL, PL0, N, N);
      timeit("x out of place, pitch N+8 -> power of two, tiles per y row, nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      P = params(W, A, N, PITCH2, PL0, PL, N, N);
      timeit("x out of place, power of two -> pitch N+8, tiles per y row, nt", [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
    }
    if (getenv("KB_STRIDESWEEP")) {   // from which power-of-two row stride on does the x pass suffer?  plane = 2^k bytes
      for (int k = 13; k <= 24; ++k) {
        const i64 plane = ((i64)1 << k) / 16, nouter = PL0 / plane;
        for (i64 pad : {(i64)0, (i64)8}) {
          ColParams<T> P = params(A, W, (i64)N * (plane + pad), (i64)N * plane, plane + pad, plane, plane, (int)nouter);
          snprintf(nm, sizeof nm, "x out of place, row stride 2^%d B + %lld B -> 2^%d B, %lld batches, nt", k, (long long)pad * 16, k, (long long)nouter);
          timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
        }
      }
      return 0;
    }
    if (getenv("KB_PADSWEEP")) {
      for (int m : {1, 2, 3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 27, 31, 33, 37, 41, 47, 63, 65, 127, 129, 255, 257}) {
        const i64 pad = 8 * m, PL = PL0 + pad;
        ColParams<T> P = params(A, W, 0, 0, PL, PL0, PL0, 1);
        snprintf(nm, sizeof nm, "x out of place, padded (+%d lines of 128 B) -> power of two, nt", m);
        timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
      }
      return 0;
    }
    {   // the R2C arrays (513 columns): the plane stride 1024 * 513 * 16 B still has 14 zero low bits
      const i64 NF = 513, PLR = (i64)N * NF;
      printf("   -- R2C layout, 513 columns (times for 0.5 of the C2C volume)\n");
      for (i64 pad : {(i64)0, (i64)8, (i64)520}) {
        const i64 PL = PLR + pad;
        ColParams<T> P = params(A, A, 0, 0, PL, PL, PLR, 1);
        snprintf(nm, sizeof nm, "r2c x in place, flattened, plane stride 1024*513+%lld, nt", (long long)pad);
        timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
        P = params(A, W, 0, 0, PL, PLR, PLR, 1);
        snprintf(nm, sizeof nm, "r2c x out of place, padded (+%lld) -> exact, nt", (long long)pad);
        timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
        P = params(A, W, 0, 0, PLR, PL, PLR, 1);
        snprintf(nm, sizeof nm, "r2c x out of place, exact -> padded (+%lld), nt", (long long)pad);
        timeit(nm, [&] { launch_k<KNT, T>(P, P.ntile_c * P.nouter); });
        P = params(A, A, PL, PL, NF, NF, NF, N);
        snprintf(nm, sizeof nm, "r2c y in place, plane stride +%lld", (long long)pad);
        timeit(nm, [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
        P = params(A, W, PLR, PL, NF, NF, NF, N);
        snprintf(nm, sizeof nm, "r2c y out of place, exact -> plane stride +%lld", (long long)pad);
        timeit(nm, [&] { launch_k<K, T>(P, P.ntile_c * P.nouter); });
      }
    }
    return 0;
  }
  if (filter[0] && strstr("wideb", filter)) {      // ... and the 42-values plans (8 - 16 threads per transform: 64 - 128 threads on 8 columns), 320, 240, 224
    {
      typedef Spec<672, 42, 2, 2, 2, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, true, 1, 1>("42x2x2x2x2"));
      vs.push_back(make_tile<S, double, 16, false, 1, 1>("42x2x2x2x2"));
      vs.push_back(make_tile<S, double, 16, true, 1, 1>("42x2x2x2x2"));
      run_all<double>(vs, 672, "", rounds);
    }
    {
      typedef Spec<336, 42, 2, 2, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, true, 0, 1>("42x2x2x2"));
      vs.push_back(make_tile<S, double, 16, true, 0, 1>("42x2x2x2"));
      vs.push_back(make_tile<S, double, 32, true, 1, 1>("42x2x2x2"));
      run_all<double>(vs, 336, "", rounds);
    }
    {
      typedef Spec<320, 40, 8> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, true, 0, 1>("40x8"));
      vs.push_back(make_tile<S, double, 16, true, 0, 1>("40x8"));
      vs.push_back(make_tile<S, double, 32, true, 1, 1>("40x8"));
      run_all<double>(vs, 320, "", rounds);
    }
    {
      typedef Spec<240, 10, 6, 2, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, true, 0, 1>("10x6x2x2"));
      vs.push_back(make_tile<S, double, 16, true, 0, 1>("10x6x2x2"));
      vs.push_back(make_tile<S, double, 32, true, 1, 1>("10x6x2x2"));
      run_all<double>(vs, 240, "", rounds);
    }
    {
      typedef Spec<448, 28, 4, 4> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, true, 0, 1>("28x4x4"));
      vs.push_back(make_tile<S, float, 32, false, 1, 1>("28x4x4"));
      vs.push_back(make_tile<S, float, 32, true, 1, 1>("28x4x4"));
      run_all<float>(vs, 448, "", rounds);
    }
    {
      typedef Spec<480, 10, 6, 2, 2, 2> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, true, 0, 1>("10x6x2x2x2"));
      vs.push_back(make_tile<S, float, 32, true, 1, 1>("10x6x2x2x2"));
      run_all<float>(vs, 480, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("wide16", filter)) {     // round 5: 16 columns for the plans whose 8-column workgroups are 2 - 3 waves (16 - 24 threads per transform)
    {
      typedef Spec<448, 28, 4, 4> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, true, 0, 1>("28x4x4"));
      vs.push_back(make_tile<S, double, 16, false, 1, 1>("28x4x4"));
      vs.push_back(make_tile<S, double, 16, true, 1, 1>("28x4x4"));
      run_all<double>(vs, 448, "", rounds);
    }
    {
      typedef Spec<480, 10, 6, 2, 2, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 0, 1>("10x6x2x2x2"));
      vs.push_back(make_tile<S, double, 16, false, 1, 1>("10x6x2x2x2"));
      vs.push_back(make_tile<S, double, 16, true, 1, 1>("10x6x2x2x2"));
      run_all<double>(vs, 480, "", rounds);
    }
    {
      typedef Spec<640, 8, 4, 4, 5> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 0, 1>("8x4x4x5"));
      vs.push_back(make_tile<S, double, 16, false, 1, 1>("8x4x4x5"));
      vs.push_back(make_tile<S, double, 16, true, 1, 1>("8x4x4x5"));
      run_all<double>(vs, 640, "", rounds);
    }
    {
      typedef Spec<720, 10, 6, 6, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, true, 1, 1>("10x6x6x2"));
      vs.push_back(make_tile<S, double, 16, false, 1, 1>("10x6x6x2"));
      vs.push_back(make_tile<S, double, 16, true, 1, 1>("10x6x6x2"));
      run_all<double>(vs, 720, "", rounds);
    }
    {
      typedef Spec<600, 10, 10, 6> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 0, 1>("10x10x6"));
      vs.push_back(make_tile<S, double, 16, true, 1, 1>("10x10x6"));
      run_all<double>(vs, 600, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("y64b", filter)) {       // ... and 1344 (42 values per thread), 1500
    {
      typedef Spec<1344, 42, 2, 2, 2, 2, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("42x2x2x2x2x2"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("42x2x2x2x2x2"));
      vs.push_back(make_tile_occ<S, double, 4, false, 1, 1, false, 2>("42x2x2x2x2x2"));
      run_all<double>(vs, 1344, "", rounds);
    }
    {
      typedef Spec<1500, 15, 10, 10> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("15x10x10"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("15x10x10"));
      run_all<double>(vs, 1500, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("y64", filter)) {        // round 5: 64-byte tiles with LDS twiddles won the y pass (rows a few KB apart) of 1440 / 1536 by 16 - 26 %
                                                   // and lost the x pass (rows MBs apart): which other lengths?
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<SD, double, 8, true, 1, 1>("8x8x4x4"));
      vs.push_back(make_tile_occ<SD, double, 4, true, 1, 1, false, 3>("8x8x4x4"));
      vs.push_back(make_tile_occ<SD, double, 4, true, 1, 1, false, 4>("8x8x4x4"));
      run_all<double>(vs, 1024, "", rounds);
    }
    {
      typedef Spec<1152, 12, 12, 4, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile_occ<S, double, 8, false, 1, 1, false, 2>("12x12x4x2"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("12x12x4x2"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 3>("12x12x4x2"));
      run_all<double>(vs, 1152, "", rounds);
    }
    {
      typedef Spec<1200, 10, 10, 6, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 16, false, 1, 1>("10x10x6x2"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("10x10x6x2"));
      run_all<double>(vs, 1200, "", rounds);
    }
    {
      typedef Spec<1280, 8, 8, 4, 5> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("8x8x4x5"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("8x8x4x5"));
      run_all<double>(vs, 1280, "", rounds);
    }
    {
      typedef Spec<1792, 28, 4, 4, 4> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("28x4x4x4"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 1>("28x4x4x4"));
      vs.push_back(make_tile_occ<S, double, 2, true, 1, 1, false, 2>("28x4x4x4"));
      run_all<double>(vs, 1792, "", rounds);
    }
    {
      typedef Spec<2048, 16, 16, 8> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("16x16x8"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 1>("16x16x8"));
      run_all<double>(vs, 2048, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("tw64", filter)) {       // round 5: 64-byte tiles WITH LDS twiddles in double precision (the trials of rounds 2 / 3 had none)
    {
      typedef Spec<1536, 8, 8, 8, 3> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("8x8x8x3"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("8x8x8x3"));
      vs.push_back(make_tile_occ<S, double, 4, false, 1, 1, false, 3>("8x8x8x3"));
      run_all<double>(vs, 1536, "", rounds);
    }
    {
      typedef Spec<1440, 10, 6, 6, 2, 2> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("10x6x6x2x2"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("10x6x6x2x2"));
      run_all<double>(vs, 1440, "", rounds);
    }
    {
      typedef Spec<2304, 24, 24, 4> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("24x24x4"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("24x24x4"));
      run_all<double>(vs, 2304, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("tw1536", filter)) {     // round 5: LDS twiddles were what made the 64-byte tiles of 1792 fp32 pay; 1536 / 1152 fp32 run without
    {
      typedef Spec<1536, 8, 8, 8, 3> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile_occ<S, float, 8, false, 1, 1, false, 3>("8x8x8x3"));      // shipped (three per CU by the API; 8 waves: 6 per SIMD, 80 registers)
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 2>("8x8x8x3"));
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 3>("8x8x8x3"));
      vs.push_back(make_tile_occ<Spec<1536, 16, 8, 4, 3>, float, 8, true, 1, 1, false, 2>("16x8x4x3"));
      run_all<float>(vs, 1536, "", rounds);
    }
    {
      typedef Spec<1152, 24, 24, 2> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile_occ<S, float, 16, false, 1, 1, false, 2>("24x24x2"));     // shipped
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 2>("24x24x2"));
      vs.push_back(make_tile_occ<Spec<1152, 12, 12, 4, 2>, float, 16, true, 1, 1, false, 2>("12x12x4x2"));
      run_all<float>(vs, 1152, "", rounds);
    }
    {
      typedef Spec<896, 28, 4, 4, 2> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, true, 1, 1>("28x4x4x2"));                   // shipped (col_pair: split + LDS twiddles)
      vs.push_back(make_tile_occ<S, float, 16, true, 1, 1, false, 2>("28x4x4x2"));
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 4>("28x4x4x2"));
      run_all<float>(vs, 896, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("narrow", filter)) {     // round 5: what 1792 fp32 gained (64-byte tiles + LDS twiddles, two workgroups per CU) at the other
                                                   // lengths that run one big workgroup per CU
    {
      typedef Spec<2048, 32, 8, 8> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, false, 1, 1>("32x8x8"));
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 2>("32x8x8"));
      vs.push_back(make_tile_occ<Spec<2048, 16, 16, 8>, float, 8, true, 1, 1, false, 2>("16x16x8"));
      run_all<float>(vs, 2048, "", rounds);
    }
    {
      typedef Spec<1440, 10, 6, 6, 2, 2> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, false, 1, 1>("10x6x6x2x2"));
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 2>("10x6x6x2x2"));
      vs.push_back(make_tile_occ<S, float, 8, false, 1, 1, false, 2>("10x6x6x2x2"));
      run_all<float>(vs, 1440, "", rounds);
    }
    {
      typedef Spec<1200, 10, 10, 6, 2> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, false, 1, 1>("10x10x6x2"));
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 2>("10x10x6x2"));
      run_all<float>(vs, 1200, "", rounds);
    }
    {
      typedef Spec<1280, 8, 8, 4, 5> S;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S, float, 16, false, 1, 1>("8x8x4x5"));
      vs.push_back(make_tile_occ<S, float, 8, true, 1, 1, false, 2>("8x8x4x5"));
      vs.push_back(make_tile_occ<S, float, 16, true, 1, 1, false, 2>("8x8x4x5"));
      run_all<float>(vs, 1280, "", rounds);
    }
    {
      typedef Spec<1792, 28, 4, 4, 4> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("28x4x4x4"));
      vs.push_back(make_tile_occ<S, double, 4, false, 1, 1, false, 2>("28x4x4x4"));
      run_all<double>(vs, 1792, "", rounds);
    }
    {
      typedef Spec<1600, 20, 20, 4> S;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S, double, 8, false, 1, 1>("20x20x4"));
      vs.push_back(make_tile_occ<S, double, 4, true, 1, 1, false, 2>("20x20x4"));
      vs.push_back(make_tile_occ<S, double, 4, false, 1, 1, false, 2>("20x20x4"));
      run_all<double>(vs, 1600, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("occ1200", filter)) {    // round 5: do two workgroups per CU really run at 1200 / 1440 under the register cap?
    {
      typedef Spec<1200, 10, 10, 6, 2> S12;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S12, double, 8, false, 1, 1>("10x10x6x2"));
      vs.push_back(make_tile_occ<S12, double, 8, false, 1, 1, false, 2>("10x10x6x2"));
      vs.push_back(make_tile_occ<S12, double, 8, false, 2, 1, false, 2>("10x10x6x2"));
      vs.push_back(make_tile_occ<S12, double, 8, false, 2, 1, false, 3>("10x10x6x2"));
      vs.push_back(make_tile_occ<S12, double, 4, false, 1, 1, false, 4>("10x10x6x2"));
      // 5 waves on 4 SIMDs: one SIMD carries two, and the passes last as long as ITS two waves (membench stamp1200).  Two
      // 128-byte tiles per workgroup: 10 waves = 3 3 2 2
      vs.push_back(make_tile<S12, double, 16, false, 1, 1>("10x10x6x2"));
      vs.push_back(make_tile<S12, double, 16, false, 1, 1, true>("10x10x6x2"));
      run_all<double>(vs, 1200, "", rounds);
    }
    {
      typedef Spec<1440, 10, 6, 6, 2, 2> S14;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S14, double, 8, false, 1, 1>("10x6x6x2x2"));
      vs.push_back(make_tile_occ<S14, double, 8, false, 2, 1, false, 2>("10x6x6x2x2"));
      vs.push_back(make_tile_occ<S14, double, 8, false, 2, 1, false, 3>("10x6x6x2x2"));
      vs.push_back(make_tile_occ<S14, double, 4, false, 1, 1, false, 3>("10x6x6x2x2"));
      vs.push_back(make_tile<S14, double, 12, false, 1, 1>("10x6x6x2x2"));      // 9 waves = 3 2 2 2 (192-byte tiles)
      vs.push_back(make_tile<S14, double, 16, false, 2, 1>("10x6x6x2x2"));      // 12 waves = 3 3 3 3, exchange by quarters (90 KB)
      run_all<double>(vs, 1440, "", rounds);
    }
    {   // single precision 1792 (0.45 of the roofline): one 1024-thread workgroup per CU as shipped, or two of 512 on 64-byte tiles
      typedef Spec<1792, 28, 4, 4, 4> S17;
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<S17, float, 16, false, 1, 1>("28x4x4x4"));
      vs.push_back(make_tile_occ<S17, float, 8, false, 1, 1, false, 2>("28x4x4x4"));
      vs.push_back(make_tile_occ<S17, float, 8, true, 1, 1, false, 2>("28x4x4x4"));
      vs.push_back(make_tile_occ<S17, float, 16, false, 2, 1, false, 2>("28x4x4x4"));     // 128-byte tiles, quarter exchange (57 KB): two per CU need <= 64 registers
      vs.push_back(make_tile<S17, float, 16, false, 1, 2>("28x4x4x4"));                   // two columns per lane: 512 threads
      run_all<float>(vs, 1792, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("pow2", filter)) {       // C2C arrays: row pitch and plane stride are powers of two
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<SD, double, 8, true, true, 1, true>("8x8x4x4"));
    vs.push_back(make_tile<SD, double, 8, true, true, 1, false>("8x8x4x4"));
    g_nf_override = 1024;
    for (int remap : {1, 2}) {
      g_remap = remap;
      printf("#### remap mode %d\n", remap);
      run_all<double>(vs, 1024, "", rounds);
    }
    std::vector<Variant<float>> vf;
    vf.push_back(make_tile<SA, float, 16, true, true, 1>("16x8x8"));
    for (int remap : {1, 2}) {
      g_remap = remap;
      printf("#### remap mode %d\n", remap);
      run_all<float>(vf, 1024, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("f32", filter)) {        // single precision, 1024: is there a two-workgroup form that wins?
    std::vector<Variant<float>> vs;
    vs.push_back(make_tile<SA, float, 16, false, false, 2>("16x8x8"));
    vs.push_back(make_tile<SA, float, 16, false, false, 2, true>("16x8x8"));
    vs.push_back(make_tile<SA, float, 16, true, true, 1>("16x8x8"));
    vs.push_back(make_tile<SA, float, 16, true, true, 1, true>("16x8x8"));
    vs.push_back(make_tile<SA, float, 16, false, false, 1>("16x8x8"));
    vs.push_back(make_tile<Spec<1024, 32, 32>, float, 16, true, true, 1>("32x32"));
    vs.push_back(make_tile<Spec<1024, 32, 32>, float, 16, true, false, 1>("32x32"));
    vs.push_back(make_tile<Spec<1024, 32, 32>, float, 16, true, false, 2>("32x32"));
    vs.push_back(make_tile<SC, float, 16, false, false, 2>("16x16x4"));
    run_all<float>(vs, 1024, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("n800", filter)) {        // 800 / 640 / 1600: three workgroups per CU without LDS twiddles?
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<800, 5, 5, 4, 4, 2>, double, 8, true, true, 1>("5x5x4x4x2"));
      vs.push_back(make_tile_occ<Spec<800, 5, 5, 4, 4, 2>, double, 8, false, true, 1, false, 3>("5x5x4x4x2"));
      vs.push_back(make_tile_occ<Spec<800, 5, 5, 4, 4, 2>, double, 8, true, true, 1, false, 2>("5x5x4x4x2"));
      vs.push_back(make_tile_occ<Spec<800, 5, 4, 5, 4, 2>, double, 8, true, true, 1, false, 2>("5x4x5x4x2"));
      vs.push_back(make_tile_occ<Spec<800, 4, 4, 5, 5, 2>, double, 8, true, true, 1, false, 2>("4x4x5x5x2"));
      run_all<double>(vs, 800, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1600, 5, 5, 4, 4, 4>, double, 8, false, true, 1>("5x5x4x4x4"));
      vs.push_back(make_tile<Spec<1600, 5, 5, 4, 4, 4>, double, 8, false, true, 1, true>("5x5x4x4x4"));
      vs.push_back(make_tile<Spec<1600, 8, 8, 5, 5>, double, 8, false, true, 1>("8x8x5x5"));
      run_all<double>(vs, 1600, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("q1536", filter)) {       // round 3: the quarter exchange (49 KB) for two workgroups per CU at 1536 / 1600 / 2048
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, 1, 1>("8x8x8x3"));
    vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, 2, 1>("8x8x8x3"));
    vs.push_back(make_tile_occ<Spec<1536, 8, 8, 8, 3>, double, 8, false, 2, 1, false, 2>("8x8x8x3"));
    vs.push_back(make_tile_occ<Spec<1536, 8, 8, 8, 3>, double, 8, false, 2, 1, true, 2>("8x8x8x3"));
    vs.push_back(make_tile_occ<Spec<1536, 24, 8, 8>, double, 8, false, 2, 1, false, 2>("24x8x8"));
    run_all<double>(vs, 1536, "", rounds);
  }
  if (filter[0] && strstr("h1536", filter)) {       // 1536 fp64: 12 values per thread on 64-byte tiles, three workgroups per CU
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, true, 1>("8x8x8x3"));
    vs.push_back(make_tile_occ<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 4, false, true, 1, false, 3>("4x4x4x4x3x2"));
    vs.push_back(make_tile_occ<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 4, false, true, 1, false, 2>("4x4x4x4x3x2"));
    vs.push_back(make_tile_occ<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 4, true, true, 1, false, 2>("4x4x4x4x3x2"));
    vs.push_back(make_tile_occ<Spec<1536, 8, 4, 4, 4, 3>, double, 4, false, true, 1, false, 3>("8x4x4x4x3"));
    run_all<double>(vs, 1536, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("f32k2", filter)) {       // single precision 2048: 64-byte tiles, two workgroups per CU
    std::vector<Variant<float>> vs;
    vs.push_back(make_tile<Spec<2048, 32, 8, 8>, float, 16, false, true, 1>("32x8x8"));
    vs.push_back(make_tile<Spec<2048, 32, 8, 8>, float, 16, false, true, 1, true>("32x8x8"));
    vs.push_back(make_tile_occ<Spec<2048, 32, 8, 8>, float, 8, false, true, 1, false, 2>("32x8x8"));
    vs.push_back(make_tile_occ<Spec<2048, 32, 8, 8>, float, 8, false, true, 1, true, 2>("32x8x8"));
    vs.push_back(make_tile_occ<Spec<2048, 16, 16, 8>, float, 8, false, true, 1, false, 2>("16x16x8"));
    vs.push_back(make_tile_occ<Spec<2048, 16, 16, 8>, float, 16, false, true, 2, false, 1>("16x16x8"));
    run_all<float>(vs, 2048, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("half", filter)) {        // 64-byte tiles and more workgroups per CU where a 128-byte tile leaves one
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, true, 1>("8x8x8x3"));
      vs.push_back(make_tile_occ<Spec<1536, 8, 8, 8, 3>, double, 4, false, true, 1, false, 3>("8x8x8x3"));
      vs.push_back(make_tile_occ<Spec<1536, 8, 8, 8, 3>, double, 4, false, true, 1, false, 2>("8x8x8x3"));
      vs.push_back(make_tile_occ<Spec<1536, 8, 8, 8, 3>, double, 4, false, false, 1, false, 1>("8x8x8x3"));
      run_all<double>(vs, 1536, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<2048, 16, 16, 8>, double, 8, false, true, 1>("16x16x8"));
      vs.push_back(make_tile_occ<Spec<2048, 16, 16, 8>, double, 4, false, true, 1, false, 2>("16x16x8"));
      vs.push_back(make_tile_occ<Spec<2048, 16, 16, 8>, double, 4, false, false, 1, false, 1>("16x16x8"));
      run_all<double>(vs, 2048, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("f32long", filter)) {     // single precision at 1152 / 1280 / 1536: register caps, fewer values per thread
    {
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, float, 16, false, true, 1>("8x8x3x3x2"));
      vs.push_back(make_tile_occ<Spec<1152, 8, 8, 3, 3, 2>, float, 16, false, true, 1, false, 2>("8x8x3x3x2"));
      vs.push_back(make_tile_occ<Spec<1152, 8, 8, 3, 3, 2>, float, 16, false, true, 1, true, 2>("8x8x3x3x2"));
      vs.push_back(make_tile_occ<Spec<1152, 4, 4, 4, 3, 3, 2>, float, 8, false, true, 1, false, 2>("4x4x4x3x3x2"));
      run_all<float>(vs, 1152, "", rounds);
    }
    {
      std::vector<Variant<float>> vs;
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, float, 16, false, true, 1>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, float, 16, false, true, 1, true>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, float, 16, true, true, 1>("8x8x8x3"));
      vs.push_back(make_tile_occ<Spec<1536, 8, 8, 8, 3>, float, 8, false, true, 1, false, 3>("8x8x8x3"));
      run_all<float>(vs, 1536, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("occ512f", filter)) {     // round 3: 512 fp32: the same question as occ512
    std::vector<Variant<float>> vs;
    vs.push_back(make_tile<Spec<512, 8, 8, 8>, float, 16, false, false, 2>("8x8x8"));                         // shipped: two columns per lane
    vs.push_back(make_tile_occ<Spec<512, 4, 4, 4, 4, 2>, float, 16, true, true, 2, false, 2>("4x4x4x4x2"));
    vs.push_back(make_tile_occ<Spec<512, 4, 4, 4, 4, 2>, float, 16, true, false, 2, false, 2>("4x4x4x4x2"));
    vs.push_back(make_tile_occ<Spec<512, 8, 8, 8>, float, 16, true, true, 2, false, 4>("8x8x8"));
    vs.push_back(make_tile_occ<Spec<512, 8, 8, 8>, float, 16, true, true, 1, false, 2>("8x8x8"));
    run_all<float>(vs, 512, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("occ2048", filter)) {     // round 3: 2048 fp64: 64-byte tiles with 8 values per thread, 2048 threads per CU
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<Spec<2048, 16, 16, 8>, double, 8, false, true, 1>("16x16x8"));                    // shipped
    vs.push_back(make_tile_occ<Spec<2048, 8, 8, 8, 4>, double, 4, true, true, 1, false, 2>("8x8x8x4"));
    vs.push_back(make_tile_occ<Spec<2048, 8, 8, 8, 4>, double, 4, false, true, 1, false, 2>("8x8x8x4"));
    vs.push_back(make_tile_occ<Spec<2048, 16, 16, 8>, double, 4, false, true, 1, false, 2>("16x16x8"));
    run_all<double>(vs, 2048, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("occ256", filter)) {      // round 3: 256 and 768 fp64: fewer values per thread, more threads per CU?
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<256, 16, 16>, double, 8, true, false, 1>("16x16"));                          // shipped
      vs.push_back(make_tile_occ<Spec<256, 4, 4, 4, 4>, double, 8, true, true, 1, false, 4>("4x4x4x4"));
      vs.push_back(make_tile_occ<Spec<256, 4, 4, 4, 4>, double, 8, true, false, 1, false, 4>("4x4x4x4"));
      vs.push_back(make_tile_occ<Spec<256, 8, 8, 4>, double, 8, true, false, 1, false, 4>("8x8x4"));
      vs.push_back(make_tile_occ<Spec<256, 8, 8, 4>, double, 8, true, true, 1, false, 8>("8x8x4"));
      run_all<double>(vs, 256, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<768, 8, 8, 4, 3>, double, 8, true, true, 1>("8x8x4x3"));                     // shipped (split, LDS twiddles)
      vs.push_back(make_tile_occ<Spec<768, 4, 4, 4, 4, 3>, double, 8, true, true, 1, false, 2>("4x4x4x4x3"));
      vs.push_back(make_tile_occ<Spec<768, 4, 4, 4, 4, 3>, double, 8, true, true, 1, false, 3>("4x4x4x4x3"));
      vs.push_back(make_tile_occ<Spec<768, 4, 4, 4, 4, 3>, double, 8, false, true, 1, false, 3>("4x4x4x4x3"));
      run_all<double>(vs, 768, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("occ512", filter)) {      // round 3: 512 fp64 (BASELINE config 2): more workgroups per CU?
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<Spec<512, 8, 8, 8>, double, 8, false, false, 1>("8x8x8"));                       // shipped: 64 KB whole-complex exchange
    vs.push_back(make_tile_occ<Spec<512, 8, 8, 8>, double, 8, true, true, 1, false, 3>("8x8x8"));           // split + LDS twiddles, 3 per CU
    vs.push_back(make_tile_occ<Spec<512, 8, 8, 8>, double, 8, true, true, 1, false, 4>("8x8x8"));           // 4 per CU (64 VGPRs)
    vs.push_back(make_tile_occ<Spec<512, 8, 8, 8>, double, 8, true, true, 1, true, 4>("8x8x8"));            // ... non-temporal
    vs.push_back(make_tile_occ<Spec<512, 8, 8, 4, 2>, double, 8, true, true, 1, false, 4>("8x8x4x2"));
    vs.push_back(make_tile_occ<Spec<512, 8, 4, 4, 4>, double, 8, true, true, 1, false, 4>("8x4x4x4"));
    vs.push_back(make_tile_occ<Spec<512, 4, 4, 4, 4, 2>, double, 8, true, true, 1, false, 2>("4x4x4x4x2"));  // E = 4, 1024 threads, 2 per CU
    run_all<double>(vs, 512, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("small", filter)) {       // 384 / 640 / 576: fewer values per thread, more threads per CU?
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<384, 8, 8, 3, 2>, double, 8, true, false, 1>("8x8x3x2"));
      vs.push_back(make_tile<Spec<384, 4, 4, 4, 3, 2>, double, 8, true, false, 1>("4x4x4x3x2"));
      vs.push_back(make_tile_occ<Spec<384, 4, 4, 4, 3, 2>, double, 8, true, false, 1, false, 3>("4x4x4x3x2"));
      vs.push_back(make_tile_occ<Spec<384, 4, 4, 4, 3, 2>, double, 8, true, true, 1, false, 4>("4x4x4x3x2"));
      vs.push_back(make_tile<Spec<384, 8, 4, 4, 3>, double, 8, true, false, 1>("8x4x4x3"));
      run_all<double>(vs, 384, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<640, 8, 4, 4, 5>, double, 8, false, false, 1>("8x4x4x5"));
      vs.push_back(make_tile<Spec<640, 4, 4, 4, 5, 2>, double, 8, false, false, 1>("4x4x4x5x2"));
      vs.push_back(make_tile_occ<Spec<640, 4, 4, 4, 5, 2>, double, 8, true, true, 1, false, 3>("4x4x4x5x2"));
      vs.push_back(make_tile_occ<Spec<640, 8, 4, 4, 5>, double, 8, true, true, 1, false, 3>("8x4x4x5"));
      vs.push_back(make_tile_occ<Spec<640, 4, 4, 4, 5, 2>, double, 8, false, false, 1, false, 2>("4x4x4x5x2"));
      run_all<double>(vs, 640, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<576, 8, 8, 3, 3>, double, 8, false, false, 1>("8x8x3x3"));
      vs.push_back(make_tile_occ<Spec<576, 4, 4, 4, 3, 3>, double, 8, false, false, 1, false, 2>("4x4x4x3x3"));
      vs.push_back(make_tile_occ<Spec<576, 4, 4, 4, 3, 3>, double, 8, true, true, 1, false, 3>("4x4x4x3x3"));
      run_all<double>(vs, 576, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("occ", filter)) {         // 1152 with 12 values per thread and a register cap for two workgroups per CU
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, double, 8, false, true, 1>("8x8x3x3x2"));
    vs.push_back(make_tile_occ<Spec<1152, 4, 4, 4, 3, 3, 2>, double, 8, false, true, 1, false, 2>("4x4x4x3x3x2"));
    vs.push_back(make_tile_occ<Spec<1152, 8, 8, 3, 3, 2>, double, 8, false, true, 1, false, 2>("8x8x3x3x2"));
    vs.push_back(make_tile_occ<Spec<1152, 4, 4, 4, 3, 3, 2>, double, 8, false, true, 1, true, 2>("4x4x4x3x3x2"));
    run_all<double>(vs, 1152, "", rounds);
    std::vector<Variant<double>> v2;
    v2.push_back(make_tile<Spec<1280, 8, 8, 4, 5>, double, 8, false, true, 1>("8x8x4x5"));
    v2.push_back(make_tile_occ<Spec<1280, 4, 4, 4, 4, 5>, double, 8, false, true, 1, false, 2>("4x4x4x4x5"));
    v2.push_back(make_tile_occ<Spec<1280, 8, 8, 4, 5>, double, 8, false, true, 1, false, 2>("8x8x4x5"));
    run_all<double>(v2, 1280, "", rounds);
    return 0;
  }
  if (filter[0] && strstr("twl", filter)) {         // LDS twiddles at the long lengths with their present plans
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, true, 1>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, true, true, 1>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, true, true, 1, true>("8x8x8x3"));
      run_all<double>(vs, 1536, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, double, 8, false, true, 1>("8x8x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, double, 8, true, true, 1>("8x8x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, double, 8, true, true, 1, true>("8x8x3x3x2"));
      run_all<double>(vs, 1152, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("long", filter)) {        // lengths above 1024 in double precision: threads per CU against values per thread
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, double, 8, false, true, 1>("8x8x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 8, 8, 3, 3, 2>, double, 8, false, true, 1, true>("8x8x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 4, 4, 4, 3, 3, 2>, double, 8, false, true, 1>("4x4x4x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 4, 4, 4, 3, 3, 2>, double, 8, false, true, 1, true>("4x4x4x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 4, 4, 4, 3, 3, 2>, double, 8, true, true, 1>("4x4x4x3x3x2"));
      vs.push_back(make_tile<Spec<1152, 4, 4, 4, 3, 3, 2>, double, 8, true, true, 1, true>("4x4x4x3x3x2"));
      run_all<double>(vs, 1152, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1280, 8, 8, 4, 5>, double, 8, false, true, 1>("8x8x4x5"));
      vs.push_back(make_tile<Spec<1280, 8, 8, 4, 5>, double, 8, false, true, 1, true>("8x8x4x5"));
      vs.push_back(make_tile<Spec<1280, 4, 4, 4, 4, 5>, double, 8, false, true, 1>("4x4x4x4x5"));
      vs.push_back(make_tile<Spec<1280, 4, 4, 4, 4, 5>, double, 8, false, true, 1, true>("4x4x4x4x5"));
      vs.push_back(make_tile<Spec<1280, 4, 4, 4, 4, 5>, double, 8, true, true, 1>("4x4x4x4x5"));
      vs.push_back(make_tile<Spec<1280, 4, 4, 4, 4, 5>, double, 8, true, true, 1, true>("4x4x4x4x5"));
      run_all<double>(vs, 1280, "", rounds);
    }
    {
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, true, 1>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 8, 8, 8, 3>, double, 8, false, true, 1, true>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 8, false, true, 1>("4x4x4x4x3x2"));
      vs.push_back(make_tile<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 8, false, true, 1, true>("4x4x4x4x3x2"));
      vs.push_back(make_tile<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 8, true, true, 1>("4x4x4x4x3x2"));
      vs.push_back(make_tile<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 8, true, true, 1, true>("4x4x4x4x3x2"));
      run_all<double>(vs, 1536, "", rounds);
    }
    return 0;
  }
  if (filter[0] && strstr("sizes", filter)) {       // other lengths whose whole-complex exchange leaves one workgroup per CU
    {
      typedef Spec<768, 8, 8, 4, 3> S768;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S768, double, 8, false, false, 1>("8x8x4x3"));
      vs.push_back(make_tile<S768, double, 8, true, true, 1>("8x8x4x3"));
      vs.push_back(make_tile<S768, double, 8, false, true, 1>("8x8x4x3"));
      vs.push_back(make_tile<Spec<768, 4, 4, 4, 4, 3>, double, 8, true, true, 1>("4x4x4x4x3"));
      run_all<double>(vs, 768, "", rounds);
    }
    {
      typedef Spec<800, 5, 5, 4, 4, 2> S800;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S800, double, 8, false, false, 1>("5x5x4x4x2"));
      vs.push_back(make_tile<S800, double, 8, true, true, 1>("5x5x4x4x2"));
      vs.push_back(make_tile<S800, double, 8, false, true, 1>("5x5x4x4x2"));
      run_all<double>(vs, 800, "", rounds);
    }
    {
      typedef Spec<1536, 8, 8, 8, 3> S1536;
      std::vector<Variant<double>> vs;
      vs.push_back(make_tile<S1536, double, 8, false, true, 1>("8x8x8x3"));
      vs.push_back(make_tile<Spec<1536, 4, 4, 4, 4, 3, 2>, double, 8, false, true, 1>("4x4x4x4x3x2"));
      vs.push_back(make_tile<Spec<1536, 8, 8, 4, 3, 2>, double, 8, false, true, 1>("8x8x4x3x2"));
      run_all<double>(vs, 1536, "", rounds);
    }
    return 0;
  }
  {
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<SA, double, 8, false, false, 1>("16x8x8"));
    vs.push_back(make_tile<SA, double, 8, false, false, 1, true>("16x8x8"));
    vs.push_back(make_tile<SA, double, 8, true, true, 1>("16x8x8"));
    vs.push_back(make_tile<SA, double, 8, true, true, 1, true>("16x8x8"));
    vs.push_back(make_tile<SD, double, 8, true, true, 1>("8x8x4x4"));
    vs.push_back(make_tile<SD, double, 8, true, true, 1, true>("8x8x4x4"));
    vs.push_back(make_tile<SD, double, 8, false, true, 1>("8x8x4x4"));
    vs.push_back(make_tile<SF, double, 8, true, true, 1>("8x8x8x2"));
    run_all<double>(vs, 1024, filter, rounds);
  }
  {
    std::vector<Variant<float>> vs;
    vs.push_back(make_tile<SA, float, 16, false, false, 2>("16x8x8"));
    vs.push_back(make_tile<SA, float, 16, false, false, 2, true>("16x8x8"));
    vs.push_back(make_tile<SA, float, 16, true, true, 2>("16x8x8"));
    vs.push_back(make_tile<SD, float, 16, true, true, 2>("8x8x4x4"));
    vs.push_back(make_tile<SD, float, 16, true, true, 2, true>("8x8x4x4"));
    vs.push_back(make_tile<SD, float, 16, true, false, 1>("8x8x4x4"));
    run_all<float>(vs, 1024, filter, rounds);
  }
  {
    std::vector<Variant<double>> vs;
    vs.push_back(make_tile<S512, double, 8, true, false, 1>("8x8x8"));
    vs.push_back(make_tile<S512, double, 8, true, false, 1, true>("8x8x8"));
    vs.push_back(make_tile<S512, double, 8, true, true, 1>("8x8x8"));
    vs.push_back(make_tile<Spec<512, 8, 4, 4, 4>, double, 8, true, false, 1>("8x4x4x4"));
    vs.push_back(make_tile<Spec<512, 8, 4, 4, 4>, double, 8, true, true, 1>("8x4x4x4"));
    run_all<double>(vs, 512, filter, rounds);
  }
  return 0;
}
