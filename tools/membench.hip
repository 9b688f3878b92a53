// membench.hip -- memory-system yardsticks for the strided-axis kernels (developer tool, not part of the library).
//
// Answers, on the box it runs on: what does a plain 16-byte-per-lane copy reach (the guide quotes 6.29 TB/s), and
// which property of the ColFft access pattern (in place, 128-byte row segments at a large pitch, loads-then-stores
// phases, row pitch) costs how much of it.  Also carries a stamped diagnostic build of the strided FFT kernel that
// records, per workgroup, when its loads landed, its passes ended and its stores drained.
//   make -C tools membench && tools/build/membench [filter]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
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

typedef double d2 __attribute__((ext_vector_type(2)));   // 16 bytes

template <int NT> __device__ __forceinline__ d2 ld(const d2* p) {
  if constexpr (NT & 1) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int NT> __device__ __forceinline__ void st(d2* p, d2 x) {
  if constexpr (NT & 2) __builtin_nontemporal_store(x, p);
  else *p = x;
}

// ---- 1. grid-stride streaming copy, U independent 16-byte loads in flight per lane -------------------------
template <int NT, int U>
__global__ __launch_bounds__(256) void k_stream(const d2* __restrict__ src, d2* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) { d2 x = v[u]; x.x += 1.0; st<NT>(dst + i + u * stride, x); }
  }
  for (; i < n; i += stride) { d2 x = ld<NT>(src + i); x.x += 1.0; st<NT>(dst + i, x); }
}

// ---- 2. one contiguous chunk per workgroup: T threads x E values, all loads, then all stores ---------------
template <int NT, int T, int E>
__global__ __launch_bounds__(T) void k_chunk(const d2* __restrict__ src, d2* __restrict__ dst, size_t n) {
  const size_t base = (size_t)blockIdx.x * (T * E) + threadIdx.x;
  d2 v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) {
    size_t i = base + (size_t)k * T;
    v[k] = ld<NT>(src + (i < n ? i : n - 1));
  }
#pragma unroll
  for (int k = 0; k < E; ++k) {
    size_t i = base + (size_t)k * T;
    d2 x = v[k];
    x.x += 1.0;
    if (i < n) st<NT>(dst + i, x);
  }
}

// ---- 3. read only / write only ------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void k_read(const d2* __restrict__ src, d2* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  d2 acc = {0.0, 0.0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 7 * stride < n; i += 8 * stride) {
    d2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld<NT>(src + i + u * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc.x == 1.2345e300) dst[threadIdx.x] = acc;   // never true: keeps the loads alive
}
template <int NT>
__global__ __launch_bounds__(256) void k_write(const d2* __restrict__, d2* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const d2 x = {1.0, (double)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) st<NT>(dst + i, x);
}

// ---- 4. the ColFft tile pattern, no FFT: COLS x 1024-row tiles, E values per thread -------------------------
template <int COLS, int E, int NT>
__global__ __launch_bounds__(1024 / E * COLS) void k_tile(ColParams<double> P) {
  constexpr int TPT = 1024 / E;
  const int bid = P.remap ? xcd_remap((int)blockIdx.x, P.ntile_c * P.nouter) : (int)blockIdx.x;
  const int outer = bid / P.ntile_c, tc = bid - outer * P.ntile_c;
  const int c = threadIdx.x % COLS, j = threadIdx.x / COLS, col = tc * COLS + c;
  if (col >= P.ncols) return;
  const d2* ip = reinterpret_cast<const d2*>(P.in + (i64)outer * P.in_outer + col);
  d2* op = reinterpret_cast<d2*>(P.out + (i64)outer * P.out_outer + col);
  d2 v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = ld<NT>(ip + row_off(P.in_map, (unsigned)(j + k * TPT)));
#pragma unroll
  for (int k = 0; k < E; ++k) {
    d2 x = v[k];
    x.x += 1.0;
    st<NT>(op + row_off(P.out_map, (unsigned)(j + k * TPT)), x);
  }
}

// ---- 4b. the same for other lengths: NR rows per tile, E values per thread (occupancy as the launch allows) ------
template <int NR, int COLS, int E, int NT>
__global__ __launch_bounds__(NR / E * COLS) void k_tile_n(ColParams<double> P) {
  constexpr int TPT = NR / E;
  const int bid = P.remap ? xcd_remap((int)blockIdx.x, P.ntile_c * P.nouter) : (int)blockIdx.x;
  const int outer = bid / P.ntile_c, tc = bid - outer * P.ntile_c;
  const int c = threadIdx.x % COLS, j = threadIdx.x / COLS, col = tc * COLS + c;
  if (col >= P.ncols) return;
  const d2* ip = reinterpret_cast<const d2*>(P.in + (i64)outer * P.in_outer + col);
  d2* op = reinterpret_cast<d2*>(P.out + (i64)outer * P.out_outer + col);
  d2 v[E];
#pragma unroll
  for (int k = 0; k < E; ++k) v[k] = ld<NT>(ip + row_off(P.in_map, (unsigned)(j + k * TPT)));
#pragma unroll
  for (int k = 0; k < E; ++k) {
    d2 x = v[k];
    x.x += 1.0;
    st<NT>(op + row_off(P.out_map, (unsigned)(j + k * TPT)), x);
  }
}

// ---- 5. stamped diagnostic build of the strided FFT (shares of a workgroup's life; never quote its run time) --
struct Stamp { unsigned long long t[8]; unsigned xcc, cu; };
__device__ __forceinline__ unsigned long long stamp_now() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long stamp_nowait() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <class S, int COLS, bool SPLIT, int WPS = (SPLIT ? 2 : 1) * (S::TPT * COLS / 256)>      // WPS: waves per SIMD the register cap admits
__global__ __launch_bounds__(S::TPT* COLS, WPS) void k_stamped(ColParams<double> P, Stamp* stamps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef double T;
  const int bid = P.remap ? xcd_remap((int)blockIdx.x, P.ntile_c * P.nouter) : (int)blockIdx.x;
  const int outer = bid / P.ntile_c, tc = bid - outer * P.ntile_c;
  const int c = threadIdx.x % COLS, j = threadIdx.x / COLS, col = tc * COLS + c;
  const bool act = col < P.ncols;
  const cx<T>* ip = P.in + (i64)outer * P.in_outer + (act ? col : P.ncols - 1);
  cx<T>* op = P.out + (i64)outer * P.out_outer + col;
  struct Slot { int c; __device__ int operator()(int pos) const { return pos * COLS + c; } };
  unsigned long long t0, t1, t2, t3, t4, rt;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
  t0 = stamp_nowait();
  cx<T> v[S::E];
#pragma unroll
  for (int k = 0; k < S::E; ++k) v[k] = ip[row_off(P.in_map, (unsigned)(j + k * S::TPT))];
  t1 = stamp_now();                                   // this wave's loads have landed
  if constexpr (SPLIT) {
    XchSplit<T, Slot> xch{reinterpret_cast<T*>(lds), Slot{c}};
    run_passes<S, 0, T>(v, j, P.tw, xch);
  } else {
    XchFull<T, Slot> xch{reinterpret_cast<cx<T>*>(lds), Slot{c}};
    run_passes<S, 0, T>(v, j, P.tw, xch);
  }
  t2 = stamp_now();                                   // passes done
  if (act) {
#pragma unroll
    for (int k = 0; k < S::E; ++k) op[row_off(P.out_map, (unsigned)(j + k * S::TPT))] = scale(v[k], P.scale);
  }
  t3 = stamp_nowait();                                // stores issued
  t4 = stamp_now();                                   // stores drained
  unsigned long long rt_end;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_end)::"memory");
  if (threadIdx.x == 0) {
    Stamp s;
    s.t[0] = t0; s.t[1] = t1; s.t[2] = t2; s.t[3] = t3; s.t[4] = t4; s.t[5] = rt; s.t[6] = rt_end; s.t[7] = 0;
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    s.xcc = xcc; s.cu = hwid;
    stamps[blockIdx.x] = s;
  }
}

// ---- 6. stamped diagnostic build of the PERSISTENT strided FFT (ColFftP): where does an iteration go? ------------
struct PStamp { unsigned long long wait, issue, passes, stores, total, rt0, rt1, ntiles; };
__device__ __forceinline__ unsigned long long stamp_lgkm() {
  unsigned long long t;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
template <class S, int COLS, int EARLY>
__global__ __launch_bounds__(S::TPT* COLS) void k_pstamped(ColParams<double> P, PStamp* stamps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  typedef double T;
  typedef ColFftP<S, T, COLS, false, 1> K;
  cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
  char* xbuf = lds + K::TW_BYTES;
  typename K::Thread th;
  const int tid = threadIdx.x, bid = blockIdx.x;
  th.c = tid % COLS;
  th.j = tid / COLS;
  th.vin = (unsigned)(((i64)th.j * P.in_map.lo + th.c) * (i64)sizeof(cx<T>));
  th.vout = (unsigned)(((i64)th.j * P.out_map.lo + th.c) * (i64)sizeof(cx<T>));
  const int nfc = P.ncols / COLS, nfull = nfc * P.nouter;
  int first, end, step;
  K::share(P, bid, nfull, first, end, step);
  unsigned long long rt0, rt1, w = 0, is = 0, ps = 0, st = 0, n = 0;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory");
  const unsigned long long tstart = stamp_nowait();
  int t = first;
  cx<T> a[1][S::E], b[1][S::E];
  if (t < end) K::template load_tile<false>(P, th, t / nfc, t % nfc, a);
  stage_twiddles<S, T>(ltw, P.tw, tid, K::THREADS);
  __syncthreads();
  auto stepf = [&](cx<T> (&cur)[1][S::E], cx<T> (&nxt)[1][S::E]) -> bool {
    const int tn = t + step;
    const int tl = tn < end ? tn : t;
    const unsigned long long s0 = stamp_nowait();
    const unsigned long long s1 = stamp_now();                 // loads of cur landed, stores of the previous tile drained
    unsigned long long s2;
    PackV<cx<T>, 1>* xb = reinterpret_cast<PackV<cx<T>, 1>*>(xbuf);
    typename K::NoPrefetch np;
    if (EARLY == 2) {
      const char* ib; char* ob;
      K::tile_base(P, tl / nfc, tl % nfc, ib, ob);
      typename K::Prefetch pf{P, th, ib, nxt};
      s2 = s1;
      K::template passes<0>(cur, th, ltw, xb, pf);
    } else if (EARLY == 1) {
      K::template load_tile<false>(P, th, tl / nfc, tl % nfc, nxt);
      s2 = stamp_nowait();
      K::template passes<0>(cur, th, ltw, xb, np);
    } else {
      s2 = s1;
      K::template passes<0>(cur, th, ltw, xb, np);
    }
    const unsigned long long s3 = stamp_lgkm();
    if (!EARLY) K::template load_tile<false>(P, th, tl / nfc, tl % nfc, nxt);
    K::template store_tile<false>(P, th, t / nfc, t % nfc, cur);
    const unsigned long long s4 = stamp_nowait();
    w += s1 - s0; is += s2 - s1; ps += s3 - s2; st += s4 - s3; n += 1;
    if (tn >= end) return false;
    t = tn;
    return true;
  };
  if (t < end) {
    for (;;) {
      if (!stepf(a, b)) break;
      if (!stepf(b, a)) break;
    }
  }
  const unsigned long long tend = stamp_now();
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
  if (threadIdx.x == 0) {
    PStamp s;
    s.wait = w; s.issue = is; s.passes = ps; s.stores = st; s.total = tend - tstart; s.rt0 = rt0; s.rt1 = rt1; s.ntiles = n;
    stamps[blockIdx.x] = s;
  }
}

// ---- harness --------------------------------------------------------------------------------------------------
static hipEvent_t e0, e1;
template <class F>
static double time_ms(F launch, int warm = 2, int reps = 5) {
  for (int i = 0; i < warm; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}
static const char* g_filter = "";
static bool want(const char* name) { return !g_filter[0] || strstr(name, g_filter); }
static void report(const char* name, double ms, double bytes) {
  printf("%-58s %7.3f ms  %6.0f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
  fflush(stdout);
}

template <int COLS, int E, int NT>
static void run_tile(const char* tag, cx<double>* in, cx<double>* out, int pitch, bool xdir, int remap) {
  const int N = 1024, NF = 513;
  char nm[160];
  snprintf(nm, sizeof nm, "tile c%d e%d nt%d %s p%d %s%s", COLS, E, NT, xdir ? "x" : "y", pitch, in == out ? "inplace" : "outofplace",
           remap ? " rm" : "");
  if (!want(nm) && !want(tag)) return;
  ColParams<double> P;
  memset(&P, 0, sizeof P);
  P.in = in; P.out = out; P.tw = nullptr; P.remap = remap; P.scale = 1.0;
  if (!xdir) {
    P.in_outer = P.out_outer = (i64)N * pitch;
    P.in_map = P.out_map = make_rowmap(0, pitch, N, N);
    P.ncols = NF; P.nouter = N;
  } else {
    P.in_outer = P.out_outer = 0;
    P.in_map = P.out_map = make_rowmap(0, (i64)N * pitch, N, N);
    P.ncols = N * pitch; P.nouter = 1;
  }
  P.ntile_c = (P.ncols + COLS - 1) / COLS;
  const int grid = P.ntile_c * P.nouter;
  const double ms = time_ms([&] { hipLaunchKernelGGL((k_tile<COLS, E, NT>), dim3(grid), dim3(1024 / E * COLS), 0, 0, P); });
  const double bytes = 2.0 * N * N * 16.0 * (xdir ? pitch : NF);
  report(nm, ms, bytes);
}

template <int NR, int COLS, int E, int NT>
static void run_tile_n(cx<double>* in, cx<double>* out, bool xdir, int lds_bytes) {
  const int N = NR, NF = NR / 2 + 1;
  char nm[160];
  snprintf(nm, sizeof nm, "tile N=%d c%d e%d nt%d %s %s lds%dK", NR, COLS, E, NT, xdir ? "x" : "y", in == out ? "inplace" : "outofplace", lds_bytes / 1024);
  ColParams<double> P;
  memset(&P, 0, sizeof P);
  P.in = in; P.out = out; P.tw = nullptr; P.remap = 1; P.scale = 1.0;
  if (!xdir) {
    P.in_outer = P.out_outer = (i64)N * NF;
    P.in_map = P.out_map = make_rowmap(0, NF, N, N);
    P.ncols = NF; P.nouter = N;
  } else {
    P.in_outer = P.out_outer = 0;
    P.in_map = P.out_map = make_rowmap(0, (i64)N * NF, N, N);
    P.ncols = N * NF; P.nouter = 1;
  }
  P.ntile_c = (P.ncols + COLS - 1) / COLS;
  const int grid = P.ntile_c * P.nouter;
  if (lds_bytes > 65536)
    CK(hipFuncSetAttribute((const void*)k_tile_n<NR, COLS, E, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  const double ms = time_ms([&] { hipLaunchKernelGGL((k_tile_n<NR, COLS, E, NT>), dim3(grid), dim3(NR / E * COLS), lds_bytes, 0, P); });
  report(nm, ms, 2.0 * N * N * 16.0 * NF);
}

template <class S, int COLS, bool SPLIT, int WPS = (SPLIT ? 2 : 1) * (S::TPT * COLS / 256)>
static void run_stamped(const char* plan, cx<double>* buf, bool xdir) {
  const int N = S::N, NF = S::N / 2 + 1, pitch = NF;
  char nm[160];
  snprintf(nm, sizeof nm, "stamped %d %s c%d%s wps%d %s", N, plan, COLS, SPLIT ? " split" : "", WPS, xdir ? "x" : "y");
  if (!want(nm)) return;
  auto twh = build_pass_twiddles<S, double>();
  cx<double>* tw = nullptr;
  CK(hipMalloc(&tw, twh.size() * sizeof(cx<double>)));
  CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<double>), hipMemcpyHostToDevice));
  ColParams<double> P;
  memset(&P, 0, sizeof P);
  P.in = buf; P.out = buf; P.tw = tw; P.remap = 1; P.scale = 1.0;
  if (!xdir) {
    P.in_outer = P.out_outer = (i64)N * pitch;
    P.in_map = P.out_map = make_rowmap(0, pitch, N, N);
    P.ncols = NF; P.nouter = N;
  } else {
    P.in_outer = P.out_outer = 0;
    P.in_map = P.out_map = make_rowmap(0, (i64)N * pitch, N, N);
    P.ncols = N * pitch; P.nouter = 1;
  }
  P.ntile_c = (P.ncols + COLS - 1) / COLS;
  const int grid = P.ntile_c * P.nouter;
  const int LDS = S::N * COLS * (SPLIT ? 8 : 16);
  CK(hipFuncSetAttribute((const void*)k_stamped<S, COLS, SPLIT, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_stamped<S, COLS, SPLIT, WPS>, S::TPT * COLS, LDS));
  Stamp* ds = nullptr;
  CK(hipMalloc(&ds, sizeof(Stamp) * grid));
  const double ms = time_ms([&] { hipLaunchKernelGGL((k_stamped<S, COLS, SPLIT, WPS>), dim3(grid), dim3(S::TPT * COLS), LDS, 0, P, ds); }, 2, 3);
  std::vector<Stamp> hs(grid);
  CK(hipMemcpy(hs.data(), ds, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
  // shares (shader cycles of wave 0): load wait, passes, store issue, store drain
  std::vector<double> ph[5];
  unsigned long long rt0 = ~0ull, rt1 = 0;
  {  // shader clock = d(memtime)/d(memrealtime) * 100 MHz; busy fraction of a CU = sum of workgroup lifetimes on it / span
    double cyc = 0, rtt = 0;
    for (auto& s : hs) { cyc += (double)(s.t[4] - s.t[0]); rtt += (double)(s.t[6] - s.t[5]); }
    unsigned long long a0 = ~0ull, a1 = 0;
    for (auto& s : hs) { a0 = std::min(a0, s.t[5]); a1 = std::max(a1, s.t[6]); }
    printf("    shader clock %.0f MHz; sum of lifetimes %.3f ms over a span of %.3f ms x 256 CUs x occ %d -> busy %.1f %%\n", cyc / rtt * 100.0,
           rtt / 100.0 / 1e3, (double)(a1 - a0) / 100.0 / 1e3, occ, rtt / ((double)(a1 - a0) * 256.0 * occ) * 100.0);
  }
  for (auto& s : hs) {
    for (int i = 0; i < 4; ++i) ph[i].push_back((double)(s.t[i + 1] - s.t[i]));
    ph[4].push_back((double)(s.t[4] - s.t[0]));
    rt0 = std::min(rt0, s.t[5]);
    rt1 = std::max(rt1, s.t[5]);
  }
  auto med = [](std::vector<double>& v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
  printf("%-34s %7.3f ms occ/CU=%d grid=%d | cycles med (p10..p90): load %.0f (%.0f..%.0f)  passes %.0f (%.0f..%.0f)  st-issue %.0f (%.0f..%.0f)  st-drain %.0f (%.0f..%.0f)  total %.0f (%.0f..%.0f) | first->last start %.1f us\n",
         nm, ms, occ, grid, med(ph[0], .5), med(ph[0], .1), med(ph[0], .9), med(ph[1], .5), med(ph[1], .1), med(ph[1], .9),
         med(ph[2], .5), med(ph[2], .1), med(ph[2], .9), med(ph[3], .5), med(ph[3], .1), med(ph[3], .9), med(ph[4], .5),
         med(ph[4], .1), med(ph[4], .9), (double)(rt1 - rt0) / 100.0);
  {  // round 5: per CU, how the phases of its resident workgroups lie against each other (shader-clock stamps of one CU share
     // a counter): the share of the CU's span in which at least one workgroup has loads or stores in flight ("memory"), at
     // least one runs its passes ("passes"), both at once ("overlapped"), and neither
    std::map<unsigned long long, std::vector<const Stamp*>> cus;
    for (auto& st : hs) cus[((unsigned long long)st.xcc << 32) | ((st.cu >> 8) & 0xFFu)].push_back(&st);
    double span = 0, mem = 0, pas = 0, both = 0, none = 0;
    for (auto& kv : cus) {
      std::vector<std::pair<unsigned long long, int>> ev;      // (time, kind): +-1 memory, +-2 passes
      unsigned long long lo = ~0ull, hi = 0;
      for (const Stamp* st : kv.second) {
        ev.push_back({st->t[0], 1}); ev.push_back({st->t[1], -1});
        ev.push_back({st->t[1], 2}); ev.push_back({st->t[2], -2});
        ev.push_back({st->t[2], 1}); ev.push_back({st->t[4], -1});
        lo = std::min(lo, st->t[0]); hi = std::max(hi, st->t[4]);
      }
      std::sort(ev.begin(), ev.end());
      int nm_ = 0, np_ = 0;
      unsigned long long prev = lo;
      for (auto& e : ev) {
        const double dt = (double)(e.first - prev);
        if (nm_ > 0) mem += dt;
        if (np_ > 0) pas += dt;
        if (nm_ > 0 && np_ > 0) both += dt;
        if (nm_ == 0 && np_ == 0) none += dt;
        prev = e.first;
        if (e.second == 1) ++nm_; else if (e.second == -1) --nm_; else if (e.second == 2) ++np_; else --np_;
      }
      span += (double)(hi - lo);
    }
    printf("    %zu CUs: of a CU's span, memory phase %.1f %%, passes %.1f %%, both at once %.1f %%, neither %.1f %%\n", cus.size(), mem / span * 100,
           pas / span * 100, both / span * 100, none / span * 100);
  }
  // effective clock: sum of per-workgroup lifetimes / (workgroups resident at once * wall) is not known here; print the
  // mean lifetime so that lifetime * grid / (256 CUs * occ) can be compared with the wall time
  double sum = 0;
  for (double x : ph[4]) sum += x;
  printf("    mean lifetime %.0f cycles; grid*lifetime/(256*occ) = %.3f ms at 2.4 GHz (wall %.3f ms)\n", sum / grid,
         sum / (256.0 * occ) / 2.4e9 * 1e3, ms);
  fflush(stdout);
  CK(hipFree(ds));
  CK(hipFree(tw));
}


template <int COLS, int E, int NT>
static void tile_mall_probe(cx<double>* buf) {
  const int N = 1024, NF = 513;
  printf("---- Infinity Cache probe, tile copy c%d e%d nt%d (in place)\n", COLS, E, NT);
  for (int np : {4, 8, 16, 32, 64}) {
    ColParams<double> P;
    memset(&P, 0, sizeof P);
    P.remap = 1; P.scale = 1.0;
    P.in_outer = P.out_outer = (i64)N * NF;
    P.in_map = P.out_map = make_rowmap(0, NF, N, N);
    P.ncols = NF; P.nouter = np;
    P.ntile_c = (P.ncols + COLS - 1) / COLS;
    const int grid = P.ntile_c * P.nouter, nsec = N / np, reps = 64;
    double t[2];
    for (int mode = 0; mode < 2; ++mode) {
      int r = 0;
      t[mode] = time_ms([&] {
        cx<double>* sec = buf + (mode == 0 ? 0 : (size_t)((r++ * 37) % nsec) * np * N * NF);
        P.in = P.out = sec;
        hipLaunchKernelGGL((k_tile<COLS, E, NT>), dim3(grid), dim3(1024 / E * COLS), 0, 0, P);
      }, 4, reps);
    }
    const double bytes = 2.0 * np * N * NF * 16.0;
    printf("   y pattern, %2d planes (%6.1f MB): same section %.4f ms (%.0f GB/s)   rotating sections %.4f ms (%.0f GB/s)\n", np, bytes / 2e6, t[0],
           bytes / t[0] / 1e6, t[1], bytes / t[1] / 1e6);
    fflush(stdout);
  }
}

template <class S, int COLS, int EARLY>
static void run_pstamped(const char* plan, cx<double>* buf, bool xdir) {
  const int N = 1024, NF = 513, pitch = 513;
  char nm[160];
  snprintf(nm, sizeof nm, "pstamped %s c%d %s %s", plan, COLS, EARLY == 2 ? "prefetch-spread" : EARLY ? "prefetch-before-passes" : "prefetch-after-passes", xdir ? "x" : "y");
  if (!want(nm)) return;
  typedef ColFftP<S, double, COLS, false, 1> K;
  auto twh = build_pass_twiddles<S, double>();
  cx<double>* tw = nullptr;
  CK(hipMalloc(&tw, twh.size() * sizeof(cx<double>)));
  CK(hipMemcpy(tw, twh.data(), twh.size() * sizeof(cx<double>), hipMemcpyHostToDevice));
  ColParams<double> P;
  memset(&P, 0, sizeof P);
  P.in = buf; P.out = buf; P.tw = tw; P.remap = 1; P.scale = 1.0;
  if (!xdir) {
    P.in_outer = P.out_outer = (i64)N * pitch;
    P.in_map = P.out_map = make_rowmap(0, pitch, N, N);
    P.ncols = NF; P.nouter = N;
  } else {
    P.in_outer = P.out_outer = 0;
    P.in_map = P.out_map = make_rowmap(0, (i64)N * pitch, N, N);
    P.ncols = N * pitch; P.nouter = 1;
  }
  P.ntile_c = (P.ncols + COLS - 1) / COLS;
  const int grid = 256;
  P.nblocks = grid;
  CK(hipFuncSetAttribute((const void*)k_pstamped<S, COLS, EARLY>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
  PStamp* ds = nullptr;
  CK(hipMalloc(&ds, sizeof(PStamp) * grid));
  const double ms = time_ms([&] { hipLaunchKernelGGL((k_pstamped<S, COLS, EARLY>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, 0, P, ds); }, 2, 3);
  std::vector<PStamp> hs(grid);
  CK(hipMemcpy(hs.data(), ds, sizeof(PStamp) * grid, hipMemcpyDeviceToHost));
  double w = 0, is = 0, ps = 0, st = 0, tot = 0, n = 0, rt = 0;
  unsigned long long a0 = ~0ull, a1 = 0, lmin = ~0ull, lmax = 0;
  for (auto& s : hs) {
    w += s.wait; is += s.issue; ps += s.passes; st += s.stores; tot += s.total; n += s.ntiles; rt += (double)(s.rt1 - s.rt0);
    a0 = std::min(a0, s.rt0); a1 = std::max(a1, s.rt1);
    lmin = std::min(lmin, s.rt1 - s.rt0); lmax = std::max(lmax, s.rt1 - s.rt0);
  }
  printf("%-52s %7.3f ms | per tile (cycles): wait %.0f  load-issue %.0f  passes %.0f  stores %.0f  = %.0f of %.0f | clock %.0f MHz | workgroup life min %.3f max %.3f ms, span %.3f ms\n",
         nm, ms, w / n, is / n, ps / n, st / n, (w + is + ps + st) / n, tot / n, tot / rt * 100.0, lmin / 100.0 / 1e3, lmax / 100.0 / 1e3,
         (double)(a1 - a0) / 100.0 / 1e3);
  fflush(stdout);
  CK(hipFree(ds));
  CK(hipFree(tw));
}

int main(int argc, char** argv) {
  g_filter = argc > 1 ? argv[1] : "";
  if (argc > 1 && !strcmp(argv[1], "stamp1200")) {  // round 5: phases of the 30-values strided kernel at one and at two workgroups per CU
    const size_t el = (size_t)1440 * 1440 * 721 + 4096;
    cx<double>* a = nullptr;
    CK(hipMalloc(&a, el * sizeof(cx<double>)));
    CK(hipMemset(a, 0, el * sizeof(cx<double>)));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    typedef Spec<1200, 10, 10, 6, 2> S12;
    typedef Spec<1024, 8, 8, 4, 4> S10;
    g_filter = "";
    for (int x = 0; x < 2; ++x) {
      run_stamped<S10, 8, true, 8>("8x8x4x4", a, x);
      run_stamped<S12, 8, true, 2>("10x10x6x2", a, x);
      run_stamped<S12, 8, true, 3>("10x10x6x2", a, x);
      run_stamped<S12, 4, true, 3>("10x10x6x2", a, x);      // 64-byte tiles: 160 threads = 3 waves, one per SIMD reserved: three resident
      run_stamped<S12, 4, false, 3>("10x10x6x2", a, x);     // whole-complex exchange (75 KB): two resident
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "tile1024w")) {  // round 5: does the bare tile pattern of 1024 get faster with wider tiles (256 / 512-byte row segments)?
    const size_t el = (size_t)1024 * 1024 * 584;
    cx<double>*a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, el * sizeof(cx<double>)));
    CK(hipMalloc(&b, el * sizeof(cx<double>)));
    CK(hipMemset(a, 0, el * sizeof(cx<double>)));
    CK(hipMemset(b, 0, el * sizeof(cx<double>)));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
      for (int x = 0; x < 2; ++x) {
        run_tile_n<1024, 8, 8, 0>(a, a, x, 80 << 10);      // as the FFT kernel runs: 1024 threads, two per CU
        run_tile_n<1024, 8, 8, 0>(a, a, x, 0);
        run_tile_n<1024, 16, 16, 0>(a, a, x, 80 << 10);    // 256-byte segments, 1024 threads, two per CU
        run_tile_n<1024, 16, 16, 0>(a, a, x, 0);
        run_tile_n<1024, 16, 32, 0>(a, a, x, 0);           // 512 threads
        run_tile_n<1024, 32, 32, 0>(a, a, x, 0);           // 512-byte segments, 1024 threads
        run_tile_n<1024, 8, 8, 0>(a, b, x, 80 << 10);
        run_tile_n<1024, 16, 16, 0>(a, b, x, 80 << 10);
      }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "tile1200")) {   // round 5: the bare tile pattern of the 30-values kernels (320 / 384 threads, one or two per CU)
    const size_t el = (size_t)1536 * 1536 * 769;
    cx<double>*a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, el * sizeof(cx<double>)));
    CK(hipMalloc(&b, el * sizeof(cx<double>)));
    CK(hipMemset(a, 0, el * sizeof(cx<double>)));
    CK(hipMemset(b, 0, el * sizeof(cx<double>)));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int x = 0; x < 2; ++x) {
      run_tile_n<1024, 8, 8, 0>(a, a, x, 80 << 10);
      run_tile_n<1200, 8, 30, 0>(a, a, x, 90 << 10);     // one workgroup per CU, as the FFT kernel runs
      run_tile_n<1200, 8, 30, 0>(a, a, x, 75 << 10);     // two
      run_tile_n<1200, 8, 30, 0>(a, a, x, 0);
      run_tile_n<1200, 8, 30, 0>(a, b, x, 90 << 10);
      run_tile_n<1200, 8, 10, 0>(a, a, x, 75 << 10);     // 960 threads, two per CU
      run_tile_n<1200, 8, 10, 0>(a, a, x, 0);
      run_tile_n<1440, 8, 30, 0>(a, a, x, 90 << 10);
      run_tile_n<1440, 8, 30, 0>(a, a, x, 0);
      run_tile_n<1440, 8, 10, 0>(a, a, x, 0);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "tilelong")) {   // the bare tile pattern at the lengths above 1024 (dynamic LDS only limits occupancy)
    const size_t el = (size_t)1536 * 1536 * 769;
    cx<double>*a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, el * sizeof(cx<double>)));
    CK(hipMalloc(&b, el * sizeof(cx<double>)));
    CK(hipMemset(a, 0, el * sizeof(cx<double>)));
    CK(hipMemset(b, 0, el * sizeof(cx<double>)));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int x = 0; x < 2; ++x) {
      run_tile_n<512, 8, 8, 0>(a, a, x, 72 << 10);
      run_tile_n<512, 8, 8, 0>(a, a, x, 40 << 10);
      run_tile_n<512, 8, 8, 0>(a, b, x, 40 << 10);
      run_tile_n<512, 8, 4, 0>(a, a, x, 72 << 10);
      run_tile_n<1024, 8, 8, 0>(a, a, x, 80 << 10);
      run_tile_n<1024, 8, 8, 0>(a, a, x, 0);
      run_tile_n<1152, 8, 24, 0>(a, a, x, 72 << 10);
      run_tile_n<1152, 8, 24, 0>(a, a, x, 0);
      run_tile_n<1152, 8, 12, 0>(a, a, x, 72 << 10);
      run_tile_n<1152, 8, 12, 0>(a, a, x, 0);
      run_tile_n<1152, 8, 24, 0>(a, b, x, 72 << 10);
      run_tile_n<1280, 8, 40, 0>(a, a, x, 80 << 10);
      run_tile_n<1280, 8, 20, 0>(a, a, x, 80 << 10);
      run_tile_n<1280, 8, 20, 0>(a, a, x, 0);
      run_tile_n<1536, 8, 24, 0>(a, a, x, 96 << 10);
      run_tile_n<1536, 8, 24, 0>(a, a, x, 48 << 10);
      run_tile_n<1536, 8, 12, 0>(a, a, x, 96 << 10);
      run_tile_n<1536, 8, 12, 0>(a, a, x, 0);
      run_tile_n<1536, 8, 24, 0>(a, b, x, 96 << 10);
    }
    return 0;
  }
  const int N = 1024;
  const size_t elems = (size_t)N * N * 584;             // complex128 elements per buffer (9.8 GB)
  const size_t n = (size_t)N * N * 513;                 // elements moved by the linear copies (8.6 GB each way)
  cx<double>*a = nullptr, *b = nullptr;
  CK(hipMalloc(&a, elems * sizeof(cx<double>)));
  CK(hipMalloc(&b, elems * sizeof(cx<double>)));
  CK(hipMemset(a, 0, elems * sizeof(cx<double>)));
  CK(hipMemset(b, 0, elems * sizeof(cx<double>)));
  {
    std::vector<cx<double>> h((size_t)4 * N * 520);
    for (size_t i = 0; i < h.size(); ++i)
      h[i] = mk<double>((double)((i * 2654435761u) % 1000) / 1000.0 - 0.5, (double)((i * 40503u) % 977) / 977.0 - 0.5);
    for (size_t off = 0; off < elems; off += h.size())
      CK(hipMemcpy(a + off, h.data(), std::min(h.size(), elems - off) * sizeof(cx<double>), hipMemcpyHostToDevice));
  }
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("buffers a=%p b=%p (a mod 2MiB = %zu, b mod 2MiB = %zu)\n", (void*)a, (void*)b, (size_t)a % (2u << 20), (size_t)b % (2u << 20));
  const d2* src = reinterpret_cast<const d2*>(a);
  d2* dsta = reinterpret_cast<d2*>(a);
  d2* dstb = reinterpret_cast<d2*>(b);
  const double bytes = 2.0 * n * 16.0;
  char nm[160];

  for (int round = 0; round < 2; ++round) {
    printf("---- linear copies, round %d\n", round);
#define STREAM(NT, U, GRID, DST, TAG)                                                                              \
  snprintf(nm, sizeof nm, "stream nt%d u%d grid%d %s", NT, U, GRID, TAG);                                            \
  if (want(nm)) report(nm, time_ms([&] { hipLaunchKernelGGL((k_stream<NT, U>), dim3(GRID), dim3(256), 0, 0, src, DST, n); }), bytes);
    STREAM(0, 1, 16384, dstb, "outofplace")
    STREAM(0, 4, 2048, dstb, "outofplace")
    STREAM(0, 4, 4096, dstb, "outofplace")
    STREAM(0, 8, 2048, dstb, "outofplace")
    STREAM(0, 8, 1024, dstb, "outofplace")
    STREAM(3, 4, 2048, dstb, "outofplace")
    STREAM(3, 8, 2048, dstb, "outofplace")
    STREAM(1, 8, 2048, dstb, "outofplace")
    STREAM(2, 8, 2048, dstb, "outofplace")
    STREAM(0, 4, 2048, dsta, "inplace")
    STREAM(0, 8, 2048, dsta, "inplace")
    STREAM(3, 8, 2048, dsta, "inplace")
    STREAM(0, 1, 16384, dsta, "inplace")
#define CHUNK(NT, T, E, DST, TAG)                                                                                  \
  snprintf(nm, sizeof nm, "chunk nt%d t%d e%d %s", NT, T, E, TAG);                                                   \
  if (want(nm))                                                                                                    \
    report(nm, time_ms([&] { hipLaunchKernelGGL((k_chunk<NT, T, E>), dim3((unsigned)((n + (size_t)T * E - 1) / ((size_t)T * E))), dim3(T), 0, 0, src, DST, n); }), bytes);
    CHUNK(0, 512, 16, dstb, "outofplace")
    CHUNK(0, 512, 16, dsta, "inplace")
    CHUNK(3, 512, 16, dstb, "outofplace")
    CHUNK(3, 512, 16, dsta, "inplace")
    CHUNK(0, 256, 8, dstb, "outofplace")
    CHUNK(0, 256, 8, dsta, "inplace")
    CHUNK(0, 256, 16, dsta, "inplace")
    CHUNK(0, 256, 4, dsta, "inplace")
    snprintf(nm, sizeof nm, "read nt0");
    if (want(nm)) report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<0>), dim3(2048), dim3(256), 0, 0, src, dstb, n); }), bytes / 2);
    snprintf(nm, sizeof nm, "read nt1");
    if (want(nm)) report(nm, time_ms([&] { hipLaunchKernelGGL((k_read<1>), dim3(2048), dim3(256), 0, 0, src, dstb, n); }), bytes / 2);
    snprintf(nm, sizeof nm, "write nt0");
    if (want(nm)) report(nm, time_ms([&] { hipLaunchKernelGGL((k_write<0>), dim3(2048), dim3(256), 0, 0, src, dstb, n); }), bytes / 2);
    snprintf(nm, sizeof nm, "write nt2");
    if (want(nm)) report(nm, time_ms([&] { hipLaunchKernelGGL((k_write<2>), dim3(2048), dim3(256), 0, 0, src, dstb, n); }), bytes / 2);
  }

  printf("---- tile pattern (ColFft addressing, no FFT)\n");
  for (int pitch : {513, 514, 516, 520, 521, 528, 529, 545, 577}) {
    run_tile<8, 16, 0>("tilepitch", a, a, pitch, false, 1);
  }
  run_tile<8, 16, 0>("tilebase", a, a, 513, false, 1);
  run_tile<8, 16, 0>("tilebase", a, b, 513, false, 1);
  run_tile<8, 16, 3>("tilebase", a, a, 513, false, 1);
  run_tile<8, 16, 3>("tilebase", a, b, 513, false, 1);
  run_tile<8, 16, 0>("tilebase", a, a, 520, false, 1);
  run_tile<8, 16, 0>("tilebase", a, b, 520, false, 1);
  run_tile<8, 16, 3>("tilebase", a, a, 520, false, 1);
  run_tile<8, 16, 3>("tilebase", a, b, 520, false, 1);
  run_tile<8, 16, 0>("tilebase", a, a, 513, true, 1);
  run_tile<8, 16, 0>("tilebase", a, b, 513, true, 1);
  run_tile<8, 16, 3>("tilebase", a, b, 520, true, 1);
  run_tile<8, 8, 0>("tilebase", a, a, 513, false, 1);
  run_tile<8, 8, 0>("tilebase", a, b, 513, false, 1);
  run_tile<8, 4, 0>("tilebase", a, a, 513, false, 1);
  run_tile<8, 32, 0>("tilebase", a, a, 513, false, 1);
  run_tile<16, 16, 0>("tilebase", a, a, 513, false, 1);
  run_tile<16, 16, 0>("tilebase", a, b, 513, false, 1);
  run_tile<16, 16, 3>("tilebase", a, b, 520, false, 1);
  run_tile<4, 16, 0>("tilebase", a, a, 513, false, 1);

  if (want("mall")) {
    tile_mall_probe<8, 16, 0>(a);
    tile_mall_probe<8, 16, 3>(a);
  }
  printf("---- stamped strided FFT (diagnostic build)\n");
  typedef Spec<1024, 16, 8, 8> SA;
  typedef Spec<1024, 8, 8, 4, 4> SD;
  run_stamped<SA, 8, false>("16x8x8", a, false);
  run_stamped<SA, 8, true>("16x8x8", a, false);
  run_stamped<SD, 8, true>("8x8x4x4", a, false);
  run_stamped<SA, 8, false>("16x8x8", a, true);
  run_stamped<SA, 8, true>("16x8x8", a, true);
  run_pstamped<SA, 8, 2>("16x8x8", a, false);
  run_pstamped<SA, 8, 2>("16x8x8", a, true);
  run_pstamped<SA, 8, 1>("16x8x8", a, false);
  run_pstamped<SA, 8, 0>("16x8x8", a, false);
  run_pstamped<SA, 8, 1>("16x8x8", a, true);
  run_pstamped<SA, 8, 0>("16x8x8", a, true);
  return 0;
}
