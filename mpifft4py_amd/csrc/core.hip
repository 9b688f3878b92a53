// core.hip -- kernel registry, twiddle cache, launch layer and the small
// data-movement kernels (box copy / mask / scale / synthetic fill).
#include <algorithm>
#include <cstdarg>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <tuple>
#include "mfft_internal.h"
#include "fft_nlz.h"

namespace mfft {

// ---------------------------------------------------------------------------
static thread_local std::string g_err;

int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
const char* last_error() { return g_err.c_str(); }

std::vector<KernelEntry>& kernel_registry() {
  static std::vector<KernelEntry> reg;
  return reg;
}

// The registry is complete once the static initialisers of the kernels_*.hip units have run; lookups go
// through hash maps built on first use (a linear scan of ~2000 entries per launch costs ~0.3 us, visible at 32^3).
static uint64_t kernel_key(int family, int n, int prec, int inv, int nt, int pad) {
  // pad codes run to 18 (registry.h: 16 + pad for the ColFft3 kernels): six bits
  return ((uint64_t)(unsigned)n << 16) | ((uint64_t)family << 10) | ((uint64_t)(pad & 63) << 4) | ((uint64_t)nt << 2) |
         ((uint64_t)inv << 1) | (uint64_t)prec;
}

const KernelEntry* find_kernel(int family, int n, int prec, int inv, int nt, int pad) {
  static std::unordered_map<uint64_t, const KernelEntry*> index;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const KernelEntry& e : kernel_registry()) index.emplace(kernel_key(e.family, e.n, e.prec, e.inv, e.nt, e.pad), &e);
  });
  auto it = index.find(kernel_key(family, n, prec, inv, nt, pad));
  return it == index.end() ? nullptr : it->second;
}

const KernelEntry* find_chirpz(int family, int n, int prec, int inv) {
  static std::unordered_map<uint64_t, const KernelEntry*> cache;
  static std::mutex mu;
  const uint64_t key = kernel_key(family, n, prec, inv, 0, 0);
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const KernelEntry* best = nullptr;
  for (const KernelEntry& e : kernel_registry())
    if (e.family == family && e.prec == prec && e.inv == inv && e.n >= 2 * n - 1 && (!best || e.n < best->n)) best = &e;
  cache.emplace(key, best);
  return best;
}

bool length_supported(int64_t n, bool real_transform) {
  if (n <= 0 || n > (1 << 20)) return false;
  if (real_transform)
    return find_kernel(FAM_R2C, (int)n, MFFT_DOUBLE, 0) != nullptr || find_chirpz(FAM_R2CZ, (int)n, MFFT_DOUBLE, 0) != nullptr ||
           (n % 2 == 0 && n >= 4 && find_chirpz(FAM_R2CZH, (int)(n / 2), MFFT_DOUBLE, 0) != nullptr) || big_length_ok(n);
  return n == 1 || find_kernel(FAM_COL, (int)n, MFFT_DOUBLE, 0) != nullptr ||
         find_chirpz(FAM_COLZ, (int)n, MFFT_DOUBLE, 0) != nullptr || big_length_ok(n);
}
bool radix_plan_exists(int contiguous, int64_t n, int prec) {
  if (n < 2 || n >= 65536) return false;
  return find_kernel(contiguous ? FAM_ROW : FAM_COL, (int)n, prec, 0) != nullptr && find_kernel(contiguous ? FAM_ROW : FAM_COL, (int)n, prec, 1) != nullptr;
}
// 1: a radix plan, 2: the one-workgroup chirp-z kernels, 3: the scratch-buffer fallback (bigfft.hip), 0: none
int length_route(int64_t n, bool real_transform, int prec) {
  if (n <= 0 || n > (1 << 20)) return 0;
  if (real_transform) {
    if (find_kernel(FAM_R2C, (int)n, prec, 0)) return 1;
    if (find_chirpz(FAM_R2CZ, (int)n, prec, 0) || (n % 2 == 0 && n >= 4 && find_chirpz(FAM_R2CZH, (int)(n / 2), prec, 0))) return 2;
  } else {
    if (n == 1 || find_kernel(FAM_COL, (int)n, prec, 0)) return 1;
    if (find_chirpz(FAM_COLZ, (int)n, prec, 0)) return 2;
  }
  return big_length_ok(n) ? 3 : 0;
}

// ---------------------------------------------------------------------------
// per-device caches: inter-pass twiddles per kernel entry, real twiddles per
// (n, prec); kernels with > 64 KiB dynamic LDS get the opt-in attribute once.
// ---------------------------------------------------------------------------
struct DevCache {
  std::map<const KernelEntry*, void*> tw;
  std::map<std::pair<int, int>, void*> rtw;
  std::map<const void*, bool> attr_done;
  std::map<std::tuple<int, int, int>, std::pair<void*, void*>> ztab;   // (n, M, prec) -> (chirp, bhat)
  std::map<std::pair<int, int>, void*> rt3;                            // (L, prec) -> twiddles of the pruned nonlinear z stage
};
static std::mutex g_cache_mu;
static std::map<int, DevCache> g_cache;

static int prepare_kernel(const KernelEntry* e, void** tw_out) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_cache_mu);
  DevCache& c = g_cache[dev];
  if (!c.attr_done[e->func]) {
    if (e->lds_bytes > 65536)
      MFFT_HIP(hipFuncSetAttribute(e->func, hipFuncAttributeMaxDynamicSharedMemorySize, e->lds_bytes));
    c.attr_done[e->func] = true;
  }
  auto it = c.tw.find(e);
  if (it == c.tw.end()) {
    const size_t bytes = (size_t)e->tw_count * elem_bytes(e->prec, true);
    std::vector<char> host(bytes);
    e->build_tw(host.data());
    void* d = nullptr;
    MFFT_HIP(hipMalloc(&d, bytes));
    MFFT_HIP(hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice));
    it = c.tw.emplace(e, d).first;
  }
  *tw_out = it->second;
  return 0;
}

static int real_twiddles(int n, int prec, void** out) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_cache_mu);
  DevCache& c = g_cache[dev];
  auto key = std::make_pair(n, prec);
  auto it = c.rtw.find(key);
  if (it == c.rtw.end()) {
    void* d = nullptr;
    if (prec == MFFT_DOUBLE) {
      auto v = build_real_twiddles<double>(n);
      MFFT_HIP(hipMalloc(&d, v.size() * sizeof(v[0])));
      MFFT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    } else {
      auto v = build_real_twiddles<float>(n);
      MFFT_HIP(hipMalloc(&d, v.size() * sizeof(v[0])));
      MFFT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    }
    it = c.rtw.emplace(key, d).first;
  }
  *out = it->second;
  return 0;
}

static int nlz3_twiddles(int L, int prec, void** out) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_cache_mu);
  DevCache& c = g_cache[dev];
  auto key = std::make_pair(L, prec);
  auto it = c.rt3.find(key);
  if (it == c.rt3.end()) {
    void* d = nullptr;
    if (prec == MFFT_DOUBLE) {
      auto v = build_nlz3_twiddles<double>(L);
      MFFT_HIP(hipMalloc(&d, v.size() * sizeof(v[0])));
      MFFT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    } else {
      auto v = build_nlz3_twiddles<float>(L);
      MFFT_HIP(hipMalloc(&d, v.size() * sizeof(v[0])));
      MFFT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
    }
    it = c.rt3.emplace(key, d).first;
  }
  *out = it->second;
  return 0;
}

// chirp-z tables of logical length n on convolution length M (fft_chirpz.h)
template <typename T>
static int upload_chirpz(int n, int M, void** chirp, void** bhat) {
  auto c = build_chirp<T>(n);
  auto b = build_chirp_filter<T>(n, M);
  MFFT_HIP(hipMalloc(chirp, c.size() * sizeof(c[0])));
  MFFT_HIP(hipMemcpy(*chirp, c.data(), c.size() * sizeof(c[0]), hipMemcpyHostToDevice));
  MFFT_HIP(hipMalloc(bhat, b.size() * sizeof(b[0])));
  MFFT_HIP(hipMemcpy(*bhat, b.data(), b.size() * sizeof(b[0]), hipMemcpyHostToDevice));
  return 0;
}
static int chirpz_tables(int n, int M, int prec, void** chirp, void** bhat) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_cache_mu);
  DevCache& c = g_cache[dev];
  auto key = std::make_tuple(n, M, prec);
  auto it = c.ztab.find(key);
  if (it == c.ztab.end()) {
    void *ch = nullptr, *bh = nullptr;
    MFFT_TRY(prec == MFFT_DOUBLE ? upload_chirpz<double>(n, M, &ch, &bh) : upload_chirpz<float>(n, M, &ch, &bh));
    it = c.ztab.emplace(key, std::make_pair(ch, bh)).first;
  }
  *chirp = it->second.first;
  *bhat = it->second.second;
  return 0;
}

static RowMap to_map(const RowSpec& r, int n) { return make_rowmap(r.hi, r.lo, r.split, n); }

// ---------------------------------------------------------------------------
template <typename T, class PT = ColParams<T>>
static int launch_col_t(const KernelEntry* e, const ColArgs& a, void* tw, hipStream_t s, const void* chirp = nullptr,
                        const void* bhat = nullptr) {
  PT P;
  if constexpr (!std::is_same<PT, ColParams<T>>::value) {
    P.chirp = static_cast<const cx<T>*>(chirp);
    P.bhat = static_cast<const cx<T>*>(bhat);
    P.n = a.n;
  }
  P.in = static_cast<const cx<T>*>(a.in);
  P.out = static_cast<cx<T>*>(a.out);
  P.tw = static_cast<const cx<T>*>(tw);
  P.in_outer = a.in_outer;
  P.out_outer = a.out_outer;
  P.in_map = to_map(a.in_rows, a.n);
  P.out_map = to_map(a.out_rows, a.n);
  P.ncols = (int)a.ncols;
  P.ntile_c = (int)((a.ncols + e->tile - 1) / e->tile);
  P.nouter = (int)a.nouter;
  static const int remap_env = getenv("MFFT_REMAP") ? atoi(getenv("MFFT_REMAP")) : -1;      // experiments: 0 none, 1 XCD-aware, 2 skewed
  P.remap = remap_env >= 0 ? remap_env : a.remap;
  P.fold = a.fold ? 1 : 0;
  P.scale = (T)a.scale;
  P.nblocks = 0;
  P.mask = a.mask;
  P.b_row_lo = a.band.row_lo; P.b_row_hi = a.band.row_hi; P.b_coff = a.band.c_off; P.b_cper = a.band.c_per;
  P.b_clim = a.band.c_lim; P.b_goff = a.band.g_off; P.b_gstep = a.band.g_step; P.b_glo = a.band.g_lo; P.b_ghi = a.band.g_hi;
  P.tile_list = a.band.on ? a.band.tile_list : nullptr;
  P.ntiles_listed = a.band.ntiles_listed;
  P.b_gzero = a.band.on ? a.band.g_zero : 0;
  if constexpr (std::is_same<PT, ColParams<T>>::value) {
    P.in_wrap = (int)a.in_wrap;
    P.in_wrap_gap = (int)a.in_wrap_gap;
  } else {
    if (a.in_wrap > 0) return set_error(MFFT_ERR_UNSUPPORTED, "wrapped input columns need a radix kernel (length %d has none)", a.n);
  }
  const int64_t grid = P.tile_list ? (int64_t)P.ntiles_listed : (int64_t)P.ntile_c * a.nouter * (e->grid_mult > 1 ? e->grid_mult : 1);
  if (grid <= 0) return 0;
  if (grid > 0x7FFFFFFF) return set_error(MFFT_ERR_UNSUPPORTED, "grid too large (%lld tiles)", (long long)grid);
  e->launch(&P, (int)grid, s);
  MFFT_HIP(hipGetLastError());
  return 0;
}

#ifndef MFFT_COL3S_DEFAULT
#define MFFT_COL3S_DEFAULT 1
#endif
int launch_col(const ColArgs& a, hipStream_t s) {
  if (a.n == 1) {   // a length-1 transform is a (scaled) copy of its single row
    if (a.in == a.out && a.in_outer == a.out_outer && a.scale == 1.0) return 0;
    BoxArgs b;
    b.src = a.in; b.dst = a.out; b.e0 = a.nouter; b.e1 = 1; b.e2 = a.ncols; b.s0 = a.in_outer; b.d0 = a.out_outer;
    b.elem = (int)elem_bytes(a.prec, true); b.scale = a.scale; b.prec = a.prec;
    return launch_box_copy(b, s);
  }
  if (a.n >= 65536)      // beyond the radix kernels' 16-bit row arithmetic: the scratch-buffer fallback or nothing
    return big_length_ok(a.n) ? big_col(a, s) : set_error(MFFT_ERR_UNSUPPORTED, "transform length %d too large", a.n);
  if (a.ncols >= (1ll << 31)) return set_error(MFFT_ERR_UNSUPPORTED, "too many columns");
  // non-temporal variant only when every row segment of the tile is a whole, private L2 line
  const int64_t per_line = 128 / (int64_t)elem_bytes(a.prec, true);
  auto aligned = [&](const void* p, int64_t outer, const RowSpec& r) {
    return ((uintptr_t)p % 128 == 0) && outer % per_line == 0 && r.lo % per_line == 0 && r.hi % per_line == 0;
  };
  // Out of place the NT variant is always ahead (1024^3 x pass 3.70 -> 3.63 ms at one workgroup per CU, 3.24 -> 3.19 ms at
  // two); in place only the kernels that run two workgroups per CU gain from it (3.42 -> 3.35 ms; one per CU: 3.57 -> 3.84 ms)
  static const int nt_mode = getenv("MFFT_NT") ? atoi(getenv("MFFT_NT")) : 1;   // 0 never, 1 by that rule, 2 always
  const KernelEntry* ent = nullptr;
  const bool nt_ok = a.allow_nt && nt_mode > 0 && !a.pad && aligned(a.in, a.in_outer, a.in_rows) && aligned(a.out, a.out_outer, a.out_rows);
  if (nt_ok) {
    ent = find_kernel(FAM_COL, a.n, a.prec, a.inverse ? 1 : 0, 1);
    if (ent && a.in == a.out && !(ent->nt_inplace || nt_mode == 2)) ent = nullptr;
  }
  const KernelEntry* e = nullptr;
  if (a.mask || a.band.on) {
    const int code = a.band.on ? 6 : 5;
    if (!a.inverse || a.pad || (a.mask && a.band.on)) return set_error(MFFT_ERR_INVALID, "a dealias mask is applied by inverse, un-padded transforms only");
    if (ent) {      // the same alignment rule picks the non-temporal build
      e = find_kernel(FAM_COL, a.n, a.prec, 1, 1, code);
      if (e && a.in == a.out && !(e->nt_inplace || nt_mode == 2)) e = nullptr;
    }
    if (!e) e = find_kernel(FAM_COL, a.n, a.prec, 1, 0, code);
    if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no masked-load kernel for length %d", a.n);
    ent = nullptr;
  }
  if (a.pad) {
    if ((a.pad == 1) != a.inverse) return set_error(MFFT_ERR_INVALID, "pad-on-load is an inverse-transform mode, truncate-on-store a forward one");
    e = find_kernel(FAM_COL, a.n, a.prec, a.inverse ? 1 : 0, 0, a.pad);
    if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no fused 3/2-rule kernel for length %d", a.n);
  }
  // N = 3 L as three sub-transforms per workgroup (fft_col3.h; pad codes 16 + pad), plain and 3/2-rule passes.  Measured at
  // 1536 (profiles/r04_col3_1536.txt): even with the ColFft plan in double precision (1536^3 pair 75.1 - 81.0 against 76.9 -
  // 78.1 ms, 3/2-rule pair of 1024^3 45.5 - 47.7 against 47.9 - 48.1), ahead in single precision on the x passes (rows a whole
  // plane apart: 9.0 -> 7.3 ms inverse, 7.5 -> 6.8 forward) and behind on the y passes (6.9 -> 7.7 - 8.1).  Default: single
  // precision, passes without an outer batch (the x passes); MFFT_COL3=1 always, MFFT_COL3=0 never.
  static const int col3_mode = getenv("MFFT_COL3") ? atoi(getenv("MFFT_COL3")) : -1;
  const bool col3_on = col3_mode > 0 || (col3_mode < 0 && a.prec == MFFT_SINGLE && (a.nouter == 1 || a.thirds > 0));
  if (col3_on && !a.mask && !a.band.on) {
    const KernelEntry* e3 = nullptr;
    if (nt_ok) e3 = find_kernel(FAM_COL, a.n, a.prec, a.inverse ? 1 : 0, 1, 16);              // same alignment rule, NT build
    if (!e3) e3 = find_kernel(FAM_COL, a.n, a.prec, a.inverse ? 1 : 0, 0, 16 + a.pad);
    if (e3) e = e3;
  }
  // The pad-on-load inverse with one third of a tile's transform per workgroup (fft_col3.h ColFft3S, pad code 33): out of
  // place only (its passes are).  3/2-rule ifftn + fftn pair of 1024^3 (profiles/r04_col3s_ab.txt): double precision x pass
  // 5.31 -> 5.16 ms, y pass 8.47 -> 9.17 (not taken); single precision x 2.99 -> 2.54, y 4.98 -> 4.35.  Default on; MFFT_COL3S=0 never.
  static const int col3s_mode = getenv("MFFT_COL3S") ? atoi(getenv("MFFT_COL3S")) : MFFT_COL3S_DEFAULT;
  // double precision: the passes without an outer batch only (y pass 8.4 -> 9.1 ms with it); MFFT_COL3S=2: every pass
  if (col3s_mode > 0 && a.thirds != 0 && a.pad == 1 && a.inverse && a.in != a.out && !a.mask && !a.band.on &&
      (col3s_mode > 1 || a.thirds > 0 || a.prec == MFFT_SINGLE || a.nouter == 1)) {
    if (const KernelEntry* es = find_kernel(FAM_COL, a.n, a.prec, 1, 0, 33)) e = es;
  }
  if (!e && ent) e = ent;
  if (!e) e = find_kernel(FAM_COL, a.n, a.prec, a.inverse ? 1 : 0, 0);
  // Round 5: the y-pass builds (registry.h register_col_ytile, nt code 2: 64-byte tiles, LDS twiddles, two workgroups per
  // CU) where the rows of a tile lie at most 64 KB apart on both sides -- the y passes of every decomposition; the x passes,
  // rows a plane apart, lose with them (1440 / 1536 fp64: y 12.95 -> 9.62 / 13.74 -> 11.55 ms, x 12.85 -> 14.49 / 12.56 -> 14.41).  Not for the
  // pruned 2/3-rule passes (their tile lists are built for the width of the general kernel).  MFFT_YTILE=0 never, 2 always.
  static const int ytile_mode = getenv("MFFT_YTILE") ? atoi(getenv("MFFT_YTILE")) : 1;
  if (ytile_mode > 0 && e && !a.band.on && e->family == FAM_COL && e->pad <= 5 && e->nt != 2) {
    const int64_t es = (int64_t)elem_bytes(a.prec, true);
    const bool near_rows = std::llabs(a.in_rows.lo) * es <= 65536 && std::llabs(a.out_rows.lo) * es <= 65536;
    if (near_rows || ytile_mode > 1)
      if (const KernelEntry* ey = find_kernel(FAM_COL, a.n, a.prec, a.inverse ? 1 : 0, 2, e->pad)) e = ey;
  }
  void* tw = nullptr;
  if (!e) {   // no radix plan for this length: chirp-z on the next compiled length >= 2n-1
    e = find_chirpz(FAM_COLZ, a.n, a.prec, a.inverse ? 1 : 0);
    if (!e && big_length_ok(a.n)) return big_col(a, s);
    if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no kernel for a complex transform of length %d", a.n);
    void *chirp = nullptr, *bhat = nullptr;
    MFFT_TRY(prepare_kernel(e, &tw));
    MFFT_TRY(chirpz_tables(a.n, e->n, a.prec, &chirp, &bhat));
    return a.prec == MFFT_DOUBLE ? launch_col_t<double, ColParamsZ<double>>(e, a, tw, s, chirp, bhat)
                                 : launch_col_t<float, ColParamsZ<float>>(e, a, tw, s, chirp, bhat);
  }
  MFFT_TRY(prepare_kernel(e, &tw));
  return a.prec == MFFT_DOUBLE ? launch_col_t<double>(e, a, tw, s) : launch_col_t<float>(e, a, tw, s);
}

template <typename T, class PT = RowParams<T>>
static int launch_row_t(const KernelEntry* e, const RowArgs& a, void* tw, hipStream_t s, const void* chirp = nullptr,
                        const void* bhat = nullptr) {
  PT P;
  if constexpr (!std::is_same<PT, RowParams<T>>::value) {
    P.chirp = static_cast<const cx<T>*>(chirp);
    P.bhat = static_cast<const cx<T>*>(bhat);
    P.n = a.n;
  }
  P.in = static_cast<const cx<T>*>(a.in);
  P.out = static_cast<cx<T>*>(a.out);
  P.tw = static_cast<const cx<T>*>(tw);
  P.in_stride = a.in_stride;
  P.out_stride = a.out_stride;
  P.nrows = a.nrows;
  P.scale = (T)a.scale;
  if constexpr (std::is_same<PT, RowParams<T>>::value) {
    P.zs = a.zs.nchunk ? make_zsplit(a.zs.q, a.zs.nchunk, a.zs.last_len, a.zs.rows_total, a.zs.pitch, a.zs.last_pitch) : ZSplit{1, 1, 0, 0, 0, 1, 0};
    P.row0 = a.zs.row0;
  }
  const int64_t grid = (a.nrows + e->tile - 1) / e->tile;
  if (grid <= 0) return 0;
  if (grid > 0x7FFFFFFF) return set_error(MFFT_ERR_UNSUPPORTED, "grid too large");
  e->launch(&P, (int)grid, s);
  MFFT_HIP(hipGetLastError());
  return 0;
}

int col_tile_width(int64_t n, int prec, bool inverse, int pad_code) {
  const KernelEntry* e = find_kernel(FAM_COL, (int)n, prec, inverse ? 1 : 0, 0, pad_code);
  return e ? e->tile : 0;
}
bool c2r_limit_supported(int64_t n, int prec) { return n >= 4 && n < 65536 && find_kernel(FAM_C2R, (int)n, prec, 1, 0, 3) != nullptr; }
bool band_fusable(int64_t n, int prec) { return n >= 2 && n < 65536 && find_kernel(FAM_COL, (int)n, prec, 1, 0, 6) != nullptr; }
bool mask_fusable(int64_t n, int prec) { return n >= 2 && n < 65536 && find_kernel(FAM_COL, (int)n, prec, 1, 0, 5) != nullptr; }

bool zsplit_limit_supported(int64_t n, int prec) {
  return n >= 2 && n < 65536 && find_kernel(FAM_R2C, (int)n, prec, 0, 0, 7) && find_kernel(FAM_C2R, (int)n, prec, 1, 0, 7);
}

bool zsplit_supported(int64_t n, int prec, bool real_transform) {
  if (n < 2 || n > 65536) return false;
  if (real_transform) return find_kernel(FAM_R2C, (int)n, prec, 0, 0, 4) && find_kernel(FAM_C2R, (int)n, prec, 1, 0, 4);
  return find_kernel(FAM_ROW, (int)n, prec, 0, 0, 4) && find_kernel(FAM_ROW, (int)n, prec, 1, 0, 4);
}

int launch_row(const RowArgs& a, hipStream_t s) {
  if (a.zs.nchunk) {
    const KernelEntry* ec = find_kernel(FAM_ROW, a.n, a.prec, a.inverse ? 1 : 0, 0, 4);
    if (!ec) return set_error(MFFT_ERR_UNSUPPORTED, "no z-chunked row kernel of length %d", a.n);
    void* twc = nullptr;
    MFFT_TRY(prepare_kernel(ec, &twc));
    return a.prec == MFFT_DOUBLE ? launch_row_t<double>(ec, a, twc, s) : launch_row_t<float>(ec, a, twc, s);
  }
  const KernelEntry* e = find_kernel(FAM_ROW, a.n, a.prec, a.inverse ? 1 : 0);
  void* tw = nullptr;
  if (!e) {
    e = find_chirpz(FAM_ROWZ, a.n, a.prec, a.inverse ? 1 : 0);
    if (!e && big_length_ok(a.n)) return big_row(a, s);
    if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no kernel for a complex transform of length %d", a.n);
    void *chirp = nullptr, *bhat = nullptr;
    MFFT_TRY(prepare_kernel(e, &tw));
    MFFT_TRY(chirpz_tables(a.n, e->n, a.prec, &chirp, &bhat));
    return a.prec == MFFT_DOUBLE ? launch_row_t<double, RowParamsZ<double>>(e, a, tw, s, chirp, bhat)
                                 : launch_row_t<float, RowParamsZ<float>>(e, a, tw, s, chirp, bhat);
  }
  MFFT_TRY(prepare_kernel(e, &tw));
  return a.prec == MFFT_DOUBLE ? launch_row_t<double>(e, a, tw, s) : launch_row_t<float>(e, a, tw, s);
}

template <typename T, class PT = RealParams<T>>
static int launch_real_t(const KernelEntry* e, const RealArgs& a, void* tw, void* rtw, hipStream_t s,
                         const void* chirp = nullptr, const void* bhat = nullptr) {
  PT P;
  if constexpr (!std::is_same<PT, RealParams<T>>::value) {
    P.chirp = static_cast<const cx<T>*>(chirp);
    P.bhat = static_cast<const cx<T>*>(bhat);
    P.n = a.n;
  }
  P.in = a.in;
  P.out = a.out;
  P.tw = static_cast<const cx<T>*>(tw);
  P.rtw = static_cast<const cx<T>*>(rtw);
  P.in_stride = a.in_stride;
  P.out_stride = a.out_stride;
  P.nrows = a.nrows;
  P.valid = a.valid > 0 ? a.valid : a.n / 2 + 1;
  P.scale = (T)a.scale;
  if constexpr (std::is_same<PT, RealParams<T>>::value) {
    P.zs = a.zs.nchunk ? make_zsplit(a.zs.q, a.zs.nchunk, a.zs.last_len, a.zs.rows_total, a.zs.pitch, a.zs.last_pitch) : ZSplit{1, 1, 0, 0, 0, 1, 0};
    P.row0 = a.zs.row0;
  }
  const int64_t grid = (a.nrows + e->tile - 1) / e->tile;
  if (grid <= 0) return 0;
  if (grid > 0x7FFFFFFF) return set_error(MFFT_ERR_UNSUPPORTED, "grid too large");
  e->launch(&P, (int)grid, s);
  MFFT_HIP(hipGetLastError());
  return 0;
}

// c2r kernels with the mirrors through LDS (registry.h c2r_mlds_candidate: built where they were measured ahead): taken where
// they exist; MFFT_C2R_MLDS=0: never
static int c2r_mlds_mode() {
  static const int m = getenv("MFFT_C2R_MLDS") ? atoi(getenv("MFFT_C2R_MLDS")) : 1;
  return m;
}
// MFFT_C2R_MLDS: 0 never, 1 (default) where measured ahead, 2 wherever a build exists, 3 the column-limited kernels of the 3/2-rule
// only.  profiles/r06_c2r_mlds.txt: the 12-values plans of 288 ... 3456 complex points that no shuffle serves gain 1 - 15 % on plain
// rows and 15 - 47 % on column-limited ones (3/2-rule pair of 576^3 fp64: bwd_z 3.08 -> 1.65 ms, of 768^3: 6.60 -> 3.91); rows of more
// than a wave's threads: real 6144 / 8192 +4 ... +15 %, real 4096 in double precision only (+10 %; single -5 %), real 3072 only
// column-limited in single precision (+5 %; plain -2 ... -9 %).  The 30- / 42-values plans have no such build (registers: -2.2 x).
static bool c2r_mlds_take(int n, int prec, bool limited) {
  const int m = c2r_mlds_mode();
  if (m <= 0) return false;
  if (m == 2) return true;
  if (m == 3) return limited;
  const int c = n / 2;
  if (c == 1000 || c == 2000) return prec == MFFT_SINGLE;      // the 20-values plans: 400 / 500 / 800 gain in both precisions
  // (800^3 bwd_z 1.75 -> 1.50 ms, under the 2/3-rule 1.65 -> 1.32; 1000^3 3.28 -> 3.07 / 3.45 -> 2.99; 1600^3 13.7 -> 13.3 / 13.2 -> 12.2),
  // real 2000 / 4000 in single precision only (double: -6 ... -11 %)
  if (c == 1536) return limited && prec == MFFT_SINGLE;
  if (c == 2048) return !limited && prec == MFFT_DOUBLE;
  return true;
}
static int launch_real(int fam, const RealArgs& a, hipStream_t s) {
  if (a.zs.nchunk) {
    // column-limited as well (3/2-rule: only the first `valid` of the n/2+1 bins exist, and those are what is chunked)
    const bool lim = a.valid > 0 && a.valid < a.n / 2 + 1;
    const KernelEntry* ec = find_kernel(fam, a.n, a.prec, fam == FAM_C2R ? 1 : 0, 0, lim ? 7 : 4);
    if (fam == FAM_C2R && c2r_mlds_take(a.n, a.prec, lim))
      if (const KernelEntry* em = find_kernel(fam, a.n, a.prec, 1, 1, lim ? 7 : 4)) ec = em;
    const int64_t real_stride_c = fam == FAM_R2C ? a.in_stride : a.out_stride;
    if (!ec || real_stride_c % 2 != 0) return set_error(MFFT_ERR_UNSUPPORTED, "no z-chunked real kernel of length %d", a.n);
    void *twc = nullptr, *rtwc = nullptr;
    MFFT_TRY(prepare_kernel(ec, &twc));
    MFFT_TRY(real_twiddles(a.n, a.prec, &rtwc));
    return a.prec == MFFT_DOUBLE ? launch_real_t<double>(ec, a, twc, rtwc, s) : launch_real_t<float>(ec, a, twc, rtwc, s);
  }
  const bool limited = a.valid > 0 && a.valid < a.n / 2 + 1;
  const KernelEntry* e = find_kernel(fam, a.n, a.prec, fam == FAM_C2R ? 1 : 0, 0, limited ? 3 : 0);
  if (fam == FAM_C2R && c2r_mlds_take(a.n, a.prec, limited))
    if (const KernelEntry* em = find_kernel(fam, a.n, a.prec, 1, 1, limited ? 3 : 0)) e = em;
  if (!e && limited) return set_error(MFFT_ERR_UNSUPPORTED, "no column-limited real kernel of length %d", a.n);
  // the radix kernels read a real row as (n/2) complex values: rows must stay 2-element aligned
  const int64_t real_stride = fam == FAM_R2C ? a.in_stride : a.out_stride;
  void *tw = nullptr, *rtw = nullptr;
  if (!e || (real_stride % 2 != 0 && !limited)) {   // no radix plan (or an odd pitch): chirp-z on the real row
    const bool r2c = fam == FAM_R2C;
    // even length and pitch: n/2 complex values + split pass (half the convolution length); else full length
    const bool half = a.n % 2 == 0 && a.n >= 4 && real_stride % 2 == 0;
    const int nz = half ? a.n / 2 : a.n;
    e = find_chirpz(half ? (r2c ? FAM_R2CZH : FAM_C2RZH) : (r2c ? FAM_R2CZ : FAM_C2RZ), nz, a.prec, r2c ? 0 : 1);
    if (!e && big_length_ok(a.n)) return big_real(!r2c, a, s);
    if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no kernel for a real transform of length %d", a.n);
    void *chirp = nullptr, *bhat = nullptr;
    MFFT_TRY(prepare_kernel(e, &tw));
    MFFT_TRY(chirpz_tables(nz, e->n, a.prec, &chirp, &bhat));
    if (half) MFFT_TRY(real_twiddles(a.n, a.prec, &rtw));
    RealArgs az = a;
    az.n = a.n;
    return a.prec == MFFT_DOUBLE ? launch_real_t<double, RealParamsZ<double>>(e, az, tw, rtw, s, chirp, bhat)
                                 : launch_real_t<float, RealParamsZ<float>>(e, az, tw, rtw, s, chirp, bhat);
  }
  if (real_stride % 2 != 0)
    return set_error(MFFT_ERR_UNSUPPORTED, "real row stride %lld must be even", (long long)real_stride);
  MFFT_TRY(prepare_kernel(e, &tw));
  MFFT_TRY(real_twiddles(a.n, a.prec, &rtw));
  return a.prec == MFFT_DOUBLE ? launch_real_t<double>(e, a, tw, rtw, s) : launch_real_t<float>(e, a, tw, rtw, s);
}

int launch_r2c(const RealArgs& a, hipStream_t s) { return launch_real(FAM_R2C, a, s); }
int launch_c2r(const RealArgs& a, hipStream_t s) { return launch_real(FAM_C2R, a, s); }

// fused nonlinear z stage (fft_nlz.h): out_f = rfft((irfft(a) x irfft(b))_f) along the contiguous axis, row by row
bool nlz_supported(int64_t n, int prec) {
  return n >= 2 && n < 65536 && (find_kernel(FAM_NLZ, (int)n, prec, 0) != nullptr || find_kernel(FAM_NLZ, (int)n, prec, 0, 0, 3) != nullptr);
}
template <typename T>
static int launch_nlz_t(const KernelEntry* e, const NlzArgs& a, void* tw, void* rt3, hipStream_t s) {
  NlzParams<T> P;
  P.rt3 = static_cast<const cx<T>*>(rt3);
  for (int f = 0; f < 3; ++f) {
    P.a[f] = static_cast<const cx<T>*>(a.a[f]);
    P.b[f] = static_cast<const cx<T>*>(a.b[f]);
    P.out[f] = static_cast<cx<T>*>(a.out[f]);
  }
  P.tw = static_cast<const cx<T>*>(tw);
  P.in_stride = a.in_stride;
  P.out_stride = a.out_stride;
  P.nrows = a.nrows;
  P.valid = a.valid > 0 && a.valid < a.n / 2 + 1 ? a.valid : a.n / 2 + 1;
  P.valid_in = a.valid_in > 0 && a.valid_in < P.valid ? a.valid_in : P.valid;
  P.scale = (T)a.scale;
  const int64_t grid = (a.nrows + 2 * e->tile - 1) / (2 * e->tile);      // a thread group works through a PAIR of rows
  if (grid <= 0) return 0;
  if (grid > 0x7FFFFFFF) return set_error(MFFT_ERR_UNSUPPORTED, "grid too large");
  e->launch(&P, (int)grid, s);
  MFFT_HIP(hipGetLastError());
  return 0;
}
int launch_nlz(const NlzArgs& a, hipStream_t s) {
  const KernelEntry* e = a.n < 65536 ? find_kernel(FAM_NLZ, a.n, a.prec, 0) : nullptr;
  // 3/2-rule rows (n = 3 L with the L + 1 bins of the un-padded mesh) also have the pruned kernel (fft_nlz.h Nlz3Fft: three
  // sub-transforms of length L in three thread groups, a third of the registers).  Measured EVEN with NlzFft at 768 (1.36 ms per
  // 73,728 rows both) and behind at 1536 (1.80 - 1.91 against 1.59 ms per 36,864 rows): profiles/r06_nlz_variants.txt -- its
  // staging and combination cost what the skipped radix-3 pass saves.  MFFT_NLZ3=1 takes it; where NlzFft has no plan it runs anyway.
  static const int nlz3_on = getenv("MFFT_NLZ3") ? atoi(getenv("MFFT_NLZ3")) : 0;
  const bool rows3 = a.n % 3 == 0 && a.valid == a.n / 3 + 1 && a.n < 65536;
  if ((nlz3_on || !e) && rows3)
    if (const KernelEntry* e3 = find_kernel(FAM_NLZ, a.n, a.prec, 0, 0, 3)) e = e3;
  if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no fused nonlinear z-stage kernel of length %d", a.n);
  for (int f = 0; f < 3; ++f)
    if (!a.a[f] || !a.b[f] || !a.out[f]) return set_error(MFFT_ERR_INVALID, "null argument");
  void *tw = nullptr, *rt3 = nullptr;
  MFFT_TRY(prepare_kernel(e, &tw));
  if (e->pad == 3) MFFT_TRY(nlz3_twiddles(a.n / 3, a.prec, &rt3));
  return a.prec == MFFT_DOUBLE ? launch_nlz_t<double>(e, a, tw, rt3, s) : launch_nlz_t<float>(e, a, tw, rt3, s);
}

// ---------------------------------------------------------------------------
// data-movement kernels
// ---------------------------------------------------------------------------
// Work unit = (row (i, j), chunk of CHUNK contiguous elements of that row); threads of
// blockDim.x stride over the chunk, blockDim.y units per block, grid-stride over units.
// Long rows (pad / truncate of whole planes) and many short rows (z chunks) both fill the chip.
constexpr int64_t BOX_CHUNK = 2048;

template <typename R, int MODE>
__global__ __launch_bounds__(256) void box_copy_kernel(const R* __restrict__ src, R* __restrict__ dst,
                                                       int64_t e1, int64_t e2, int64_t s0, int64_t s1,
                                                       int64_t d0, int64_t d1, R scale, int64_t nunits, int64_t nchunks) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; u < nunits; u += (int64_t)gridDim.x * blockDim.y) {
    const int64_t row = u / nchunks, c = u - row * nchunks;
    const int64_t i = row / e1, j = row - i * e1;
    const R* sp = src + i * s0 + j * s1;
    R* dp = dst + i * d0 + j * d1;
    const int64_t k1 = (c + 1) * BOX_CHUNK < e2 ? (c + 1) * BOX_CHUNK : e2;
    for (int64_t k = c * BOX_CHUNK + threadIdx.x; k < k1; k += blockDim.x) {
      R v = sp[k] * scale;
      if (MODE == 1) v += dp[k];
      dp[k] = v;
    }
  }
}

struct alignas(16) vec16 { double a, b; };
template <int MODE>
__global__ __launch_bounds__(256) void box_copy16_kernel(const vec16* __restrict__ src, vec16* __restrict__ dst,
                                                         int64_t e1, int64_t e2, int64_t s0, int64_t s1,
                                                         int64_t d0, int64_t d1, int64_t nunits, int64_t nchunks) {
  for (int64_t u = (int64_t)blockIdx.x * blockDim.y + threadIdx.y; u < nunits; u += (int64_t)gridDim.x * blockDim.y) {
    const int64_t row = u / nchunks, c = u - row * nchunks;
    const int64_t i = row / e1, j = row - i * e1;
    const vec16* sp = src + i * s0 + j * s1;
    vec16* dp = dst + i * d0 + j * d1;
    const int64_t k1 = (c + 1) * BOX_CHUNK < e2 ? (c + 1) * BOX_CHUNK : e2;
    for (int64_t k = c * BOX_CHUNK + threadIdx.x; k < k1; k += blockDim.x) dp[k] = sp[k];
  }
}

int launch_box_copy(const BoxArgs& a, hipStream_t s) {
  const int64_t nrows = a.e0 * a.e1;
  if (nrows <= 0 || a.e2 <= 0) return 0;
  const int unit = a.elem;    // bytes per element
  const bool plain = (a.mode == 0 && a.scale == 1.0);
  auto shape = [&](int64_t e2, int* bx, int* by, int64_t* nchunks, int64_t* nunits, unsigned* grid) {
    *nchunks = (e2 + BOX_CHUNK - 1) / BOX_CHUNK;
    const int64_t run = e2 < BOX_CHUNK ? e2 : BOX_CHUNK;
    int x = 64;
    while (x < 256 && x < run) x *= 2;
    *bx = x;
    *by = 256 / x;
    *nunits = nrows * *nchunks;
    int64_t g = (*nunits + *by - 1) / *by;
    if (g > 65536 * 2) g = 65536 * 2;
    *grid = (unsigned)g;
  };
  int bx, by;
  int64_t nchunks, nunits;
  unsigned grid;
  if (plain && unit == 16) {
    shape(a.e2, &bx, &by, &nchunks, &nunits, &grid);
    hipLaunchKernelGGL(box_copy16_kernel<0>, dim3(grid), dim3(bx, by), 0, s, static_cast<const vec16*>(a.src),
                       static_cast<vec16*>(a.dst), a.e1, a.e2, a.s0, a.s1, a.d0, a.d1, nunits, nchunks);
    MFFT_HIP(hipGetLastError());
    return 0;
  }
  // scalar path in units of the real type
  const int rbytes = a.prec == MFFT_DOUBLE ? 8 : 4;
  const int per = unit / rbytes;            // reals per element
  const int64_t e2 = a.e2 * per;
  shape(e2, &bx, &by, &nchunks, &nunits, &grid);
#define MFFT_BOX(R, MODE)                                                                                   \
  hipLaunchKernelGGL((box_copy_kernel<R, MODE>), dim3(grid), dim3(bx, by), 0, s,                            \
                     static_cast<const R*>(a.src), static_cast<R*>(a.dst), a.e1, e2, a.s0 * per, a.s1 * per, \
                     a.d0 * per, a.d1 * per, (R)a.scale, nunits, nchunks)
  if (a.prec == MFFT_DOUBLE) {
    if (a.mode == 1) MFFT_BOX(double, 1); else MFFT_BOX(double, 0);
  } else {
    if (a.mode == 1) MFFT_BOX(float, 1); else MFFT_BOX(float, 0);
  }
#undef MFFT_BOX
  MFFT_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void mask_kernel(cx<T>* fu, const uint8_t* mask, size_t count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const T m = (T)mask[i];
    cx<T> v = fu[i];
    v.x *= m;
    v.y *= m;
    fu[i] = v;
  }
}

int launch_mask(void* fu, const uint8_t* mask, size_t count, int prec, hipStream_t s) {
  if (count == 0) return 0;
  size_t grid = (count + 255) / 256;
  if (grid > 16384) grid = 16384;
  if (prec == MFFT_DOUBLE)
    hipLaunchKernelGGL(mask_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s, static_cast<cx<double>*>(fu), mask, count);
  else
    hipLaunchKernelGGL(mask_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, static_cast<cx<float>*>(fu), mask, count);
  MFFT_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void line_nyquist_kernel(cx<T>* rows, int64_t nrows, int64_t pitch, int64_t coln) {
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    cx<T>* p = rows + r * pitch;
    const cx<T> c0 = p[0], cn = p[coln];
    p[0] = mk<T>(c0.x - cn.y, (T)0);
    p[coln] = mk<T>(cn.x, (T)0);
  }
}

int launch_line_nyquist(void* rows, int64_t nrows, int64_t pitch, int64_t coln, int prec, hipStream_t s) {
  if (nrows <= 0) return 0;
  const unsigned grid = (unsigned)std::min<int64_t>((nrows + 255) / 256, 4096);
  if (prec == MFFT_DOUBLE)
    hipLaunchKernelGGL(line_nyquist_kernel<double>, dim3(grid), dim3(256), 0, s, static_cast<cx<double>*>(rows), nrows, pitch, coln);
  else
    hipLaunchKernelGGL(line_nyquist_kernel<float>, dim3(grid), dim3(256), 0, s, static_cast<cx<float>*>(rows), nrows, pitch, coln);
  MFFT_HIP(hipGetLastError());
  return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* d, size_t count, T sc) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
    d[i] *= sc;
}

int launch_scale(void* data, size_t count, double scale, int prec, hipStream_t s) {
  if (count == 0) return 0;
  size_t grid = (count + 255) / 256;
  if (grid > 16384) grid = 16384;
  if (prec == MFFT_DOUBLE)
    hipLaunchKernelGGL(scale_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s, static_cast<double*>(data), count, scale);
  else
    hipLaunchKernelGGL(scale_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, static_cast<float*>(data), count, (float)scale);
  MFFT_HIP(hipGetLastError());
  return 0;
}

// counter-based uniform [0,1) generator (splitmix64 of the element index)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(T* d, size_t count, uint64_t seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
    const uint64_t r = splitmix64(seed * 0xD1B54A32D192ED03ull + i);
    d[i] = (T)((double)(r >> 11) * (1.0 / 9007199254740992.0));
  }
}

int launch_fill_uniform(void* data, size_t count, int prec, uint64_t seed, hipStream_t s) {
  if (count == 0) return 0;
  size_t grid = (count + 255) / 256;
  if (grid > 16384) grid = 16384;
  if (prec == MFFT_DOUBLE)
    hipLaunchKernelGGL(fill_kernel<double>, dim3((unsigned)grid), dim3(256), 0, s, static_cast<double*>(data), count, seed);
  else
    hipLaunchKernelGGL(fill_kernel<float>, dim3((unsigned)grid), dim3(256), 0, s, static_cast<float*>(data), count, seed);
  MFFT_HIP(hipGetLastError());
  return 0;
}

}  // namespace mfft
