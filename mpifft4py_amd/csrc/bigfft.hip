// bigfft.hip -- transforms of ANY length (round 5): the fallback behind the radix plans (plans.h) and the one-workgroup
// chirp-z kernels (fft_chirpz.h: n <= 4096).
//
// numpy / FFTW take every n (reference: serialFFT/numpy_fft.py:25-46, pyfftw_fft.py:26-203); until round 4 a complex length
// above 4096 without a radix plan, and every length above 8192, was MFFT_ERR_UNSUPPORTED.  Here such a transform runs as
// Bluestein's convolution over a power-of-two length M >= 2n - 1 that does NOT have to fit a workgroup: the length-M
// transforms are done in "four steps" over a scratch buffer in HBM with the library's own radix kernels --
//
//     y[m]   = x[m] c[m]  (m < n), 0 (n <= m < M)          c[m] = exp(-i pi m^2 / n)            gather kernel
//     Y      = FFT_M(y):  M = M1 M2, index m = m1 M2 + m2
//                strided transforms of length M1 (rows M2 apart)                                 launch_col
//                times W_M^(m2 k1)                                                               table kernel
//                contiguous transforms of length M2                                              launch_row
//              -> Y in the order [k1][k2] (bin k1 + M1 k2): never un-permuted, because
//     Z      = Y . Bhat   with Bhat = FFT_M(b) / M computed ONCE per (n, precision) by the same three launches
//     z      = the three launches mirrored (inverse rows, conj table, inverse columns): natural order again
//     X[k]   = z[k] c[k] scale                                                                   scatter kernel
//
// -- 8 launches and ~10 passes over the scratch per batch of vectors, against 1 launch and 2 passes of a radix plan: a
// completeness path (0.05 - 0.1 of the roofline), not a fast one.
// COMPOSITE lengths n = n1 n2 whose factors both have radix plans (16384 = 128 x 128, 10000 = 80 x 125, 65536, 2^20 ...) skip
// Bluestein: gather, the three launches of the four-step transform at length n itself (table W_n^(m2 k1)), and a scatter
// that reads bin k1 + n1 k2 from [k1][k2] -- 5 launches, ~6 passes.  Real transforms go through the complex one (r2c: imaginary
// parts zero, the first n/2 + 1 bins stored; c2r: Hermitian extension on load, real parts stored; the imaginary parts of
// bins 0 and n/2 are ignored as numpy's irfft ignores them).  Inverse transforms use the swap identity of fft_core.h.
// Not offered here: the fused 3/2-rule / 2/3-rule passes and the z-chunked real kernels (their callers ask
// can_fuse_pad / mask_fusable / zsplit_supported first and take the copy-based routes).
#include <hip/hip_runtime.h>

#include <cmath>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

#include "fft_kernels.h"
#include "mfft_internal.h"
#include "twiddle.h"

namespace mfft {

namespace {

// where element k of vector v lives: (v / ncols) * outer + (v % ncols) * cstride + (k / split) * hi + (k % split) * lo
struct VecMap {
  i64 outer, cstride, ncols;
  i64 hi, lo, split;
};
MFFT_HD i64 vec_off(const VecMap& m, i64 v, i64 k) {
  const i64 o = v / m.ncols, c = v - o * m.ncols;
  const i64 q = k / m.split, r = k - q * m.split;
  return o * m.outer + c * m.cstride + q * m.hi + r * m.lo;
}

enum { BZ_C2C = 0, BZ_R2C = 1, BZ_C2R = 2 };

// thread t -> (vector, position): position fastest for contiguous vectors, vector fastest for strided ones (vfast)
MFFT_HD void bz_index(i64 t, i64 nvec, i64 len, bool vfast, i64* v, i64* k) {
  if (vfast) { *k = t / nvec; *v = t - *k * nvec; }
  else { *v = t / len; *k = t - *v * len; }
}

// scratch[v][m] = in(v, m) * chirp[m] for m < n, 0 for n <= m < M
template <typename T, int MODE, bool INV>
__global__ void bz_gather(const void* in, cx<T>* scratch, const cx<T>* chirp, VecMap map, i64 nvec, int n, int M, int valid,
                          bool vfast) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nvec * (i64)M) return;
  i64 v, m;
  bz_index(t, nvec, M, vfast, &v, &m);
  cx<T> x = mk<T>((T)0, (T)0);
  if (m < n) {
    if constexpr (MODE == BZ_R2C) {
      x.x = static_cast<const T*>(in)[vec_off(map, v, m)];
    } else if constexpr (MODE == BZ_C2R) {
      // Hermitian extension of the n/2 + 1 stored bins (of which only `valid` exist in memory)
      const i64 kk = m <= n / 2 ? m : n - m;
      if (kk < valid) {
        x = static_cast<const cx<T>*>(in)[vec_off(map, v, kk)];
        if (m > n / 2) x.y = -x.y;
        if (kk == 0 || (n % 2 == 0 && kk == n / 2)) x.y = (T)0;
      }
    } else {
      x = static_cast<const cx<T>*>(in)[vec_off(map, v, m)];
    }
    if (INV) x = swapri(x);
    if (chirp) x = x * chirp[m];
  }
  scratch[v * (i64)M + m] = x;
}

// out(v, k) = scratch[v][k] * chirp[k] * scale
// (perm1 > 0: the composite route -- bin k = k1 + perm1 * k2 sits at [k1][k2] = k1 * (M / perm1) + k2; no chirp there)
template <typename T, int MODE, bool INV>
__global__ void bz_scatter(const cx<T>* scratch, void* out, const cx<T>* chirp, VecMap map, i64 nvec, int n, int M, int nout,
                           T scale, bool vfast, int perm1) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nvec * (i64)nout) return;
  i64 v, k;
  bz_index(t, nvec, nout, vfast, &v, &k);
  i64 src = k;
  if (perm1 > 0) { const i64 k2 = k / perm1, k1 = k - k2 * perm1; src = k1 * (M / perm1) + k2; }
  cx<T> x = scratch[v * (i64)M + src];
  if (chirp) x = x * chirp[k];
  if (INV) x = swapri(x);
  x = mfft::scale(x, scale);
  if constexpr (MODE == BZ_C2R) static_cast<T*>(out)[vec_off(map, v, k)] = x.x;
  else static_cast<cx<T>*>(out)[vec_off(map, v, k)] = x;
}

// z[i] *= table[i % M] (or its conjugate)
template <typename T, bool CONJ>
__global__ void bz_mul(cx<T>* z, const cx<T>* table, i64 count, int M) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  cx<T> w = table[t % M];
  if (CONJ) w.y = -w.y;
  z[t] = z[t] * w;
}

struct BigTables {
  bool composite = false;    // M = n = M1 * M2 with radix plans for both factors: no chirp, no filter
  int M = 0, M1 = 0, M2 = 0;
  void* chirp = nullptr;     // c[m], m < n
  void* step = nullptr;      // W_M^(m2 k1) at [k1 * M2 + m2]
  void* bhat = nullptr;      // FFT_M(b) / M in the order the three forward launches leave it
};
std::mutex g_big_mu;
std::map<std::tuple<int, int, int>, BigTables> g_big;          // (device, n, prec)
std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> g_scratch;   // (device, stream) -> buffer

inline unsigned nblocks(i64 count) { return (unsigned)((count + 255) / 256); }

int big_forward(void* z, i64 nvec, const BigTables& t, int prec, bool inverse, hipStream_t s) {
  ColArgs c;
  c.in = z; c.out = z; c.n = t.M1; c.prec = prec; c.inverse = inverse; c.nouter = nvec; c.ncols = t.M2;
  c.in_outer = c.out_outer = t.M; c.in_rows.lo = c.out_rows.lo = t.M2; c.scale = 1.0; c.allow_nt = false;
  RowArgs r;
  r.in = z; r.out = z; r.n = t.M2; r.prec = prec; r.inverse = inverse; r.in_stride = r.out_stride = t.M2;
  r.nrows = nvec * t.M1; r.scale = 1.0;
  const i64 count = nvec * (i64)t.M;
  auto mul = [&](bool conj) -> int {
    if (prec == MFFT_DOUBLE) {
      if (conj) hipLaunchKernelGGL((bz_mul<double, true>), dim3(nblocks(count)), dim3(256), 0, s, static_cast<cx<double>*>(z), static_cast<const cx<double>*>(t.step), count, t.M);
      else hipLaunchKernelGGL((bz_mul<double, false>), dim3(nblocks(count)), dim3(256), 0, s, static_cast<cx<double>*>(z), static_cast<const cx<double>*>(t.step), count, t.M);
    } else {
      if (conj) hipLaunchKernelGGL((bz_mul<float, true>), dim3(nblocks(count)), dim3(256), 0, s, static_cast<cx<float>*>(z), static_cast<const cx<float>*>(t.step), count, t.M);
      else hipLaunchKernelGGL((bz_mul<float, false>), dim3(nblocks(count)), dim3(256), 0, s, static_cast<cx<float>*>(z), static_cast<const cx<float>*>(t.step), count, t.M);
    }
    MFFT_HIP(hipGetLastError());
    return 0;
  };
  if (!inverse) {
    MFFT_TRY(launch_col(c, s));
    MFFT_TRY(mul(false));
    MFFT_TRY(launch_row(r, s));
  } else {
    // the inverse kernels compute swap(fft(swap(.))) = the unnormalised inverse DFT, whose twiddle is the conjugate
    MFFT_TRY(launch_row(r, s));
    MFFT_TRY(mul(true));
    MFFT_TRY(launch_col(c, s));
  }
  return 0;
}

// n = n1 * n2 with a strided radix plan of length n1 and a contiguous one of length n2 (the most balanced such pair), or 0
int composite_split(int n, int prec) {
  int best = 0;
  for (int n2 = 2; (long long)n2 * n2 <= (long long)n * 8192 && n2 <= 8192; ++n2) {
    if (n % n2) continue;
    const int n1 = n / n2;
    if (n1 < 2 || n1 > 8192 || !radix_plan_exists(0, n1, prec) || !radix_plan_exists(1, n2, prec)) continue;
    auto skew = [](int a, int b) { return a > b ? (double)a / b : (double)b / a; };
    if (!best || skew(n1, n2) < skew(n / best, best)) best = n2;
  }
  return best;
}

template <typename T>
int build_tables(int n, int prec, BigTables* t) {
  const long double pi = 3.141592653589793238462643383279503L;
  if (const int n2 = composite_split(n, prec)) {
    t->composite = true;
    t->M = n; t->M2 = n2; t->M1 = n / n2;
    std::vector<cx<T>> step((size_t)n);
    for (int k1 = 0; k1 < t->M1; ++k1)
      for (int m2 = 0; m2 < t->M2; ++m2) {
        const long long q = ((long long)k1 * m2) % n;
        const long double a = 2.0L * pi * (long double)q / (long double)n;
        step[(size_t)k1 * t->M2 + m2] = mk<T>((T)cosl(a), (T)(-sinl(a)));
      }
    MFFT_HIP(hipMalloc(&t->step, step.size() * sizeof(cx<T>)));
    MFFT_HIP(hipMemcpy(t->step, step.data(), step.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
    return 0;
  }
  int M = 2;
  while (M < 2 * n - 1) M *= 2;
  int lg = 0;
  while ((1 << lg) < M) ++lg;
  t->M = M;
  t->M1 = 1 << ((lg + 1) / 2);
  t->M2 = M / t->M1;
  auto c = build_chirp<T>(n);
  std::vector<cx<T>> step((size_t)M), b((size_t)M, mk<T>((T)0, (T)0));
  for (int k1 = 0; k1 < t->M1; ++k1)
    for (int m2 = 0; m2 < t->M2; ++m2) {
      const long long q = ((long long)k1 * m2) % M;
      const long double a = 2.0L * pi * (long double)q / (long double)M;
      step[(size_t)k1 * t->M2 + m2] = mk<T>((T)cosl(a), (T)(-sinl(a)));
    }
  // b[m] = b[M - m] = conj(c[m]) for |m| < n, scaled by 1 / M so that the inverse launches need no normalisation
  for (int m = 0; m < n; ++m) {
    const cx<T> v = mk<T>(c[m].x / (T)M, -c[m].y / (T)M);
    b[m] = v;
    if (m) b[M - m] = v;
  }
  MFFT_HIP(hipMalloc(&t->chirp, c.size() * sizeof(cx<T>)));
  MFFT_HIP(hipMemcpy(t->chirp, c.data(), c.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  MFFT_HIP(hipMalloc(&t->step, step.size() * sizeof(cx<T>)));
  MFFT_HIP(hipMemcpy(t->step, step.data(), step.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  MFFT_HIP(hipMalloc(&t->bhat, b.size() * sizeof(cx<T>)));
  MFFT_HIP(hipMemcpy(t->bhat, b.data(), b.size() * sizeof(cx<T>), hipMemcpyHostToDevice));
  MFFT_TRY(big_forward(t->bhat, 1, *t, prec, false, nullptr));
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

int tables_for(int n, int prec, BigTables* out) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_big_mu);
  const auto key = std::make_tuple(dev, n, prec);
  auto it = g_big.find(key);
  if (it == g_big.end()) {
    BigTables t;
    MFFT_TRY(prec == MFFT_DOUBLE ? build_tables<double>(n, prec, &t) : build_tables<float>(n, prec, &t));
    it = g_big.emplace(key, t).first;
  }
  *out = it->second;
  return 0;
}

// one scratch buffer per (device, stream): launches of one stream run in order, so its buffer is free again when the next
// call on that stream reaches it; grown on demand, kept for the life of the process (at most BIG_SCRATCH_BYTES + one vector)
constexpr size_t BIG_SCRATCH_BYTES = (size_t)256 << 20;
int scratch_for(hipStream_t s, size_t bytes, void** p) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_big_mu);
  auto& e = g_scratch[std::make_pair(dev, s)];
  if (e.second < bytes) {
    if (e.first) {
      // growth frees the old buffer: never under a capture (MFFT_GRAPH=1) -- a graph captured earlier on this stream would
      // keep the old pointer.  The first allocation is the full budget, so growth only happens for a single vector beyond it.
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
      if (cs != hipStreamCaptureStatusNone) return set_error(MFFT_ERR_INVALID, "four-step scratch would grow inside a stream capture");
      MFFT_HIP(hipStreamSynchronize(s));
      MFFT_HIP(hipFree(e.first));
      e = {nullptr, 0};
    }
    const size_t want = std::max(bytes, BIG_SCRATCH_BYTES);
    MFFT_HIP(hipMalloc(&e.first, want));
    e.second = want;
  }
  *p = e.first;
  return 0;
}

}  // namespace (the release hook below is called from plan.hip)
// A plan's stream is going away: its scratch buffer (up to 256 MiB) goes with it instead of staying for the life of the process.
void big_release_stream(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return; }
  std::lock_guard<std::mutex> lk(g_big_mu);
  auto it = g_scratch.find(std::make_pair(dev, s));
  if (it == g_scratch.end()) return;
  if (it->second.first) {
    (void)hipStreamSynchronize(s);
    (void)hipFree(it->second.first);
  }
  g_scratch.erase(it);
}
namespace {

template <typename T, int MODE>
int run_big_t(const void* in, void* out, const VecMap& imap, const VecMap& omap, i64 nvec, int n, int nout, int valid, bool inverse,
              double scale, bool vfast, int prec, hipStream_t s) {
  BigTables t;
  MFFT_TRY(tables_for(n, prec, &t));
  const size_t per_vec = (size_t)t.M * sizeof(cx<T>);
  i64 chunk = (i64)(BIG_SCRATCH_BYTES / per_vec);
  if (chunk < 1) chunk = 1;
  if (chunk > nvec) chunk = nvec;
  void* z = nullptr;
  MFFT_TRY(scratch_for(s, (size_t)chunk * per_vec, &z));
  const cx<T>* chirp = static_cast<const cx<T>*>(t.chirp);
  for (i64 v0 = 0; v0 < nvec; v0 += chunk) {
    const i64 nv = std::min(chunk, nvec - v0);
    // vectors [v0, v0 + nv): the maps address vector v0 + v, which for the two-level vector index is not a plain offset
    VecMap im = imap, om = omap;
    const void* inp = in;
    void* outp = out;
    // shift by whole vectors through the base pointers where the map is linear in v (one column group or unit outer)
    auto shift = [&](const VecMap& m, i64 v) { return (v / m.ncols) * m.outer + (v % m.ncols) * m.cstride; };
    const bool linear_ok = (v0 % imap.ncols == 0 || imap.ncols >= nvec) && (v0 % omap.ncols == 0 || omap.ncols >= nvec);
    if (!linear_ok) return set_error(MFFT_ERR_INTERNAL, "big transform: chunk boundary inside a column group");
    const size_t ies = MODE == BZ_R2C ? sizeof(T) : sizeof(cx<T>), oes = MODE == BZ_C2R ? sizeof(T) : sizeof(cx<T>);
    inp = static_cast<const char*>(in) + (size_t)shift(imap, v0) * ies;
    outp = static_cast<char*>(out) + (size_t)shift(omap, v0) * oes;
    const i64 gcount = nv * (i64)t.M, scount = nv * (i64)nout;
    if (inverse)
      hipLaunchKernelGGL((bz_gather<T, MODE, true>), dim3(nblocks(gcount)), dim3(256), 0, s, inp, static_cast<cx<T>*>(z), chirp, im, nv, n, t.M, valid, vfast);
    else
      hipLaunchKernelGGL((bz_gather<T, MODE, false>), dim3(nblocks(gcount)), dim3(256), 0, s, inp, static_cast<cx<T>*>(z), chirp, im, nv, n, t.M, valid, vfast);
    MFFT_HIP(hipGetLastError());
    MFFT_TRY(big_forward(z, nv, t, prec, false, s));
    if (!t.composite) {
      hipLaunchKernelGGL((bz_mul<T, false>), dim3(nblocks(gcount)), dim3(256), 0, s, static_cast<cx<T>*>(z), static_cast<const cx<T>*>(t.bhat), gcount, t.M);
      MFFT_HIP(hipGetLastError());
      MFFT_TRY(big_forward(z, nv, t, prec, true, s));
    }
    const int perm1 = t.composite ? t.M1 : 0;
    if (inverse)
      hipLaunchKernelGGL((bz_scatter<T, MODE, true>), dim3(nblocks(scount)), dim3(256), 0, s, static_cast<const cx<T>*>(z), outp, chirp, om, nv, n, t.M, nout, (T)scale, vfast, perm1);
    else
      hipLaunchKernelGGL((bz_scatter<T, MODE, false>), dim3(nblocks(scount)), dim3(256), 0, s, static_cast<const cx<T>*>(z), outp, chirp, om, nv, n, t.M, nout, (T)scale, vfast, perm1);
    MFFT_HIP(hipGetLastError());
  }
  return 0;
}

VecMap rows_map(i64 stride, i64 nrows) { return VecMap{0, stride, nrows > 0 ? nrows : 1, 0, 1, (i64)1 << 62}; }
VecMap cols_map(i64 outer, i64 ncols, const RowSpec& r) {
  VecMap m{outer, 1, ncols, 0, r.lo, (i64)1 << 62};
  if (r.split > 0) {
    if (r.split == 1) m.lo = r.hi;
    else { m.hi = r.hi; m.split = r.split; }
  }
  return m;
}

}  // namespace

bool big_length_ok(int64_t n) { return n >= 2 && n <= MFFT_BIG_MAX_LENGTH; }

// chunks of the strided form must end on column-group boundaries: process one outer batch (or a run of whole ones) per
// chunk by letting run_big_t's chunk be a multiple of ncols -- done here by splitting the call per group of outer batches
int big_col(const ColArgs& a, hipStream_t s) {
  if (a.pad || a.mask || a.band.on || a.in_wrap)
    return set_error(MFFT_ERR_UNSUPPORTED, "length %d has no radix plan: the fused 3/2-rule / 2/3-rule passes are not available for it", a.n);
  BigTables tb;
  MFFT_TRY(tables_for(a.n, a.prec, &tb));
  const size_t per_vec = (size_t)tb.M * elem_bytes(a.prec, true);
  const i64 per_outer = a.ncols;
  i64 outers = std::max<i64>(1, (i64)(BIG_SCRATCH_BYTES / per_vec) / per_outer);     // whole outer batches per call
  for (i64 o0 = 0; o0 < a.nouter; o0 += outers) {
    const i64 no = std::min(outers, a.nouter - o0);
    const size_t es = elem_bytes(a.prec, true);
    const void* in = static_cast<const char*>(a.in) + (size_t)(o0 * a.in_outer) * es;
    void* out = static_cast<char*>(a.out) + (size_t)(o0 * a.out_outer) * es;
    const VecMap im = cols_map(a.in_outer, a.ncols, a.in_rows), om = cols_map(a.out_outer, a.ncols, a.out_rows);
    // inside one call the chunking of run_big_t must not cut a column group unless there is a single group
    i64 done = 0;
    const i64 nvec = no * a.ncols;
    const i64 fit = std::max<i64>(1, (i64)(BIG_SCRATCH_BYTES / per_vec));
    if (no == 1 && fit < nvec) {        // one batch wider than the scratch: columns in runs (linear in v inside one group)
      for (done = 0; done < nvec; done += fit) {
        const i64 nv = std::min(fit, nvec - done);
        VecMap i1 = im, o1 = om;
        i1.ncols = o1.ncols = nv;
        const void* ip = static_cast<const char*>(in) + (size_t)done * es;
        void* op = static_cast<char*>(out) + (size_t)done * es;
        MFFT_TRY(a.prec == MFFT_DOUBLE
                     ? (run_big_t<double, BZ_C2C>(ip, op, i1, o1, nv, a.n, a.n, a.n, a.inverse, a.scale, true, a.prec, s))
                     : (run_big_t<float, BZ_C2C>(ip, op, i1, o1, nv, a.n, a.n, a.n, a.inverse, a.scale, true, a.prec, s)));
      }
    } else {
      MFFT_TRY(a.prec == MFFT_DOUBLE
                   ? (run_big_t<double, BZ_C2C>(in, out, im, om, nvec, a.n, a.n, a.n, a.inverse, a.scale, true, a.prec, s))
                   : (run_big_t<float, BZ_C2C>(in, out, im, om, nvec, a.n, a.n, a.n, a.inverse, a.scale, true, a.prec, s)));
    }
  }
  return 0;
}

int big_row(const RowArgs& a, hipStream_t s) {
  if (a.zs.nchunk) return set_error(MFFT_ERR_UNSUPPORTED, "length %d has no radix plan: no z-chunked kernel for it", a.n);
  const VecMap im = rows_map(a.in_stride, a.nrows), om = rows_map(a.out_stride, a.nrows);
  return a.prec == MFFT_DOUBLE
             ? run_big_t<double, BZ_C2C>(a.in, a.out, im, om, a.nrows, a.n, a.n, a.n, a.inverse, a.scale, false, a.prec, s)
             : run_big_t<float, BZ_C2C>(a.in, a.out, im, om, a.nrows, a.n, a.n, a.n, a.inverse, a.scale, false, a.prec, s);
}

int big_real(bool c2r, const RealArgs& a, hipStream_t s) {
  if (a.zs.nchunk) return set_error(MFFT_ERR_UNSUPPORTED, "real length %d has no radix plan: no z-chunked kernel for it", a.n);
  const int nbins = a.n / 2 + 1;
  const int valid = a.valid > 0 && a.valid < nbins ? a.valid : nbins;
  const VecMap im = rows_map(a.in_stride, a.nrows), om = rows_map(a.out_stride, a.nrows);
  if (c2r)
    return a.prec == MFFT_DOUBLE
               ? run_big_t<double, BZ_C2R>(a.in, a.out, im, om, a.nrows, a.n, a.n, valid, true, a.scale, false, a.prec, s)
               : run_big_t<float, BZ_C2R>(a.in, a.out, im, om, a.nrows, a.n, a.n, valid, true, a.scale, false, a.prec, s);
  return a.prec == MFFT_DOUBLE
             ? run_big_t<double, BZ_R2C>(a.in, a.out, im, om, a.nrows, a.n, valid, valid, false, a.scale, false, a.prec, s)
             : run_big_t<float, BZ_R2C>(a.in, a.out, im, om, a.nrows, a.n, valid, valid, false, a.scale, false, a.prec, s);
}

}  // namespace mfft
