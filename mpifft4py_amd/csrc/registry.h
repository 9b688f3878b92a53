// registry.h -- table of compiled gfx950 kernel instantiations.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "fft_kernels.h"
#include "fft_chirpz.h"
#include "fft_col3.h"
#include "twiddle.h"
#include "plans.h"

namespace mfft {

enum Family {
  FAM_COL = 0, FAM_ROW = 1, FAM_R2C = 2, FAM_C2R = 3,
  // chirp-z (Bluestein) variants: entry.n is the convolution length M, the logical length is a launch parameter
  FAM_COLZ = 4, FAM_ROWZ = 5, FAM_R2CZ = 6, FAM_C2RZ = 7,
  FAM_R2CZH = 8, FAM_C2RZH = 9,    // even real length: chirp-z of n/2 complex values + split pass
  FAM_NLZ = 10                     // fused nonlinear z stage (fft_nlz.h): entry.n is the REAL length of a z row, tile = row PAIRS per workgroup
};

struct KernelEntry {
  int family;
  int n;            // transform length (R2C/C2R: the REAL length)
  int prec;         // 0 single, 1 double
  int inv;          // COL/ROW: 1 = inverse
  int nt;           // COL: 1 = non-temporal variant (128-byte aligned rows only)
  int nt_inplace;   // COL, nt = 1: also faster than the regular variant when the transform is in place
  int pad;          // COL: 1 = zero-padded input (inverse), 2 = truncated output (forward); ROW / R2C / C2R: 3 = column-limited
                    // (3/2-rule), 4 = complex side split into z chunks (fused pencil pack / unpack), 7 = both (3/2-rule pencils)
  int tile;         // COLS (COL) or ROWS (others)
  int threads;
  int lds_bytes;
  int tw_count;     // entries of the inter-pass twiddle table
  int grid_mult;    // COL: workgroups per tile (3 for ColFft3S, fft_col3.h: one third of a tile's transform each)
  void (*build_tw)(void* host_dst);
  void (*launch)(const void* params, int grid, hipStream_t s);
  const void* func;
  const char* name;
};

std::vector<KernelEntry>& kernel_registry();
const KernelEntry* find_kernel(int family, int n, int prec, int inv, int nt = 0, int pad = 0);
// chirp-z kernel with the smallest convolution length M >= 2n-1 (nullptr: n too long)
const KernelEntry* find_chirpz(int family, int n, int prec, int inv);

// ---- default tiling heuristics (measured on MI355X at 1024^3) -----
// Strided-axis kernel: tiles are 128 bytes wide (one L2 line per row segment; with
// narrower tiles every line is fetched once per tile that touches it).  The
// exchange buffer may take up to 128 KiB of the CU's 160 KiB LDS; if whole complex
// values do not fit, real and imaginary parts are exchanged one after the other.
// Single precision: one lane may own VEC = 2 adjacent columns (16 bytes per lane, twiddles loaded once).  That pays
// only while a thread holds at most 32 complex values and the workgroup stays at 512 threads (2^a <= 1024: +5..17 %);
// measured with kbench2 (profiles/r01_kbench2_long_and_fp32.txt): VEC = 1 is ahead for the E = 20/24/40 plans
// (640: +20 %, 1280: 2x, 768: +5 %, 1536: +12 %) and for 2048 / 4096, where VEC = 2 needs 1024 threads and spills.
// Two (or three) workgroups per CU instead of one: when the whole-complex exchange buffer of a 128-byte tile takes
// between 80 and 128 KiB of the CU's 160 KiB (640 < N <= 1024), the kernel uses the split re/im exchange and LDS
// twiddles (and one column per lane), so that one workgroup's loads and stores overlap another's passes.  Measured with
// kbench3 (interleaved A/B, profiles/r02_kbench3_variants.txt), y in place / x in place / x out of place, ms:
//   fp64 1024 as 8x8x4x4 (plans.h)  3.52 -> 3.44   3.57 -> 3.35   3.63 -> 3.19      (as 16x8x8: y 3.52 -> 3.68, rejected)
//   fp64  768 (8x8x4x3)             1.69 -> 1.39   1.62 -> 1.40   1.76 -> 1.36
//   fp64  800 (5x5x4x4x2)           2.40 -> 2.07   2.42 -> 1.90   2.45 -> 1.87
//   fp32 1024 (16x8x8, 1024 thr.)   1.95 -> 1.74   1.82 -> 1.76   1.86 -> 1.63      (two columns per lane + split: 2.1 - 2.8)
// Non-temporal accesses help these kernels in double precision (in place too) and hurt in single (8 bytes per lane).
template <class S, typename T> constexpr bool col_pair() {
  constexpr long long tile = (long long)S::N * 128;
  constexpr int lanes = 128 / (int)sizeof(cx<T>);
  if (S::N == 512 && sizeof(T) == 8 && S::E == 4) return true;      // plans.h MFFT_COLPLANS_F64_B: 1024 threads, two per CU
  return S::NP > 1 && tile > 81920 && tile <= 131072 && S::TPT * lanes <= 1024 &&
         (S::N != 1024 || sizeof(T) == 4 || S::E == 8);
}
template <class S, typename T> constexpr int col_vec() {
  constexpr int lanes16 = 16 / (int)sizeof(cx<T>) > 0 ? 16 / (int)sizeof(cx<T>) : 1;   // columns in 16 bytes
  constexpr int full = 128 / (int)sizeof(cx<T>);                                         // columns in a 128-byte tile
  if (col_pair<S, T>()) return 1;
  return (lanes16 > 1 && S::E * lanes16 <= 32 && S::TPT * (full / lanes16) <= 512) ? lanes16 : 1;
}
// Round 5, two 128-byte tiles per workgroup (256-byte row segments).  A workgroup's waves go round the four SIMDs of its CU,
// so 5 waves (length 1200 in double precision: 40 threads per transform x 8 columns = 320 threads) put two on one SIMD and
// the passes of the whole workgroup last as long as THAT SIMD's two waves (tools/membench stamp1200: passes 24.8 k cycles =
// 2 x the 12.4 k cycles one wave issues).  A second workgroup cannot hide it: the dispatcher admits a workgroup only where
// EVERY SIMD has registers for ceil(waves / 4) of its waves (tools/occ_probe: 5 waves of 168 registers -> one resident per
// CU where the occupancy API answers two; 10 waves of 96 -> one; 4 waves of 168 -> two, three), so the second workgroup of
// round 5's capped build never ran.  With 16 columns the one resident workgroup has 10 waves = 3 3 2 2 and twice the loads
// in flight: 1200 fp64 strided pass 8.05 / 7.30 / 7.80 -> 6.44 / 6.42 / 6.61 ms (kbench3 occ1200: y in place, x in place, x out
// of place; profiles/r05_wave_placement.txt).  Only where the split exchange of 16 columns fits the LDS (N <= 1280), the
// workgroup stays within 1024 threads and the kernel is alone on its CU anyway (N > 1024): of the plans, 1200.
#ifndef MFFT_COL_WIDE
#define MFFT_COL_WIDE 1
#endif
template <class S, typename T> constexpr bool col_wide() {
  if (!MFFT_COL_WIDE) return false;
  constexpr int cols = 128 / (int)sizeof(cx<T>);
  constexpr int waves = (S::TPT * cols + 63) / 64;
  return sizeof(T) == 8 && S::NP > 1 && S::N > 1024 && waves % 4 == 1 && S::TPT * 2 * cols <= 1024 &&
         (long long)S::N * 2 * cols * (int)sizeof(T) <= 163840;
}
// The same idea at the other end: plans with 8 or 16 threads per transform make 64- or 128-thread workgroups on 128-byte
// tiles -- one or two waves that each reserve a wave's registers on ALL four SIMDs (the rule above), at 150 - 260 registers
// two or three such workgroups = 128 - 384 threads per CU.  With 256 threads (one wave per SIMD; 256- or 512-byte row
// segments, split exchange) the same registers hold twice the threads.  kbench3 wide16 / wideb, double precision, y in place /
// x in place / x out of place, ms:   448 (28x4x4) 0.332 / 0.317 / 0.330 -> 0.287 / 0.302 / 0.287;   480 (10x6x2x2x2) 0.447 / 0.414 / 0.410 ->
// 0.369 / 0.377 / 0.372;   672 (42x2x2x2x2) 1.387 / 1.276 / 1.339 -> 1.221 / 1.169 / 1.219;   336 0.180 / 0.162 / 0.174 -> 0.164 / 0.152 / 0.159;
// 320 (40x8) 0.117 / 0.105 / 0.128 -> 0.107 / 0.100 / 0.121;   240 0.062 / 0.059 / 0.057 -> 0.046 / 0.045 / 0.044;   single precision 448
// 0.160 / 0.145 / 0.158 -> 0.149 / 0.138 / 0.150.   Not: 640 (+-2 %), 600, 720 (their three-wave workgroups stay ahead), 480 fp32.
template <class S, typename T> constexpr int col_wide_small() {        // columns per workgroup, 0: the default
  if (!MFFT_COL_WIDE || S::NP < 2) return 0;
  if (sizeof(T) == 8 && S::TPT == 16 && (S::N == 448 || S::N == 480 || S::N == 672)) return 16;
  if (sizeof(T) == 8 && S::TPT == 8 && (S::N == 240 || S::N == 320 || S::N == 336)) return 32;
  if (sizeof(T) == 4 && S::TPT == 16 && S::N == 448) return 32;
  return 0;
}
template <class S, typename T> constexpr int col_cols() {
  constexpr int vec = col_vec<S, T>();
  if (col_wide_small<S, T>() > 0) return col_wide_small<S, T>();
  if (col_wide<S, T>()) return 2 * (128 / (int)sizeof(cx<T>));
  int cols = 128 / (int)sizeof(cx<T>);
  while (cols > vec && S::TPT * (cols / vec) > 1024) cols /= 2;
  while (cols > vec && (long long)S::N * cols * (int)sizeof(T) > 131072) cols /= 2;   // even split must fit
  while (S::TPT * (cols / vec) < 64) cols *= 2;      // at least one full wave
  return cols;
}
template <class S, typename T> constexpr bool col_split() {
  if (col_wide_small<S, T>() > 0) return true;
  return S::NP > 1 && (col_pair<S, T>() || (long long)S::N * col_cols<S, T>() * (int)sizeof(cx<T>) > 131072);
}
template <class S, typename T> constexpr bool col_twlds() {
  if (col_wide_small<S, T>() > 0) return S::N != 672;      // (672: 1.22 / 1.17 / 1.22 without against 1.27 / 1.20 / 1.26 with the table in LDS)
  return S::NP > 1 && (col_pair<S, T>() || (!col_split<S, T>() &&
         (long long)S::N * col_cols<S, T>() * (int)sizeof(cx<T>) + S::TW * (int)sizeof(cx<T>) <= 65536));
}
// Contiguous-axis kernels keep one exchange buffer per row, so the rows a CU holds at once (its threads in flight) are
// set by LDS.  The plans with 12 or 20 values per thread (3- and 5-smooth lengths >= 384) in double precision had 6 - 10
// rows = 380 - 500 threads per CU; they drop the LDS copy of the twiddles (it is per workgroup) and exchange real and
// imaginary parts one after the other (half the buffer).  z stages (r2c / c2r) of the R2C pair, ms, before -> without
// LDS twiddles -> with the split exchange as well: 1152^3 5.49/6.04 -> 5.02/4.96; 1280^3 8.60/10.03 -> 7.05/7.24;
// 1536^3 12.43/12.18 -> 11.38/10.76 (profiles/r02_row_kernels_long_lengths.txt).
// (single precision: measured worse -- 1152^3 z stages 2.71 / 2.74 -> 2.95 / 3.58 ms, 1536^3 6.4 / 5.7 -> 6.9 / 6.3)
// Round 3, the 30-values-per-thread plans (lengths with 3 and 5 among their factors): 8 or 16 threads per row, so the 40 KB
// budget meant 64 - 128 threads per workgroup and 4 workgroups = 256 - 512 threads per CU.  From length 120 on they run
// lean with half the LDS budget per workgroup -- in double precision every contiguous-axis kernel, in single precision
// the c2r kernels only (z stages r2c / c2r of the R2C pair, ms, 40 KB + LDS twiddles -> lean, profiles/r03_mixed_radix_15.txt:
// fp64 960^3 6.30 / 4.11 -> 4.90 / 4.08, 720^3 2.88 / 2.15 -> 2.72 / 1.93, 480^3 0.80 / 0.54 -> 0.64 / 0.43;
// fp32 960^3 2.25 / 3.41 -> 2.71 / 2.56, 720^3 0.94 / 1.47 -> 1.11 / 1.05, 1200^3 4.45 / 6.48 -> 5.14 / 4.91).
// (their double-precision kernels then get a register cap as well: row_occ_wgs below)
// Round 5, the 42-values plans (21 * 2^a) in double precision: their contiguous-axis kernels hold 280 - 430 registers (AGPRs
// included), so a CU admits ONE workgroup whatever its size (the dispatcher reserves a wave's registers on every SIMD per
// started group of four waves: section "Round 5" of DESIGN.md 4) -- and the 40 KB LDS budget below made that workgroup 64
// threads: one wave per CU (1344^3: c2r 21.4 ms, r2c 16.6, where a strided pass takes 12 - 13).  They take 256 threads (one
// wave per SIMD) with the split exchange in up to 128 KB instead.
#ifndef MFFT_ROW_WIDE42
#define MFFT_ROW_WIDE42 1
#endif
// Measured (1344^3 / 672^3 / 336^3 pairs, stage ms): c2r 21.36 -> 19.24 / 2.58 -> 2.23 / 0.33 -> 0.22, r2c 16.57 -> 17.57 / 1.72 -> 1.91 / 0.20 -> 0.21:
// the c2r kernels take it, the others keep their rows.
template <class S, typename T, bool C2R = false> constexpr bool row_wide42() {
  return MFFT_ROW_WIDE42 && C2R && sizeof(T) == 8 && S::E % 21 == 0 && S::N >= 168 && S::NP > 1;
}
template <class S, typename T, bool C2R = false> constexpr bool row_lean15() {
  return S::E % 15 == 0 && S::N >= 120 && (sizeof(T) == 8 || C2R);
}
template <class S, typename T, bool C2R = false> constexpr bool row_lean() {
  return (sizeof(T) == 8 && S::E >= 12 && S::N >= 384) || row_lean15<S, T, C2R>();
}
template <class S, typename T, bool C2R = false> constexpr bool row_split() {
  // the c2r kernels of the E = 20 plans lose with it (1280: 7.2 -> 8.3 ms)
  if (row_wide42<S, T, C2R>()) return true;
  return S::NP > 1 && row_lean<S, T, C2R>() && !(C2R && S::E >= 20 && !row_lean15<S, T, C2R>());
}
template <class S, typename T, bool SPLIT, bool C2R = false> constexpr int row_rows_n() {
  int rows = 256 / S::TPT;
  if (rows < 1) rows = 1;
  const long long per_row = (long long)(S::N + S::N / S::R(0) + 1) * (int)(SPLIT ? sizeof(T) : sizeof(cx<T>));
  const long long budget = row_wide42<S, T, C2R>() && SPLIT ? 131072 : row_lean15<S, T, C2R>() ? 20480 : 40960;
  while (rows > 1 && per_row * rows > budget && S::TPT * (rows / 2) >= 64) rows /= 2;   // never below one wave
  return rows;
}
template <class S, typename T, bool C2R = false> constexpr int row_rows() { return row_rows_n<S, T, row_split<S, T, C2R>(), C2R>(); }
template <class S, typename T, bool C2R = false> constexpr bool row_twlds() {
  return S::NP > 1 && !row_lean<S, T, C2R>() && !row_wide42<S, T, C2R>();
}

#ifndef MFFT_COL_OCC_R5
#define MFFT_COL_OCC_R5 1
#endif
// Workgroups per CU the strided kernel's REGISTER allocation must leave room for (0: whatever the compiler takes).
// 1152 in double precision runs 768 threads with ~88 VGPRs: one register more than two workgroups per CU allow
// (6 waves per SIMD x 85); capped, both fit (LDS 2 x 72 KiB) and one's loads overlap the other's passes.
// Single precision (kbench3 f32long, profiles/r02_kbench3_long_lengths.txt): 1152 (8x8x3x3x2, 768 threads) with the cap
// for two workgroups 3.7 - 3.9 -> 2.9 - 3.0 ms per pass; 1536 as 64-byte tiles (8 columns, 512 threads, 48 KiB) with the
// cap for three 7.9 - 8.7 -> 7.3 - 7.8 ms (the same tiling changes nothing in double precision).
// Round 5: scripts/kernel_regs.py over every strided kernel -- which plans hold fewer workgroups per CU by REGISTERS than their
// LDS would admit, and by how little?  Capped and measured against the uncapped library in one session
// (profiles/r05_col_occupancy_caps.txt, pairs):  single precision 720 (30 values per thread, 384 threads: 110 -> 94 VGPRs, no
// scratch, THREE workgroups) 4.22 / 4.27 -> 4.13 / 3.97 ms;  single precision 2304 (768 threads: 84 -> 80, 16 bytes of scratch,
// two workgroups) 170.7 -> 163.6 ms (y passes 29.6 / 32.3 -> 26.7 / 27.3): kept.  1200 / 2400 (30 values, 320 / 640 threads: 178
// -> 168 VGPRs in double precision with 36 bytes of scratch, 115 -> 89 in single with none; two workgroups fit their 75 KiB
// of split exchange): double precision 42.7 / 43.0 -> 43.3 / 42.0 (nothing), single precision 24.6 / 23.9 -> 24.8 / 25.4 and
// 281 -> 288 (worse): NOT kept -- a second workgroup does not help these five-pass kernels, whatever it is they wait for.
// (768 and 750 looked the same but their LDS twiddle tables leave room for two workgroups only, which their registers
// allow already; 1440 / 1536 / 1792 need the four-round exchange for a second workgroup, which costs 450 - 560 bytes of
// scratch under the cap: round 3's result stands.)
// Round 5, single precision 1792 (28x4x4x4; 0.45 of the roofline): one 1024-thread workgroup per CU on 128-byte tiles (112 KB
// of split exchange) against two 512-thread workgroups on 64-byte tiles with LDS twiddles (69 KB each; 8 waves = two per
// SIMD, so the dispatcher's rule above admits the second at <= 128 registers): kbench3 occ1200, y in place / x in place / x
// out of place 12.52 / 13.94 / 13.56 -> 10.42 / 11.67 / 11.72 ms (without the LDS twiddles 11.54 / 13.59 / 13.27).
#ifndef MFFT_COL_NARROW_F32
#define MFFT_COL_NARROW_F32 1
#endif
// The same at 1440 (10x6x6x2x2: 768 threads on 128-byte tiles -> two workgroups of 384: 7.55 / 7.66 / 7.93 -> 7.17 / 7.23 / 7.32 ms) and
// at 2048, which returns from the 32x8x8 plan of round 1 (plans.h MFFT_COLPLANS_F32_C) to 16x16x8: 16 values per thread, 1024
// threads on 64-byte tiles, 64 registers, two workgroups per CU as the double-precision 1024 kernel has them: 17.14 / 19.37 /
// 17.73 -> 15.39 / 16.89 / 17.30 ms (32x8x8 on 64-byte tiles: 16.86 / 17.54 / 18.15).  Not at 1200 (3.95 / 4.02 / 4.01 -> 4.08 / 3.97 /
// 4.01), 1280 (3.94 / 3.77 / 3.60 -> 4.54 / 4.08 / 4.41), nor in double precision at 1600 (15.1 / 14.1 / 14.5 -> 18.7 / 17.5 / 17.2) and 1792
// (22.4 / 22.0 / 22.1 -> 20.6 / 22.7 / 21.6): kbench3 narrow, profiles/r05_wave_placement.txt.
template <class S, typename T> constexpr bool col_narrow_f32() {
  return MFFT_COL_NARROW_F32 && sizeof(T) == 4 &&
         ((S::N == 1792 && S::E == 28) || (S::N == 1440 && S::E == 30) || (S::N == 2048 && S::E == 16));
}
template <class S, typename T> constexpr int col_wgs() {
  if (sizeof(T) == 8) return ((S::N == 1152 && S::E == 12) || (S::N == 512 && S::E == 4)) ? 2 : 0;
  if (S::N == 1152 && S::E == 24) return 2;
  if (S::N == 1536 && S::E == 24) return 3;
  if (MFFT_COL_OCC_R5 && S::N == 2304 && S::E == 24) return 2;
  // (1728 in single precision -- 24x24x3, 576 threads = 9 waves on 64-byte tiles, 0.40 of the roofline -- capped at 80 registers for two
  // workgroups with the split exchange, 2 x 55 KB instead of one with 110 KB: 1728^3 pair 70.5 - 71.1 -> 70.3 ms, nothing; not kept)
  if (MFFT_COL_OCC_R5 && S::N == 720 && S::E == 30) return 3;
  // (16 + waves per SIMD: the cap that the dispatcher's rule asks of TWO workgroups, 2 x ceil(waves / 4) -- the plain
  // "workgroups x threads / 256" of mfft_kern_occ gives 3 for the 6 waves of 1440, and the second workgroup would stay out)
  if (col_narrow_f32<S, T>()) return 16 + 2 * ((S::TPT * 8 / 64 + 3) / 4);
  return 0;
}
constexpr int col_wgs_count(int w) { return w >= 16 ? 2 : w; }
// ---- generic __global__ wrapper + launch thunks ------------------------------
template <class K, class P>
__global__ __launch_bounds__(K::THREADS) void mfft_kern(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  K::body(p, (int)blockIdx.x, (int)threadIdx.x, lds);
}
// the same with a register cap: WGS workgroups per CU = WGS * THREADS / 256 waves per SIMD
// (WGS >= 16: WGS - 16 waves per SIMD, said directly)
template <class K, class P, int WGS>
__global__ __launch_bounds__(K::THREADS, WGS >= 16 ? WGS - 16 : (WGS * K::THREADS + 255) / 256) void mfft_kern_occ(P p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  K::body(p, (int)blockIdx.x, (int)threadIdx.x, lds);
}

template <class K, class P, int WGS = 0>
void launch_thunk(const void* params, int grid, hipStream_t s) {
  if constexpr (WGS > 1)
    hipLaunchKernelGGL((mfft_kern_occ<K, P, WGS>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, s, *static_cast<const P*>(params));
  else
    hipLaunchKernelGGL((mfft_kern<K, P>), dim3(grid), dim3(K::THREADS), K::LDS_BYTES, s, *static_cast<const P*>(params));
}

template <class S, typename T>
void tw_thunk(void* dst) {
  auto v = build_pass_twiddles<S, T>();
  memcpy(dst, v.data(), v.size() * sizeof(cx<T>));
}

template <class K, class P, class S, typename T, int WGS = 0>
KernelEntry make_entry(int family, int n, int inv, int tile, const char* name) {
  KernelEntry e;
  e.family = family;
  e.n = n;
  e.prec = sizeof(T) == 8 ? 1 : 0;
  e.inv = inv;
  e.nt = 0;
  e.nt_inplace = 0;
  e.pad = 0;
  e.tile = tile;
  e.threads = K::THREADS;
  e.lds_bytes = K::LDS_BYTES;
  e.tw_count = S::TW;
  e.grid_mult = 1;
  e.build_tw = &tw_thunk<S, T>;
  e.launch = &launch_thunk<K, P, WGS>;
  if constexpr (WGS > 1) e.func = reinterpret_cast<const void*>(&mfft_kern_occ<K, P, WGS>);
  else e.func = reinterpret_cast<const void*>(&mfft_kern<K, P>);
  e.name = name;
  return e;
}

// (the masked-load / pruned 2/3-rule variants of the plans capped in round 5 hold 10 - 40 registers more than the plain
// kernels: under the same cap they would spill 100 - 170 bytes, so they keep the compiler's allocation)
template <class S, typename T> constexpr int col_wgs_mask() {
  constexpr bool r5 = sizeof(T) == 4 && ((S::N == 720 && S::E == 30) || (S::N == 2304 && S::E == 24));
  return r5 ? 0 : col_wgs<S, T>();
}

// non-temporal variants (for tiles whose rows are whole lines): not for the single-precision kernels with two workgroups per
// CU (8 bytes per lane: measured worse, section header), nor for 2048 in single precision, whose 64-register cap they
// overrun by 68 - 148 bytes of scratch
template <class S, typename T> constexpr bool col_has_nt() {
  return S::N >= 256 && !(col_pair<S, T>() && sizeof(T) == 4) && !(col_narrow_f32<S, T>() && S::N == 2048);
}
template <class S, typename T>
void register_col(const char* name) {
  auto& reg = kernel_registry();
  constexpr int W = col_wgs<S, T>();
  constexpr int WM = col_wgs_mask<S, T>();
  constexpr int C = ((sizeof(T) == 4 && S::N == 1536 && W == 3) || col_narrow_f32<S, T>()) ? 8 : col_cols<S, T>();      // see col_wgs
  // (1536 in single precision, 64-byte tiles since round 2: with LDS twiddles 7.25 / 7.73 / 7.46 -> 6.70 / 7.34 / 7.20 ms although only two
  // of its workgroups then fit a CU instead of three -- kbench3 tw1536)
  constexpr bool CT = col_twlds<S, T>() || col_narrow_f32<S, T>() || (sizeof(T) == 4 && S::N == 1536 && W == 3);
  constexpr bool CS = col_split<S, T>() || (W > 1 && (long long)S::N * C * (int)sizeof(cx<T>) * col_wgs_count(W) > 163840);
  constexpr int CV = col_narrow_f32<S, T>() ? 1 : col_vec<S, T>();
  reg.push_back(make_entry<ColFft<S, T, C, false, CT, CS, CV>, ColParams<T>, S, T, W>(FAM_COL, S::N, 0, C, name));
  reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV>, ColParams<T>, S, T, W>(FAM_COL, S::N, 1, C, name));
  if constexpr (S::E % 3 == 0 && S::N >= 6) {   // 3/2-rule lengths: pad-on-load (inverse) / truncate-on-store (forward)
    reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV, false, 1>, ColParams<T>, S, T, W>(FAM_COL, S::N, 1, C, name));
    reg.back().pad = 1;
    reg.push_back(make_entry<ColFft<S, T, C, false, CT, CS, CV, false, 2>, ColParams<T>, S, T, W>(FAM_COL, S::N, 0, C, name));
    reg.back().pad = 2;
  }
  // 2/3-rule: inverse transform with the dealias mask applied on load (pad = 5)
  reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV, false, 3>, ColParams<T>, S, T, WM>(FAM_COL, S::N, 1, C, name));
  reg.back().pad = 5;
  if constexpr (col_has_nt<S, T>()) {
    reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV, true, 3>, ColParams<T>, S, T, WM>(FAM_COL, S::N, 1, C, name));
    reg.back().pad = 5;
    reg.back().nt = 1;
    reg.back().nt_inplace = (col_pair<S, T>() && S::N != 512) ? 1 : 0;
  }
  // pruned 2/3-rule passes (pad = 6)
  reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV, false, 4>, ColParams<T>, S, T, WM>(FAM_COL, S::N, 1, C, name));
  reg.back().pad = 6;
  if constexpr (col_has_nt<S, T>()) {
    reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV, true, 4>, ColParams<T>, S, T, WM>(FAM_COL, S::N, 1, C, name));
    reg.back().pad = 6;
    reg.back().nt = 1;
    reg.back().nt_inplace = (col_pair<S, T>() && S::N != 512) ? 1 : 0;
  }
  if constexpr (col_has_nt<S, T>()) {     // aligned-row (non-temporal) variants
    reg.push_back(make_entry<ColFft<S, T, C, false, CT, CS, CV, true>, ColParams<T>, S, T, W>(FAM_COL, S::N, 0, C, name));
    reg.back().nt = 1;
    reg.back().nt_inplace = (col_pair<S, T>() && S::N != 512) ? 1 : 0;
    reg.push_back(make_entry<ColFft<S, T, C, true, CT, CS, CV, true>, ColParams<T>, S, T, W>(FAM_COL, S::N, 1, C, name));
    reg.back().nt = 1;
    reg.back().nt_inplace = (col_pair<S, T>() && S::N != 512) ? 1 : 0;
  }
}

// Round 5, y-pass builds (KernelEntry::nt == 2; core.hip launch_col takes them when the rows of a tile lie at most 64 KB
// apart): 64-byte tiles with LDS twiddles, two workgroups per CU by LDS alone (3 - 4 waves each, so the dispatcher's rule
// leaves them 256 registers).  kbench3 tw64, double precision, y in place / x in place / x out of place, ms:
//   1440  12.95 / 12.85 / 13.43 -> 9.62 / 14.49 / 14.39        1536  13.74 / 12.56 / 13.37 -> 11.55 / 14.41 / 14.04
// -- the y pass, whose neighbouring tiles share their 128-byte lines within a few MB, gains 16 - 26 %; the x pass, whose rows
// lie MBs apart (every row of a tile on its own page, twice the tiles), loses as much: so the build is chosen by the stride.
// Without the LDS twiddles the same tiles gain nothing (1440: 13.7 / 13.7 / 13.3; rounds 2 and 3 tried only that).
#ifndef MFFT_COL_YTILE
#define MFFT_COL_YTILE 1
#endif
template <class S, typename T> constexpr bool col_ytile() {
  // (1200: 5.99 -> 5.62 ms on the y pass against its 16-column build, kbench3 y64)
  return MFFT_COL_YTILE && sizeof(T) == 8 && ((S::N == 1440 && S::E == 30) || (S::N == 1536 && S::E == 24) || (S::N == 1200 && S::E == 30));
}
template <class S, typename T>
void register_col_ytile(const char* name) {
  if constexpr (col_ytile<S, T>()) {
    auto& reg = kernel_registry();
    constexpr int C = 64 / (int)sizeof(cx<T>);
    auto add = [&](KernelEntry e, int pad) {
      e.nt = 2;
      e.pad = pad;
      reg.push_back(e);
    };
    add(make_entry<ColFft<S, T, C, false, true, 1, 1>, ColParams<T>, S, T>(FAM_COL, S::N, 0, C, name), 0);
    add(make_entry<ColFft<S, T, C, true, true, 1, 1>, ColParams<T>, S, T>(FAM_COL, S::N, 1, C, name), 0);
    if constexpr (S::E % 3 == 0) {
      add(make_entry<ColFft<S, T, C, true, true, 1, 1, false, 1>, ColParams<T>, S, T>(FAM_COL, S::N, 1, C, name), 1);
      add(make_entry<ColFft<S, T, C, false, true, 1, 1, false, 2>, ColParams<T>, S, T>(FAM_COL, S::N, 0, C, name), 2);
    }
    add(make_entry<ColFft<S, T, C, true, true, 1, 1, false, 3>, ColParams<T>, S, T>(FAM_COL, S::N, 1, C, name), 5);
  }
}

// ColFft3 (fft_col3.h): N = 3 L as three sub-transforms of the plan SL.  Registered under pad codes 16 (plain), 17 (pad
// on load, inverse), 18 (truncate on store, forward): launch_col chooses between them and the ColFft kernels of the same
// length (MFFT_COL3=0 / 1 force either in the same process).
template <class SL, typename T>
void tw3_thunk(void* dst) {
  auto v = build_col3_twiddles<SL, T>();
  memcpy(dst, v.data(), v.size() * sizeof(cx<T>));
}
template <class K, class SL, typename T, int WGS>
KernelEntry make_entry3(int inv, int nt, int pad, int tile, const char* name) {
  KernelEntry e;
  e.family = FAM_COL;
  e.n = 3 * SL::N;
  e.prec = sizeof(T) == 8 ? 1 : 0;
  e.inv = inv;
  e.nt = nt;
  e.nt_inplace = nt;                    // two workgroups per CU (or 1024 threads): non-temporal in place too
  e.pad = pad;
  e.tile = tile;
  e.threads = K::THREADS;
  e.lds_bytes = K::LDS_BYTES;
  e.tw_count = K::TW;
  e.grid_mult = 1;
  e.build_tw = &tw3_thunk<SL, T>;
  e.launch = &launch_thunk<K, ColParams<T>, WGS>;
  if constexpr (WGS > 1) e.func = reinterpret_cast<const void*>(&mfft_kern_occ<K, ColParams<T>, WGS>);
  else e.func = reinterpret_cast<const void*>(&mfft_kern<K, ColParams<T>>);
  e.name = name;
  return e;
}
template <class SL, typename T>
void register_col3(const char* name) {
  auto& reg = kernel_registry();
  constexpr int VEC = (sizeof(T) == 4) ? 2 : 1;                   // 16 bytes per lane
  constexpr int C = 128 / (int)sizeof(cx<T>);                     // 128-byte tiles
  constexpr int THR = SL::TPT * (C / VEC);
  // whole-complex exchange while two workgroups of it (and their twiddles) fit the CU's 160 KB, else real / imaginary parts
  constexpr int SPL = ((long long)SL::N * C * (int)sizeof(cx<T>) + SL::TW * (int)sizeof(cx<T>) <= 81920) ? 0 : 1;
  constexpr int W = THR <= 512 ? 2 : 0;                           // register cap for two workgroups per CU
  reg.push_back(make_entry3<ColFft3<SL, T, C, false, true, SPL, VEC, false, 0>, SL, T, W>(0, 0, 16, C, name));
  reg.push_back(make_entry3<ColFft3<SL, T, C, true, true, SPL, VEC, false, 0>, SL, T, W>(1, 0, 16, C, name));
  reg.push_back(make_entry3<ColFft3<SL, T, C, false, true, SPL, VEC, true, 0>, SL, T, W>(0, 1, 16, C, name));
  reg.push_back(make_entry3<ColFft3<SL, T, C, true, true, SPL, VEC, true, 0>, SL, T, W>(1, 1, 16, C, name));
  reg.push_back(make_entry3<ColFft3<SL, T, C, true, true, SPL, VEC, false, 1>, SL, T, W>(1, 0, 17, C, name));
  reg.push_back(make_entry3<ColFft3<SL, T, C, false, true, SPL, VEC, false, 2>, SL, T, W>(0, 0, 18, C, name));
  // pad code 33: the pad-on-load inverse with one third of a tile's transform per workgroup (ColFft3S), registers capped for
  // two workgroups per CU (1024 threads at L = 512: 64 VGPRs, what the length-L ColFft kernel runs with)
  reg.push_back(make_entry3<ColFft3S<SL, T, C, true, true, SPL, VEC, false, 1>, SL, T, 2>(1, 0, 33, C, name));
  reg.back().grid_mult = 3;
}

// Register cap of the contiguous-axis kernels: the 30-values-per-thread plans in double precision come out at 256 VGPRs plus
// 4 - 36 AGPRs (c2c and r2c), i.e. ONE wave per SIMD; capped for two (512 threads per CU) they spill a few registers and
// run faster (MFFT_ROW_OCC=0 switches it off): z stage of the R2C pair, r2c: 480^3 0.62 -> 0.38 ms, 720^3 2.14 -> 1.41,
// 960^3 4.1 -> 3.0, 1200^3 9.3 -> 7.5, 1440^3 17.9 -> 12.5 (with the scheduling fences of fft_kernels.h R2CFft).
#ifndef MFFT_ROW_OCC_C2R
#define MFFT_ROW_OCC_C2R 1
#endif
#ifndef MFFT_ROW_OCC
#define MFFT_ROW_OCC 1
#endif
// Round 5: the 28-values-per-thread plans (7 * 2^a) as well.  Their c2r kernels sat at 242 - 248 VGPRs when radix 7 was
// measured (896^3 bwd_z 2.05 ms, profiles/r04_radix7_sweep.txt) and came out at 262 - 287 (256 + AGPRs = ONE wave per
// SIMD) after an unrelated edit of C2RFft two commits later: 4.2 ms in every sweep since, found by the bisect of
// profiles/r05_radix7_c2r_bisect.txt.  Capped, 448 / 896 / 1792 spill 20 - 100 bytes of scratch per lane; the short
// lengths (56 ... 224 with LDS twiddles) would spill 370: not capped.
template <class S, typename T> constexpr int row_occ_wgs(int threads) {
  return (MFFT_ROW_OCC && sizeof(T) == 8 && ((S::E % 15 == 0 && S::N >= 120) || (S::E == 28 && S::N >= 448)) && threads <= 256)
             ? 512 / threads : 0;   // shorter: 500+ bytes of scratch
}

// Round 4: real kernels whose threads per transform do not divide a wave (10, 12, 15, 20, 24, 30 ... threads) can run
// wave-packed (fft_kernels.h R2CFft / C2RFft WP: whole rows per wave, the mirrored bin through a wave shuffle -- r2c without
// the LDS round trip of its post-pass, c2r with every bin loaded once).  Measured (profiles/r04_wave_packed_real_kernels.txt,
// z stages of the R2C pair against the dense layout): it pays for the 30-values-per-thread plans, whose dense kernels
// hold the mirrors in E more registers under a two-wave cap -- r2c at every length (720^3 1.43 -> 1.35 ms, 900^3 3.09 ->
// 2.69, 1440^3 12.4 -> 11.3; 600 / 1200 unchanged), c2r in single precision (720^3 0.98 -> 0.73, 900^3 1.66 -> 1.51) and in
// double precision at 10, 12 and 20 threads per transform (600^3 0.77 -> 0.70, 720^3 1.28 -> 1.18, 1200^3 even; 15 and 24
// threads lose 10 - 13 %: 900^3, 1440^3) -- and LOSES for the 12- and 20-values-per-thread plans (1152^3: r2c 4.65 -> 5.41,
// c2r 4.75 -> 6.04 ms; 1000^3 r2c 3.31 -> 3.84), which keep the dense layout: their dense kernels run at full occupancy and
// a quarter of idle lanes costs more than the second load of a mirror pair.
#ifndef MFFT_REAL_WP
#define MFFT_REAL_WP 1
#endif
template <class S> constexpr bool wave_packable() { return MFFT_REAL_WP && S::TPT < 64 && 64 % S::TPT != 0 && S::E % 15 == 0 && S::E % 2 == 0; }   // (register pairing: even E)
template <class S, typename T> constexpr bool r2c_wave_packed() { return wave_packable<S>(); }
template <class S, typename T> constexpr bool c2r_wave_packed() {
  return wave_packable<S>() && (sizeof(T) == 4 || S::TPT == 10 || S::TPT == 12 || S::TPT == 20);
}
template <class S, typename T, bool C2R> constexpr int real_rows() {       // rows per workgroup of the r2c / c2r kernels
  constexpr int r = row_rows<S, T, C2R>();
  constexpr bool wp = C2R ? c2r_wave_packed<S, T>() : r2c_wave_packed<S, T>();
  if constexpr (wp) {
    constexpr int rpw = 64 / S::TPT;
    return r / rpw > 0 ? r / rpw * rpw : rpw;                              // whole waves of rows
  } else {
    return r;
  }
}
template <class S, typename T, bool C2R> constexpr int real_threads() {
  constexpr bool wp = C2R ? c2r_wave_packed<S, T>() : r2c_wave_packed<S, T>();
  if constexpr (wp) return real_rows<S, T, C2R>() / (64 / S::TPT) * 64;
  else return S::TPT * real_rows<S, T, C2R>();
}
template <class S, typename T> constexpr int c2r_rows() { return real_rows<S, T, true>(); }
template <class S, typename T> constexpr int c2r_threads() { return real_threads<S, T, true>(); }
template <class S, typename T> constexpr int r2c_rows() { return real_rows<S, T, false>(); }
template <class S, typename T> constexpr int r2c_threads() { return real_threads<S, T, false>(); }

// Round 5, the census of scripts/kernel_regs.py applied to the contiguous-axis kernels: plans whose PLAIN kernels sit one to
// fourteen registers above the next occupancy step while their LDS admits the extra workgroup, and which compile under the
// cap with little or no scratch (family: 0 c2c rows, 1 r2c, 2 c2r): single precision c2r of real 2048 (130 -> 128 VGPRs, no
// scratch: four workgroups of 256 threads instead of three), c2c rows of 720 / 1200 in single precision (172 - 178 -> 168,
// none: five workgroups instead of four), rows and r2c of 500 in double precision (182 -> 168: none / 20 bytes, six instead
// of four), the 20-values rows of 800 / 2000 in double precision (130 - 132 -> 128: 12 - 16 bytes, four instead of three).
// MEASURED against the library without them, alternating, one session (profiles/r05_row_occupancy_caps.txt, z stages in ms):
// 2048^3 fp32 c2r 13.9 / 13.8 -> 14.1 / 13.7, 1000^3 fp64 r2c 3.32 / 3.30 -> 3.31 / 3.32, C2C rows 1200 fp32 6.5 - 6.6 -> the
// same, 800 fp64 3.0 - 3.1 -> the same, 500 fp64 0.70 - 0.72 -> 0.72 - 0.74, 720 fp32 1.13 - 1.17 -> 1.22 - 1.23 (6 % WORSE):
// nothing gained anywhere -- the extra workgroup is not what these kernels wait for.  OFF; MFFT_ROW_OCC_R5=1 builds them.
#ifndef MFFT_ROW_OCC_R5
#define MFFT_ROW_OCC_R5 0
#endif
template <class S, typename T, int FAM> constexpr int row_occ_r5(int threads) {
  if (!MFFT_ROW_OCC_R5) return 0;
  if (sizeof(T) == 4 && FAM == 2 && S::N == 1024 && S::E == 16) return 1024 / threads;
  if (sizeof(T) == 4 && FAM == 0 && (S::N == 720 || S::N == 1200) && S::E == 30) return 768 / threads;       // three waves per SIMD
  if (sizeof(T) == 8 && FAM != 2 && S::N == 500 && S::E == 20) return 768 / threads;
  if (sizeof(T) == 8 && FAM == 0 && (S::N == 800 || S::N == 2000) && S::E == 20) return 1024 / threads;
  return 0;
}

#ifndef MFFT_C2R_MLDS
#define MFFT_C2R_MLDS 1
#endif
// Measured on every plan it applies to (profiles/r06_c2r_mlds.txt, z stage of (256, 256, n) meshes, second load -> LDS mirrors):
// the 12-values plans of 1152 / 2304 complex points gain 4 - 11 % (real 2304: 0.549 -> 0.528 ms fp64, 0.357 -> 0.321 fp32; real 4608: 1.10
// -> 0.98 / 0.75 -> 0.68), in single precision also 576 (real 1152: 0.150 -> 0.139) and the 20-values plan of 1000 (real 2000: 0.390 ->
// 0.368); the 30-values plans in double precision -- whose 1.31 x fetch at 1440^3 was the reason to try -- LOSE 2.2 x (the split
// exchange parks E more reals: 226 -> 289 VGPRs + 33 AGPRs, one wave per SIMD; 1440^3 bwd_z 10.1 -> 23.9 ms) and keep the second load,
// as do 500 / 576 / 1000 in double precision (+-4 %, +10 %).  Built for the winners only.
template <class S, typename T> constexpr bool c2r_mlds_candidate() {
  if (!MFFT_C2R_MLDS || S::NP < 2 || c2r_wave_packed<S, T>() || (S::TPT <= 64 && 64 % S::TPT == 0)) return false;
  if (S::E == 12 && (S::N == 1152 || S::N == 2304)) return true;
  // the 27 * 2^a row plans (plans.h group T) and, for the column-limited kernels of the 3/2-rule, their 9 * 2^a neighbours:
  // which of the builds is TAKEN is core.hip c2r_mlds_take's rule
  if (S::E == 12 && (S::N == 432 || S::N == 864 || S::N == 1728 || S::N == 3456 || S::N == 288 || S::N == 576)) return true;
  // rows of more than a wave's threads (real 3072 / 6144 with 12 values per thread, real 4096 / 8192 with 16)
  if ((S::E == 12 && (S::N == 1536 || S::N == 3072)) || (S::E == 16 && (S::N == 2048 || S::N == 4096))) return true;
  // the 20-values plans of the 25 * 2^a and 125 * 2^a lengths (10 - 100 threads per row, no shuffle, not wave-packed)
  if (S::E == 20 && (S::N == 400 || S::N == 500 || S::N == 800 || S::N == 1000 || S::N == 2000)) return true;
  return false;
}
template <class S, typename T>
void register_rows(const char* name) {
  auto& reg = kernel_registry();
  constexpr int R = row_rows<S, T>();
  constexpr int WO = row_occ_wgs<S, T>(S::TPT * R);
  constexpr int WO5 = row_occ_r5<S, T, 0>(S::TPT * R) ? row_occ_r5<S, T, 0>(S::TPT * R) : WO;      // the plain kernels only
  constexpr bool WPC = c2r_wave_packed<S, T>();
  constexpr bool WPR = r2c_wave_packed<S, T>();
  // c2r: only where the mirrored bins come through wave shuffles (threads per transform a power of two up to 64, or the
  // wave-packed layout: 12 - 28 bytes of scratch under the cap); the variants that load both bins would spill 470 - 680
  // bytes per lane (with scheduling fences every five values as well)
  constexpr int WOC = (MFFT_ROW_OCC_C2R && (WPC || (S::TPT <= 64 && 64 % S::TPT == 0))) ? row_occ_wgs<S, T>(c2r_threads<S, T>()) : 0;
  // the 20-values-per-thread plans in double precision: their column-limited / chunked c2r kernels come out at 256 VGPRs +
  // 10 - 40 AGPRs = one wave per SIMD; capped for two: 1000^3 2/3-rule c2r stage 5.6 -> 4.7 ms (the plain kernel, which reads
  // half as much again, takes 3.3)
  constexpr int WOV = (MFFT_ROW_OCC_C2R && sizeof(T) == 8 && S::E == 20 && S::N >= 160 && c2r_threads<S, T>() <= 256)
                          ? 512 / c2r_threads<S, T>() : WOC;
  constexpr bool RT = row_twlds<S, T>();
  constexpr bool SP = row_split<S, T>();
  constexpr int RC = c2r_rows<S, T>();            // the c2r kernels may differ
  constexpr int RR = r2c_rows<S, T>();            // ... and the r2c kernels (wave-packed layout)
  constexpr int WOR = row_occ_wgs<S, T>(r2c_threads<S, T>());
  constexpr int WOR5 = row_occ_r5<S, T, 1>(r2c_threads<S, T>()) ? row_occ_r5<S, T, 1>(r2c_threads<S, T>()) : WOR;
  constexpr int WOC5 = row_occ_r5<S, T, 2>(c2r_threads<S, T>());       // plain c2r only
  constexpr bool SC = row_split<S, T, true>();
  constexpr bool RTC = row_twlds<S, T, true>();
  reg.push_back(make_entry<RowFft<S, T, R, false, RT, false, SP>, RowParams<T>, S, T, WO5>(FAM_ROW, S::N, 0, R, name));
  reg.push_back(make_entry<RowFft<S, T, R, true, RT, false, SP>, RowParams<T>, S, T, WO5>(FAM_ROW, S::N, 1, R, name));
  reg.push_back(make_entry<R2CFft<S, T, RR, RT, false, false, SP, WPR>, RealParams<T>, S, T, WOR5>(FAM_R2C, 2 * S::N, 0, RR, name));
  reg.push_back(make_entry<C2RFft<S, T, RC, RTC, false, false, SC, WPC>, RealParams<T>, S, T, (WOC5 ? WOC5 : WOC)>(FAM_C2R, 2 * S::N, 1, RC, name));
  // pencil decompositions: the z-chunk pack / unpack fused into the stores / loads (pad = 4)
  reg.push_back(make_entry<RowFft<S, T, R, false, RT, true, SP>, RowParams<T>, S, T, WO>(FAM_ROW, S::N, 0, R, name));
  reg.back().pad = 4;
  reg.push_back(make_entry<RowFft<S, T, R, true, RT, true, SP>, RowParams<T>, S, T, WO>(FAM_ROW, S::N, 1, R, name));
  reg.back().pad = 4;
  reg.push_back(make_entry<R2CFft<S, T, RR, RT, false, true, SP, WPR>, RealParams<T>, S, T, WOR>(FAM_R2C, 2 * S::N, 0, RR, name));
  reg.back().pad = 4;
  reg.push_back(make_entry<C2RFft<S, T, RC, RTC, false, true, SC, WPC>, RealParams<T>, S, T, WOV>(FAM_C2R, 2 * S::N, 1, RC, name));
  reg.back().pad = 4;
  if constexpr (S::N % 3 == 0 && S::N >= 6) {   // 3/2-rule lengths: column-limited real transforms (pad = 3)
    reg.push_back(make_entry<R2CFft<S, T, RR, RT, true, false, SP, WPR>, RealParams<T>, S, T, WOR>(FAM_R2C, 2 * S::N, 0, RR, name));
    reg.back().pad = 3;
    // ... and with the kept columns split into the z chunks of the pencils' exchange (pad = 7): the 3/2-rule pencil
    // transforms write / read the exchange blocks themselves (pencil.py:511-632, 758-883 do it in the MPI datatypes)
    reg.push_back(make_entry<R2CFft<S, T, RR, RT, true, true, SP, WPR>, RealParams<T>, S, T, WOR>(FAM_R2C, 2 * S::N, 0, RR, name));
    reg.back().pad = 7;
    reg.push_back(make_entry<C2RFft<S, T, RC, RTC, true, true, SC, WPC>, RealParams<T>, S, T, WOV>(FAM_C2R, 2 * S::N, 1, RC, name));
    reg.back().pad = 7;
  }
  if constexpr (S::N >= 4) {                    // column-limited c2r: 3/2-rule lengths and the pruned 2/3-rule (any length)
    reg.push_back(make_entry<C2RFft<S, T, RC, RTC, true, false, SC, WPC>, RealParams<T>, S, T, WOV>(FAM_C2R, 2 * S::N, 1, RC, name));
    reg.back().pad = 3;
  }
  // Round 6: c2r kernels that no wave shuffle serves, with the mirrors through LDS instead of a second load (C2RFft MLDS;
  // KernelEntry::nt = 1; core.hip launch_real takes them where they exist, MFFT_C2R_MLDS=0: never)
  if constexpr (c2r_mlds_candidate<S, T>()) {
    reg.push_back(make_entry<C2RFft<S, T, RC, RTC, false, false, SC, false, true>, RealParams<T>, S, T, WOC>(FAM_C2R, 2 * S::N, 1, RC, name));
    reg.back().nt = 1;
    reg.push_back(make_entry<C2RFft<S, T, RC, RTC, false, true, SC, false, true>, RealParams<T>, S, T, WOV>(FAM_C2R, 2 * S::N, 1, RC, name));
    reg.back().nt = 1;
    reg.back().pad = 4;
    reg.push_back(make_entry<C2RFft<S, T, RC, RTC, true, false, SC, false, true>, RealParams<T>, S, T, WOV>(FAM_C2R, 2 * S::N, 1, RC, name));
    reg.back().nt = 1;
    reg.back().pad = 3;
    if constexpr (S::N % 3 == 0 && S::N >= 6) {
      reg.push_back(make_entry<C2RFft<S, T, RC, RTC, true, true, SC, false, true>, RealParams<T>, S, T, WOV>(FAM_C2R, 2 * S::N, 1, RC, name));
      reg.back().nt = 1;
      reg.back().pad = 7;
    }
  }
}

// chirp-z variants (fft_chirpz.h): every plan of length >= 16 also serves as the convolution length M
// of the arbitrary-length kernels
// chirp-z tiles: a 1024-thread workgroup is capped at 128 VGPRs and the two chained transforms then spill
// (M = 2048: 160 bytes of scratch per lane); there a half-width (64-byte) tile with 512 threads and whole-complex
// exchanges is faster (1000^3 fp64: 10.3 -> 8.2 ms per pass).  With 512 threads or fewer the 128-byte tile with
// split exchanges stays ahead (720^3, M = 1536: 3.5 vs 4.5 ms).
template <class S, typename T> constexpr int colz_cols() {
  // (the wide tiles of round 5's col_wide_small are a measurement of the radix kernels: the chirp-z kernels keep theirs)
  constexpr int c = col_wide_small<S, T>() > 0 ? 128 / (int)sizeof(cx<T>) : col_cols<S, T>(), v = col_vec<S, T>();
  return (col_split<S, T>() && S::TPT * (c / v) > 512 && c / 2 >= v && c / 2 >= 64 / (int)sizeof(cx<T>) &&
          (long long)S::N * (c / 2) * (int)sizeof(cx<T>) <= 131072 && S::TPT * (c / 2 / v) >= 64) ? c / 2 : c;
}
template <class S, typename T> constexpr bool colz_split() {
  return S::NP > 1 && (long long)S::N * colz_cols<S, T>() * (int)sizeof(cx<T>) > 131072;
}
template <class S, typename T>
void register_col_z(const char* name) {
  if constexpr (S::N >= 16) {
    auto& reg = kernel_registry();
    constexpr int C = colz_cols<S, T>();
    constexpr bool CS = colz_split<S, T>();
    constexpr int CV = col_vec<S, T>();
    reg.push_back(make_entry<ColFftZ<S, T, C, false, CS, CV>, ColParamsZ<T>, S, T>(FAM_COLZ, S::N, 0, C, name));
    reg.push_back(make_entry<ColFftZ<S, T, C, true, CS, CV>, ColParamsZ<T>, S, T>(FAM_COLZ, S::N, 1, C, name));
  }
}
template <class S, typename T>
void register_rows_z(const char* name) {
  if constexpr (S::N >= 16) {
    auto& reg = kernel_registry();
    constexpr int R = row_rows_n<S, T, false>();     // the chirp-z kernels exchange whole complex values
    reg.push_back(make_entry<RowFftZ<S, T, R, 0, false>, RowParamsZ<T>, S, T>(FAM_ROWZ, S::N, 0, R, name));
    reg.push_back(make_entry<RowFftZ<S, T, R, 0, true>, RowParamsZ<T>, S, T>(FAM_ROWZ, S::N, 1, R, name));
    reg.push_back(make_entry<RowFftZ<S, T, R, 1, false>, RealParamsZ<T>, S, T>(FAM_R2CZ, S::N, 0, R, name));
    reg.push_back(make_entry<RowFftZ<S, T, R, 2, true>, RealParamsZ<T>, S, T>(FAM_C2RZ, S::N, 1, R, name));
    reg.push_back(make_entry<RowFftZ<S, T, R, 3, false>, RealParamsZ<T>, S, T>(FAM_R2CZH, S::N, 0, R, name));
    reg.push_back(make_entry<RowFftZ<S, T, R, 4, true>, RealParamsZ<T>, S, T>(FAM_C2RZH, S::N, 1, R, name));
  }
}

// a plan of the main list: strided kernels always, row kernels unless the length has an override
// (the 30-values-per-thread plans of the lengths divisible by 15 are not offered as chirp-z convolution lengths: every
// length they could serve has a cheaper 2-, 3- or 5-smooth neighbour, and each plan costs ~40 more kernels)
template <class S, typename T>
void register_plan(const char* name) {
  constexpr bool chirp = S::N % 15 != 0 && S::N % 7 != 0;      // (nor the 28-values-per-thread plans of 7 * 2^a)
  if constexpr (!mfft_has_col_override<T>(S::N)) {
    register_col<S, T>(name);
    register_col_ytile<S, T>(name);
    if constexpr (chirp) register_col_z<S, T>(name);
  }
  if constexpr (!mfft_has_row_override_t<T>(S::N)) {
    register_rows<S, T>(name);
    if constexpr (chirp) register_rows_z<S, T>(name);
  }
}
template <class S, typename T>
void register_colplan(const char* name) {
  register_col<S, T>(name);
  register_col_ytile<S, T>(name);
  register_col_z<S, T>(name);
}
template <class S, typename T>
void register_rowplan(const char* name) {
  register_rows<S, T>(name);
  register_rows_z<S, T>(name);
}

struct PlanRegistrar {
  template <class F> explicit PlanRegistrar(F f) { f(); }
};

}  // namespace mfft
