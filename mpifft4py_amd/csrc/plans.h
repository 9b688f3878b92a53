// plans.h -- the compile-time radix plans (length, radix sequence) that get
// instantiated as gfx950 kernels.  First pass first; the first pass needs no
// twiddles, so the largest radix goes there.  E = lcm(radices) values per
// thread, TPT = N/E threads per transform.
//
// Lengths covered: 2^a (2..4096), 3*2^a (6..3072), 5*2^a (10..2560): these are
// what the reference's power-of-two meshes and their 3/2-rule padded
// counterparts (slab.py:75-76, 487-489) produce; 9*2^a (18..2304) are the 3/2-rule
// images of the 3*2^a meshes, 27*2^a (54..3456, group T) theirs in turn, 25*2^a (50..1600) and 125*2^a (250..2000: 1000^3 is a mesh people run) round off the
// 5-smooth sizes; groups L and M (below) the lengths with both 3 and 5 among their factors.  Every
// other length goes through the chirp-z kernels (fft_chirpz.h).  The groups only exist
// so the instantiations can be compiled in parallel translation units.
#pragma once

#define MFFT_PLANS_A(X) X(2, 2) X(4, 4) X(8, 8) X(16, 16) X(32, 8, 4) X(64, 8, 8) X(128, 16, 8) X(256, 16, 16)
#define MFFT_PLANS_B(X) X(512, 8, 8, 8) X(1024, 16, 8, 8)
#define MFFT_PLANS_C(X) X(2048, 16, 16, 8) X(4096, 16, 16, 16)
#define MFFT_PLANS_D(X) X(6, 6) X(12, 12) X(24, 24) X(48, 12, 4) X(96, 24, 4) X(192, 24, 8)
#define MFFT_PLANS_E(X) X(384, 24, 8, 2) X(768, 24, 8, 4) X(1536, 8, 8, 8, 3) X(3072, 8, 8, 4, 4, 3)
#define MFFT_PLANS_F(X) X(10, 10) X(20, 20) X(40, 20, 2) X(80, 20, 4) X(160, 40, 4)
#define MFFT_PLANS_G(X) X(320, 40, 8) X(640, 8, 4, 4, 5) X(1280, 8, 8, 4, 5) X(2560, 8, 8, 8, 5)
#define MFFT_PLANS_H(X) X(18, 6, 3) X(36, 12, 3) X(72, 24, 3) X(144, 24, 6) X(288, 24, 12) X(576, 24, 24)
#define MFFT_PLANS_I(X) X(1152, 24, 24, 2) X(2304, 24, 24, 4)
#define MFFT_PLANS_J(X) X(50, 10, 5) X(100, 20, 5) X(200, 20, 10) X(400, 20, 20) X(800, 20, 20, 2) X(1600, 20, 20, 4)
#define MFFT_PLANS_K(X) X(250, 10, 5, 5) X(500, 20, 5, 5) X(1000, 20, 10, 5) X(2000, 20, 20, 5)
// Round 3: the 9-, 25- and 125-smooth groups (H - K) and the row plans of groups G - I use the prime-factor butterflies of
// radix 12 = 3x4, 20 = 5x4, 24 = 3x8 (fft_core.h BflyPFA) so that, with the same number of values per thread, a
// transform takes 2 - 4 passes instead of 4 - 6 (profiles/r03_composite_radix.txt: 1000^3 fp32 pair 14.8 -> 11.3 ms,
// 1600^3 fp64 103.5 -> 87.4, 800^3 fp64 11.6 -> 10.1, 576^3 fp64 3.85 -> 3.5).  The 3- and 5-smooth groups D - G use them
// (24 = 3x8, 40 = 5x8) up to 768 / 320 and in their row plans: worth 3 - 8 % at 192 ... 768, nothing measurable
// beyond (640 / 1280 / 2560 / 1536 / 3072 strided keep the radix-8 sequences).
// Round 3: lengths with BOTH 3 and 5 among their factors (15 * 2^a, 45 * 2^a, 75 * 2^a, 225 * 2^a: 720, 900, 960, 1200 ... are meshes
// people run, and went through chirp-z at 0.12 - 0.22 of the roofline).  E = lcm(radices) must contain 15, so these plans
// hold 30 values per thread and radix 4 is out (it would make it 60); the prime-factor butterflies of radix 6, 10, 15, 30
// (fft_core.h BflyPFA: no twiddles inside, index maps are register renamings) keep the pass count at 1 - 6.
// With radices 5, 3, 2 only (6 - 8 passes) the same lengths ran 8 - 30 % slower: 720^3 pair 10.5 -> 8.4 ms, 1200^3 58.8 -> 45.1.
#define MFFT_PLANS_L(X) X(30, 30) X(60, 10, 6) X(90, 15, 6) X(120, 10, 6, 2) X(150, 15, 10) X(180, 10, 6, 3) \
  X(240, 10, 6, 2, 2) X(300, 10, 10, 3) X(360, 10, 6, 6) X(450, 15, 10, 3) X(480, 10, 6, 2, 2, 2)
#define MFFT_PLANS_M(X) X(600, 10, 10, 6) X(720, 10, 6, 6, 2) X(900, 15, 10, 6) X(960, 10, 6, 2, 2, 2, 2) \
  X(1200, 10, 10, 6, 2) X(1440, 10, 6, 6, 2, 2) X(1800, 15, 10, 6, 2)
// Group N: the 3/2-rule images of the longest 5-, 25- and 125-smooth meshes (1280 -> 1920, 2560 -> 3840, 1600 -> 2400,
// 500 / 1000 / 2000 -> 750 / 1500 / 3000): chirp-z (1920, 1500, 750) or no kernel at all (beyond 2048) before.
#define MFFT_PLANS_N(X) X(750, 15, 10, 5) X(1500, 15, 10, 10) X(1920, 10, 6, 2, 2, 2, 2, 2) X(2400, 10, 10, 6, 2, 2) \
  X(3000, 10, 10, 10, 3) X(3840, 10, 6, 2, 2, 2, 2, 2, 2)
// Group O (round 4): 7 * 2^a -- 896, 1792, 3584 are ordinary meshes (numpy / FFTW take every n: numpy_fft.py:25-46) and went
// through chirp-z (896) or had no kernel at all.  28 values per thread (radix 28 = the prime-factor butterfly 7 x 4, no
// inner twiddles), radix-4 / 2 passes after it.
#define MFFT_PLANS_O(X) X(14, 14) X(28, 28) X(56, 28, 2) X(112, 28, 4) X(224, 28, 4, 2) X(448, 28, 4, 4) \
  X(896, 28, 4, 4, 2) X(1792, 28, 4, 4, 4) X(3584, 28, 4, 4, 4, 2)
// Group P (round 4): 8192, first of all as the convolution length M of the chirp-z kernels, which makes EVERY length up to
// 4096 a supported one (2 n - 1 <= 8192), as it is for numpy / FFTW (numpy_fft.py:25-46); real rows up to 16384.
#define MFFT_PLANS_P(X) X(8192, 32, 16, 16)
// Group Q (round 5): the 3-, 5-, 7- and 9-smooth lengths between 4096 and 8192 -- 4608, 5120, 6144, 7168 are ordinary FFTW /
// numpy sizes (numpy_fft.py:25-46 takes every n) and were MFFT_ERR_UNSUPPORTED as complex lengths (their real rows already
// worked through the half-length plans); 192 / 128 / 256 / 256 threads per transform, strided tiles of 2 - 4 columns.
#define MFFT_PLANS_Q(X) X(4608, 24, 24, 8) X(5120, 40, 8, 4, 4) X(6144, 24, 8, 8, 4) X(7168, 28, 4, 4, 4, 4)
// Group R (round 5): 21 * 2^a -- 672, 1344, 2688 are 7-smooth meshes FFTW users pick (numpy_fft.py:25-46 takes every n) and ran
// through the one-workgroup chirp-z kernels at ~0.17 of the roofline.  E = lcm(radices) must contain 3 and 7, and a radix-4
// pass would make it 84: 42 values per thread (84 VGPRs of data in single precision, 168 in double: one wave per SIMD there),
// the prime-factor butterfly 42 = 6 x 7 first (no twiddles inside), radix-2 passes for the rest.  (35 * 2^a would need 70.)
#define MFFT_PLANS_R(X) X(42, 42) X(84, 42, 2) X(168, 42, 2, 2) X(336, 42, 2, 2, 2) X(672, 42, 2, 2, 2, 2) \
  X(1344, 42, 2, 2, 2, 2, 2) X(2688, 42, 2, 2, 2, 2, 2, 2)

// Group S (round 6): 35 * 2^a in SINGLE precision only -- 560, 1120, 2240 are 7-smooth meshes that ran through the one-workgroup
// chirp-z kernels at 0.11 - 0.19 of the roofline (profiles/r06_any_n_sweep.txt).  E = lcm(radices) must contain 5 and 7 and a
// radix-4 pass would make it 140: 70 values per thread (140 VGPRs of data in single precision; double precision would need
// 280 and keeps chirp-z), the prime-factor butterfly 70 = 7 x 10 first (no twiddles inside), radix-2 passes for the rest.
#define MFFT_PLANS_S(X) X(70, 70) X(140, 70, 2) X(280, 70, 2, 2) X(560, 70, 2, 2, 2) X(1120, 70, 2, 2, 2, 2) \
  X(2240, 70, 2, 2, 2, 2, 2)
#define MFFT_ROWPLANS_S(X)

// Group T (round 6): 27 * 2^a -- the 3/2-rule images of the 9 * 2^a meshes (288 -> 432, 576 -> 864, 1152 -> 1728, 2304 -> 3456:
// slab.py:75-76, 487-489), which had no radix plan, so a 3/2-rule transform of 576^3 or 1152^3 had no fused pad / truncate passes
// (MFFT_ERR_UNSUPPORTED) and the plain transforms of these lengths ran through chirp-z.  24 values per thread like 576 ... 2304
// for the strided kernels, 12 for the contiguous-axis kernels (MFFT_ROWPLANS_T) and the fused nonlinear z stage.
#define MFFT_PLANS_T(X) X(54, 6, 3, 3) X(108, 12, 3, 3) X(216, 24, 3, 3) X(432, 24, 6, 3) X(864, 24, 12, 3) X(1728, 24, 24, 3) \
  X(3456, 24, 24, 6)
#define MFFT_ROWPLANS_T(X) X(216, 12, 6, 3) X(432, 12, 12, 3) X(864, 12, 12, 6) X(1728, 12, 12, 12) X(3456, 12, 12, 12, 2)
// Groups U and V (round 6): the remaining 3/2-rule images of planned meshes -- 135 * 2^a (720 -> 1080, 1440 -> 2160, 360 -> 540), 1350 / 2700 /
// 2250 (of 900 / 1800 / 1500) and the odd halves their real axes need (675, 1125: 15 values per thread), all with the 30-values prime-factor
// butterfly first like groups L - N (also the odd halves 75 / 135 / 225 / 375 of the real axes of 100 / 180 / 300 / 500 and 2880 / 3600, the
// images of 1920 / 2400); 81 * 2^a (432 -> 648, 864 -> 1296, 1728 -> 2592) with 12 values per thread.  Before, a 3/2-rule
// transform of those meshes took the copy-based route around chirp-z transforms (as the 9 * 2^a meshes did before group T).
#define MFFT_PLANS_U(X) X(270, 30, 3, 3) X(540, 30, 6, 3) X(1080, 30, 6, 6) X(2160, 30, 6, 6, 2) X(1350, 30, 15, 3) X(2700, 30, 30, 3) \
  X(2250, 30, 15, 5) X(675, 15, 15, 3) X(1125, 15, 15, 5) \
  X(75, 15, 5) X(135, 15, 3, 3) X(225, 15, 15) X(375, 15, 5, 5) X(2880, 30, 6, 2, 2, 2, 2) X(3600, 30, 30, 2, 2)
#define MFFT_ROWPLANS_U(X)
#define MFFT_PLANS_V(X) X(162, 6, 3, 3, 3) X(324, 12, 3, 3, 3) X(648, 12, 6, 3, 3) X(1296, 12, 12, 3, 3) X(2592, 12, 12, 6, 3)
#define MFFT_ROWPLANS_V(X)
// Group W (round 6): 63 * 2^a -- the 3/2-rule images of the 21 * 2^a meshes (168 -> 252, 336 -> 504, 672 -> 1008, 1344 -> 2016): the 42-values
// butterfly of group R first, a radix-6 pass, radix-2 passes.  (The images of 35 * 2^a would need 210 values per thread: chirp-z.)
#define MFFT_PLANS_W(X) X(126, 42, 3) X(252, 42, 6) X(504, 42, 6, 2) X(1008, 42, 6, 2, 2) X(2016, 42, 6, 2, 2, 2)
#define MFFT_ROWPLANS_W(X)

// Row-family overrides (RowFft / R2CFft / C2RFft of complex length N): along the contiguous
// axis a transform's LDS exchange buffer is private, so large E (few threads per row) starves
// the CU of waves.  For the 3- and 5-smooth lengths >= 96 the row kernels therefore use radix
// 4/2 passes around one radix-3/5 pass (E = 12 / 20) instead of the strided kernels' E = 24 / 40.
// A length listed here is NOT given row kernels by its MFFT_PLANS_* entry.
#define MFFT_ROWPLANS_D(X) X(96, 12, 4, 2) X(192, 12, 4, 4)
#define MFFT_ROWPLANS_E(X) X(384, 12, 4, 4, 2) X(768, 12, 4, 4, 4) X(1536, 12, 4, 4, 4, 2) X(3072, 12, 4, 4, 4, 4)
#define MFFT_ROWPLANS_F(X) X(160, 20, 4, 2)
#define MFFT_ROWPLANS_G(X) X(320, 20, 4, 4) X(640, 20, 4, 4, 2) X(1280, 20, 4, 4, 4) X(2560, 20, 4, 4, 4, 2)
#define MFFT_ROWPLANS_H(X) X(144, 12, 12) X(288, 12, 12, 2) X(576, 12, 12, 4)
#define MFFT_ROWPLANS_I(X) X(1152, 12, 12, 4, 2) X(2304, 12, 12, 4, 4)
#define MFFT_ROWPLANS_J(X)
#define MFFT_ROWPLANS_K(X)
#define MFFT_ROWPLANS_L(X)
#define MFFT_ROWPLANS_M(X)
#define MFFT_ROWPLANS_N(X)
#define MFFT_ROWPLANS_O(X)
#define MFFT_ROWPLANS_P(X)
#define MFFT_ROWPLANS_Q(X)
#define MFFT_ROWPLANS_R(X)
#define MFFT_ROWPLANS_A(X)
#define MFFT_ROWPLANS_B(X)
#define MFFT_ROWPLANS_C(X)
#define MFFT_FOR_EACH_ROWPLAN(X) \
  MFFT_ROWPLANS_D(X) MFFT_ROWPLANS_E(X) MFFT_ROWPLANS_F(X) MFFT_ROWPLANS_G(X) MFFT_ROWPLANS_H(X) MFFT_ROWPLANS_I(X) \
  MFFT_ROWPLANS_T(X)

// Double precision only: the contiguous-axis kernels of 500 keep 5x5x5x4 -- with 20x5x5 the c2r kernel of real length 1000
// doubled its time (1000^3 bwd_z 3.3 -> 6.6 ms) while single precision gains from it (2.03 -> 1.88 ms).
#define MFFT_ROWPLANS_F64_K(X) X(500, 5, 5, 5, 4)
// true if complex length n takes its row kernels from MFFT_ROWPLANS_*
constexpr bool mfft_has_row_override(int n) {
  return n == 96 || n == 192 || n == 384 || n == 768 || n == 1536 || n == 3072 || n == 160 || n == 320 || n == 640 ||
         n == 1280 || n == 2560 || n == 144 || n == 288 || n == 576 || n == 1152 || n == 2304 || n == 216 || n == 432 ||
         n == 864 || n == 1728 || n == 3456;
}

template <typename T> constexpr bool mfft_has_row_override_t(int n) { return mfft_has_row_override(n) || (sizeof(T) == 8 && n == 500); }

// Strided-kernel overrides for SINGLE precision: the longest lengths run with E = 32 plans, one column per lane,
// 1024 threads and 128-byte tiles (kbench2: 2048 2.7 -> 3.8 TB/s, 4096 2.0 -> 3.1 TB/s; the E = 16 plans would need
// two columns per lane at 1024 threads, which spills).  A length listed here gets its fp32 strided kernels from
// this list, everything else from its MFFT_PLANS_* entry.
// (round 5: 2048 is back on its 16x16x8 plan, on 64-byte tiles with two workgroups per CU: registry.h col_narrow_f32)
#define MFFT_COLPLANS_F32_C(X) X(4096, 32, 32, 4)
// Strided-kernel override for DOUBLE precision: 1024 runs as 8x8x4x4 (E = 8, 1024 threads, 60 VGPRs) with LDS twiddles and
// the split re/im exchange, i.e. 80 KB of LDS, so that TWO workgroups share a CU and one's loads and stores overlap the
// other's passes.  Interleaved A/B at 1024^3 (kbench3, profiles/r02_kbench3_variants.txt): y in place 3.52 -> 3.44 ms,
// x in place 3.57 -> 3.35 ms (with non-temporal accesses, which only pay in place at two workgroups per CU),
// x out of place 3.63 -> 3.19 ms.  The same change loses in single precision (1.85 -> 2.4 ms) and is neutral at 512.
// Round 3, 512 in double precision (BASELINE config 2): 4x4x4x4x2 (E = 4, 1024 threads, ~40 KB of LDS with LDS twiddles and the
// split exchange), two workgroups per CU = 2048 threads instead of 1024 (kbench3 `occ512`, profiles/r03_kbench3_occ512.txt):
// y in place 0.478 -> 0.422 ms, x in place / out of place 0.478 -> 0.456 ms; 8x8x8 with the split exchange at 4 workgroups per CU
// gets half of that (0.446 / 0.460).
#define MFFT_COLPLANS_F64_B(X) X(1024, 8, 8, 4, 4) X(512, 4, 4, 4, 4, 2)
// 384 and 1152 in double precision: 12 instead of 24 values per thread, i.e. twice the threads per CU (384: 3 workgroups of
// 256 instead of 128 threads; 1152: 768 threads and, with the register cap of registry.h col_wgs, two workgroups per CU
// instead of one).  kbench3 `small` / `occ` (profiles/r02_kbench3_long_lengths.txt), y in place / x in place / x out of
// place, ms:  384: 0.231 -> 0.184, 0.217 -> 0.186, 0.245 -> 0.191;  1152: 6.72 -> 5.64, 7.14 -> 5.45, 7.36 -> 5.44.
// The same exchange of plans is neutral at 576, 640, 1280 and loses at 1536 (1024 threads, one workgroup either way).
// 768 in double precision keeps 8x8x4x3 for the strided kernels: with 24x8x4 the inverse y pass of the 768^3 pair was 8 %
// slower in three sessions (1.40 -> 1.52 ms) while single precision gains 5 % overall from it.
#define MFFT_COLPLANS_F64_E(X) X(384, 12, 4, 4, 2) X(768, 8, 8, 4, 3)
#define MFFT_COLPLANS_F64_I(X) X(1152, 12, 12, 4, 2)
// 900 in double precision: the strided kernels keep the plain 5x5x3x3x2x2 sequence -- with the composite radices they
// came out 15 % slower (x / y passes of the 900^3 pair 2.5 / 2.25 -> 3.0 / 2.6 ms, the same for 10x10x3x3, 10x15x6 and
// 10x6x15), the only length of groups L and M where that happened; the contiguous-axis kernels gain from 15x10x6 like the rest.
#define MFFT_COLPLANS_F64_M(X) X(900, 5, 5, 3, 3, 2, 2)
// Round 4: lengths N = 3 L whose strided transforms (plain and 3/2-rule pad / truncate) ALSO exist as three length-L
// sub-transforms per workgroup (fft_col3.h ColFft3: a third of the exchange buffer): 1536, the 3/2-rule image of 1024.
// X(N, L, radices of the length-L sub-plan).  The sub-plan holds FOUR values per thread (12 for the three thirds: 48 VGPRs
// of data in double precision, 1024 threads, no spills under the 128-register cap); with eight (8 x 8 x 8: the 24 values of
// the ColFft plan) the kernel needs 140+ registers and two workgroups per CU are out of reach (-Rpass-analysis: 22 - 224
// registers spilled under any cap), and 3072 = 3 x 1024 would need 2048 threads.  The ColFft kernels of the same length
// stay: they are what double precision and the y passes run (core.hip launch_col has the measurements; MFFT_COL3=0 / 1).
#define MFFT_COL3PLANS_E(X) X(1536, 512, 4, 4, 4, 4, 2)
template <typename T> constexpr bool mfft_has_col_override(int n) {
  return (sizeof(T) == 4 && n == 4096) || (sizeof(T) == 8 && (n == 1024 || n == 512 || n == 384 || n == 768 || n == 1152 || n == 900));
}

// Round 6: complex plans of the fused nonlinear z stage (fft_nlz.h NlzFft: X(M, radices) with M the REAL length of a z row --
// two real rows ride on one complex transform of length M).  Powers of two (dealias None / 2/3-rule) with 8 values per
// thread and the 3 * 2^a images of the 3/2-rule with 12: the kernel parks 2 E complex + E real values per thread next to
// the working set of a transform, so few values per thread (many threads per row) is what keeps it at two waves per SIMD.
#define MFFT_NLZPLANS_P2(X) X(16, 8, 2) X(32, 8, 4) X(64, 8, 8) X(128, 8, 4, 4) X(256, 8, 8, 4) X(512, 8, 8, 8) \
  X(1024, 8, 8, 4, 4) X(2048, 8, 8, 8, 4) X(4096, 8, 8, 8, 8)
#define MFFT_NLZPLANS_3(X) X(12, 12) X(24, 12, 2) X(48, 12, 4) X(96, 12, 4, 2) X(192, 12, 4, 4) X(384, 12, 4, 4, 2) \
  X(768, 12, 4, 4, 4) X(1536, 12, 4, 4, 4, 2) X(3072, 12, 4, 4, 4, 4)
// the 9 * 2^a meshes (dealias None / 2/3-rule) and their 3/2-rule images 27 * 2^a
#define MFFT_NLZPLANS_9(X) X(144, 12, 12) X(288, 12, 12, 2) X(576, 12, 12, 4) X(1152, 12, 12, 4, 2) X(2304, 12, 12, 4, 4) \
  X(216, 12, 6, 3) X(432, 12, 12, 3) X(864, 12, 12, 6) X(1728, 12, 12, 12) X(3456, 12, 12, 12, 2) \
  X(324, 12, 3, 3, 3) X(648, 12, 6, 3, 3) X(1296, 12, 12, 3, 3) X(2592, 12, 12, 6, 3)

// ... and the sub-plans of its pruned 3/2-rule flavour (Nlz3Fft: X(L, radices) with L = N/2 = M/3, three sub-transforms of
// length L per row in three thread groups)
#define MFFT_NLZ3PLANS(X) X(4, 4) X(8, 8) X(16, 4, 4) X(32, 8, 4) X(64, 8, 8) X(128, 8, 4, 4) X(256, 4, 4, 4, 4) X(512, 4, 4, 4, 4, 2) \
  X(1024, 8, 8, 4, 4)
#define MFFT_FOR_EACH_PLAN(X)                                                                                     \
  MFFT_PLANS_A(X) MFFT_PLANS_B(X) MFFT_PLANS_C(X) MFFT_PLANS_D(X) MFFT_PLANS_E(X) MFFT_PLANS_F(X) MFFT_PLANS_G(X) \
  MFFT_PLANS_H(X) MFFT_PLANS_I(X) MFFT_PLANS_J(X) MFFT_PLANS_K(X) MFFT_PLANS_L(X) MFFT_PLANS_M(X) MFFT_PLANS_N(X) \
  MFFT_PLANS_O(X) MFFT_PLANS_P(X) MFFT_PLANS_Q(X) MFFT_PLANS_R(X) MFFT_PLANS_T(X) MFFT_PLANS_U(X) MFFT_PLANS_V(X) MFFT_PLANS_W(X)
