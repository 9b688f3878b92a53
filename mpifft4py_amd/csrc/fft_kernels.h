// fft_kernels.h -- workgroup-level FFT kernel bodies (HIP, gfx950).
//
// Four kernel families, all built on fft_core.h:
//   col_fft_body : batched c2c along a STRIDED axis.  A workgroup owns a tile of
//                  COLS adjacent (contiguous-in-memory) columns x N rows; lanes
//                  run fastest over the columns so every global access is a
//                  COLS*sizeof(complex) contiguous segment (128 B for fp64/COLS=8).
//                  Input and output rows use two-level addressing
//                  row r -> (r / split) * hi + (r % split) * lo, which is what lets
//                  the slab/pencil pack and unpack steps (reference slab.py:403,
//                  cython/maths.pyx:21-31, the Alltoallw sub-array types of
//                  slab.py:199-211 / pencil.py:218-246) be fused into the FFT's own
//                  loads/stores instead of being separate full-volume copies.
//   row_fft_body : batched c2c along the CONTIGUOUS axis (lanes along the row).
//   r2c_body     : real -> half-complex along the contiguous axis: length-N real
//                  row = length-N/2 complex FFT + split post-pass (reference:
//                  rfft / rfft2 / rfftn last-axis stage, numpy_fft.py:39-44,67-72).
//   c2r_body     : inverse of r2c (irfft; Im of the k=0 and k=N/2 bins ignored,
//                  as pocketfft/FFTW c2r do).
//
// The bodies are written once and compiled (a) by hipcc as __device__ code that
// the __global__ wrappers in kernels.hip call, and (b) by g++ for the fibre-based
// workgroup emulator (emu_test.cpp), where MFFT_BARRIER() yields to a scheduler.
#pragma once
#include "fft_core.h"

#if defined(__HIPCC__)
#define MFFT_D __device__ __forceinline__
#define MFFT_BARRIER() __syncthreads()
// (Rounds 3 - 4; since round 5 the cure sits where the fault is, fft_core.h pass_compute / MFFT_LAUNDER_MODE 4, and this one is
// compiled only with MFFT_LAUNDER_MODE=1.)
// hipcc (ROCm 7.2) miscompiles the contiguous-axis kernels of the 30-values-per-thread plans when the thread's index inside
// its transform has a known power-of-two range (j = tid % 8 or tid % 16: lengths 240 and 480 -- a third of the output bins
// wrong, the same bins for every radix order and every number of rows per workgroup, while the workgroup emulator and the
// strided kernels of the same plans, whose j = tid / columns has no such range, are right; tools/rowcheck.hip).  Passing j
// through an empty asm hides the range from the optimiser and the results are exact again.  Applied to those plans only: the
// other kernels are verified as compiled (tests/test_gpu_stages.py) and keep their code.
#if defined(__HIP_DEVICE_COMPILE__)
#define MFFT_OPAQUE(x) asm volatile("" : "+v"(x))
#else
#define MFFT_OPAQUE(x) ((void)0)
#endif
// (round 5: the 42-values plans of 21 * 2^a show the same fault -- 336: j = tid % 8, a tenth of the bins wrong on the device, exact in
// the emulator -- and get the same treatment)
// ORIGIN: hide j's range here, at its origin (rounds 3 - 4 for every contiguous-axis kernel; since round 5 the c2r kernels only --
// they come out with FEWER registers that way: dense c2r of 1500 / 1800 / 750 points in double precision 216 - 230 VGPRs against
// 258 + AGPRs, the wave-packed single-precision ones 162 against 200 - 290 -- while c2c rows and r2c keep j's range and hide it
// only from the twiddle index, run_passes below: r2c 2 -> 3 waves per SIMD in single precision, 100 - 236 -> 12 - 56 bytes of
// scratch under the double-precision cap).
template <class S, bool ORIGIN = false> MFFT_D int row_thread_index(int tid) {
  int j = tid % S::TPT;
  if constexpr (::mfft::launder_plan<S>()) {
#if MFFT_LAUNDER_MODE == 1
    MFFT_OPAQUE(j);
#elif MFFT_LAUNDER_MODE == 4
    if constexpr (ORIGIN) MFFT_OPAQUE(j);
#endif
  }
  return j;
}
#else
#define MFFT_OPAQUE(x) ((void)0)
template <class S, bool ORIGIN = false> inline int row_thread_index(int tid) { return tid % S::TPT; }
#define MFFT_D inline
namespace mfft { void emu_barrier(); }
#define MFFT_BARRIER() ::mfft::emu_barrier()
#endif

// Value of `x` in lane `src` of the caller's wave (64 lanes).  Device: one cross-lane read.  Host emulator: a round trip
// through a per-workgroup slot array between two barriers (emu_test.cpp emu_shfl) -- every thread of the workgroup must
// execute the same sequence of shuffles, which the kernels guarantee (they sit in fully unrolled, uniform loops) -- so
// that the lane arithmetic of the shuffle paths is checked on the CPU as well.
// WAVE_SHFL_DIRECT: the same on the device WITHOUT the function in between.  Through the `__forceinline__` template wrapper the
// shuffles of the dense c2r kernels compiled to 13 - 45 more VGPRs for the 28-values plans -- 263 / 287 instead of 250 / 242,
// i.e. one wave per SIMD instead of two and twice the time (896^3 bwd_z 2.05 -> 4.2 ms from round 4's commit d98a6c4 on; found
// in round 5 by reverting that commit's hunks one at a time, profiles/r05_radix7_c2r_bisect.txt) -- while the wave-packed
// kernels, which were tuned with the wrapper, lose registers the other way round (fp32 c2r of 1200 / 1500 / 1800: 163 -> 200
// VGPRs with the direct form), and so does one dense 20-values plan (c2r of 640: 248 -> 170 VGPRs and 11 % slower).  So: the
// dense c2r kernels of the 28-values plans call the builtin directly, everything else keeps the wrapper;
// scripts/kernel_regs.py --diff shows such moves before they reach a GPU.
#if defined(__HIPCC__)
#define WAVE_SHFL_DIRECT(x, src) __shfl((x), (src), 64)
template <typename T> __device__ __forceinline__ T wave_shfl(T x, int src) { return __shfl(x, src, 64); }
#else
#define WAVE_SHFL_DIRECT(x, src) wave_shfl((x), (src))
namespace mfft { double emu_shfl(double x, int src_lane); }
template <typename T> inline T wave_shfl(T x, int src) { return (T)::mfft::emu_shfl((double)x, src); }
#endif

// nothing is scheduled across it (device code only)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MFFT_NO_R2C_FENCE)
#define MFFT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MFFT_SCHED_FENCE() ((void)0)
#endif

namespace mfft {

typedef long long i64;

// exact r / d for 0 <= r < 2^16, 1 <= d < 2^16 with m = floor(2^32 / d) + 1
MFFT_HD unsigned fastdiv(unsigned r, unsigned m) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(r, m);
#else
  return (unsigned)(((unsigned long long)r * m) >> 32);
#endif
}

// two-level row addressing
struct RowMap {
  i64 hi, lo;        // element strides
  unsigned split;    // rows per group (>= n means "no split": offset = r*lo)
  unsigned magic;    // floor(2^32/split)+1
};
// rows 0..n-1; split >= n (or 0) means plain stride `lo`
inline RowMap make_rowmap(i64 hi, i64 lo, i64 split, i64 n) {
  RowMap m;
  if (split <= 0 || split >= n) {          // single group
    m.hi = 0; m.lo = lo; m.split = 0x7FFFFFFFu; m.magic = 1u;   // q = 0 for r < 2^31
  } else if (split == 1) {                 // every row its own group
    m.hi = 0; m.lo = hi; m.split = 0x7FFFFFFFu; m.magic = 1u;
  } else {
    m.hi = hi; m.lo = lo; m.split = (unsigned)split;
    m.magic = (unsigned)(0x100000000ull / (unsigned long long)split + 1ull);
  }
  return m;
}
MFFT_HD i64 row_off(const RowMap& m, unsigned r) {
  unsigned q = fastdiv(r, m.magic);
  unsigned rem = r - q * m.split;
  return (i64)q * m.hi + (i64)rem * m.lo;
}

// Complex side of a contiguous-axis kernel split into z chunks (pencil decompositions): the pack / unpack copies
// around the z-splitting exchange -- the Alltoallw sub-array types of the reference, pencil.py:218-246, 971-999 --
// fused into the kernel's own stores / loads.  Position k of row r lives in block l = min(k / q, nchunk - 1), a
// (rows_total, len_l) array with len_l = q except for the last block (last_len: q + 1 with the Nyquist column).
struct ZSplit {
  unsigned q, magic;     // chunk length, floor(2^32 / q) + 1
  int nchunk;            // 0: plain rows
  int last_len;
  i64 block;             // elements between the starts of consecutive blocks: rows_total * pitch
  int pitch, last_pitch; // elements between consecutive rows of a block (>= its chunk length; round 4: rows of 129 / 257
                         // elements start on cache lines for the strided pass that reads them next)
};
inline ZSplit make_zsplit(i64 q, int nchunk, i64 last_len, i64 rows_total, i64 pitch = 0, i64 last_pitch = 0) {
  ZSplit z;
  z.q = (unsigned)q;
  z.magic = (unsigned)(0x100000000ull / (unsigned long long)q + 1ull);
  z.nchunk = nchunk;
  z.last_len = (int)last_len;
  z.pitch = (int)(pitch > 0 ? pitch : q);
  z.last_pitch = (int)(last_pitch > 0 ? last_pitch : last_len);
  z.block = rows_total * (i64)z.pitch;
  return z;
}
// element offset of (row, k); *ok = the position exists (the last chunk may drop the Nyquist column)
MFFT_HD i64 zsplit_off(const ZSplit& z, i64 row, int k, bool* ok) {
  int l = (int)fastdiv((unsigned)k, z.magic);
  if (l >= z.nchunk) l = z.nchunk - 1;
  const int kk = k - l * (int)z.q;
  const bool last = l == z.nchunk - 1;
  *ok = kk < (last ? z.last_len : (int)z.q);
  return (i64)l * z.block + row * (i64)(last ? z.last_pitch : z.pitch) + kk;
}

template <typename T>
struct ColParams {
  const cx<T>* in;
  cx<T>* out;
  const cx<T>* tw;       // Spec::TW inter-pass twiddles (device memory)
  i64 in_outer, out_outer;
  RowMap in_map, out_map;
  int ncols;             // contiguous columns per outer batch
  int ntile_c;           // ceil(ncols / COLS)
  int nouter;
  int remap;             // 1: XCD-aware block -> tile mapping, 2: the same with skewed entry points (power-of-two tile ranges)
  int fold;              // PAD == 2: 1 = add the Nyquist row N/3 into row 2N/3 before it is stored
  T scale;
  int nblocks;           // persistent experiment (tools/fft_persist_experiment.h): workgroups launched; unused by ColFft
  const unsigned char* mask;   // PAD == 3: one byte per element of `in` (same offsets); 0 = the element reads as zero
  // PAD == 4 (pruned 2/3-rule pass): rows [b_row_lo, b_row_hi) read as zero without being loaded; column j of outer
  // batch o has z = (b_coff + j) % b_cper and y = b_goff + (b_coff + j) / b_cper + o * b_gstep and is KEPT iff
  // z < b_clim and (y < b_glo or y >= b_ghi); tiles without a kept column do nothing at all
  int b_row_lo, b_row_hi, b_coff, b_cper, b_clim, b_goff, b_gstep, b_glo, b_ghi;
  const int* tile_list;  // PAD == 4: the tiles that hold a kept column (the launch has one workgroup per entry); null: all tiles
  int ntiles_listed;
  int b_gzero;           // PAD == 4: 1 = columns of a removed y are not skipped but read as zeros and WRITTEN (the transform of
                         // zeros): for outputs that somebody reads whole, e.g. a chunk that goes through an exchange;
                         // 2 = the same for the columns of a removed z too, and no tile is skipped: the output is complete
                         // (the pencils' first inverse pass: the band parameters instead of one mask byte per element)
  // Round 5, wrapped INPUT columns: the tiles follow the (compact) output, whose ncols columns are contiguous, while the
  // input keeps in_wrap columns per row of a pitched buffer: column c is read from c + (c / in_wrap) * in_wrap_gap.  The pass
  // that leaves a line-aligned intermediate (rows of 520 instead of 513 bins) for the caller's compact spectrum stores
  // whole cache lines this way (its loads straddle lines instead, which costs little).  0: plain columns.
  int in_wrap = 0, in_wrap_gap = 0;
};

template <typename T>
struct RowParams {
  const cx<T>* in;
  cx<T>* out;
  const cx<T>* tw;
  i64 in_stride, out_stride;   // row strides in complex elements
  i64 nrows;
  T scale;
  ZSplit zs;                   // CHUNK kernels: the chunked side (output of a forward, input of an inverse transform)
  i64 row0;                    // first row of this launch inside the chunk blocks
};

template <typename T>
struct RealParams {            // r2c: in = real rows, out = complex rows; c2r the reverse
  const void* in;
  void* out;
  const cx<T>* tw;             // twiddles of the length-N/2 complex transform
  const cx<T>* rtw;            // exp(-2 pi i k / N), k = 0..N/2-1
  i64 in_stride, out_stride;   // row strides in elements of the respective type
  i64 nrows;
  int valid;                   // complex columns that exist in memory (N/2+1 normally; fewer for the
                               // 3/2-rule: r2c stores only the first `valid`, c2r reads the rest as 0)
  T scale;
  ZSplit zs;                   // CHUNK kernels: the complex side is split into z chunks
  i64 row0;                    // first row of this launch inside the chunk blocks
};

// position -> padded LDS slot for row-major (lane-along-row) exchange buffers:
// one pad element every PD positions breaks the stride-R0 bank aliasing of the
// first autosort scatter.
template <int PD> MFFT_HD int padpos(int pos) {
  if constexpr (PD > 0) return pos + pos / PD;
  else return pos;
}
template <int N, int PD> constexpr int padded_len() { return PD > 0 ? N + N / PD + 1 : N; }
template <int PD> struct PadSlot {
  MFFT_D int operator()(int pos) const { return padpos<PD>(pos); }
};

// ---------------------------------------------------------------------------
// LDS exchange policies.  slot(pos) maps a transform position to an element
// index of the exchange buffer (column-interleaved or padded row layouts).
//   XchFull  : whole complex values, buffer of N complex per transform
//   XchSplit : real parts, then imaginary parts, through a buffer of N reals
//              per transform (half the LDS, two more barriers per exchange)
// ---------------------------------------------------------------------------
template <typename T, class Slot>
struct XchFull {
  cx<T>* buf;
  Slot slot;
  template <class S, int P>
  MFFT_D void exchange(cx<T> (&v)[S::E], int j, bool pre_barrier) {
    if (pre_barrier) MFFT_BARRIER();              // everyone finished the previous gather
    pass_scatter<S, P>(j, [&](int pos, int reg) { buf[slot(pos)] = v[reg]; });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) { v[reg] = buf[slot(pos)]; });
  }
  MFFT_D void put(int pos, cx<T> val) { buf[slot(pos)] = val; }
  MFFT_D cx<T> get(int pos) { return buf[slot(pos)]; }
};

template <typename T, class Slot>
struct XchSplit {
  T* buf;
  Slot slot;
  template <class S, int P>
  MFFT_D void exchange(cx<T> (&v)[S::E], int j, bool pre_barrier) {
    if (pre_barrier) MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) { buf[slot(pos)] = v[reg].x; });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) { v[reg].x = buf[slot(pos)]; });
    MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) { buf[slot(pos)] = v[reg].y; });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) { v[reg].y = buf[slot(pos)]; });
  }
};

// generic pass driver: runs passes P..NP-1 with LDS exchanges in between.
// On entry v holds the inputs of pass 0 (positions j + k*TPT).
template <class S, int P, typename T, class TwPtr, class Xch, bool HIDE = true>
MFFT_D void run_passes(cx<T> (&v)[S::E], int j, TwPtr tw, Xch& xch) {
#if MFFT_LAUNDER_MODE == 4
  if constexpr (HIDE && P > 0 && launder_plan<S>()) {        // see fft_core.h MFFT_LAUNDER_MODE: the twiddle index must not see j's range
    int jc = j;
    MFFT_HIDE_RANGE(jc);
    pass_compute<S, P, T>(v, jc, tw);
  } else
#endif
  pass_compute<S, P, T>(v, j, tw);
  if constexpr (P + 1 < S::NP) {
    xch.template exchange<S, P>(v, j, P > 0);
    run_passes<S, P + 1, T, TwPtr, Xch, HIDE>(v, j, tw, xch);
  }
}

// ---- multi-column variants: each thread owns VEC adjacent columns (VEC register
// sets), so that one lane moves 16 bytes even in single precision and the
// twiddles of a pass are loaded once for all of its columns.
template <typename U, int VEC>
struct PackV {
  U e[VEC];
};

template <typename T, int VEC, class Slot>
struct XchFullV {
  PackV<cx<T>, VEC>* buf;
  Slot slot;
  template <class S, int P>
  MFFT_D void exchange(cx<T> (&v)[VEC][S::E], int j, bool pre_barrier) {
    if (pre_barrier) MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) {
      PackV<cx<T>, VEC> p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) p.e[i] = v[i][reg];
      buf[slot(pos)] = p;
    });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) {
      const PackV<cx<T>, VEC> p = buf[slot(pos)];
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i][reg] = p.e[i];
    });
  }
};

template <typename T, int VEC, class Slot>
struct XchSplitV {
  PackV<T, VEC>* buf;
  Slot slot;
  template <class S, int P>
  MFFT_D void exchange(cx<T> (&v)[VEC][S::E], int j, bool pre_barrier) {
    if (pre_barrier) MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) {
      PackV<T, VEC> p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) p.e[i] = v[i][reg].x;
      buf[slot(pos)] = p;
    });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) {
      const PackV<T, VEC> p = buf[slot(pos)];
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i][reg].x = p.e[i];
    });
    MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) {
      PackV<T, VEC> p;
#pragma unroll
      for (int i = 0; i < VEC; ++i) p.e[i] = v[i][reg].y;
      buf[slot(pos)] = p;
    });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) {
      const PackV<T, VEC> p = buf[slot(pos)];
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i][reg].y = p.e[i];
    });
  }
};

// The same in FOUR rounds through a buffer of N/2 reals per column: real parts of the positions below N/2, real parts of the
// rest, then the imaginary parts likewise.  A gathering thread knows at compile time which half a register comes from
// (register k reads position j + k TPT: the lower half for k < E/2); a scattering thread decides per value.  Half the LDS
// of XchSplitV for twice the barriers and store instructions: worth it only where it buys a second workgroup per CU.
template <typename T, int VEC, class Slot>
struct XchQuarterV {
  PackV<T, VEC>* buf;
  Slot slot;
  template <class S, int P>
  MFFT_D void exchange(cx<T> (&v)[VEC][S::E], int j, bool pre_barrier) {
    component<S, P, 0>(v, j, pre_barrier);
    component<S, P, 1>(v, j, true);
  }
  // One component: the registers gathered in the first round (k < E/2) still hold values the second round has to scatter,
  // so the first round's gathers wait in E/2 temporaries until the second round's scatters are out.
  template <class S, int P, int COMP>
  MFFT_D void component(cx<T> (&v)[VEC][S::E], int j, bool pre_barrier) {
    static_assert(S::E % 2 == 0, "the quarter exchange splits the registers in two halves");
    constexpr int H = S::N / 2, EH = S::E / 2;
    T low[VEC][EH];
    if (pre_barrier) MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) {
      if (pos < H) {
        PackV<T, VEC> p;
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.e[i] = COMP == 0 ? v[i][reg].x : v[i][reg].y;
        buf[slot(pos)] = p;
      }
    });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) {
      if (reg < EH) {
        const PackV<T, VEC> p = buf[slot(pos)];
#pragma unroll
        for (int i = 0; i < VEC; ++i) low[i][reg < EH ? reg : 0] = p.e[i];
      }
    });
    MFFT_BARRIER();
    pass_scatter<S, P>(j, [&](int pos, int reg) {
      if (pos >= H) {
        PackV<T, VEC> p;
#pragma unroll
        for (int i = 0; i < VEC; ++i) p.e[i] = COMP == 0 ? v[i][reg].x : v[i][reg].y;
        buf[slot(pos - H)] = p;
      }
    });
    MFFT_BARRIER();
    pass_gather<S>(j, [&](int pos, int reg) {
      if (reg >= EH) {
        const PackV<T, VEC> p = buf[slot(pos - H)];
#pragma unroll
        for (int i = 0; i < VEC; ++i) (COMP == 0 ? v[i][reg].x : v[i][reg].y) = p.e[i];
      }
    });
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
      for (int k = 0; k < EH; ++k) (COMP == 0 ? v[i][k].x : v[i][k].y) = low[i][k];
  }
};

// PB0: also synchronise before the FIRST scatter (persistent kernels: the previous tile's last gather)
template <class S, int P, typename T, int VEC, class TwPtr, class Xch, bool PB0 = false>
MFFT_D void run_passes_v(cx<T> (&v)[VEC][S::E], int j, TwPtr tw, Xch& xch) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) pass_compute<S, P, T>(v[i], j, tw);
  if constexpr (P + 1 < S::NP) {
    xch.template exchange<S, P>(v, j, P > 0 || PB0);
    run_passes_v<S, P + 1, T, VEC, TwPtr, Xch, PB0>(v, j, tw, xch);
  }
}

// bijective XCD-aware block remap: workgroups are dealt round-robin over the 8
// XCDs (MI355X_MICROARCH.md, "Workgroup dispatch"), so block b lands on XCD b%8.
// Give each XCD a contiguous range of tiles so that neighbouring tiles, which
// share boundary cache lines when rows are not 128-byte aligned, share an L2.
MFFT_HD int xcd_remap(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int x = b & 7, i = b >> 3;
  return x * q + (x < r ? x : r) + i;
}
// The same with XCD x entering its range 64*x tiles later (and wrapping around).  When the tile range is a power of two
// (C2C arrays: rows a power of two apart), the eight ranges start a power of two apart as well, so the eight windows
// of tiles in flight would sit on the SAME memory channels; shifted by 8 KB each they cover eight different sets.
MFFT_HD int xcd_remap_skew(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int x = b & 7, i = b >> 3;
  const int cnt = q + (x < r ? 1 : 0);
  int k = i + 64 * x;
  if (cnt > 0) k %= cnt;
  return x * q + (x < r ? x : r) + k;
}

// stage the twiddle table into LDS (cooperatively), returns pointer to it
template <class S, typename T>
MFFT_D void stage_twiddles(cx<T>* lds_tw, const cx<T>* gtw, int tid, int nthreads) {
  for (int i = tid; i < S::TW; i += nthreads) lds_tw[i] = gtw[i];
}

// ---------------------------------------------------------------------------
// strided-axis c2c
// ---------------------------------------------------------------------------
// NT: non-temporal global loads/stores.  Only for tiles whose rows are 128-byte
// aligned (every L2 line belongs to exactly one workgroup): measured +12 % on
// such layouts, 2x slower on unaligned ones, where neighbouring tiles share lines.
//
// PAD fuses the 3/2-rule copies (reference slab.py:516-536, pencil.py:351-379) into the transform,
// for N = 3n/2 (so E is a multiple of 3 and the regions are whole register ranges):
//   PAD == 1: the input has n = 2N/3 physical rows; logical rows [N/3, 2N/3) are zeros
//             (copy_to_padded: low half to the front, high half to the back)
//   PAD == 3: (any plan) the 2/3-rule instead: `fu * dealias` (slab.py:237-245) fused into the load of the first
//             inverse pass -- one mask byte per element, no masked copy of the spectrum
//   PAD == 4: (any plan) the 2/3-rule when the mask is the product of three 1-D band conditions (what
//             get_dealias_filter builds): removed rows are not loaded, tiles of removed columns are not processed
//   PAD == 2: only logical rows [0, N/3) and [2N/3, N) are stored, to n physical rows
//             (copy_from_padded); with P.fold the Nyquist rows N/3 and 2N/3, which sit in the
//             same thread (j = 0), are summed as `fu[n/2:] += fp[-n/2:]` does.
// SPLIT: 0 whole complex values through LDS, 1 real then imaginary parts (XchSplitV), 2 in four rounds (XchQuarterV)
template <class S, typename T, int COLS, bool INV, bool TWLDS, int SPLIT = 0, int VEC = 1, bool NT = false,
          int PAD = 0>
struct ColFft {
  static_assert(COLS % VEC == 0, "VEC must divide COLS");
  static_assert(PAD == 0 || PAD == 3 || PAD == 4 || S::E % 3 == 0, "pad/truncate fusion needs a radix-3 plan");
  static constexpr int KLO = S::E / 3, KHI = 2 * (S::E / 3);     // register ranges of the three row regions
  static constexpr int NSKIP = S::N / 3;
  static constexpr int CG = COLS / VEC;            // lanes along the contiguous axis
  static constexpr int THREADS = S::TPT * CG;
  static constexpr int TW_BYTES = TWLDS ? (int)(S::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = S::NP > 1 ? (int)(S::N * COLS * (SPLIT ? sizeof(T) : sizeof(cx<T>)) / (SPLIT == 2 ? 2 : 1)) : 0;
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;

  struct Slot {
    int c;
    MFFT_D int operator()(int pos) const { return pos * CG + c; }
  };
  typedef PackV<cx<T>, VEC> GPack;                 // VEC adjacent complex values in global memory
#if defined(__HIPCC__)
  typedef T vec_t __attribute__((ext_vector_type(2 * VEC)));
  static MFFT_D GPack load_pack(const cx<T>* src) {
    if constexpr (NT) {
      const vec_t x = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(src));
      GPack g;
#pragma unroll
      for (int i = 0; i < VEC; ++i) g.e[i] = mk<T>(x[2 * i], x[2 * i + 1]);
      return g;
    } else {
      return *reinterpret_cast<const GPack*>(src);
    }
  }
  static MFFT_D void store_pack(cx<T>* dst, const GPack& g) {
    if constexpr (NT) {
      vec_t x;
#pragma unroll
      for (int i = 0; i < VEC; ++i) { x[2 * i] = g.e[i].x; x[2 * i + 1] = g.e[i].y; }
      __builtin_nontemporal_store(x, reinterpret_cast<vec_t*>(dst));
    } else {
      *reinterpret_cast<GPack*>(dst) = g;
    }
  }
#else
  static MFFT_D GPack load_pack(const cx<T>* src) { return *reinterpret_cast<const GPack*>(src); }
  static MFFT_D void store_pack(cx<T>* dst, const GPack& g) { *reinterpret_cast<GPack*>(dst) = g; }
#endif

  static MFFT_D void body(const ColParams<T>& P, int bid_raw, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    int bid;
    if constexpr (PAD == 4) {                      // neighbours in the list are neighbours in memory: the same XCD-aware order
      if (P.tile_list) bid = P.tile_list[P.remap ? xcd_remap(bid_raw, P.ntiles_listed) : bid_raw];
      else bid = P.remap ? xcd_remap(bid_raw, P.ntile_c * P.nouter) : bid_raw;
    } else {
      bid = P.remap == 2 ? xcd_remap_skew(bid_raw, P.ntile_c * P.nouter)
          : P.remap    ? xcd_remap(bid_raw, P.ntile_c * P.nouter) : bid_raw;
    }
    const int outer = bid / P.ntile_c;
    const int tc = bid - outer * P.ntile_c;
    const int c = tid % CG;
    const int j = tid / CG;
    const int col = tc * COLS + c * VEC;
    const int nact = P.ncols - col;                // columns of this thread inside the array (may be <= 0)
    i64 icol = col;
    int first = VEC;                               // wrapped input: columns i >= first of this lane lie in_wrap_gap further
    if (P.in_wrap > 0) {
      const int q = col / P.in_wrap;
      icol += (i64)q * P.in_wrap_gap;
      if (VEC > 1 && col - q * P.in_wrap + VEC > P.in_wrap) first = P.in_wrap - (col - q * P.in_wrap);
    }
    const cx<T>* ip = P.in + (i64)outer * P.in_outer + icol;
    cx<T>* op = P.out + (i64)outer * P.out_outer + col;
    bool tile_zero = false;                        // b_gzero == 2: a tile of removed columns only is written as zeros
    if constexpr (PAD == 4) {                      // nothing downstream reads the columns the mask removes
      int t = P.b_coff + tc * COLS;
      int z = t % P.b_cper, y = P.b_goff + t / P.b_cper + outer * P.b_gstep;
      bool any = false;
      for (int i = 0; i < COLS && tc * COLS + i < P.ncols; ++i) {
        any = any || (z < P.b_clim && (P.b_gzero == 1 || y < P.b_glo || y >= P.b_ghi));
        if (++z == P.b_cper) { z = 0; ++y; }
      }
      if (!any) {                                  // the same for every thread of the workgroup
        if (P.b_gzero != 2) return;
        tile_zero = true;                          // complete output wanted: no loads, no passes, zeros stored
      }
    }
    // b_gzero: which of this thread's VEC columns belong to a removed y (b_gzero == 2: or to a removed z) and are
    // therefore transformed as zeros
    bool zcol[VEC];
    bool zero_col = false;                         // all of them
    if constexpr (PAD == 4) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) zcol[i] = false;
      if (P.b_gzero) {
        zero_col = true;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const int t = P.b_coff + col + i;
          const int yq = t / P.b_cper;
          const int y = P.b_goff + yq + outer * P.b_gstep;
          zcol[i] = !(y < P.b_glo || y >= P.b_ghi) || (P.b_gzero == 2 && t - yq * P.b_cper >= P.b_clim);
          zero_col = zero_col && zcol[i];
        }
      }
    }

    cx<T> v[VEC][S::E];
    if (PAD == 4 && tile_zero) {
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i][k] = mk<T>((T)0, (T)0);
      }
    } else {
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      unsigned r = (unsigned)(j + k * S::TPT);
      if (PAD == 1) {
        if (k >= KLO && k < KHI) {               // the zero band of the padded spectrum: nothing to load
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i][k] = mk<T>((T)0, (T)0);
          continue;
        }
        if (k >= KHI) r -= NSKIP;
      }
      const cx<T>* src = ip + row_off(P.in_map, r);
      bool zero_row = false;
      if constexpr (PAD == 4) {                    // a removed row: re-read row 0 (a cache hit) and drop the value -- no branch
                                                   // (skipping the load of a register whose rows are all removed, a workgroup-
                                                   // uniform branch, was measured: 1024^3 x pass 1.84 -> 2.03 ms, fp32 1.51 -> 2.05)
        zero_row = zero_col || ((int)r >= P.b_row_lo && (int)r < P.b_row_hi);
        if (zero_row) src = ip + row_off(P.in_map, 0u);
      }
      if (nact >= VEC && first == VEC) {
        const GPack g = load_pack(src);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i][k] = INV ? swapri(g.e[i]) : g.e[i];
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          cx<T> x = mk<T>((T)0, (T)0);
          if (i < nact) x = src[i + (i >= first ? P.in_wrap_gap : 0)];
          v[i][k] = INV ? swapri(x) : x;
        }
      }
      if constexpr (PAD == 4) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (zero_row || zcol[i]) v[i][k] = mk<T>((T)0, (T)0);
      }
      if constexpr (PAD == 3) {                    // 2/3-rule: the dealias mask applied while the spectrum is read
        const unsigned char* mp = P.mask + (src - P.in);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const unsigned char keep = i < nact ? mp[i] : (unsigned char)0;
          if (!keep) v[i][k] = mk<T>((T)0, (T)0);
        }
      }
    }
    if constexpr (TWLDS && S::NP > 1) {
      stage_twiddles<S, T>(ltw, P.tw, tid, THREADS);
      MFFT_BARRIER();
    }
    if constexpr (SPLIT == 2) {
      XchQuarterV<T, VEC, Slot> xch{reinterpret_cast<PackV<T, VEC>*>(lds + TW_BYTES), Slot{c}};
      if constexpr (TWLDS) run_passes_v<S, 0, T, VEC>(v, j, (const cx<T>*)ltw, xch);
      else run_passes_v<S, 0, T, VEC>(v, j, P.tw, xch);
    } else if constexpr (SPLIT) {
      XchSplitV<T, VEC, Slot> xch{reinterpret_cast<PackV<T, VEC>*>(lds + TW_BYTES), Slot{c}};
      if constexpr (TWLDS) run_passes_v<S, 0, T, VEC>(v, j, (const cx<T>*)ltw, xch);
      else run_passes_v<S, 0, T, VEC>(v, j, P.tw, xch);
    } else {
      XchFullV<T, VEC, Slot> xch{reinterpret_cast<PackV<cx<T>, VEC>*>(lds + TW_BYTES), Slot{c}};
      if constexpr (TWLDS) run_passes_v<S, 0, T, VEC>(v, j, (const cx<T>*)ltw, xch);
      else run_passes_v<S, 0, T, VEC>(v, j, P.tw, xch);
    }
    }     // !tile_zero

#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      unsigned r = (unsigned)(j + k * S::TPT);
      if (PAD == 2) {
        if (k >= KLO && k < KHI) continue;       // truncated band
        if (k >= KHI) r -= NSKIP;
        if (k == KHI && j == 0 && P.fold) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) v[i][k] = v[i][k] + v[i][KLO];
        }
      }
      cx<T>* dst = op + row_off(P.out_map, r);
      if (nact >= VEC) {
        GPack g;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          const cx<T> x = scale(v[i][k], P.scale);
          g.e[i] = INV ? swapri(x) : x;
        }
        store_pack(dst, g);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (i < nact) {
            const cx<T> x = scale(v[i][k], P.scale);
            dst[i] = INV ? swapri(x) : x;
          }
        }
      }
    }
  }
};

// exchange policy of the contiguous-axis kernels: whole complex values, or (SPLIT) real and imaginary parts one after the
// other through half the LDS -- twice the rows per CU for the plans with many values per thread
template <bool SPLIT, typename T, class Slot> struct RowXch {
  typedef XchFull<T, Slot> type;
  typedef cx<T> elem;
};
template <typename T, class Slot> struct RowXch<true, T, Slot> {
  typedef XchSplit<T, Slot> type;
  typedef T elem;
};

// ---------------------------------------------------------------------------
// contiguous-axis c2c
// ---------------------------------------------------------------------------
// CHUNK: the output (forward) / input (inverse) rows are split into z chunks (ZSplit), see above.
template <class S, typename T, int ROWS, bool INV, bool TWLDS, bool CHUNK = false, bool SPLIT = false>
struct RowFft {
  typedef typename RowXch<SPLIT, T, PadSlot<S::R(0)>>::elem XE;
  static constexpr int THREADS = S::TPT * ROWS;
  static constexpr int PD = S::R(0);
  static constexpr int PLEN = padded_len<S::N, PD>();
  static constexpr int TW_BYTES = TWLDS ? (int)(S::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = S::NP > 1 ? (int)(PLEN * ROWS * sizeof(XE)) : 0;
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;

  static MFFT_D void body(const RowParams<T>& P, int bid, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    const int rl = tid / S::TPT;
    const int j = row_thread_index<S>(tid);
    XE* xch = reinterpret_cast<XE*>(lds + TW_BYTES) + rl * PLEN;
    const i64 row = (i64)bid * ROWS + rl;
    const bool active = row < P.nrows;
    // rows past the end re-read the last row (and store nothing): unconditional loads let the
    // compiler issue all E of them before the first wait
    const cx<T>* ip = P.in + (active ? row : P.nrows - 1) * P.in_stride;
    cx<T>* op = P.out + row * P.out_stride;

    const i64 crow = P.row0 + (active ? row : P.nrows - 1);     // CHUNK: row inside the chunk blocks
    cx<T> v[S::E];
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      cx<T> x;
      if constexpr (CHUNK && INV) {          // every position of a complex row exists: the load stays unconditional
        bool ok;
        x = P.in[zsplit_off(P.zs, crow, j + k * S::TPT, &ok)];
      } else {
        x = ip[j + k * S::TPT];
      }
      v[k] = INV ? swapri(x) : x;
    }
    if constexpr (TWLDS && S::NP > 1) {
      stage_twiddles<S, T>(ltw, P.tw, tid, THREADS);
      MFFT_BARRIER();
    }
    typename RowXch<SPLIT, T, PadSlot<PD>>::type xc{xch, PadSlot<PD>{}};
    if constexpr (TWLDS) run_passes<S, 0, T>(v, j, (const cx<T>*)ltw, xc);
    else run_passes<S, 0, T>(v, j, P.tw, xc);

    if (active) {
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        cx<T> x = scale(v[k], P.scale);
        x = INV ? swapri(x) : x;
        if constexpr (CHUNK && !INV) {
          bool ok;
          const i64 o = zsplit_off(P.zs, crow, j + k * S::TPT, &ok);
          if (ok) P.out[o] = x;
        } else {
          op[j + k * S::TPT] = x;
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// real -> half-complex along the contiguous axis.  S describes M = N/2.
// ---------------------------------------------------------------------------
// LIMIT: only the first P.valid complex columns exist in memory (3/2-rule): r2c does not
// store the others, c2r reads them as zeros.  A template flag so that the regular kernels keep
// unconditional loads (a runtime test costs the c2r kernel 7 % at 1024^3).
// WP: the wave-packed thread layout described at C2RFft (threads per transform that do not divide a wave): the mirrored
// value Z[M - pos] of the split post-pass comes through a wave shuffle instead of an LDS round trip with three more
// barriers and E more live registers (the r2c stage of 720^3 / 900^3 took 1.8 x the c2r stage's time, round 3).
template <class S, typename T, int ROWS, bool TWLDS, bool LIMIT = false, bool CHUNK = false, bool SPLIT = false, bool WP = false>
struct R2CFft {
  typedef typename RowXch<SPLIT, T, PadSlot<S::R(0)>>::elem XE;
  static constexpr int M = S::N;
  static constexpr int RPW = WP ? 64 / S::TPT : 1;
  static_assert(!WP || (S::TPT < 64 && 64 % S::TPT != 0 && ROWS % (64 / S::TPT) == 0), "wave-packed rows: whole waves of rows");
  static constexpr int THREADS = WP ? ROWS / RPW * 64 : S::TPT * ROWS;
  static constexpr int PD = S::R(0);
  static constexpr int PLEN = padded_len<M, PD>();
  static constexpr int TW_BYTES = TWLDS ? (int)(S::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = (int)(PLEN * ROWS * sizeof(XE));
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;

  static MFFT_D void body(const RealParams<T>& P, int bid, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    int rl, j, lane0 = 0;
    bool dup = false;
    if constexpr (WP) {            // see C2RFft
      const int lane = tid & 63;
      int lr = lane / S::TPT;
      dup = lr >= RPW;
      if (dup) lr = RPW - 1;
      lane0 = lr * S::TPT;
      j = dup ? lane - RPW * S::TPT : lane - lane0;
#if MFFT_LAUNDER_MODE == 1
      if constexpr (S::E % 15 == 0) MFFT_OPAQUE(j);
#endif
      rl = (tid >> 6) * RPW + lr;
    } else {
      rl = tid / S::TPT;
      j = row_thread_index<S>(tid);
    }
    XE* xch = reinterpret_cast<XE*>(lds + TW_BYTES) + rl * PLEN;
    const i64 row = (i64)bid * ROWS + rl;
    const bool inrange = row < P.nrows;
    const bool active = inrange && !dup;
    // a row of N reals read as N/2 complex (x[2n], x[2n+1])
    const cx<T>* ip =
        reinterpret_cast<const cx<T>*>(static_cast<const T*>(P.in) + (inrange ? row : P.nrows - 1) * P.in_stride);
    cx<T>* op = static_cast<cx<T>*>(P.out) + row * P.out_stride;

    cx<T> v[S::E];
#pragma unroll
    for (int k = 0; k < S::E; ++k) v[k] = ip[j + k * S::TPT];   // unconditional: see RowFft
    if constexpr (TWLDS && S::NP > 1) {
      stage_twiddles<S, T>(ltw, P.tw, tid, THREADS);
      MFFT_BARRIER();
    }
    typename RowXch<SPLIT, T, PadSlot<PD>>::type xc{xch, PadSlot<PD>{}};
    if constexpr (TWLDS) run_passes<S, 0, T>(v, j, (const cx<T>*)ltw, xc);
    else run_passes<S, 0, T>(v, j, P.tw, xc);

    // split post-pass: X[k] = E[k] + w_k O[k],  E = (Z[k] + conj Z[M-k])/2,
    // O = -i (Z[k] - conj Z[M-k])/2,  w_k = exp(-2 pi i k / N)
    const T half = (T)0.5;
    auto put = [&](int pos, cx<T> val) {
      if constexpr (CHUNK) {
        bool ok;
        const i64 o = zsplit_off(P.zs, P.row0 + row, pos, &ok);
        if (ok) static_cast<cx<T>*>(P.out)[o] = val;
      } else {
        op[pos] = val;
      }
    };
    auto emit_w = [&](int pos, cx<T> zk, cx<T> zpartner, cx<T> w) {
      if (pos == 0) {
        put(0, mk<T>((zk.x + zk.y) * P.scale, (T)0));
        if (!LIMIT || M < P.valid) put(M, mk<T>((zk.x - zk.y) * P.scale, (T)0));
      } else if (!LIMIT || pos < P.valid) {
        const cx<T> zm = conj(zpartner);
        const cx<T> e = scale(zk + zm, half);
        const cx<T> o = mul_mi(scale(zk - zm, half));
        put(pos, scale(e + w * o, P.scale));
      }
    };
    auto emit = [&](int pos, cx<T> zk, cx<T> zpartner) { emit_w(pos, zk, zpartner, P.rtw[pos]); };
    constexpr bool SHFL = WP || (S::TPT <= 64 && (64 % S::TPT) == 0);
    if constexpr (SHFL) {
      // Z[M-pos] of (lane j, register k) is register E-1-k of lane TPT-j of the same row: one
      // wave shuffle instead of an LDS round trip and two barriers (lane 0: its own register E-k)
      const int src = WP ? lane0 + (j == 0 ? 0 : S::TPT - j) : (tid & 63) - j + ((S::TPT - j) & (S::TPT - 1));
      // The post-pass twiddles exp(-2 pi i pos / N), loaded BEFORE the loop where the plan is small (<= 128 bytes of them per
      // thread).  Left inside `if (active)`, hipcc loads each one right before its use and waits for it with
      // s_waitcnt vmcnt(0) -- which on gfx9 also waits for the previous step's STORE to be acknowledged: the eight steps
      // of the single-precision kernel of 1024 ran one memory round trip after the other (read in the ISA, round 4)
      constexpr bool RTW_EARLY = S::E * (int)sizeof(cx<T>) <= 128;      // 192 measured: no gain at 768 / 1536, fewer waves
      cx<T> rt[RTW_EARLY ? S::E : 1];
      if constexpr (RTW_EARLY) {
#pragma unroll
        for (int k = 0; k < S::E; ++k) rt[k] = P.rtw[j + k * S::TPT];
      }
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const cx<T> give = v[S::E - 1 - k];
        cx<T> pm = mk<T>(wave_shfl(give.x, src), wave_shfl(give.y, src));
        if (j == 0) pm = v[(S::E - k) % S::E];
        if constexpr (RTW_EARLY) {
          if (active) emit_w(j + k * S::TPT, v[k], pm, rt[k]);
        } else
        if (active) emit(j + k * S::TPT, v[k], pm);
        // 30 values per thread: keep the scheduler from hoisting every shuffle above the first store (all mirrors live at
        // once = 120 more registers in double precision, one wave per SIMD)
        if constexpr (S::E % 15 == 0) {
          if (k % 3 == 2) MFFT_SCHED_FENCE();
        }
      }
    } else if constexpr (SPLIT) {
      // the mirrored partners through the half-size buffer: real parts, then imaginary parts (the imaginary parts are
      // consumed as they are read: only the real parts of the mirrors are held)
      T pmx[S::E];
      if constexpr (S::NP > 1) MFFT_BARRIER();
#pragma unroll
      for (int k = 0; k < S::E; ++k) xch[padpos<PD>(j + k * S::TPT)] = v[k].x;
      MFFT_BARRIER();
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int pos = j + k * S::TPT;
        pmx[k] = xch[padpos<PD>(pos == 0 ? 0 : M - pos)];
      }
      MFFT_BARRIER();
#pragma unroll
      for (int k = 0; k < S::E; ++k) xch[padpos<PD>(j + k * S::TPT)] = v[k].y;
      MFFT_BARRIER();
      if (active) {
#pragma unroll
        for (int k = 0; k < S::E; ++k) {
          const int pos = j + k * S::TPT;
          emit(pos, v[k], mk<T>(pmx[k], xch[padpos<PD>(pos == 0 ? 0 : M - pos)]));
          if constexpr (S::E % 15 == 0) {
            if (k % 3 == 2) MFFT_SCHED_FENCE();
          }
        }
      }
    } else {
      if constexpr (S::NP > 1) MFFT_BARRIER();
#pragma unroll
      for (int k = 0; k < S::E; ++k) xc.put(j + k * S::TPT, v[k]);
      MFFT_BARRIER();
      if (active) {
#pragma unroll
        for (int k = 0; k < S::E; ++k) {
          const int pos = j + k * S::TPT;
          emit(pos, v[k], xc.get(pos == 0 ? 0 : M - pos));
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// half-complex -> real along the contiguous axis.  S describes M = N/2.
// out = irfft(in) * N * scale   (scale = 1/N gives numpy's irfft)
// ---------------------------------------------------------------------------
// WP ("wave-packed", round 4): for transforms whose TPT threads do not divide a wave (24, 30, 48, 60 ... threads: the
// lengths with 3 and 5 among their factors, 9 * 2^a, 125 * 2^a) a wave holds RPW = 64 / TPT whole rows and its last
// 64 - RPW * TPT lanes DUPLICATE the first threads of its last row (same loads, same LDS writes, same shuffles: harmless,
// they store nothing).  No row straddles a wave any more, so the mirrored bin X[M - pos] comes through a wave shuffle
// as it does for the power-of-two thread counts, and every bin is read from memory ONCE (the unpacked layout loaded
// both bins of a pair: 1.23 x the algorithmic HBM traffic at 720^3, profiles/r03_720_pmc_traffic.json).
// MLDS (round 6): where no wave shuffle reaches the mirrored bin (threads per transform that neither divide a wave nor run
// wave-packed: 15, 24, 30, 48 ... threads), every bin is still read from memory ONCE and the mirrors come through the
// exchange buffer, which is idle before the first pass -- instead of a second global load per bin (1.31 x the algorithmic
// fetch at 1440^3, profiles/r05_ytile_builds.txt).
template <class S, typename T, int ROWS, bool TWLDS, bool LIMIT = false, bool CHUNK = false, bool SPLIT = false, bool WP = false,
          bool MLDS = false>
struct C2RFft {
  typedef typename RowXch<SPLIT, T, PadSlot<S::R(0)>>::elem XE;
  static constexpr int M = S::N;
  static constexpr int RPW = WP ? 64 / S::TPT : 1;                      // rows per wave (wave-packed layout)
  static_assert(!WP || (S::TPT < 64 && 64 % S::TPT != 0 && ROWS % (64 / S::TPT) == 0), "wave-packed rows: whole waves of rows");
  static constexpr int THREADS = WP ? ROWS / RPW * 64 : S::TPT * ROWS;
  static constexpr int PD = S::R(0);
  static constexpr int PLEN = padded_len<M, PD>();
  static constexpr int TW_BYTES = TWLDS ? (int)(S::TW * sizeof(cx<T>)) : 0;
  static constexpr int XCH_BYTES = S::NP > 1 ? (int)(PLEN * ROWS * sizeof(XE)) : 0;
  static constexpr int LDS_BYTES = TW_BYTES + XCH_BYTES;

  static MFFT_D void body(const RealParams<T>& P, int bid, int tid, char* lds) {
    cx<T>* ltw = reinterpret_cast<cx<T>*>(lds);
    int rl, j, lane0 = 0;          // row of the workgroup, thread of the row, (WP) lane of the row's thread 0
    bool dup = false;              // (WP) a lane that duplicates a thread of the wave's last row
    if constexpr (WP) {
      const int lane = tid & 63;
      int lr = lane / S::TPT;
      dup = lr >= RPW;
      if (dup) lr = RPW - 1;
      lane0 = lr * S::TPT;
      j = dup ? lane - RPW * S::TPT : lane - lane0;
      if constexpr (S::E % 15 == 0) MFFT_OPAQUE(j);         // see row_thread_index (c2r: at the origin)
      rl = (tid >> 6) * RPW + lr;
    } else {
      rl = tid / S::TPT;
      j = row_thread_index<S, true>(tid);
    }
    XE* xch = reinterpret_cast<XE*>(lds + TW_BYTES) + rl * PLEN;
    const i64 row = (i64)bid * ROWS + rl;
    const bool inrange = row < P.nrows;
    const bool active = inrange && !dup;
    // rows past the end re-read the last row (unconditional loads: see RowFft) and store nothing
    const cx<T>* ip = static_cast<const cx<T>*>(P.in) + (inrange ? row : P.nrows - 1) * P.in_stride;
    cx<T>* op = reinterpret_cast<cx<T>*>(static_cast<T*>(P.out) + row * P.out_stride);

    // bin `pos` of this row; CHUNK: out of its z chunk, a position the last chunk does not hold (dropped Nyquist
    // column) reads as zero -- the load itself stays unconditional (clamped offset + select), see RowFft
    const i64 crow = P.row0 + (inrange ? row : P.nrows - 1);
    auto bin = [&](int pos) -> cx<T> {
      if constexpr (CHUNK) {
        bool ok;
        const i64 o = zsplit_off(P.zs, crow, pos, &ok);
        const cx<T> x = static_cast<const cx<T>*>(P.in)[ok ? o : (i64)0];
        if constexpr (sizeof(T) == 4) return keep_bits(x, ok);      // not a select: hipcc made a branch of it and serialised the loads
        else return ok ? x : mk<T>((T)0, (T)0);                     // (double precision: measured as it was, see LIMIT below)
      } else {
        return ip[pos];
      }
    };
    // pre-pass: Z[k] = (X[k] + conj X[M-k]) + i conj(w_k) (X[k] - conj X[M-k])
    // (twice the textbook value; the factor is folded into the normalisation)
    cx<T> v[S::E];
    constexpr bool SHFL = WP || (S::TPT <= 64 && (64 % S::TPT) == 0);
    if constexpr (SHFL) {
      // Every bin is read from memory ONCE.  The mirrored partner X[M-pos] of (lane j, register k)
      // is register E-1-k of lane TPT-j of the same row (all inside one wave), fetched with a
      // wave shuffle; lane 0's partners are its own registers E-k and the extra bin X[M].
      // The bins are loaded straight into v and turned into the pre-pass values IN PLACE, two registers
      // (k, E-1-k) at a time: a second register array would cost E complex registers (256 VGPRs for the
      // E = 20 plans in fp64).  Lane 0 only ever reads its own shuffle operand, so it simply offers the
      // registers it needs itself: X[M] first, then its register E-k, carried over from the previous step.
      static_assert(S::E % 2 == 0, "register pairing needs an even number of values per thread");
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int pos = j + k * S::TPT;
        if constexpr (LIMIT && sizeof(T) == 4) {
          // columns >= valid read as zero.  Single precision: the load stays UNCONDITIONAL (position 0 instead, a cache hit,
          // and keep_bits).  With a branch around it hipcc waits for every load on its own inside its branch -- twelve round
          // trips to HBM one after the other: c2r of the 1024^3 3/2-rule pair 7.7 ms against 5.1 for r2c with the same
          // bytes; 5.2 now (profiles/r04_serialised_loads.txt)
          const bool ok = pos < P.valid;
          v[k] = keep_bits(bin(ok ? pos : 0), ok);
        } else if constexpr (LIMIT) {
          // double precision: the branches compile to predicated loads that are all in flight together, and the loads of
          // the missing columns are skipped: 9.2 ms; the unconditional form compiles for fewer registers and then waits for
          // every (cached) twiddle load of the pre-pass on its own: 10.1 ms
          v[k] = mk<T>((T)0, (T)0);
          if (pos < P.valid) v[k] = bin(pos);
        } else {
          v[k] = bin(pos);
        }
      }
      cx<T> carry = mk<T>((T)0, (T)0);
      if (j == 0 && (!LIMIT || M < P.valid)) carry = bin(M);
      // lane of thread TPT-j (mod TPT) of this row
      const int src = WP ? lane0 + (j == 0 ? 0 : S::TPT - j) : (tid & 63) - j + ((S::TPT - j) & (S::TPT - 1));
      auto prepass = [&](cx<T> xk, cx<T> pm, int pos) {
        cx<T> xm = conj(pm);
        if (pos == 0) {              // imaginary parts of the k=0 and k=N/2 bins are ignored
          xk.y = (T)0;
          xm.y = (T)0;
        }
        const cx<T> e = xk + xm;
        const cx<T> dd = xk - xm;
        return swapri(e + mul_pi(dd * conj(P.rtw[pos])));   // inverse transform through the swap identity
      };
#pragma unroll
      for (int k = 0; k < S::E / 2; ++k) {
        constexpr int E = S::E;
        const int kp = E - 1 - k;
        const cx<T> a = v[k], b = v[kp];
        const cx<T> give1 = j == 0 ? carry : b;          // partner of position j + k*TPT
        const cx<T> give2 = j == 0 ? v[k + 1] : a;       // partner of position j + kp*TPT (lane 0: its register E-kp = k+1)
        cx<T> pm1, pm2;
        // (direct form for the 28-values plans only: with it the c2r of 640 complex points -- 20 values, 248 -> 170 VGPRs --
        // came out 11 % SLOWER, 1280^3 bwd_z 6.17 -> 6.86 ms in the round-5 sweep; every other plan keeps round 4's code)
        if constexpr (WP || S::E % 7 != 0) {
          pm1 = mk<T>(wave_shfl(give1.x, src), wave_shfl(give1.y, src));
          pm2 = mk<T>(wave_shfl(give2.x, src), wave_shfl(give2.y, src));
        } else {
          pm1 = mk<T>(WAVE_SHFL_DIRECT(give1.x, src), WAVE_SHFL_DIRECT(give1.y, src));
          pm2 = mk<T>(WAVE_SHFL_DIRECT(give2.x, src), WAVE_SHFL_DIRECT(give2.y, src));
        }
        carry = b;                                       // lane 0's partner register of the next step
        v[k] = prepass(a, pm1, j + k * S::TPT);
        v[kp] = prepass(b, pm2, j + kp * S::TPT);
        if constexpr (S::E % 15 == 0) {        // see R2CFft: keep the shuffles from all being hoisted to the front
          if (k % 3 == 2) MFFT_SCHED_FENCE();
        }
      }
    } else if constexpr (MLDS && S::NP > 1) {
      // every bin once into v, position M (thread 0's partner of position 0) beside it; the mirrors through LDS
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int pos = j + k * S::TPT;
        if constexpr (LIMIT) {
          const bool ok = pos < P.valid;
          v[k] = keep_bits(bin(ok ? pos : 0), ok);
        } else {
          v[k] = bin(pos);
        }
      }
      cx<T> xM = mk<T>((T)0, (T)0);
      if (j == 0 && (!LIMIT || M < P.valid)) xM = bin(M);
      auto prepass = [&](cx<T> xk, cx<T> pm, int pos) {
        cx<T> xm = conj(pm);
        if (pos == 0) {              // imaginary parts of the k=0 and k=N/2 bins are ignored
          xk.y = (T)0;
          xm.y = (T)0;
        }
        const cx<T> e = xk + xm;
        const cx<T> dd = xk - xm;
        return swapri(e + mul_pi(dd * conj(P.rtw[pos])));
      };
      if constexpr (SPLIT) {         // real parts, then imaginary parts, through the half-size buffer (slot padpos(M) exists: PLEN > M + M/PD)
        T pmx[S::E];
#pragma unroll
        for (int k = 0; k < S::E; ++k) xch[padpos<PD>(j + k * S::TPT)] = v[k].x;
        if (j == 0) xch[padpos<PD>(M)] = xM.x;
        MFFT_BARRIER();
#pragma unroll
        for (int k = 0; k < S::E; ++k) pmx[k] = xch[padpos<PD>(M - (j + k * S::TPT))];
        MFFT_BARRIER();
#pragma unroll
        for (int k = 0; k < S::E; ++k) xch[padpos<PD>(j + k * S::TPT)] = v[k].y;
        if (j == 0) xch[padpos<PD>(M)] = xM.y;
        MFFT_BARRIER();
#pragma unroll
        for (int k = 0; k < S::E; ++k) {
          const int pos = j + k * S::TPT;
          v[k] = prepass(v[k], mk<T>(pmx[k], xch[padpos<PD>(M - pos)]), pos);
        }
      } else {
#pragma unroll
        for (int k = 0; k < S::E; ++k) xch[padpos<PD>(j + k * S::TPT)] = v[k];
        if (j == 0) xch[padpos<PD>(M)] = xM;
        MFFT_BARRIER();
#pragma unroll
        for (int k = 0; k < S::E; ++k) {
          const int pos = j + k * S::TPT;
          v[k] = prepass(v[k], xch[padpos<PD>(M - pos)], pos);
        }
      }
      MFFT_BARRIER();                // the buffer goes back to the passes' exchanges
    } else {
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int pos = j + k * S::TPT;
        cx<T> z;
        {
          cx<T> xk, xm;
          if constexpr (LIMIT) {          // unconditional loads of clamped positions + selects (see the shuffle path)
            const bool okk = pos < P.valid, okm = (M - pos) < P.valid;
            xk = keep_bits(bin(okk ? pos : 0), okk);
            xm = conj(keep_bits(bin(okm ? M - pos : 0), okm));
          } else {
            xk = bin(pos);
            xm = conj(bin(M - pos));
          }
          if (pos == 0) {            // imaginary parts of the k=0 and k=N/2 bins are ignored
            xk.y = (T)0;
            xm.y = (T)0;
          }
          const cx<T> e = xk + xm;
          const cx<T> d = xk - xm;
          const cx<T> o = d * conj(P.rtw[pos]);
          z = e + mul_pi(o);
        }
        v[k] = swapri(z);            // inverse transform through the swap identity
      }
    }
    if constexpr (TWLDS && S::NP > 1) {
      stage_twiddles<S, T>(ltw, P.tw, tid, THREADS);
      MFFT_BARRIER();
    }
    typename RowXch<SPLIT, T, PadSlot<PD>>::type xc{xch, PadSlot<PD>{}};
    if constexpr (TWLDS) run_passes<S, 0, T, const cx<T>*, decltype(xc), false>(v, j, (const cx<T>*)ltw, xc);
    else run_passes<S, 0, T, const cx<T>*, decltype(xc), false>(v, j, P.tw, xc);

    if (active) {
      const T s = P.scale;
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        cx<T> x = swapri(v[k]);
        op[j + k * S::TPT] = scale(x, s);
      }
    }
  }
};

}  // namespace mfft
