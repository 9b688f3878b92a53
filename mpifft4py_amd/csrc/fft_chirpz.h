// fft_chirpz.h -- transforms of ARBITRARY length n (primes, 7-smooth, 9*2^a, ...) on top of the
// compiled radix plans: Bluestein's chirp-z identity, evaluated entirely inside one workgroup.
//
// The reference accepts any mesh its FFT backend accepts (numpy/FFTW take every n;
// mpiFFT4py/serialFFT/numpy_fft.py:25-107); the radix plans of plans.h cover 2^a, 3*2^a, 5*2^a.
// Every other length goes through
//     X[k] = w[k] * sum_m (x[m] w[m]) conj(w)[k-m],      w[m] = exp(-i pi m^2 / n)
// i.e. a cyclic convolution of length M >= 2n-1 with a fixed filter, M being one of the compiled
// plans S.  Because the register layout of fft_core.h has the same position set (j + k*TPT) on
// input and output, the two length-M transforms chain in registers:
//     load n rows (x chirp, zeros above)  ->  FFT_M  ->  x bhat  ->  IFFT_M (swap identity)
//     ->  x chirp  ->  store n rows
// with exactly the global-memory pattern of the native kernels (same tiles, same fused pack /
// unpack row maps).  HBM traffic is therefore unchanged; the price is ~4-5x the butterfly work
// and an exchange buffer sized for M instead of n.  Tables (chirp: n entries, bhat = FFT_M(filter)/M:
// M entries) are built on the host in long double (twiddle.h) and read through L1/L2.
#pragma once
#include "fft_kernels.h"

namespace mfft {

template <typename T>
struct ColParamsZ : ColParams<T> {
  const cx<T>* chirp;
  const cx<T>* bhat;
  int n;                 // logical transform length (2n-1 <= S::N)
};
template <typename T>
struct RowParamsZ : RowParams<T> {
  const cx<T>* chirp;
  const cx<T>* bhat;
  int n;
};
template <typename T>
struct RealParamsZ : RealParams<T> {      // n = REAL length (any parity), rtw unused
  const cx<T>* chirp;
  const cx<T>* bhat;
  int n;
};

// the convolution in registers: on entry v = chirp-modulated input at positions j + k*TPT (zero for
// positions >= n), on exit swapri(v[k]) is the convolution at position j + k*TPT
template <class S, typename T, int VEC, class Xch>
MFFT_D void chirpz_conv_v(cx<T> (&v)[VEC][S::E], int j, const cx<T>* tw, const cx<T>* bhat, Xch& xch) {
  run_passes_v<S, 0, T, VEC>(v, j, tw, xch);
  if constexpr (S::NP > 1) MFFT_BARRIER();       // the exchange buffer is reused by the second transform
#pragma unroll
  for (int k = 0; k < S::E; ++k) {
    const cx<T> b = bhat[j + k * S::TPT];
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i][k] = swapri(v[i][k] * b);
  }
  run_passes_v<S, 0, T, VEC>(v, j, tw, xch);
}
template <class S, typename T, class Xch>
MFFT_D void chirpz_conv(cx<T> (&v)[S::E], int j, const cx<T>* tw, const cx<T>* bhat, Xch& xch) {
  run_passes<S, 0, T>(v, j, tw, xch);
  if constexpr (S::NP > 1) MFFT_BARRIER();
#pragma unroll
  for (int k = 0; k < S::E; ++k) v[k] = swapri(v[k] * bhat[j + k * S::TPT]);
  run_passes<S, 0, T>(v, j, tw, xch);
}

// ---------------------------------------------------------------------------
// strided-axis c2c of runtime length P.n (tile geometry and row maps as ColFft)
// ---------------------------------------------------------------------------
template <class S, typename T, int COLS, bool INV, bool SPLIT, int VEC>
struct ColFftZ {
  static_assert(COLS % VEC == 0, "VEC must divide COLS");
  static constexpr int CG = COLS / VEC;
  static constexpr int THREADS = S::TPT * CG;
  static constexpr int LDS_BYTES = S::NP > 1 ? (int)(S::N * COLS * (SPLIT ? sizeof(T) : sizeof(cx<T>))) : 0;
  struct Slot {
    int c;
    MFFT_D int operator()(int pos) const { return pos * CG + c; }
  };
  typedef PackV<cx<T>, VEC> GPack;

  static MFFT_D void body(const ColParamsZ<T>& P, int bid_raw, int tid, char* lds) {
    const int bid = P.remap ? xcd_remap(bid_raw, P.ntile_c * P.nouter) : bid_raw;
    const int outer = bid / P.ntile_c;
    const int tc = bid - outer * P.ntile_c;
    const int c = tid % CG;
    const int j = tid / CG;
    const int col = tc * COLS + c * VEC;
    const int nact = P.ncols - col;
    const int n = P.n;
    const cx<T>* ip = P.in + (i64)outer * P.in_outer + col;
    cx<T>* op = P.out + (i64)outer * P.out_outer + col;

    // Loads are unconditional (rows >= n re-read row n-1 and are zeroed by a select afterwards): with the
    // value used inside a branch the compiler waits after every single load.
    cx<T> v[VEC][S::E];
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      const int r = j + k * S::TPT;
      const cx<T>* src = ip + row_off(P.in_map, (unsigned)(r < n ? r : n - 1));
      if (nact >= VEC) {
        const GPack g = *reinterpret_cast<const GPack*>(src);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i][k] = g.e[i];
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i][k] = i < nact ? src[i] : mk<T>((T)0, (T)0);
      }
    }
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      const int r = j + k * S::TPT;
      const cx<T> w = P.chirp[r < n ? r : n - 1];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const cx<T> x = (INV ? swapri(v[i][k]) : v[i][k]) * w;
        v[i][k] = r < n ? x : mk<T>((T)0, (T)0);
      }
    }
    if constexpr (SPLIT) {
      XchSplitV<T, VEC, Slot> xch{reinterpret_cast<PackV<T, VEC>*>(lds), Slot{c}};
      chirpz_conv_v<S, T, VEC>(v, j, P.tw, P.bhat, xch);
    } else {
      XchFullV<T, VEC, Slot> xch{reinterpret_cast<PackV<cx<T>, VEC>*>(lds), Slot{c}};
      chirpz_conv_v<S, T, VEC>(v, j, P.tw, P.bhat, xch);
    }
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      const int r = j + k * S::TPT;
      if (r < n) {
        const cx<T> w = scale(P.chirp[r], P.scale);
        cx<T>* dst = op + row_off(P.out_map, (unsigned)r);
        if (nact >= VEC) {
          GPack g;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const cx<T> x = swapri(v[i][k]) * w;
            g.e[i] = INV ? swapri(x) : x;
          }
          *reinterpret_cast<GPack*>(dst) = g;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            if (i < nact) {
              const cx<T> x = swapri(v[i][k]) * w;
              dst[i] = INV ? swapri(x) : x;
            }
        }
      }
    }
  }
};

// ---------------------------------------------------------------------------
// contiguous-axis transforms.  KIND 0: c2c of runtime length P.n.
// Real rows of length P.n, full-length flavours (any n, any row pitch):
//   KIND 1: real -> half-complex (n/2+1 bins stored);
//   KIND 2: half-complex -> real (Hermitian extension built on load).
// Real rows of EVEN length and even pitch, half-length flavours (the n reals are n/2 complex values, one
// chirp-z of length n/2 plus the split pass of the radix kernels: half the convolution length):
//   KIND 3: real -> half-complex;  KIND 4: half-complex -> real.
// c2r ignores the imaginary parts of bin 0 and, for even n, bin n/2 (as pocketfft/FFTW do).
// ---------------------------------------------------------------------------
template <class S, typename T, int ROWS, int KIND, bool INV>
struct RowFftZ {
  static constexpr int THREADS = S::TPT * ROWS;
  static constexpr int PD = S::R(0);
  static constexpr int PLEN = padded_len<S::N, PD>();
  static constexpr int LDS_BYTES = (S::NP > 1 || KIND == 3) ? (int)(PLEN * ROWS * sizeof(cx<T>)) : 0;

  template <class PZ>
  static MFFT_D void body(const PZ& P, int bid, int tid, char* lds) {
    const int rl = tid / S::TPT;
    const int j = tid % S::TPT;
    cx<T>* xch = reinterpret_cast<cx<T>*>(lds) + rl * PLEN;
    const i64 row = (i64)bid * ROWS + rl;
    const bool active = row < P.nrows;
    const int nh = P.n / 2;
    const int n = KIND >= 3 ? nh : P.n;              // logical length of the complex chirp-z
    const i64 lrow = active ? row : P.nrows - 1;     // rows past the end re-read the last row, store nothing

    // unconditional loads (positions >= n re-read position n-1 and are zeroed by a select): see ColFftZ
    cx<T> v[S::E];
    cx<T> vm[KIND == 4 ? S::E : 1];                  // KIND 4: the mirrored bins X[n - r]
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      const int r = j + k * S::TPT;
      const int rc = r < n ? r : n - 1;
      if constexpr (KIND == 0) {
        const cx<T>* ip = static_cast<const cx<T>*>(P.in) + lrow * P.in_stride;
        v[k] = ip[rc];
      } else if constexpr (KIND == 1) {
        const T* ip = static_cast<const T*>(P.in) + lrow * P.in_stride;
        v[k] = mk<T>(ip[rc], (T)0);
      } else if constexpr (KIND == 2) {
        const cx<T>* ip = static_cast<const cx<T>*>(P.in) + lrow * P.in_stride;
        v[k] = ip[rc <= nh ? rc : n - rc];
      } else if constexpr (KIND == 3) {              // (x[2r], x[2r+1]) as one complex value
        const cx<T>* ip = reinterpret_cast<const cx<T>*>(static_cast<const T*>(P.in) + lrow * P.in_stride);
        v[k] = ip[rc];
      } else {
        const cx<T>* ip = static_cast<const cx<T>*>(P.in) + lrow * P.in_stride;
        v[k] = ip[rc];
        vm[k] = ip[n - rc];
      }
    }
#pragma unroll
    for (int k = 0; k < S::E; ++k) {
      const int r = j + k * S::TPT;
      const int rc = r < n ? r : n - 1;
      cx<T> x = v[k];
      if constexpr (KIND == 0) {
        if (INV) x = swapri(x);
      } else if constexpr (KIND == 2) {
        if (r > nh) x = conj(x);              // Hermitian extension
        if (r == 0 || 2 * r == n) x.y = (T)0;
        x = swapri(x);                        // inverse through the swap identity
      } else if constexpr (KIND == 4) {
        // Z[r] = (X[r] + conj X[n-r]) + i conj(w_r) (X[r] - conj X[n-r])   (twice the textbook value)
        cx<T> xm = conj(vm[k]);
        if (r == 0) { x.y = (T)0; xm.y = (T)0; }
        x = swapri((x + xm) + mul_pi((x - xm) * conj(P.rtw[rc])));
      }
      x = x * P.chirp[rc];
      v[k] = r < n ? x : mk<T>((T)0, (T)0);
    }
    XchFull<T, PadSlot<PD>> xc{xch, PadSlot<PD>{}};
    chirpz_conv<S, T>(v, j, P.tw, P.bhat, xc);
    if constexpr (KIND == 3) {
      // split pass: X[r] = E[r] + w_r O[r], E = (Z[r] + conj Z[n-r])/2, O = -i (Z[r] - conj Z[n-r])/2; partner via LDS
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int r = j + k * S::TPT;
        v[k] = swapri(v[k]) * P.chirp[r < n ? r : n - 1];
      }
      if constexpr (S::NP > 1) MFFT_BARRIER();       // everyone finished the last gather of the convolution
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int r = j + k * S::TPT;
        if (r < n) xc.put(r, v[k]);
      }
      MFFT_BARRIER();
      if (active) {
        cx<T>* op = static_cast<cx<T>*>(P.out) + row * P.out_stride;
        const T half = (T)0.5;
#pragma unroll
        for (int k = 0; k < S::E; ++k) {
          const int r = j + k * S::TPT;
          if (r == 0) {
            op[0] = mk<T>((v[k].x + v[k].y) * P.scale, (T)0);
            op[n] = mk<T>((v[k].x - v[k].y) * P.scale, (T)0);
          } else if (r < n) {
            const cx<T> zm = conj(xc.get(n - r));
            const cx<T> e = scale(v[k] + zm, half);
            const cx<T> o = mul_mi(scale(v[k] - zm, half));
            op[r] = scale(e + P.rtw[r] * o, P.scale);
          }
        }
      }
      return;
    }
    if (active) {
#pragma unroll
      for (int k = 0; k < S::E; ++k) {
        const int r = j + k * S::TPT;
        if (r < n) {
          const cx<T> x = swapri(v[k]) * scale(P.chirp[r], P.scale);
          if constexpr (KIND == 0) {
            cx<T>* op = static_cast<cx<T>*>(P.out) + row * P.out_stride;
            op[r] = INV ? swapri(x) : x;
          } else if constexpr (KIND == 1) {
            cx<T>* op = static_cast<cx<T>*>(P.out) + row * P.out_stride;
            if (r <= nh) op[r] = x;
          } else if constexpr (KIND == 2) {
            T* op = static_cast<T*>(P.out) + row * P.out_stride;
            op[r] = x.y;                      // Re of the un-swapped value
          } else {
            cx<T>* op = reinterpret_cast<cx<T>*>(static_cast<T*>(P.out) + row * P.out_stride);
            op[r] = swapri(x);                // (x[2r], x[2r+1])
          }
        }
      }
    }
  }
};

}  // namespace mfft
