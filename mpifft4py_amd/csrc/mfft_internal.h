// mfft_internal.h -- internal C++ interfaces of libmpifft4py_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/mpifft4py_amd.h"
#include "registry.h"

namespace mfft {

// ---- errors ------------------------------------------------------------------
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
const char* last_error();

#define MFFT_HIP(call)                                                                   \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return ::mfft::set_error(MFFT_ERR_HIP, "%s failed: %s (%s:%d)", #call,             \
                               hipGetErrorString(e_), __FILE__, __LINE__);               \
  } while (0)

#define MFFT_TRY(call)                  \
  do {                                  \
    int rc_ = (call);                   \
    if (rc_ != 0) return rc_;           \
  } while (0)

inline size_t elem_bytes(int prec, bool complex_) { return (prec == MFFT_DOUBLE ? 8 : 4) * (complex_ ? 2 : 1); }

// ---- kernel launch layer -------------------------------------------------------
struct RowSpec {       // row r -> (r / split) * hi + (r % split) * lo   (elements)
  int64_t hi = 0, lo = 0, split = 0;   // split <= 0: single group
};

struct ColArgs {
  const void* in = nullptr;
  void* out = nullptr;
  int n = 0;             // transform length
  int prec = MFFT_DOUBLE;
  bool inverse = false;
  int64_t in_outer = 0, out_outer = 0;
  RowSpec in_rows, out_rows;
  int64_t ncols = 0;     // contiguous columns per outer batch
  int64_t nouter = 1;
  double scale = 1.0;
  int remap = 1;         // XCD-aware tile order
  bool allow_nt = true;  // use the non-temporal variant when the layout is 128-byte aligned
  int pad = 0;           // 1: input has 2n/3 physical rows (zero band skipped); 2: output truncated to 2n/3 rows
  bool fold = false;     // pad == 2: sum the two Nyquist rows (R2C convention)
  int64_t in_wrap = 0, in_wrap_gap = 0;   // wrapped input columns (fft_kernels.h ColParams::in_wrap): column c is read from
                                          // c + (c / in_wrap) * in_wrap_gap; radix kernels only
  int thirds = -1;       // pad == 1 inverse: one third of a tile's transform per workgroup (ColFft3S)?  -1: launch_col's rule
                         // (single precision, or a pass without an outer batch), 1: yes where the kernel exists, 0: no
  const uint8_t* mask = nullptr;   // inverse transforms: one byte per element of `in` (same element offsets), 0 = reads as zero
                                   // (the 2/3-rule `fu * dealias` of slab.py:237-245 without a masked copy of the spectrum)
  // inverse transforms, 2/3-rule mask of band form (fft_kernels.h ColParams b_*): rows [row_lo, row_hi) are zero and
  // not loaded, tiles whose columns are all removed are skipped
  struct Band {
    bool on = false;
    int row_lo = 0, row_hi = 0, c_off = 0, c_per = 1 << 30, c_lim = 1 << 30, g_off = 0, g_step = 0, g_lo = 1 << 30, g_hi = 1 << 30;
    const int* tile_list = nullptr;     // device array of the tiles with a kept column (tile = outer * ntile_c + tc), or null
    int ntiles_listed = 0;
    int g_zero = 0;                     // 1: columns of a removed y are transformed as zeros and written instead of skipped;
                                        // 2: columns of a removed z as well, no tile is skipped (complete output)
  } band;
};
bool c2r_limit_supported(int64_t n, int prec);   // a c2r kernel of real length n that reads only the first `valid` bins exists
bool band_fusable(int64_t n, int prec);
int col_tile_width(int64_t n, int prec, bool inverse, int pad_code);   // columns per tile of the strided kernel that would run   // a pruned (band) strided inverse kernel of length n exists
int launch_col(const ColArgs& a, hipStream_t s);
bool mask_fusable(int64_t n, int prec);   // a strided inverse kernel of length n that applies a mask on load exists

// complex side of a contiguous-axis transform split into z chunks (fft_kernels.h, ZSplit): the pack / unpack of the
// pencils' z-splitting exchange fused into the transform.  nchunk = 0: plain rows.
struct ZSplitArgs {
  int nchunk = 0;
  int64_t q = 0, last_len = 0;       // chunk length; length of the last chunk
  int64_t rows_total = 0, row0 = 0;  // rows of every chunk block; first row of this launch inside them
  int64_t pitch = 0, last_pitch = 0; // elements between the rows of a block / of the last block; 0 = its chunk length
};

struct RowArgs {
  const void* in = nullptr;
  void* out = nullptr;
  int n = 0;
  int prec = MFFT_DOUBLE;
  bool inverse = false;
  int64_t in_stride = 0, out_stride = 0, nrows = 0;
  double scale = 1.0;
  ZSplitArgs zs;         // forward: of `out`, inverse: of `in`
};
int launch_row(const RowArgs& a, hipStream_t s);

struct RealArgs {
  const void* in = nullptr;
  void* out = nullptr;
  int n = 0;             // REAL length
  int prec = MFFT_DOUBLE;
  int64_t in_stride = 0, out_stride = 0, nrows = 0;   // in elements of the respective types
  double scale = 1.0;
  int valid = 0;         // complex columns present in memory (0 = all n/2+1)
  ZSplitArgs zs;         // of the complex side
};
int launch_r2c(const RealArgs& a, hipStream_t s);
int launch_c2r(const RealArgs& a, hipStream_t s);

// fused nonlinear z stage (fft_nlz.h): rows of half-spectra of two vector fields in, rows of the half-spectra of their
// cross product out (may alias the inputs row for row)
struct NlzArgs {
  const void* a[3] = {nullptr, nullptr, nullptr};
  const void* b[3] = {nullptr, nullptr, nullptr};
  void* out[3] = {nullptr, nullptr, nullptr};
  int n = 0;             // REAL length of a z row
  int prec = MFFT_DOUBLE;
  int64_t in_stride = 0, out_stride = 0, nrows = 0;   // complex elements
  int valid = 0;         // bins per row present in memory (0 = all n/2+1)
  int valid_in = 0;      // bins per INPUT row, where fewer than `valid` exist (pruned 2/3-rule); 0 = valid
  double scale = 1.0;    // applied to the product (1 / n^2: both inverse transforms normalised)
};
bool nlz_supported(int64_t n, int prec);
int launch_nlz(const NlzArgs& a, hipStream_t s);

// strided 3-D box copy: dst[i][j][k] = src[i][j][k] over extents e0,e1,e2 with
// element strides (k contiguous); elem = bytes per element (8 or 16)
struct BoxArgs {
  const void* src = nullptr;
  void* dst = nullptr;
  int64_t e0 = 1, e1 = 1, e2 = 1;
  int64_t s0 = 0, s1 = 0;     // src strides of dims 0,1 (dim 2 contiguous)
  int64_t d0 = 0, d1 = 0;     // dst strides
  int elem = 16;
  int mode = 0;               // 0 copy, 1 accumulate (dst += src)
  double scale = 1.0;         // applied to src (complex/real agnostic)
  int prec = MFFT_DOUBLE;
};
int launch_box_copy(const BoxArgs& a, hipStream_t s);
// rows (nrows, pitch) of complex values: col0 <- (Re col0 - Im colN, 0), colN <- (Re colN, 0)   (line.py:231, 27-39)
int launch_line_nyquist(void* rows, int64_t nrows, int64_t pitch, int64_t coln, int prec, hipStream_t s);
int launch_mask(void* fu, const uint8_t* mask, size_t count, int prec, hipStream_t s);
int launch_scale(void* data, size_t count_real, double scale, int prec, hipStream_t s);
int launch_fill_uniform(void* data, size_t count, int prec, uint64_t seed, hipStream_t s);

bool length_supported(int64_t n, bool real_transform);
int length_route(int64_t n, bool real_transform, int prec = MFFT_DOUBLE);   // 1 radix plan, 2 one-workgroup chirp-z, 3 bigfft.hip, 0 none
// bigfft.hip: any length up to MFFT_BIG_MAX_LENGTH through Bluestein's convolution over a four-step power-of-two transform in
// a scratch buffer (the fallback behind the radix plans and the one-workgroup chirp-z kernels); plain transforms only
#define MFFT_BIG_MAX_LENGTH (1 << 20)
bool big_length_ok(int64_t n);
void big_release_stream(hipStream_t s);   // frees the scratch buffer bigfft.hip keeps for a stream (plan destruction)
bool radix_plan_exists(int contiguous, int64_t n, int prec);   // a plain strided (0) / contiguous-axis c2c (1) radix kernel of length n
int big_col(const ColArgs& a, hipStream_t s);
int big_row(const RowArgs& a, hipStream_t s);
int big_real(bool c2r, const RealArgs& a, hipStream_t s);
bool zsplit_supported(int64_t n, int prec, bool real_transform);   // radix kernels with fused z-chunk pack exist for n
bool zsplit_limit_supported(int64_t n, int prec);                  // ... that are column-limited as well (3/2-rule pencils)
hipStream_t plan_stream(mfft_plan_t plan);   // the plan's compute stream (nullptr plan -> default stream)

}  // namespace mfft
