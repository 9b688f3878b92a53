// plan.hip -- slab / pencil 3-D transform executor behind mfft_plan_t.
//
// Restates, for device-resident data and HIP kernels, the stage ordering of
//   slab  R2C/C2C : mpiFFT4py/slab.py:349-443 (fftn), 214-308 (ifftn), 743-772, 638-669
//   pencil R2CY   : mpiFFT4py/pencil.py:730-754 (fftn), 483-507 (ifftn)
//   pencil R2CX   : mpiFFT4py/pencil.py:1312-1337 (fftn), 1082-1105 (ifftn)
//   3/2-rule      : slab.py:250-268, 310-344, 372-386, 445-483; pencil.py:604-632,
//                   858-883, 1196-1224, 1440-1475
// with the pack / unpack copies (slab.py:403, cython/maths.pyx:21-31 and the
// Alltoallw sub-array types) folded into the strided FFT kernels' two-level row
// addressing wherever the split axis is not the contiguous one.
#include <algorithm>
#include <cmath>
#include <memory>
#include <cstring>
#include <string>
#include "comm.h"
#include "mfft_internal.h"
#include "relay_plan.h"

#ifndef MFFT_FWD_OOP_DEFAULT
#define MFFT_FWD_OOP_DEFAULT 0      // one-rank forward y / x passes out of place: see mfft_plan_s::fwd_out_of_place
#endif
#ifndef MFFT_P1_XPAD_DEFAULT
#define MFFT_P1_XPAD_DEFAULT 0      // one-rank real transforms: cache lines added to slow plane pitches of the intermediate (p1_plane_pad; measured: no gain)
#endif

using namespace mfft;

namespace {

#ifndef MFFT_DEFAULT_COMM_CUS
#define MFFT_DEFAULT_COMM_CUS 0
#endif

struct StageTimer {
  std::string name;
  double alg_bytes = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  double total_ms = 0;
  int64_t calls = 0;
};

struct Chunk {
  int64_t len, start;
};

struct GraphEntry {            // a captured transform: same direction, buffers and dealias mode
  bool forward;
  const void* in;
  void* out;
  int dealias;
  hipGraphExec_t exec;
};

struct Sched {                 // one all-to-all-v inside a group, bytes
  std::vector<int> peers;
  std::vector<size_t> sc, sd, rc, rd;
  std::vector<int> part;       // pencils: group id of EVERY rank for this exchange (all groups exchange at once); empty: none
};

std::vector<Chunk> pencil_chunks(int64_t n, int size) {   // pencil.py:80-90
  std::vector<Chunk> c(size);
  const int64_t q = n / size, r = n % size;
  for (int i = 0; i < size; ++i) c[i] = Chunk{q + ((r == 1 && i == size - 1) ? 1 : 0), q * i};
  return c;
}

void compute_dims(int n, int* p1, int* p2) {   // MPI.Compute_dims(n, 2): balanced, non-increasing
  int best1 = n, best2 = 1;
  for (int a = 1; a * a <= n; ++a)
    if (n % a == 0) {
      best1 = n / a;
      best2 = a;
    }
  *p1 = best1;
  *p2 = best2;
}

}  // namespace

struct mfft_plan_s {
  mfft_comm_s* comm = nullptr;
  mfft_plan_desc d;
  int P = 1, rank = 0, dev = 0;
  hipStream_t stream = nullptr;
  int prec = MFFT_DOUBLE;
  bool r2c = true;
  int64_t N0 = 0, N1 = 0, N2 = 0, Nf = 0;
  size_t es = 16, rs = 8;       // bytes per complex / per "real-space" element
  // slab
  int64_t Np0 = 0, Np1 = 0;
  // pencil
  int P1 = 1, P2 = 1, c0 = 0, c1 = 0;
  int64_t N1_0 = 0, N1_1 = 0, N2_0 = 0, N2_1 = 0;   // N0/P1, N1/P1, N0/P2, N1/P2
  std::vector<int> group0, group1, world;
  std::vector<Chunk> zc;        // z chunks of the first exchange
  int64_t q = 0, zstart = 0;    // my z extent in spectral space
  // 3/2-rule
  int64_t M0 = 0, M1 = 0, M2 = 0, Mf = 0;
  void* work3 = nullptr;        // y output of the pipelined inverse
  size_t work3_bytes = 0;
  void* work[3] = {nullptr, nullptr, nullptr};
  size_t work_bytes[3] = {0, 0, 0};
  uint8_t* mask = nullptr;
  size_t mask_count = 0;
  bool timing = false;
  std::vector<StageTimer> timers;
  bool use_graphs = false;      // single rank, small mesh: replay captured hipGraphs
  std::vector<GraphEntry> graphs;
  // exchange pipeline (slab, P > 1): kz slices, a communication stream and events
  int nslice = 1;
  int nbatch = 1;               // pencils: batches of rows pipelined through the exchanges (X: both together, Y: one after the other)
  std::vector<hipEvent_t> ev2_compute, ev2_comm;
  hipStream_t cstream = nullptr;
  int comm_cus = 0;             // CUs reserved for the communication stream (0: no CU masks)
  std::vector<hipEvent_t> ev_compute, ev_comm;
  std::vector<Chunk> kslice;    // (len, start) of each kz slice

  ~mfft_plan_s() {
    for (void* w : work)
      if (w) (void)wfree(w);
    if (mask) (void)hipFree(mask);
    if (band_tiles) (void)hipFree(band_tiles);
    if (work3) (void)wfree(work3);
    for (void* b : {nlx, nly, nlr, pcomp})
      if (b) (void)dev_free(b);
    for (void* b : nlw)
      if (b) (void)wfree(b);
    for (auto& t : timers) {
      for (auto& e : t.pending) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
      for (auto& e : t.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    }
    drop_graphs();
    for (hipEvent_t e : ev_compute) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev_comm) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev2_compute) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev2_comm) (void)hipEventDestroy(e);
    if (cstream) { big_release_stream(cstream); (void)hipStreamDestroy(cstream); }
    if (stream) { big_release_stream(stream); (void)hipStreamDestroy(stream); }
  }

  int walloc(void** p, size_t bytes) { return comm ? comm->work_alloc(p, bytes) : dev_alloc(p, bytes); }
  int wfree(void* p) { return comm ? comm->work_free(p) : dev_free(p); }

  void drop_graphs() {
    for (auto& g : graphs)
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
    graphs.clear();
  }

  int ensure_work(int i, size_t bytes) {
    if (work_bytes[i] >= bytes) return 0;
    drop_graphs();               // captured sequences hold the old buffer address
    if (work[i]) MFFT_TRY(wfree(work[i]));
    work[i] = nullptr;
    work_bytes[i] = 0;
    MFFT_TRY(walloc(&work[i], bytes));         // from the communicator: work buffers are what exchanges send from
    work_bytes[i] = bytes;
    return 0;
  }

  StageTimer* timer(const char* name, double alg_bytes) {
    for (auto& t : timers)
      if (t.name == name) return &t;
    timers.emplace_back();
    timers.back().name = name;
    timers.back().alg_bytes = alg_bytes;
    return &timers.back();
  }

  template <class F>
  int stage(const char* name, double alg_bytes, F f) { return stage_on(stream, name, alg_bytes, f); }

  template <class F>
  int stage_on(hipStream_t stream, const char* name, double alg_bytes, F f) {
    if (!timing) return f();
    // NOTE: pointers into `timers` are not kept across calls (vector may grow)
    StageTimer* t = timer(name, alg_bytes);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!t->pool.empty()) {
      ev = t->pool.back();
      t->pool.pop_back();
    } else {
      MFFT_HIP(hipEventCreate(&ev.first));
      MFFT_HIP(hipEventCreate(&ev.second));
    }
    MFFT_HIP(hipEventRecord(ev.first, stream));
    int rc = f();
    MFFT_HIP(hipEventRecord(ev.second, stream));
    t = timer(name, alg_bytes);
    t->pending.push_back(ev);
    return rc;
  }

  int collect_timing() {
    for (auto& t : timers) {
      for (auto& e : t.pending) {
        MFFT_HIP(hipEventSynchronize(e.second));
        float ms = 0;
        MFFT_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        t.total_ms += ms;
        t.calls += 1;
        t.pool.push_back(e);
      }
      t.pending.clear();
    }
    return 0;
  }

  // ---- kernel helpers (all on this->stream) -----------------------------------
  int r2c_rows(const void* in, void* out, int64_t nrows, int64_t n, int64_t in_stride, int64_t out_stride, double scale = 1.0,
               int valid = 0) {
    RealArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = prec; a.in_stride = in_stride; a.out_stride = out_stride;
    a.nrows = nrows; a.scale = scale; a.valid = valid;
    return launch_r2c(a, stream);
  }
  int c2r_rows(const void* in, void* out, int64_t nrows, int64_t n, int64_t in_stride, int64_t out_stride, double scale,
               int valid = 0) {
    RealArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = prec; a.in_stride = in_stride; a.out_stride = out_stride;
    a.nrows = nrows; a.scale = scale; a.valid = valid;
    return launch_c2r(a, stream);
  }
  int c2c_rows(const void* in, void* out, int64_t nrows, int64_t n, int64_t in_stride, int64_t out_stride, bool inv, double scale) {
    RowArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = prec; a.inverse = inv; a.in_stride = in_stride;
    a.out_stride = out_stride; a.nrows = nrows; a.scale = scale;
    return launch_row(a, stream);
  }
  // z-axis stage of the forward / backward transform (real or complex flavour)
  int z_forward(const void* in, void* out, int64_t nrows, int64_t nz, int64_t nzf) {
    if (r2c) return r2c_rows(in, out, nrows, nz, nz, nzf);
    return c2c_rows(in, out, nrows, nz, nz, nzf, false, 1.0);
  }
  int z_backward(const void* in, void* out, int64_t nrows, int64_t nz, int64_t nzf) {
    if (r2c) return c2r_rows(in, out, nrows, nz, nzf, nz, 1.0 / (double)nz);
    return c2c_rows(in, out, nrows, nz, nzf, nz, true, 1.0 / (double)nz);
  }
  // z stage with the z-chunk pack / unpack of the pencils fused in (fft_kernels.h, ZSplit): rows [row0, row0 + nrows)
  // of the Pz blocks (rows_total, len_l) that the z-splitting exchange sends / has received
  bool zfuse = false;
  bool xpad_on = true;          // xplane_pad(): MFFT_NO_XPAD=1 clears it (A/B runs; must be the same on every rank)
  bool zpitch_on = true;        // zrow_pitch(): MFFT_NO_ZPITCH=1 clears it (likewise)
  int pad_align = -1;           // pad_pitch(): MFFT_PAD_ALIGN = 0 never, 1 always, unset: where it was measured to pay
  int pad_align_inv = 1;        // inverse flavour of that route (MFFT_PAD_ALIGN_INV = 1 | 2 | 3, see slab_backward_padded_fused)
  bool xpass_inplace = false;   // MFFT_XPASS_INPLACE=1: the x pass behind an exchange runs in place on the receive buffer (rounds 1 - 3)
  // the same for the fused 3/2-rule pencil transforms: real length M2, the Nf kept columns split into the z chunks
  bool zfuse_pad() const {
    return getenv("MFFT_NO_ZFUSE") == nullptr && !d.line2d && !d.drop_nyquist && !zc.empty() && zc[0].len < 65536 &&
           M2 % 2 == 0 && zsplit_limit_supported(M2, prec);
  }
  // Row pitch of a z chunk of `len` columns in the blocks of the FORWARD z-splitting exchange of the Y-ALIGNED pencil
  // (round 4).  The rank that holds the Nyquist column has q = 129 (257 ...) columns; its x pass -- in place on the received
  // (N0, N1/P2, q) -- then reads rows N1/P2 * q elements apart: 512 * 129 * 16 bytes = 2^20 + 2^13 at 1024^3 on the 4 x 2 grid,
  // the slowest pitch there is (xplane_pad: 0.64 against 0.47 ms, profiles/r04_rank_shapes.txt), and the slowest rank sets
  // the pace of the transform.  The z kernel therefore writes such rows a whole number of cache lines apart (fft_kernels.h
  // ZSplit pitch: 129 -> 136 columns, the chunk of that one destination grows by 5 %), the x pass runs over N1/P2 * 136
  // columns (the unused ones ride along), the pitch stays through the second exchange and the y pass reads it.
  // NOT for the x-aligned pencil: there only the y pass would see the pitch, and a strided pass that reads line-aligned
  // rows but must write compact ones is SLOWER than compact -> compact (scripts/ypass_pitch_ab.py, profiles/r04_ypass_pitch.txt:
  // (256, 1024, 257) 0.49 -> 0.56 ms; aligned on both sides it would be 0.41, but the result's layout is the caller's) --
  // the y-aligned plan pays the same 0.04 ms in its y pass and wins 0.17 in the x pass.  Only the fused z kernels, only
  // chunks of 64 columns and more, only forward.
  // Round 5, the x-aligned pencil after all -- for the one case where its y pass gains: chunks whose rows are a multiple of 2^13
  // bytes (BASELINE config 5: 1024 complex64 columns per rank of the 4 x 2 grid).  Such rows are line-aligned already; what
  // hurts is that the 2048 rows a y transform gathers then lie a power of two apart (the memory-channel hash folds them onto
  // few channels): (512, 2048, 1024) axis 1 takes 4.10 ms per rank, 3.82 with the rows one cache line further apart
  // (profiles/r04_rank_shapes.txt).  The z kernel leaves that line between its rows, the chunk grows by 1.6 %, the y pass
  // reads the pitch and writes the compact blocks of the second exchange as before (a store-side pitch costs nothing).
  int64_t zrow_pitch(int64_t len, bool forward) const {
    if (!forward || !zpitch_on || !zfuse || d.drop_nyquist || zc.size() < 2 || len < 64) return len;
    const int64_t per_line = (int64_t)(128 / es);
    if (d.decomp == MFFT_PENCIL_Y) return (len + per_line - 1) / per_line * per_line;
    if (d.decomp == MFFT_PENCIL_X && (len * (int64_t)es) % 8192 == 0) return len + per_line;
    return len;
  }
  int64_t zsend_elems(int64_t rows) const {      // elements of the forward z exchange's send blocks for `rows` rows
    int64_t t = 0;
    for (const Chunk& c : zc) t += rows * zrow_pitch(c.len, true);
    return t;
  }
  ZSplitArgs zsplit(int64_t rows_total, int64_t row0, bool forward = false) const {
    ZSplitArgs z;
    z.nchunk = (int)zc.size(); z.q = zc[0].len; z.last_len = zc.back().len; z.rows_total = rows_total; z.row0 = row0;
    z.pitch = zrow_pitch(z.q, forward); z.last_pitch = zrow_pitch(z.last_len, forward);
    return z;
  }
  int z_forward_chunked(const void* in, void* blocks, int64_t nrows, int64_t row0, int64_t rows_total) {
    if (r2c) {
      RealArgs a;
      a.in = in; a.out = blocks; a.n = (int)N2; a.prec = prec; a.in_stride = N2; a.out_stride = Nf; a.nrows = nrows; a.scale = 1.0;
      a.zs = zsplit(rows_total, row0, true);
      return launch_r2c(a, stream);
    }
    RowArgs a;
    a.in = in; a.out = blocks; a.n = (int)N2; a.prec = prec; a.inverse = false; a.in_stride = N2; a.out_stride = Nf; a.nrows = nrows;
    a.scale = 1.0; a.zs = zsplit(rows_total, row0, true);
    return launch_row(a, stream);
  }
  int z_backward_chunked(const void* blocks, void* out, int64_t nrows, int64_t row0, int64_t rows_total) {
    if (r2c) {
      RealArgs a;
      a.in = blocks; a.out = out; a.n = (int)N2; a.prec = prec; a.in_stride = Nf; a.out_stride = N2; a.nrows = nrows;
      a.scale = 1.0 / (double)N2; a.zs = zsplit(rows_total, row0);
      return launch_c2r(a, stream);
    }
    RowArgs a;
    a.in = blocks; a.out = out; a.n = (int)N2; a.prec = prec; a.inverse = true; a.in_stride = Nf; a.out_stride = N2; a.nrows = nrows;
    a.scale = 1.0 / (double)N2; a.zs = zsplit(rows_total, row0);
    return launch_row(a, stream);
  }
  int col(const void* in, void* out, int64_t n, bool inv, int64_t nouter, int64_t ncols, int64_t in_outer, RowSpec in_rows,
          int64_t out_outer, RowSpec out_rows, double scale = 0.0) {
    ColArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = prec; a.inverse = inv; a.nouter = nouter; a.ncols = ncols;
    a.in_outer = in_outer; a.out_outer = out_outer; a.in_rows = in_rows; a.out_rows = out_rows;
    a.scale = scale != 0.0 ? scale : (inv ? 1.0 / (double)n : 1.0);
    // 2/3-rule (fuse_mask below): a pass that reads the caller's spectrum applies the dealias mask while it loads
    if (mask_src && in >= mask_src && static_cast<const char*>(in) < static_cast<const char*>(mask_src) + (size_t)local_complex_alloc_native() * es) {
      const int64_t off = (static_cast<const char*>(in) - static_cast<const char*>(mask_src)) / (int64_t)es;
      if (lband_use) {           // pencils, the reference's own filter: its three 1-D conditions instead of the bytes
        a.band = local_band(d.decomp == MFFT_PENCIL_Y ? off / (N1 * q) : 0);
        a.scale = scale != 0.0 ? scale : 1.0 / (double)n;
      } else {
        a.mask = mask + off;
      }
    }
    return launch_col(a, stream);
  }
  // `fu * dealias` of the reference's ifftn (slab.py:237-245, pencil.py:455-462) without the masked copy: when the first
  // inverse pass (length first_len, reading fu) has a masked-load kernel, remember fu and let col() hand the mask down.
  // Returns false when the copy is needed after all (chirp-z lengths, unit axes, MFFT_NO_MASK_FUSION=1).
  // The 2/3-rule's own mask (get_dealias_filter: three 1-D conditions |k| < kmax) recognised when it is set: x and y
  // keep [0, a) and [b, N), z keeps [0, a2).  One GPU, real data: the inverse then never loads the removed rows, skips
  // the tiles of removed columns and reads a2 bins per z row (pruned passes).
  bool band_ok = false;
  bool band_allzero = false;    // P > 1: every ky of this rank is removed (its x pass is a memset)
  int ba0 = 0, bb0 = 0, ba1 = 0, bb1 = 0, ba2 = 0;
  int col_band(const void* in, void* out, int64_t n, int64_t nouter, int64_t ncols, int64_t in_outer, RowSpec in_rows,
               int64_t out_outer, RowSpec out_rows, const ColArgs::Band& b) {
    ColArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = prec; a.inverse = true; a.nouter = nouter; a.ncols = ncols;
    a.in_outer = in_outer; a.out_outer = out_outer; a.in_rows = in_rows; a.out_rows = out_rows;
    a.scale = 1.0 / (double)n;
    a.band = b;
    a.band.on = true;
    return launch_col(a, stream);
  }
  void detect_band(const uint8_t* m) {
    band_ok = false;
    band_allzero = false;
    // every rank must take the same route (the pruned exchange has other counts): agree on the outcome below.
    // status: 0 = not a band mask (or no kernels), 1 = band mask, 2 = this rank's local mask is all zeros -- its ky range
    // lies wholly inside the removed band (1024^3 over 8 ranks: ky in [342, 683) covers ranks 3 and 4) -- which is
    // compatible with whatever band the others see: it adopts their (a0, b0, a2) and contributes zeros.
    int st = 0, a0 = 0, b0 = 0, a1 = 0, b1 = 0, a2 = 0;
    std::vector<int> list;
    if (d.decomp == MFFT_SLAB && r2c && N0 >= 2 && N1 >= 2 && N2 >= 4 && N2 % 2 == 0 && band_fusable(N0, prec) &&
        band_fusable(N1, prec) && c2r_limit_supported(N2, prec))
      st = analyse_band(m, &a0, &b0, &a1, &b1, &a2, &list);
    if (P > 1) {
      const double none = -1e18;                 // neutral element of the max-reduction
      double v[8] = {st == 0 ? 1.0 : 0.0, st == 1 ? 1.0 : 0.0, none, none, none, none, none, none};
      if (st == 1) { v[2] = a0; v[3] = -a0; v[4] = b0; v[5] = -b0; v[6] = a2; v[7] = -a2; }
      if (comm->allreduce_host(v, 8, 1) != 0) return;
      if (v[0] != 0.0 || v[1] != 1.0 || v[2] != -v[3] || v[4] != -v[5] || v[6] != -v[7]) return;   // somebody disagrees, has another mask, or nobody has a band
      if (st == 2) {                             // all my ky are removed: [g_lo, g_hi) = every local ky
        a0 = (int)v[2]; b0 = (int)v[4]; a2 = (int)v[6]; a1 = 0; b1 = (int)Np1;
        band_allzero = true;
      }
    } else if (st != 1) {
      return;
    }
    ba0 = a0; bb0 = b0; ba1 = a1; bb1 = b1; ba2 = a2;
    if (P == 1) {
      if (a1 < 1) return;                      // the y pass redirects removed rows to row 0, which must be a kept one
      if (band_tiles) (void)hipFree(band_tiles);
      band_tiles = nullptr;
      band_ntiles = (int)list.size();
      if (list.empty() || hipMalloc(reinterpret_cast<void**>(&band_tiles), list.size() * sizeof(int)) != hipSuccess ||
          hipMemcpy(band_tiles, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        return;
      }
    }
    band_ok = true;
  }
  // local mask (N0, Np1, Nf) == m0[kx] & m1[ky] & m2[kz] with the zeros of m0 and m1 one run each and those of m2 a tail?
  // 0: no, 1: yes, 2: the local mask is all zeros
  int analyse_band(const uint8_t* m, int* a0, int* b0, int* a1, int* b1, int* a2, std::vector<int>* list) const {
    const int64_t n1 = Np1;
    std::vector<uint8_t> m0(N0, 0), m1(n1, 0), m2(Nf, 0);
    for (int64_t i = 0; i < N0; ++i)
      for (int64_t j = 0; j < n1; ++j) {
        const uint8_t* row = m + (i * n1 + j) * Nf;
        uint8_t any = 0;
        for (int64_t k = 0; k < Nf; ++k) { any |= row[k]; m2[k] |= row[k]; }
        m0[i] |= any; m1[j] |= any;
      }
    for (auto* v : {&m0, &m1, &m2}) for (auto& x : *v) x = x ? 1 : 0;
    {
      bool any = false;
      for (uint8_t x : m2) any = any || x;
      if (!any) return 2;
    }
    for (int64_t i = 0; i < N0; ++i)          // the mask must BE the product of the three (values other than 0 / 1 are weights, not a filter)
      for (int64_t j = 0; j < n1; ++j) {
        const uint8_t* row = m + (i * n1 + j) * Nf;
        const uint8_t ij = m0[i] & m1[j];
        for (int64_t k = 0; k < Nf; ++k) if (row[k] != (uint8_t)(ij & m2[k])) return 0;
      }
    auto run = [](const std::vector<uint8_t>& v, int* a, int* b) {   // zeros form one run [a, b) (none: a = b = first index after the ones)
      const int n = (int)v.size();
      int lo = 0;
      while (lo < n && v[lo]) ++lo;
      int hi = lo;
      while (hi < n && !v[hi]) ++hi;
      for (int i = hi; i < n; ++i) if (!v[i]) return false;
      *a = lo; *b = hi;
      return true;
    };
    int z0 = 0, z1 = 0;
    if (!run(m0, a0, b0) || *a0 < 1 || !run(m1, a1, b1) || !run(m2, &z0, &z1) || z1 != (int)Nf || z0 < 1) return 0;
    *a2 = z0;
    if (P == 1) {      // x pass: tiles of the flattened (ky, kz) columns that hold a kept column, in memory order
      const int w = col_tile_width(N0, prec, true, 6);
      if (w <= 0) return 0;
      const int64_t ncols = n1 * Nf, ntile = (ncols + w - 1) / w;
      for (int64_t t = 0; t < ntile; ++t) {
        bool any = false;
        for (int64_t c = t * w; c < std::min(ncols, (t + 1) * w) && !any; ++c) any = m1[c / Nf] && m2[c % Nf];
        if (any) list->push_back((int)t);
      }
    }
    return 1;
  }
  // Pencils (R2C): the same recognition, LOCAL to the rank and without any change of layout -- the first inverse pass
  // (x for the X alignment, y for Y) runs the band kernel in its "complete output" mode (ColFft PAD == 4, b_gzero = 2)
  // instead of loading one mask byte per element: removed rows are not loaded, removed columns are transformed as zeros.
  bool lband_ok = false;
  int lb_row_lo = 0, lb_row_hi = 0, lb_g_lo = 0, lb_g_hi = 0, lb_c_lim = 0;
  void detect_band_local(const uint8_t* m) {
    lband_ok = false;
    if (d.decomp == MFFT_SLAB || !r2c || d.drop_nyquist || d.line2d) return;
    const bool X = d.decomp == MFFT_PENCIL_X;
    const int64_t D0 = X ? N0 : N2_0, D1 = X ? N1_1 : N1, D2 = q;
    if (D0 < 1 || D1 < 1 || D2 < 1 || !band_fusable(X ? N0 : N1, prec) || (X ? N0 : N1) < 2) return;
    std::vector<uint8_t> m0(D0, 0), m1(D1, 0), m2(D2, 0);
    for (int64_t i = 0; i < D0; ++i)
      for (int64_t j = 0; j < D1; ++j) {
        const uint8_t* row = m + (i * D1 + j) * D2;
        uint8_t any = 0;
        for (int64_t k = 0; k < D2; ++k) { any |= row[k]; m2[k] |= row[k]; }
        m0[i] |= any; m1[j] |= any;
      }
    for (auto* v : {&m0, &m1, &m2}) for (auto& x : *v) x = x ? 1 : 0;
    for (int64_t i = 0; i < D0; ++i)
      for (int64_t j = 0; j < D1; ++j) {
        const uint8_t* row = m + (i * D1 + j) * D2;
        const uint8_t ij = m0[i] & m1[j];
        for (int64_t k = 0; k < D2; ++k) if (row[k] != (uint8_t)(ij & m2[k])) return;      // not a product of 1-D filters
      }
    auto run = [](const std::vector<uint8_t>& v, int* a, int* b) {   // zeros form one run [a, b)
      const int n = (int)v.size();
      int lo = 0;
      while (lo < n && v[lo]) ++lo;
      int hi = lo;
      while (hi < n && !v[hi]) ++hi;
      for (int i = hi; i < n; ++i) if (!v[i]) return false;
      *a = lo; *b = hi;
      return true;
    };
    int a0, b0, a1, b1, z0, z1;
    if (!run(m0, &a0, &b0) || !run(m1, &a1, &b1) || !run(m2, &z0, &z1) || z1 != (int)D2) return;   // kz: a kept prefix
    if (X) { lb_row_lo = a0; lb_row_hi = b0; lb_g_lo = a1; lb_g_hi = b1; }
    else   { lb_row_lo = a1; lb_row_hi = b1; lb_g_lo = a0; lb_g_hi = b0; }
    lb_c_lim = z0;
    lband_ok = getenv("MFFT_NO_PRUNE") == nullptr || atoi(getenv("MFFT_NO_PRUNE")) == 0;
  }
  // first inverse pass of a pencil plan over rows [g0, g0 + nouter) of the g axis (X: one launch, the g axis is folded
  // into the columns; Y: batches of local kx rows)
  ColArgs::Band local_band(int64_t g0) const {
    ColArgs::Band b;
    b.on = true;
    b.row_lo = lb_row_lo; b.row_hi = lb_row_hi; b.c_lim = lb_c_lim; b.g_lo = lb_g_lo; b.g_hi = lb_g_hi; b.g_zero = 2;
    if (d.decomp == MFFT_PENCIL_X) { b.c_off = 0; b.c_per = (int)q; b.g_off = 0; b.g_step = 0; }
    else                           { b.c_off = 0; b.c_per = 1 << 30; b.g_off = (int)g0; b.g_step = 1; }
    return b;
  }
  int* band_tiles = nullptr;
  int band_ntiles = 0;
  const void* mask_src = nullptr;
  bool lband_use = false;       // this call's first pass takes the band kernel (set by fuse_mask, cleared with mask_src)
  int fuse_mask(const void* fu, int64_t first_len, bool* fused) {
    const size_t cnt = (size_t)local_complex_count();
    if (!mask || mask_count != cnt) return set_error(MFFT_ERR_INVALID, "2/3-rule requested but no dealias mask of %zu entries was set", cnt);
    const bool off = getenv("MFFT_NO_MASK_FUSION") && atoi(getenv("MFFT_NO_MASK_FUSION")) != 0;
    *fused = !off && first_len >= 2 && mask_fusable(first_len, prec);
    mask_src = *fused ? fu : nullptr;
    lband_use = *fused && lband_ok && d.decomp != MFFT_SLAB && !(getenv("MFFT_NO_PRUNE") && atoi(getenv("MFFT_NO_PRUNE")) != 0);
    return 0;
  }
  int col_pad(const void* in, void* out, int64_t n, bool inv, int pad, bool fold, int64_t nouter, int64_t ncols,
              int64_t in_outer, RowSpec in_rows, int64_t out_outer, RowSpec out_rows, double scale, int64_t in_wrap = 0,
              int64_t in_wrap_gap = 0, int thirds = -1) {
    ColArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = prec; a.inverse = inv; a.nouter = nouter; a.ncols = ncols;
    a.in_outer = in_outer; a.out_outer = out_outer; a.in_rows = in_rows; a.out_rows = out_rows;
    a.scale = scale; a.pad = pad; a.fold = fold;
    a.in_wrap = in_wrap; a.in_wrap_gap = in_wrap_gap; a.thirds = thirds;
    return launch_col(a, stream);
  }
  // One-rank fused 3/2-rule transforms (round 5): the two intermediates belong to the plan, so their z rows get a pitch of
  // whole cache lines (513 bins -> 520 in double precision: rows of 8208 bytes never start on a line, and a 128-byte tile
  // row then costs two lines on either side of the y pass).  The plain transform measured the same idea in round 4
  // (profiles/r04_ypass_pitch.txt: y pass 3.46 -> 2.96 ms with rows of 520 on both sides) and could not use it -- it has
  // one work buffer less and its x passes touch the caller's compact array on the wrong side; here the inverse x pass
  // stores whole lines per y row (its loads straddle), the y pass and the real transform see aligned rows, and the
  // forward x pass tiles the compact OUTPUT and wraps its input columns (ColParams::in_wrap).  MFFT_PAD_ALIGN=0: compact.
  // Measured (profiles/r05_pad_align_ab.txt, 3/2-rule pair): 1024^3 fp64 45.3 -> 43.9 ms (y passes 8.7 / 7.8 -> 6.9 / 7.0 ms, the x
  // passes give part of it back: their misaligned side costs 0.3 - 1.1 ms); 768^3 fp64 even; 512^3 fp64 and 1024^3 fp32 LOSE
  // 2 - 3 % (rows of 4 KiB: the y pass gains less than the x pass pays).  Default: double precision, rows of 8 KiB and more.
  int64_t pad_pitch() const {
    if (pad_align == 0 || P != 1) return Nf;
    if (pad_align < 0 && !(prec == MFFT_DOUBLE && Nf * (int64_t)es >= 8192)) return Nf;
    const int64_t line = 128 / (int64_t)es;
    return (Nf + line - 1) / line * line;
  }
  // A strided pass whose rows lie a multiple of 64 KiB apart reads 12 - 30 % slower than one whose rows are one 128-byte
  // line further apart (every row of a tile meets the same memory channels; profiles/r02_power_of_two_stride.txt);
  // the store side does not care.  Elements to add to such a row stride in an intermediate buffer, 0 when it is harmless.
  int64_t plane_pad(int64_t stride_elems) const {
    return (stride_elems * (int64_t)es) % 65536 == 0 ? (int64_t)(128 / es) : 0;
  }
  // The same idea carried THROUGH an exchange (round 4): the strided x pass that follows an exchange reads the received
  // chunks, whose x rows lie N1/P * Nf (slab; N1/P * kz in the kz-slice pipeline), N1/P1 * q (x-aligned pencil, forward) or
  // N1/P2 * q (y-aligned pencil, inverse) elements apart.  Measured alone on the device (profiles/r04_xpass_stride_map.txt,
  // r04_xpass_kernel_ab.txt; 1024 and 2048 rows, out of place, GB/s of algorithmic traffic against ~5000 for a pitch with
  // one more cache line): a power of two 4100 - 4700; 2^a + 2^(a-7) -- the Nyquist-holding ranks of the 4 x 2 pencil grid at
  // 1024^3: 512 * 129 elements -- 2100 - 3300 (the memory-channel hash folds address bits seven apart: every row of a tile
  // lands on the same channels); 2^a + 2^(a-8) 4100 - 4500 (256 * 257 elements); 2^20 + 2^11 (the slab over 8 ranks:
  // 128 * 513) 4600.  For those pitches the transform that WRITES the send blocks leaves one cache line between
  // consecutive x rows (a store-side pitch costs nothing), every chunk grows by that line per x row
  // (mfft_plan_exchange_schedule / _pieces report it: 64 KiB on a 2 GiB chunk at BASELINE config 5), and the x pass reads
  // the padded rows out of place into the caller's compact array: config 5's x pass 5.52 -> 4.46 ms per rank, the y-aligned
  // pencil's at 1024^3 0.63 -> 0.47.  MFFT_NO_XPAD=1 switches it off (every rank alike).
  // One rank (round 4): the x rows (planes) of a rank's own spectrum lie N1 * Nf elements apart -- for a power-of-two mesh
  // N * (N/2 + 1) * es = 2^a + 2^(a - log2 N + 1) bytes, which for N = 256 and 512 (N = 1024 in single precision) is one of
  // the pitches the strided x pass reads slowly (slow_pitch_pad).  The caller's array keeps its layout, so the route puts the
  // pass that READS it first or last and gives the intermediate planes `lines` cache lines more: forward y out of place into
  // padded planes, x out of them into the result; inverse y first (it reads rows, the plane pitch does not matter to it)
  // into padded planes, x out of them.  Complex data with power-of-two planes took this route since round 2 (plane_pad) and
  // keeps it.  For real data it is OFF: the x pass alone gains 4 - 10 % from the pad when it is timed by itself
  // (profiles/r04_xpass_stride_map.txt), inside the transform the pairs of 256^3, 512^3 (fp64, fp32) and 1024^3 fp32 come
  // out the same to +-1 % with 0 - 3 lines (profiles/r04_p1_xpad_ab.txt) -- the y pass that has to go first / out of place
  // gives back what the x pass wins.  MFFT_P1_XPAD = lines switches it on (read when a plan is created).
  int p1_xpad_lines = MFFT_P1_XPAD_DEFAULT;     // read when the plan is created
  int64_t p1_plane_pad() const {
    if (nat_pitch()) return slow_pitch_pad(N1 * Zp);      // pitched rows: 1024 x 520 x 16 B is a multiple of 64 KiB
    if (const int64_t c = r2c ? 0 : plane_pad(N1 * Nf)) return c;
    if (p1_xpad_lines <= 0 || d.line2d) return 0;
    return slow_pitch_pad(N1 * Nf) ? (int64_t)p1_xpad_lines * (int64_t)(128 / es) : 0;
  }
  int64_t slow_pitch_pad(int64_t stride_elems) const {
    const unsigned long long b = (unsigned long long)stride_elems * (unsigned long long)es;
    if (b < 65536) return 0;                              // small blocks live in the caches
    bool slow = b % 65536 == 0;
    if (!slow && __builtin_popcountll(b) == 2) {
      const int hi = 63 - __builtin_clzll(b), lo = __builtin_ctzll(b);
      slow = hi - lo == 7 || hi - lo == 8 || (hi - lo == 9 && hi <= 20);
    }
    return slow ? (int64_t)(128 / es) : 0;
  }
  int64_t xplane_pad(bool forward) const {                // one pitch for the whole exchange (not the kz-slice pipeline)
    if (!xpad_on || P == 1 || d.line2d || d.drop_nyquist) return 0;
    if (d.decomp == MFFT_SLAB) return (forward && nbatch <= 1 && nslice <= 1) ? slow_pitch_pad(Np1 * Nf) : 0;
    if (d.decomp == MFFT_PENCIL_X) return (forward && P1 > 1) ? slow_pitch_pad(N1_1 * q) : 0;
    return (!forward && P2 > 1) ? slow_pitch_pad(N2_1 * q) : 0;
  }
  // kz-slice pipeline of the slab: x-row pitch of slice s in the exchanged layout, and where the slice starts
  int64_t slice_pitch(int s, bool forward) const {
    const int64_t w = Np1 * kslice[s].len;
    return w + ((forward && xpad_on) ? slow_pitch_pad(w) : 0);
  }
  size_t slice_offset(int s, bool forward) const {        // elements of all earlier slices in the send / receive buffers
    size_t o = 0;
    for (int t = 0; t < s; ++t) o += (size_t)(P * Np0 * slice_pitch(t, forward));
    return o;
  }
  // One-rank forward transform: y and x passes out of place through a work buffer of the size of the spectrum instead
  // of in place on the result.  MFFT_FWD_OOP=1 / 0 forces it on / off; default: off (measured).
  // When on by default it would still need room: the buffer exists already (the inverse uses the same one), or a
  // quarter of the free HBM covers it.
  bool fwd_out_of_place(size_t cbytes) {
    static const int mode = getenv("MFFT_FWD_OOP") ? atoi(getenv("MFFT_FWD_OOP")) : MFFT_FWD_OOP_DEFAULT;
    if (mode <= 0) return false;
    if (work_bytes[0] >= cbytes) return true;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
    return cbytes <= fr / 4;
  }
  static RowSpec plain(int64_t stride) { RowSpec r; r.lo = stride; r.hi = 0; r.split = 0; return r; }
  static RowSpec two_level(int64_t split, int64_t hi, int64_t lo) { RowSpec r; r.split = split; r.hi = hi; r.lo = lo; return r; }

  int exchange(const std::vector<int>& grp, const void* send, const std::vector<size_t>& sc, const std::vector<size_t>& sd,
               void* recv, const std::vector<size_t>& rc, const std::vector<size_t>& rd, hipStream_t on = nullptr) {
    return comm->alltoallv(send, sc.data(), sd.data(), recv, rc.data(), rd.data(), grp.data(), (int)grp.size(),
                           on ? on : stream, on && on != stream ? 1 : 0);
  }
  int exchange_equal(const std::vector<int>& grp, const void* send, void* recv, size_t chunk_bytes, hipStream_t on = nullptr) {
    const int n = (int)grp.size();
    std::vector<size_t> c(n, chunk_bytes), dsp(n);
    for (int i = 0; i < n; ++i) dsp[i] = (size_t)i * chunk_bytes;
    return exchange(grp, send, c, dsp, recv, c, dsp, on);
  }
  int box(const void* src, void* dst, int64_t e0, int64_t e1, int64_t e2, int64_t s0, int64_t s1, int64_t d0, int64_t d1,
          int mode = 0, double scale = 1.0) {
    BoxArgs b;
    b.src = src; b.dst = dst; b.e0 = e0; b.e1 = e1; b.e2 = e2; b.s0 = s0; b.s1 = s1; b.d0 = d0; b.d1 = d1;
    b.elem = (int)es; b.mode = mode; b.scale = scale; b.prec = prec;
    return launch_box_copy(b, stream);
  }
  int zero(void* p, size_t bytes) {
    MFFT_HIP(hipMemsetAsync(p, 0, bytes, stream));
    return 0;
  }

  // copy src (n along `axis`) into the zero-initialised padded dst (npad along
  // axis): low half to the front, high half to the back (slab.py:518-523).
  // shapes: src (a0, n, a2) -> dst (a0, npad, a2) viewed with the axis in the middle.
  int pad_axis(const void* src, void* dst, int64_t a0, int64_t n, int64_t npad, int64_t a2, double scale) {
    const char* s = static_cast<const char*>(src);
    char* dd = static_cast<char*>(dst);
    MFFT_TRY(zero(dst, (size_t)(a0 * npad * a2) * es));
    const int64_t h = n / 2;
    MFFT_TRY(box(s, dd, a0, 1, h * a2, n * a2, 0, npad * a2, 0, 0, scale));
    MFFT_TRY(box(s + (size_t)(h * a2) * es, dd + (size_t)((npad - (n - h)) * a2) * es, a0, 1, (n - h) * a2, n * a2, 0,
                 npad * a2, 0, 0, scale));
    return 0;
  }
  // truncation with Nyquist fold (slab.py:529-533): dst[:n/2+1] = src[:n/2+1]; dst[n/2:] += src[-n/2:]
  // src may have a longer contiguous run (a2s >= a2): only the first a2 are taken.
  int trunc_axis(const void* src, void* dst, int64_t a0, int64_t n, int64_t npad, int64_t a2, int64_t a2s, double scale,
                 bool fold = true) {
    const char* s = static_cast<const char*>(src);
    char* dd = static_cast<char*>(dst);
    const int64_t h = n / 2;
    if (!fold) {   // plain corner copies: dst[:n/2] = src[:n/2]; dst[n/2:] = src[-n/2:]   (slab.py:736-739, 796-797)
      MFFT_TRY(box(s, dd, a0, h, a2, npad * a2s, a2s, n * a2, a2, 0, scale));
      MFFT_TRY(box(s + (size_t)((npad - (n - h)) * a2s) * es, dd + (size_t)(h * a2) * es, a0, n - h, a2, npad * a2s, a2s,
                   n * a2, a2, 0, scale));
      return 0;
    }
    MFFT_TRY(zero(dst, (size_t)(a0 * n * a2) * es));
    MFFT_TRY(box(s, dd, a0, h + 1, a2, npad * a2s, a2s, n * a2, a2, 0, scale));
    MFFT_TRY(box(s + (size_t)((npad - h) * a2s) * es, dd + (size_t)(h * a2) * es, a0, h, a2, npad * a2s, a2s, n * a2, a2, 1, scale));
    return 0;
  }

  // 3/2-rule normalisation: padsize per padded axis (slab.py:256, 330; line.py:184, 287 for the 2-D class)
  double padscale() const {
    double v = 1.0;
    for (int64_t n : {N0, N1, N2}) if (n > 1) v *= d.padsize;
    return v;
  }
  int sched(int which, bool forward, bool padded, Sched* out) const;
  int run_sched(const Sched& sc, const void* send, void* recv, hipStream_t on = nullptr) {
    return comm->alltoallv_part(send, sc.sc.data(), sc.sd.data(), recv, sc.rc.data(), sc.rd.data(), sc.peers.data(),
                                (int)sc.peers.size(), on ? on : stream, on && on != stream ? 1 : 0,
                                sc.part.empty() ? nullptr : sc.part.data());
  }
  // group id of every rank for the exchange inside comm0 (consecutive ranks: same rank / P1) or comm1 (same rank % P1)
  void fill_part(bool comm0, std::vector<int>* part) const {
    part->resize(P);
    for (int r = 0; r < P; ++r) (*part)[r] = comm0 ? r / P1 : r % P1;
  }
  int xchg(int which, bool forward, bool padded, const void* send, void* recv) {
    Sched sc;
    MFFT_TRY(sched(which, forward, padded, &sc));
    return run_sched(sc, send, recv);
  }
  int slab_forward(const void* u, void* fu);
  int slab_backward(const void* fu, void* u, bool masked);
  int slab_forward_pipelined(const void* u, void* fu);
  int slab_backward_pipelined(const void* src, void* u, bool pruned = false);
  int slab_forward_padded(const void* u, void* fu);
  int slab_backward_padded(const void* fu, void* u);
  bool can_fuse_pad() const;
  int slab_forward_padded_fused(const void* u, void* fu);
  int slab_backward_padded_fused(const void* fu, void* u);
  int slab_forward_rows(const void* u, void* fu);
  int slab_backward_rows(const void* src, void* u, bool pruned = false);
  int pencil_forward_pipelined_x(const void* u, void* fu);
  int pencil_backward_pipelined_x(const void* src, void* u);
  int pencil_forward_pipelined_y(const void* u, void* fu);
  int pencil_backward_pipelined_y(const void* src, void* u);
  int sched_rows(int which, bool forward, int64_t i0, int64_t mb, Sched* out) const;
  // pieces of the pipelined exchanges (host only; the executors and mfft_plan_exchange_pieces share it)
  int npieces() const { return nbatch > 1 ? nbatch : nslice > 1 ? nslice : 1; }
  int piece_sched(int which, bool forward, int piece, Sched* out) const;
  int ensure_work3(size_t bytes) {
    if (work3 && work3_bytes >= bytes) return 0;
    if (work3) MFFT_TRY(wfree(work3));
    work3 = nullptr;
    work3_bytes = 0;
    MFFT_TRY(walloc(&work3, bytes));
    work3_bytes = bytes;
    return 0;
  }
  int pencil_forward_padded_fused(const void* u, void* fu);
  int pencil_backward_padded_fused(const void* fu, void* u);
  int pencil_forward(const void* u, void* fu);
  int pencil_backward(const void* fu, void* u, bool masked);
  int pencil_forward_padded(const void* u, void* fu);
  int pencil_backward_padded(const void* fu, void* u);
  int apply_mask_copy(const void* fu, void** masked_out);
  // ---- round 6: pitched spectrum (mfft_plan_desc::complex_pitch) ----
  // The caller's complex array keeps its logical shape but its z rows lie Zp >= Nf elements apart (whole cache lines:
  // 513 -> 520 bins in double precision), so that every strided pass and both real transforms meet line-aligned rows.
  // One-rank slab R2C plans run on such arrays natively (nat_pitch); every other plan converts at the boundary through a
  // compact copy of its own (correct everywhere, fast where it was asked for).
  int64_t Zp = 0;               // row pitch of the caller's spectrum in complex elements; 0: compact rows of Nf
  void* pcomp = nullptr;        // compact copy for the plans that do not run on pitched rows natively
  size_t pcomp_bytes = 0;
  bool conv_now = false;        // exec(): this call runs on the compact copy (a route without a pitched flavour)
  bool pitched() const { return Zp > 0; }
  bool nat_pitch() const { return Zp > 0 && !conv_now && d.decomp == MFFT_SLAB && P == 1 && r2c && !d.line2d && !d.drop_nyquist; }
  int64_t Zc() const { return nat_pitch() ? Zp : Nf; }          // row pitch the one-rank slab routes run with
  void cdims(int64_t* d0, int64_t* d1, int64_t* d2) const {     // local complex extents
    if (d.decomp == MFFT_SLAB) { *d0 = N0; *d1 = Np1; *d2 = Nf; }
    else if (d.decomp == MFFT_PENCIL_X) { *d0 = N0; *d1 = N1_1; *d2 = q; }
    else { *d0 = N2_0; *d1 = N1; *d2 = q; }
  }
  int64_t local_complex_alloc_native() const { return nat_pitch() ? N0 * Np1 * Zp : local_complex_count(); }   // what the routes see
  int64_t local_complex_alloc() const {                          // elements of the caller's (possibly pitched) spectrum
    int64_t a, b, c;
    cdims(&a, &b, &c);
    return a * b * (pitched() ? Zp : c);
  }
  int repitch(const void* src, void* dst, bool to_pitched) {     // compact <-> pitched copy of one local spectrum
    int64_t a, b, c;
    cdims(&a, &b, &c);
    return to_pitched ? box(src, dst, a, b, c, b * c, c, b * Zp, Zp) : box(src, dst, a, b, c, b * Zp, Zp, b * c, c);
  }
  // ---- round 6: the nonlinear term a x b of a pseudo-spectral step as one operation (fft_nlz.h) ----
  void* nlx = nullptr;          // fused route: the six spectra after their inverse x pass, (L0, N1, Za) each
  size_t nlx_bytes = 0;
  void* nly = nullptr;          // ... and a batch of their x planes after the inverse y pass, (mb, L1, Za) each
  size_t nly_bytes = 0;
  void* nlr = nullptr;          // composed route: nine real-space work arrays
  size_t nlr_bytes = 0;
  int ensure_buf(void** b, size_t* have, size_t bytes) {
    if (*b && *have >= bytes) return 0;
    drop_graphs();
    if (*b) MFFT_TRY(dev_free(*b));
    *b = nullptr;
    *have = 0;
    MFFT_TRY(dev_alloc(b, bytes));
    *have = bytes;
    return 0;
  }
  bool nonlinear_fusable(int dealias) const;
  int nonlinear_cross_fused_ranks(const void* a, const void* b, void* out, int dealias);
  void* nlw[2] = {nullptr, nullptr};     // several ranks: the six x-pass outputs / exchange buffers (from the communicator: exchanged)
  size_t nlw_bytes[2] = {0, 0};
  int ensure_xbuf(int i, size_t bytes) {
    if (nlw[i] && nlw_bytes[i] >= bytes) return 0;
    drop_graphs();
    if (nlw[i]) MFFT_TRY(wfree(nlw[i]));
    nlw[i] = nullptr;
    nlw_bytes[i] = 0;
    MFFT_TRY(walloc(&nlw[i], bytes));
    nlw_bytes[i] = bytes;
    return 0;
  }
  int64_t local_real_count(bool padded) const;
  int nonlinear_cross(const void* a, const void* b, void* out, int dealias);
  int exec(bool forward, const void* in, void* out, int dealias);       // one transform, pitched callers' arrays converted where needed
  int nonlinear_cross_fused(const void* a, const void* b, void* out, int dealias);
  int nonlinear_cross_composed(const void* a, const void* b, void* out, int dealias);
  int64_t local_complex_count() const {
    if (d.decomp == MFFT_SLAB) return N0 * Np1 * Nf;
    if (d.decomp == MFFT_PENCIL_X) return N0 * N1_1 * q;
    return N2_0 * N1 * q;
  }
};

// ===========================================================================
// exchange schedules (host only)
//   slab:   which = 0, equal chunks over all ranks (padded: x extent M0/P)
//   pencil: which = 0 -> the z-splitting exchange (uneven last chunk), X: comm1, Y: comm0
//           which = 1 -> the other exchange (equal chunks),            X: comm0, Y: comm1
// ===========================================================================
int mfft_plan_s::sched(int which, bool forward, bool padded, Sched* o) const {
  auto equal = [&](const std::vector<int>& grp, size_t chunk) {
    const int n = (int)grp.size();
    o->peers = grp;
    o->sc.assign(n, chunk);
    o->rc.assign(n, chunk);
    o->sd.resize(n);
    o->rd.resize(n);
    for (int i = 0; i < n; ++i) o->sd[i] = o->rd[i] = (size_t)i * chunk;
  };
  if (d.decomp == MFFT_SLAB) {
    if (which != 0) return set_error(MFFT_ERR_INVALID, "slab plans have one exchange");
    const int64_t x = padded ? M0 / P : Np0;
    equal(world, (size_t)(x * (Np1 * Nf + (padded ? 0 : xplane_pad(forward)))) * es);
    return 0;
  }
  const bool X = d.decomp == MFFT_PENCIL_X;
  const int64_t m = padded ? M0 / P1 : N1_0, n = padded ? M1 / P2 : N2_1;
  fill_part(which == 0 ? !X : X, &o->part);
  if (which == 0) {
    const std::vector<int>& gz = X ? group1 : group0;
    const int Pz = (int)gz.size();
    o->peers = gz;
    o->sc.resize(Pz); o->sd.resize(Pz); o->rc.resize(Pz); o->rd.resize(Pz);
    size_t off = 0;
    for (int l = 0; l < Pz; ++l) {
      // (the fused 3/2-rule transforms have chunked kernels of their own and keep compact rows: padded -> no pitch)
      const int64_t pl = padded ? zc[l].len : zrow_pitch(zc[l].len, forward), pq = padded ? q : zrow_pitch(q, forward);
      const size_t uneven = (size_t)(m * n * pl) * es, even = (size_t)(m * n * pq) * es;
      if (forward) { o->sc[l] = uneven; o->sd[l] = off; o->rc[l] = even; o->rd[l] = (size_t)l * even; }
      else         { o->sc[l] = even; o->sd[l] = (size_t)l * even; o->rc[l] = uneven; o->rd[l] = off; }
      off += uneven;
    }
    return 0;
  }
  if (which == 1) {
    const int64_t xp = padded ? 0 : xplane_pad(forward);     // one cache line between x rows when they are 64 KiB multiples apart
    if (X) equal(group0, (size_t)(m * (N1_1 * q + xp)) * es);
    else   equal(group1, (size_t)(N2_0 * (n * (padded ? q : zrow_pitch(q, forward)) + xp)) * es);   // forward: the z kernel's row pitch travels on
    return 0;
  }
  return set_error(MFFT_ERR_INVALID, "pencil plans have two exchanges");
}

// Schedule of ONE piece of a pipelined exchange, displacements relative to the whole send / receive buffers:
//   slab, kz slices   : slice s of the packed [s][p][i][j][kz_s] layout (equal chunks)
//   slab, row batches : rows [i0, i0+mb) of every peer block of the packed (P, Np0, Np1, Nf) layout
//   pencil X          : rows [i0, i0+mb) of the blocks of exchange `which`
// An un-pipelined plan has one piece: the whole exchange.
int mfft_plan_s::piece_sched(int which, bool forward, int piece, Sched* o) const {
  if (piece < 0 || piece >= npieces()) return set_error(MFFT_ERR_INVALID, "piece %d of %d", piece, npieces());
  if (npieces() == 1) return sched(which, forward, false, o);
  if (d.decomp == MFFT_SLAB) {
    if (which != 0) return set_error(MFFT_ERR_INVALID, "slab plans have one exchange");
    if (nbatch > 1) {
      const int64_t i0 = Np0 * piece / nbatch, mb = Np0 * (piece + 1) / nbatch - i0;
      o->peers = world;
      o->sc.assign(P, (size_t)(mb * Np1 * Nf) * es);
      o->rc = o->sc;
      o->sd.resize(P);
      o->rd.resize(P);
      // [r][i][j][k] of the packed layout IS row r*Np0 + i of (N0, Np1, Nf): same blocks in both directions
      for (int r = 0; r < P; ++r) o->sd[r] = o->rd[r] = (size_t)((r * Np0 + i0) * Np1 * Nf) * es;
      return 0;
    }
    const size_t boff = slice_offset(piece, forward) * es, chunk = (size_t)(Np0 * slice_pitch(piece, forward)) * es;
    o->peers = world;
    o->sc.assign(P, chunk);
    o->rc = o->sc;
    o->sd.resize(P);
    o->rd.resize(P);
    for (int r = 0; r < P; ++r) o->sd[r] = o->rd[r] = boff + (size_t)r * chunk;
    return 0;
  }
  // X: both exchanges in batches of the m local x rows; Y: the z-splitting exchange in batches of the m local x rows,
  // the x-chunk exchange in batches of the N2_0 rows a rank owns after it
  const int64_t m = (d.decomp == MFFT_PENCIL_Y && which == 1) ? N2_0 : N1_0;
  const int64_t i0 = m * piece / nbatch, mb = m * (piece + 1) / nbatch - i0;
  return sched_rows(which, forward, i0, mb, o);
}

// ===========================================================================
// slab
// ===========================================================================
int mfft_plan_s::slab_forward(const void* u, void* fu) {
  const double Cb = (double)(N0 * Np1 * Nf) * es;            // local complex bytes
  const double Rb = (double)(Np0 * N1 * N2) * rs;            // local real-space bytes
  if (P == 1) {
    const int64_t Z = Zc();          // row pitch of the spectrum and of the intermediates: Nf, or the caller's pitch
    if (const int64_t xpad = p1_plane_pad()) {
      MFFT_TRY(stage("fwd_z", Rb + Cb, [&] { return z_forward(u, fu, N0 * N1, N2, Z); }));
      // power-of-two plane stride: the y transform writes planes one cache line apart from that, the x transform reads them
      const int64_t pl = N1 * Z + xpad;
      MFFT_TRY(ensure_work(0, (size_t)(N0 * pl) * es));
      void* A = work[0];
      MFFT_TRY(stage("fwd_y", 2 * Cb, [&] { return col(fu, A, N1, false, N0, Nf, N1 * Z, plain(Z), pl, plain(Z)); }));
      MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(A, fu, N0, false, 1, N1 * Z, 0, plain(pl), 0, plain(N1 * Z)); }));
      return 0;
    }
    MFFT_TRY(stage("fwd_z", Rb + Cb, [&] { return z_forward(u, fu, N0 * N1, N2, Z); }));
    if (fwd_out_of_place((size_t)Cb)) {
      // y transform into the work buffer (the one the inverse uses anyway), x transform out of it into the result: both
      // passes out of place (MFFT_FWD_OOP, see fwd_out_of_place)
      MFFT_TRY(ensure_work(0, (size_t)(N0 * N1 * Z) * es));
      void* A = work[0];
      MFFT_TRY(stage("fwd_y", 2 * Cb, [&] { return col(fu, A, N1, false, N0, Nf, N1 * Z, plain(Z), N1 * Z, plain(Z)); }));
      MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(A, fu, N0, false, 1, N1 * Z, 0, plain(N1 * Z), 0, plain(N1 * Z)); }));
      return 0;
    }
    MFFT_TRY(stage("fwd_y", 2 * Cb, [&] { return col(fu, fu, N1, false, N0, Nf, N1 * Z, plain(Z), N1 * Z, plain(Z)); }));
    MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(fu, fu, N0, false, 1, N1 * Z, 0, plain(N1 * Z), 0, plain(N1 * Z)); }));
    return 0;
  }
  if (nbatch > 1) return slab_forward_rows(u, fu);
  if (nslice > 1) return slab_forward_pipelined(u, fu);
  // x rows of the exchanged layout lie S elements apart: Np1 * Nf, plus one cache line where that is a 64 KiB multiple
  const int64_t S = Np1 * Nf + xplane_pad(true);
  const size_t cb = (size_t)std::max(Np0 * N1 * Nf, N0 * S) * es;
  MFFT_TRY(ensure_work(0, cb));
  MFFT_TRY(ensure_work(1, cb));
  void *A = work[0], *B = work[1];
  MFFT_TRY(stage("fwd_z", Rb + Cb, [&] { return z_forward(u, A, Np0 * N1, N2, Nf); }));
  // y transform writes straight into the packed (P, Np0, Np1, Nf) send layout (slab.py:403)
  MFFT_TRY(stage("fwd_y", 2 * Cb, [&] {
    return col(A, B, N1, false, Np0, Nf, N1 * Nf, plain(Nf), S, two_level(Np1, Np0 * S, Nf));
  }));
  if (xpass_inplace && S == Np1 * Nf) {           // round 1 - 3: receive into the result, x transform in place
    MFFT_TRY(stage("fwd_a2a", 0, [&] { return xchg(0, true, false, B, fu); }));
    MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(fu, fu, N0, false, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf)); }));
    return 0;
  }
  // A is free again (the y transform has read it): the chunks land there and the x transform runs OUT of place into the
  // result -- no more memory, and an out-of-place pass is the faster one (1024 fp64: 3.19 against 3.35 ms)
  MFFT_TRY(stage("fwd_a2a", 0, [&] { return xchg(0, true, false, B, A); }));
  MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(A, fu, N0, false, 1, Np1 * Nf, 0, plain(S), 0, plain(Np1 * Nf)); }));
  return 0;
}

int mfft_plan_s::apply_mask_copy(const void* fu, void** masked_out) {
  const size_t cnt = (size_t)local_complex_count();
  if (!mask || mask_count != cnt) return set_error(MFFT_ERR_INVALID, "2/3-rule requested but no dealias mask of %zu entries was set", cnt);
  const size_t cna = (size_t)local_complex_alloc_native();      // pitched rows: the device mask has the same pitch
  MFFT_TRY(ensure_work(2, cna * es));
  MFFT_HIP(hipMemcpyAsync(work[2], fu, cna * es, hipMemcpyDeviceToDevice, stream));
  MFFT_TRY(launch_mask(work[2], mask, cna, prec, stream));
  *masked_out = work[2];
  return 0;
}

int mfft_plan_s::slab_backward(const void* fu, void* u, bool masked) {
  const double Cb = (double)(N0 * Np1 * Nf) * es;
  const double Rb = (double)(Np0 * N1 * N2) * rs;
  const void* src = fu;
  struct MaskScope {            // the mask belongs to this call only
    mfft_plan_s* p;
    ~MaskScope() { p->mask_src = nullptr; }
  } mask_scope{this};
  if (masked && P == 1 && band_ok && !(getenv("MFFT_NO_PRUNE") && atoi(getenv("MFFT_NO_PRUNE")) != 0)) {
    if (!mask || mask_count != (size_t)local_complex_count()) return set_error(MFFT_ERR_INVALID, "2/3-rule requested but no dealias mask was set");
    const double keep0 = 1.0 - (double)(bb0 - ba0) / (double)N0, keep1 = 1.0 - (double)(bb1 - ba1) / (double)N1, keep2 = (double)ba2 / (double)Nf;
    ColArgs::Band bx, by;
    if (nat_pitch()) {
      // pitched rows (round 6): the tile list below was made for rows of Nf bins, so the x pass takes one outer batch per ky
      // (removed ky: nothing launched does anything) with the kept kz of it as its columns -- the form the pruned exchange uses
      const int64_t Z = Zp;
      MFFT_TRY(ensure_work(0, (size_t)(N0 * N1 * Z) * es));
      void* Aw = work[0];
      bx.row_lo = ba0; bx.row_hi = bb0; bx.g_off = 0; bx.g_step = 1; bx.g_lo = ba1; bx.g_hi = bb1;
      by.row_lo = ba1; by.row_hi = bb1; by.c_lim = ba2;
      MFFT_TRY(stage("bwd_x", Cb * keep1 * keep2 * (keep0 + 1.0), [&] {
        return col_band(fu, Aw, N0, N1, ba2, Z, plain(N1 * Z), Z, plain(N1 * Z), bx);
      }));
      MFFT_TRY(stage("bwd_y", Cb * keep2 * (keep1 + 1.0), [&] {
        return col_band(Aw, Aw, N1, N0, ba2, N1 * Z, plain(Z), N1 * Z, plain(Z), by);
      }));
      MFFT_TRY(stage("bwd_z", Rb + Cb * keep2, [&] { return c2r_rows(Aw, u, N0 * N1, N2, Z, N2, 1.0 / (double)N2, ba2); }));
      return 0;
    }
    MFFT_TRY(ensure_work(0, (size_t)(N0 * N1 * Nf) * es));
    void* Aw = work[0];
    bx.row_lo = ba0; bx.row_hi = bb0; bx.c_off = 0; bx.c_per = (int)Nf; bx.c_lim = ba2; bx.g_off = 0; bx.g_step = 0; bx.g_lo = ba1; bx.g_hi = bb1;
    bx.tile_list = band_tiles; bx.ntiles_listed = band_ntiles;
    by.row_lo = ba1; by.row_hi = bb1; by.c_lim = ba2;        // columns = kz of one x plane: only the first a2 are launched
    MFFT_TRY(stage("bwd_x", Cb * keep1 * keep2 * (keep0 + 1.0), [&] {
      return col_band(fu, Aw, N0, 1, N1 * Nf, 0, plain(N1 * Nf), 0, plain(N1 * Nf), bx);
    }));
    MFFT_TRY(stage("bwd_y", Cb * keep2 * (keep1 + 1.0), [&] {
      return col_band(Aw, Aw, N1, N0, ba2, N1 * Nf, plain(Nf), N1 * Nf, plain(Nf), by);
    }));
    MFFT_TRY(stage("bwd_z", Rb + Cb * keep2, [&] { return c2r_rows(Aw, u, N0 * N1, N2, Nf, N2, 1.0 / (double)N2, ba2); }));
    return 0;
  }
  if (masked && P > 1 && band_ok && (nbatch > 1 || nslice > 1) && !(getenv("MFFT_NO_PRUNE") && atoi(getenv("MFFT_NO_PRUNE")) != 0)) {
    if (!mask || mask_count != (size_t)local_complex_count()) return set_error(MFFT_ERR_INVALID, "2/3-rule requested but no dealias mask was set");
    return nbatch > 1 ? slab_backward_rows(fu, u, true)          // the row-batch exchange pipeline, pruned
                      : slab_backward_pipelined(fu, u, true);    // the kz-slice exchange pipeline, pruned
  }
  if (masked && P > 1 && band_ok && nbatch <= 1 && nslice <= 1 && !(getenv("MFFT_NO_PRUNE") && atoi(getenv("MFFT_NO_PRUNE")) != 0)) {
    // pruned inverse over P ranks (blocking exchange): the x pass reads the kept kx rows and writes (N0, Np1, a2) -- the
    // kept kz bins only, zeros for the ky this rank's mask removes --, so the exchange carries a2 / Nf of the bytes; the y
    // pass and c2r work on rows of a2 bins
    if (!mask || mask_count != (size_t)local_complex_count()) return set_error(MFFT_ERR_INVALID, "2/3-rule requested but no dealias mask was set");
    const int64_t a2 = ba2, per_line = (int64_t)(128 / es);
    const int64_t ap = (a2 + per_line - 1) / per_line * per_line;      // rows of the compact layout start on cache lines
    const size_t cbp = (size_t)(Np0 * N1 * ap) * es;
    MFFT_TRY(ensure_work(0, cbp));
    MFFT_TRY(ensure_work(1, cbp));
    void *Aw = work[0], *Bw = work[1];
    const double keep0 = 1.0 - (double)(bb0 - ba0) / (double)N0, keep2 = (double)a2 / (double)Nf;
    ColArgs::Band bx;
    bx.row_lo = ba0; bx.row_hi = bb0; bx.g_off = 0; bx.g_step = 1; bx.g_lo = ba1; bx.g_hi = bb1; bx.g_zero = 1;
    MFFT_TRY(stage("bwd_x", Cb * keep2 * (keep0 + 1.0), [&] {
      if (band_allzero) return zero(Aw, cbp);        // nothing of this rank's spectrum survives the mask
      return col_band(fu, Aw, N0, Np1, a2, Nf, plain(Np1 * Nf), ap, plain(Np1 * ap), bx);
    }));
    MFFT_TRY(stage("bwd_a2a", 0, [&] { return exchange_equal(world, Aw, Bw, (size_t)(Np0 * Np1 * ap) * es); }));
    MFFT_TRY(stage("bwd_y", 2 * Cb * keep2, [&] {
      return col(Bw, Aw, N1, true, Np0, a2, Np1 * ap, two_level(Np1, Np0 * Np1 * ap, ap), N1 * ap, plain(ap));
    }));
    MFFT_TRY(stage("bwd_z", Rb + Cb * keep2, [&] { return c2r_rows(Aw, u, Np0 * N1, N2, ap, N2, 1.0 / (double)N2, (int)a2); }));
    return 0;
  }
  if (masked) {
    bool fused = false;
    MFFT_TRY(fuse_mask(fu, (P == 1 && p1_plane_pad()) ? N1 : N0, &fused));
    if (!fused) {
      void* m = nullptr;
      MFFT_TRY(stage("bwd_mask", 2 * Cb, [&] { return apply_mask_copy(fu, &m); }));
      src = m;
    }
  }
  const int64_t Z = P == 1 ? Zc() : Nf;
  const size_t cb = (size_t)(Np0 * N1 * Z) * es;
  if (const int64_t xpad = P == 1 ? p1_plane_pad() : 0) {
    // slow plane stride (p1_plane_pad): y first (into padded planes), x out of them -- complex data: into the result, z in
    // place there; real data: into a second work buffer that c2r reads
    const int64_t pl = N1 * Z + xpad;
    MFFT_TRY(ensure_work(0, (size_t)(N0 * pl) * es));
    void* Ap = work[0];
    void* X = u;
    if (r2c) {
      MFFT_TRY(ensure_work(1, cb));
      X = work[1];
    }
    MFFT_TRY(stage("bwd_y", 2 * Cb, [&] { return col(src, Ap, N1, true, N0, Nf, N1 * Z, plain(Z), pl, plain(Z)); }));
    MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(Ap, X, N0, true, 1, N1 * Z, 0, plain(pl), 0, plain(N1 * Z)); }));
    MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward(X, u, N0 * N1, N2, Z); }));
    return 0;
  }
  MFFT_TRY(ensure_work(0, cb));
  void* A = work[0];
  if (P == 1 && Z != Nf) {       // pitched rows everywhere (the caller's array and the intermediate alike)
    MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(src, A, N0, true, 1, N1 * Z, 0, plain(N1 * Z), 0, plain(N1 * Z)); }));
    MFFT_TRY(stage("bwd_y", 2 * Cb, [&] { return col(A, A, N1, true, N0, Nf, N1 * Z, plain(Z), N1 * Z, plain(Z)); }));
    MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward(A, u, N0 * N1, N2, Z); }));
    return 0;
  }
  if (P == 1) {
    // (Round 4 measured two more one-rank routes and removed them again -- a line-aligned intermediate (rows of N2/2 + 1 bins
    // rounded up to whole cache lines: one per cent at 1024^3 for the inverse, a loss forward and at 512^3) and a forward
    // transform whose x pass alone runs out of place (gains only at 2048^3 in single precision): profiles/r04_aligned_route_ab.txt,
    // r04_fwd_oop_ab.txt; the last commit that has the switches MFFT_ALIGNED / MFFT_FWD_OOP=2 is a7fb791.)
    MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(src, A, N0, true, 1, N1 * Nf, 0, plain(N1 * Nf), 0, plain(N1 * Nf)); }));
    MFFT_TRY(stage("bwd_y", 2 * Cb, [&] { return col(A, A, N1, true, N0, Nf, N1 * Nf, plain(Nf), N1 * Nf, plain(Nf)); }));
    MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward(A, u, N0 * N1, N2, Nf); }));
    return 0;
  }
  if (nbatch > 1) return slab_backward_rows(src, u);
  if (nslice > 1) return slab_backward_pipelined(src, u);
  MFFT_TRY(ensure_work(1, cb));
  void* B = work[1];
  MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(src, A, N0, true, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf)); }));
  MFFT_TRY(stage("bwd_a2a", 0, [&] { return xchg(0, false, false, A, B); }));
  // y transform reads the (P, Np0, Np1, Nf) receive layout directly (transpose_Uc fused, maths.pyx:21-31)
  MFFT_TRY(stage("bwd_y", 2 * Cb, [&] {
    return col(B, A, N1, true, Np0, Nf, Np1 * Nf, two_level(Np1, Np0 * Np1 * Nf, Nf), N1 * Nf, plain(Nf));
  }));
  MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward(A, u, Np0 * N1, N2, Nf); }));
  return 0;
}

// ---- exchange pipeline (P > 1): the spectrum is cut into kz slices; slice s is
// transformed along y (written packed), exchanged on the communication stream
// while slice s+1 is transformed, and the x transform of slice s starts as soon
// as its exchange has landed.  Per slice the send layout is (P, Np0, Np1, kzs),
// the receive layout (N0, Np1, kzs).
int mfft_plan_s::slab_forward_pipelined(const void* u, void* fu) {
  const double Cb = (double)(N0 * Np1 * Nf) * es, Rb = (double)(Np0 * N1 * N2) * rs;
  // per slice the x rows of the exchanged layout lie slice_pitch() elements apart (Np1 * kz, plus a cache line where that
  // pitch reads slowly: xplane_pad's rule, slice by slice)
  const size_t cb = std::max((size_t)(Np0 * N1 * Nf), slice_offset(nslice, true)) * es;
  for (int i = 0; i < 3; ++i) MFFT_TRY(ensure_work(i, cb));
  char *A = static_cast<char*>(work[0]), *B = static_cast<char*>(work[1]), *Cr = static_cast<char*>(work[2]);
  char* out = static_cast<char*>(fu);
  MFFT_TRY(stage("fwd_z", Rb + Cb, [&] { return z_forward(u, A, Np0 * N1, N2, Nf); }));
  for (int s = 0; s < nslice; ++s) {
    const int64_t k0 = kslice[s].start, kz = kslice[s].len, S = slice_pitch(s, true);
    const size_t boff = slice_offset(s, true) * es;
    MFFT_TRY(stage("fwd_y", 2 * Cb / nslice, [&] {
      return col(A + (size_t)k0 * es, B + boff, N1, false, Np0, kz, N1 * Nf, plain(Nf), S, two_level(Np1, Np0 * S, kz));
    }));
    MFFT_HIP(hipEventRecord(ev_compute[s], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[s], 0));
    MFFT_TRY(stage_on(cstream, "fwd_a2a", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, true, s, &sc));
      return run_sched(sc, B, Cr, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[s], cstream));
  }
  for (int s = 0; s < nslice; ++s) {
    const int64_t k0 = kslice[s].start, kz = kslice[s].len;
    const size_t boff = slice_offset(s, true) * es;
    MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[s], 0));
    MFFT_TRY(stage("fwd_x", 2 * Cb / nslice, [&] {
      return col(Cr + boff, out + (size_t)k0 * es, N0, false, Np1, kz, kz, plain(slice_pitch(s, true)), Nf, plain(Np1 * Nf));
    }));
  }
  return 0;
}

// pruned (2/3-rule, band mask): slices that start at or beyond the a2 kept kz bins are not transformed or exchanged at
// all (every rank knows a2), the x pass does not load the removed kx rows and writes zeros for the ky its rank's mask
// removes, c2r reads a2 bins per row.
int mfft_plan_s::slab_backward_pipelined(const void* src, void* u, bool pruned) {
  const double Cb = (double)(N0 * Np1 * Nf) * es, Rb = (double)(Np0 * N1 * N2) * rs;
  ColArgs::Band bx;
  bx.row_lo = ba0; bx.row_hi = bb0; bx.g_off = 0; bx.g_step = 1; bx.g_lo = ba1; bx.g_hi = bb1; bx.g_zero = 1;
  auto kept = [&](int s) { return !pruned || kslice[s].start < ba2; };
  const size_t cb = (size_t)(Np0 * N1 * Nf) * es;
  for (int i = 0; i < 2; ++i) MFFT_TRY(ensure_work(i, cb));
  // work[2] may hold the masked copy of the spectrum (src): use a 4th buffer for the y output
  MFFT_TRY(ensure_work3(cb));
  char *A = static_cast<char*>(work[0]), *B = static_cast<char*>(work[1]), *A2 = static_cast<char*>(work3);
  const char* in = static_cast<const char*>(src);
  for (int s = 0; s < nslice; ++s) {
    if (!kept(s)) continue;
    const int64_t k0 = kslice[s].start, kz = kslice[s].len;
    const size_t boff = (size_t)(P * Np0 * Np1 * k0) * es;
    MFFT_TRY(stage("bwd_x", 2 * Cb / nslice, [&] {
      if (pruned && band_allzero) return zero(A + boff, (size_t)(N0 * Np1 * kz) * es);
      if (pruned) return col_band(in + (size_t)k0 * es, A + boff, N0, Np1, kz, Nf, plain(Np1 * Nf), kz, plain(Np1 * kz), bx);
      return col(in + (size_t)k0 * es, A + boff, N0, true, Np1, kz, Nf, plain(Np1 * Nf), kz, plain(Np1 * kz));
    }));
    MFFT_HIP(hipEventRecord(ev_compute[s], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[s], 0));
    MFFT_TRY(stage_on(cstream, "bwd_a2a", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, false, s, &sc));
      return run_sched(sc, A, B, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[s], cstream));
  }
  for (int s = 0; s < nslice; ++s) {
    if (!kept(s)) continue;
    const int64_t k0 = kslice[s].start, kz = kslice[s].len;
    const size_t boff = (size_t)(P * Np0 * Np1 * k0) * es;
    MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[s], 0));
    MFFT_TRY(stage("bwd_y", 2 * Cb / nslice, [&] {
      return col(B + boff, A2 + (size_t)k0 * es, N1, true, Np0, kz, Np1 * kz, two_level(Np1, Np0 * Np1 * kz, kz),
                 N1 * Nf, plain(Nf));
    }));
  }
  if (pruned) {
    MFFT_TRY(stage("bwd_z", Rb + Cb * (double)ba2 / (double)Nf, [&] {
      return c2r_rows(A2, u, Np0 * N1, N2, Nf, N2, 1.0 / (double)N2, ba2);
    }));
    return 0;
  }
  MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward(A2, u, Np0 * N1, N2, Nf); }));
  return 0;
}

// ---- slab, second pipeline flavour (`pipeline` < 0): batches of local x rows -----------------------------
// z and y transforms of batch b+1 overlap the exchange of batch b; the exchange delivers straight into the output
// array (its receive layout (N0, Np1, Nf) IS the output layout), where the x transform then runs in place over whole
// rows.  Compared with the kz slices: the z transform is overlapped instead of the x transform, no strided kz
// sub-columns, one work buffer less in the forward direction.  Which one is faster depends on the links; bench.py
// measures both.
int mfft_plan_s::slab_forward_rows(const void* u, void* fu) {
  const double Cb = (double)(N0 * Np1 * Nf) * es, Rb = (double)(Np0 * N1 * N2) * rs;
  const size_t cb = (size_t)(Np0 * N1 * Nf) * es;
  for (int i = 0; i < 2; ++i) MFFT_TRY(ensure_work(i, cb));
  char *A = static_cast<char*>(work[0]), *Bk = static_cast<char*>(work[1]);
  const char* in = static_cast<const char*>(u);
  const int B = nbatch;
  for (int b = 0; b < B; ++b) {
    const int64_t i0 = Np0 * b / B, mb = Np0 * (b + 1) / B - i0;
    MFFT_TRY(stage("fwd_z", (Rb + Cb) / B, [&] {
      return z_forward(in + (size_t)(i0 * N1 * N2) * rs, A + (size_t)(i0 * N1 * Nf) * es, mb * N1, N2, Nf);
    }));
    MFFT_TRY(stage("fwd_y", 2 * Cb / B, [&] {
      return col(A + (size_t)(i0 * N1 * Nf) * es, Bk + (size_t)(i0 * Np1 * Nf) * es, N1, false, mb, Nf, N1 * Nf, plain(Nf),
                 Np1 * Nf, two_level(Np1, Np0 * Np1 * Nf, Nf));
    }));
    MFFT_HIP(hipEventRecord(ev_compute[b], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[b], 0));
    MFFT_TRY(stage_on(cstream, "fwd_a2a", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, true, b, &sc));
      return run_sched(sc, Bk, fu, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[b], cstream));
  }
  MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[B - 1], 0));
  MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(fu, fu, N0, false, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf)); }));
  return 0;
}

int mfft_plan_s::slab_backward_rows(const void* src, void* u, bool pruned) {
  const double Cb = (double)(N0 * Np1 * Nf) * es, Rb = (double)(Np0 * N1 * N2) * rs;
  // pruned (2/3-rule, band mask): every array between the x pass and c2r holds rows of `w` = a2 (padded to a cache line)
  // kept kz bins instead of Nf, the exchange pieces shrink with them; see slab_backward
  const int64_t per_line = (int64_t)(128 / es);
  const int64_t w = pruned ? ((int64_t)ba2 + per_line - 1) / per_line * per_line : Nf, nz = pruned ? (int64_t)ba2 : Nf;
  const size_t cb = (size_t)(Np0 * N1 * w) * es;
  for (int i = 0; i < 2; ++i) MFFT_TRY(ensure_work(i, cb));
  const bool src_in_work2 = work[2] != nullptr && src == work[2];     // the masked copy of the spectrum
  if (src_in_work2) MFFT_TRY(ensure_work3(cb));
  else MFFT_TRY(ensure_work(2, cb));
  char *A = static_cast<char*>(work[0]), *Bk = static_cast<char*>(work[1]);
  char* A2 = static_cast<char*>(src_in_work2 ? work3 : work[2]);
  char* out = static_cast<char*>(u);
  const int B = nbatch;
  MFFT_TRY(stage("bwd_x", 2 * Cb, [&] {
    if (pruned && band_allzero) return zero(A, cb);
    if (pruned) {
      ColArgs::Band bx;
      bx.row_lo = ba0; bx.row_hi = bb0; bx.g_off = 0; bx.g_step = 1; bx.g_lo = ba1; bx.g_hi = bb1; bx.g_zero = 1;
      return col_band(src, A, N0, Np1, nz, Nf, plain(Np1 * Nf), w, plain(Np1 * w), bx);
    }
    return col(src, A, N0, true, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf));
  }));
  MFFT_HIP(hipEventRecord(ev_compute[0], stream));
  MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[0], 0));
  for (int b = 0; b < B; ++b) {
    MFFT_TRY(stage_on(cstream, "bwd_a2a", 0, [&] {
      Sched sc;
      if (pruned) {      // the same blocks as piece_sched's, with rows of w bins
        const int64_t i0 = Np0 * b / B, mb = Np0 * (b + 1) / B - i0;
        sc.peers = world;
        sc.sc.assign(P, (size_t)(mb * Np1 * w) * es);
        sc.rc = sc.sc;
        sc.sd.resize(P);
        sc.rd.resize(P);
        for (int r = 0; r < P; ++r) sc.sd[r] = sc.rd[r] = (size_t)((r * Np0 + i0) * Np1 * w) * es;
      } else {
        MFFT_TRY(piece_sched(0, false, b, &sc));
      }
      return run_sched(sc, A, Bk, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[b], cstream));
  }
  for (int b = 0; b < B; ++b) {
    const int64_t i0 = Np0 * b / B, mb = Np0 * (b + 1) / B - i0;
    MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[b], 0));
    MFFT_TRY(stage("bwd_y", 2 * Cb / B, [&] {
      return col(Bk + (size_t)(i0 * Np1 * w) * es, A2 + (size_t)(i0 * N1 * w) * es, N1, true, mb, nz, Np1 * w,
                 two_level(Np1, Np0 * Np1 * w, w), N1 * w, plain(w));
    }));
    MFFT_TRY(stage("bwd_z", (Rb + Cb) / B, [&] {
      if (pruned)
        return c2r_rows(A2 + (size_t)(i0 * N1 * w) * es, out + (size_t)(i0 * N1 * N2) * rs, mb * N1, N2, w, N2, 1.0 / (double)N2, ba2);
      return z_backward(A2 + (size_t)(i0 * N1 * Nf) * es, out + (size_t)(i0 * N1 * N2) * rs, mb * N1, N2, Nf);
    }));
  }
  return 0;
}

// ---- 3/2-rule, slab (R2C: slab.py:310-344, 445-483; P == 1: 250-268, 372-386) ----
// fused 3/2-rule (R2C, padsize 1.5): the zero band is never materialised -- the x and y
// inverse transforms read the un-padded rows and skip the band (ColFft PAD = 1), c2r reads the
// missing kz columns as zeros; forward: r2c stores only the kept columns, the y and x transforms
// store only the kept rows and fold the Nyquist row in registers (PAD = 2).  Six kernels per
// pair, like the un-padded path; pack / unpack ride on the two-level row maps.
bool mfft_plan_s::can_fuse_pad() const {
  if (!r2c || d.padsize != 1.5 || d.drop_nyquist || d.line2d) return false;
  if (N0 % 2 || N1 % 2 || 2 * M0 != 3 * N0 || 2 * M1 != 3 * N1 || 2 * M2 != 3 * N2) return false;
  return find_kernel(FAM_COL, (int)M0, prec, 1, 0, 1) && find_kernel(FAM_COL, (int)M0, prec, 0, 0, 2) &&
         find_kernel(FAM_COL, (int)M1, prec, 1, 0, 1) && find_kernel(FAM_COL, (int)M1, prec, 0, 0, 2) &&
         find_kernel(FAM_R2C, (int)M2, prec, 0, 0, 3) && find_kernel(FAM_C2R, (int)M2, prec, 1, 0, 3) &&
         getenv("MFFT_NO_PAD_FUSION") == nullptr;
}

int mfft_plan_s::slab_backward_padded_fused(const void* fu, void* u) {
  const double sc3 = padscale();
  const int64_t Mp0 = M0 / P;
  if (const int64_t Za = nat_pitch() ? Zp : pad_pitch(); Za != Nf) {          // one rank, line-aligned z rows in both intermediates
    MFFT_TRY(ensure_work(0, (size_t)(M0 * N1 * Za) * es));
    MFFT_TRY(ensure_work(2, (size_t)(M0 * M1 * Za) * es));
    void *W0 = work[0], *W2 = work[2];
    if (nat_pitch()) {                   // the caller's rows have the pitch already: whole planes of N1 * Za columns
      MFFT_TRY(stage("bwd_x", 0, [&] {
        return col_pad(fu, W0, M0, true, 1, false, 1, N1 * Za, 0, plain(N1 * Za), 0, plain(N1 * Za), sc3 / (double)M0);
      }));
      MFFT_TRY(stage("bwd_y", 0, [&] {
        return col_pad(W0, W2, M1, true, 1, false, M0, Nf, N1 * Za, plain(Za), M1 * Za, plain(Za), 1.0 / (double)M1);
      }));
    } else if (pad_align_inv == 3) {     // the y pass converts: x compact -> compact, y compact -> pitched
      MFFT_TRY(stage("bwd_x", 0, [&] {
        return col_pad(fu, W0, M0, true, 1, false, 1, N1 * Nf, 0, plain(N1 * Nf), 0, plain(N1 * Nf), sc3 / (double)M0);
      }));
      MFFT_TRY(stage("bwd_y", 0, [&] {
        return col_pad(W0, W2, M1, true, 1, false, M0, Nf, N1 * Nf, plain(Nf), M1 * Za, plain(Za), 1.0 / (double)M1);
      }));
    } else {
      MFFT_TRY(stage("bwd_x", 0, [&] {     // one outer batch per y row: compact rows in, pitched rows out
        return col_pad(fu, W0, M0, true, 1, false, N1, Nf, Nf, plain(N1 * Nf), Za, plain(N1 * Za), sc3 / (double)M0, 0, 0,
                       pad_align_inv == 2 ? -1 : 1);
      }));
      MFFT_TRY(stage("bwd_y", 0, [&] {
        return col_pad(W0, W2, M1, true, 1, false, M0, Nf, N1 * Za, plain(Za), M1 * Za, plain(Za), 1.0 / (double)M1);
      }));
    }
    MFFT_TRY(stage("bwd_z", 0, [&] { return c2r_rows(W2, u, M0 * M1, M2, Za, M2, 1.0 / (double)M2, (int)Nf); }));
    return 0;
  }
  MFFT_TRY(ensure_work(0, (size_t)(M0 * Np1 * Nf) * es));
  MFFT_TRY(ensure_work(1, (size_t)(M0 * Np1 * Nf) * es));
  MFFT_TRY(ensure_work(2, (size_t)(Mp0 * M1 * Nf) * es));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  MFFT_TRY(stage("bwd_x", 0, [&] {
    return col_pad(fu, W0, M0, true, 1, false, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf), sc3 / (double)M0);
  }));
  if (P > 1) {
    MFFT_TRY(stage("bwd_a2a", 0, [&] { return xchg(0, false, true, W0, W1); }));
    MFFT_TRY(stage("bwd_y", 0, [&] {
      return col_pad(W1, W2, M1, true, 1, false, Mp0, Nf, Np1 * Nf, two_level(Np1, Mp0 * Np1 * Nf, Nf), M1 * Nf, plain(Nf),
                     1.0 / (double)M1);
    }));
  } else {
    MFFT_TRY(stage("bwd_y", 0, [&] {
      return col_pad(W0, W2, M1, true, 1, false, Mp0, Nf, N1 * Nf, plain(Nf), M1 * Nf, plain(Nf), 1.0 / (double)M1);
    }));
  }
  MFFT_TRY(stage("bwd_z", 0, [&] { return c2r_rows(W2, u, Mp0 * M1, M2, Nf, M2, 1.0 / (double)M2, (int)Nf); }));
  return 0;
}

int mfft_plan_s::slab_forward_padded_fused(const void* u, void* fu) {
  const double isc3 = 1.0 / padscale();
  const int64_t Mp0 = M0 / P;
  if (const int64_t Za = nat_pitch() ? Zp : pad_pitch(); Za != Nf) {          // one rank, line-aligned z rows in both intermediates
    MFFT_TRY(ensure_work(0, (size_t)(M0 * N1 * Za) * es));
    MFFT_TRY(ensure_work(2, (size_t)(M0 * M1 * Za) * es));
    void *W0 = work[0], *W2 = work[2];
    MFFT_TRY(stage("fwd_z", 0, [&] { return r2c_rows(u, W2, M0 * M1, M2, M2, Za, 1.0, (int)Nf); }));
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col_pad(W2, W0, M1, false, 2, true, M0, Nf, M1 * Za, plain(Za), N1 * Za, plain(Za), 1.0);
    }));
    if (nat_pitch()) {                   // pitched result: whole planes of N1 * Za columns, no conversion
      MFFT_TRY(stage("fwd_x", 0, [&] {
        return col_pad(W0, fu, M0, false, 2, true, 1, N1 * Za, 0, plain(N1 * Za), 0, plain(N1 * Za), isc3);
      }));
      return 0;
    }
    MFFT_TRY(stage("fwd_x", 0, [&] {     // tiles of the compact result; input column c = (y, z) sits at y * Za + z
      return col_pad(W0, fu, M0, false, 2, true, 1, N1 * Nf, 0, plain(N1 * Za), 0, plain(N1 * Nf), isc3, Nf, Za - Nf);
    }));
    return 0;
  }
  MFFT_TRY(ensure_work(0, (size_t)(M0 * Np1 * Nf) * es));
  MFFT_TRY(ensure_work(1, (size_t)(M0 * Np1 * Nf) * es));
  MFFT_TRY(ensure_work(2, (size_t)(Mp0 * M1 * Nf) * es));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  MFFT_TRY(stage("fwd_z", 0, [&] { return r2c_rows(u, W2, Mp0 * M1, M2, M2, Nf, 1.0, (int)Nf); }));
  void* xin = W0;
  if (P > 1) {
    // truncate + fold in y, written straight into the packed (P, Mp0, Np1, Nf) send layout
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col_pad(W2, W0, M1, false, 2, true, Mp0, Nf, M1 * Nf, plain(Nf), Np1 * Nf, two_level(Np1, Mp0 * Np1 * Nf, Nf), 1.0);
    }));
    MFFT_TRY(stage("fwd_a2a", 0, [&] { return xchg(0, true, true, W0, W1); }));
    xin = W1;
  } else {
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col_pad(W2, W0, M1, false, 2, true, Mp0, Nf, M1 * Nf, plain(Nf), N1 * Nf, plain(Nf), 1.0);
    }));
  }
  MFFT_TRY(stage("fwd_x", 0, [&] {
    return col_pad(xin, fu, M0, false, 2, true, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf), isc3);
  }));
  return 0;
}

int mfft_plan_s::slab_backward_padded(const void* fu, void* u) {
  if (P > 1 && P > N0 / 2) return set_error(MFFT_ERR_INVALID, "number of ranks cannot exceed N[0]/2 for the 3/2-rule");
  if (can_fuse_pad()) return slab_backward_padded_fused(fu, u);
  const double sc3 = padscale();
  const int64_t Mp0 = M0 / P;
  // W0: (M0, Np1, Nf) padded in x; W1: (Mp0, N1, Nf) after the exchange; then (Mp0, M1, Nf), (Mp0, M1, Mf)
  MFFT_TRY(ensure_work(0, (size_t)std::max(M0 * Np1 * Nf, Mp0 * M1 * Mf) * es));
  MFFT_TRY(ensure_work(1, (size_t)std::max(Mp0 * N1 * Nf, Mp0 * M1 * Nf) * es));
  MFFT_TRY(ensure_work(2, (size_t)(Mp0 * M1 * Nf) * es));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  MFFT_TRY(stage("pad_x", 0, [&] { return pad_axis(fu, W0, 1, N0, M0, Np1 * Nf, sc3); }));
  MFFT_TRY(stage("bwd_x", 0, [&] { return col(W0, W0, M0, true, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf)); }));
  const void* yin = W0;
  if (P > 1) {
    MFFT_TRY(stage("bwd_a2a", 0, [&] { return xchg(0, false, true, W0, W1); }));
    // unpack (P, Mp0, Np1, Nf) -> (Mp0, N1, Nf)
    MFFT_TRY(stage("unpack", 0, [&] {
      for (int p = 0; p < P; ++p)
        MFFT_TRY(box(static_cast<char*>(W1) + (size_t)p * (Mp0 * Np1 * Nf) * es,
                     static_cast<char*>(W2) + (size_t)(p * Np1 * Nf) * es, Mp0, 1, Np1 * Nf, Np1 * Nf, 0, N1 * Nf, 0));
      return 0;
    }));
    yin = W2;
  }
  // pad y: (Mp0, N1, Nf) -> (Mp0, M1, Nf)
  void* ypad = (yin == W2) ? W1 : W2;
  MFFT_TRY(stage("pad_y", 0, [&] { return pad_axis(yin, ypad, Mp0, N1, M1, Nf, 1.0); }));
  MFFT_TRY(stage("bwd_y", 0, [&] { return col(ypad, ypad, M1, true, Mp0, Nf, M1 * Nf, plain(Nf), M1 * Nf, plain(Nf)); }));
  // pad z: (Mp0*M1, Nf) -> (Mp0*M1, Mf); one-sided for the half spectrum, two-sided for C2C (slab.py:815-817)
  MFFT_TRY(stage("pad_z", 0, [&] {
    if (!r2c) return pad_axis(ypad, W0, Mp0 * M1, N2, M2, 1, 1.0);
    MFFT_TRY(zero(W0, (size_t)(Mp0 * M1 * Mf) * es));
    return box(ypad, W0, 1, Mp0 * M1, Nf, 0, Nf, 0, Mf);
  }));
  MFFT_TRY(stage("bwd_z", 0, [&] {
    if (!r2c) return c2c_rows(W0, u, Mp0 * M1, M2, M2, M2, true, 1.0 / (double)M2);
    return c2r_rows(W0, u, Mp0 * M1, M2, Mf, M2, 1.0 / (double)M2);
  }));
  return 0;
}

int mfft_plan_s::slab_forward_padded(const void* u, void* fu) {
  if (P > 1 && P > N0 / 2) return set_error(MFFT_ERR_INVALID, "number of ranks cannot exceed N[0]/2 for the 3/2-rule");
  if (can_fuse_pad()) return slab_forward_padded_fused(u, fu);
  const double isc3 = 1.0 / padscale();
  const int64_t Mp0 = M0 / P;
  MFFT_TRY(ensure_work(0, (size_t)std::max(Mp0 * M1 * Mf, M0 * Np1 * Nf) * es));
  MFFT_TRY(ensure_work(1, (size_t)std::max(Mp0 * N1 * Nf, M0 * Np1 * Nf) * es));
  MFFT_TRY(ensure_work(2, (size_t)std::max(M0 * Np1 * Nf, r2c ? (int64_t)0 : Mp0 * M1 * Nf) * es));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  MFFT_TRY(stage("fwd_z", 0, [&] {
    if (!r2c) return c2c_rows(u, W0, Mp0 * M1, M2, M2, M2, false, 1.0);
    return r2c_rows(u, W0, Mp0 * M1, M2, M2, Mf);
  }));
  MFFT_TRY(stage("fwd_y", 0, [&] { return col(W0, W0, M1, false, Mp0, Mf, M1 * Mf, plain(Mf), M1 * Mf, plain(Mf)); }));
  // truncate y and z: (Mp0, M1, Mf) -> (Mp0, N1, Nf)   (slab.py:459 / C2C: 782 copy_from_padded axis 1).
  // The reference's C2C folds the Nyquist modes of y and z for P > 1 and does plain corner
  // copies (no fold) on one rank (slab.py:736-739); both are reproduced.
  const bool c2c_fold = P > 1;
  MFFT_TRY(stage("trunc_y", 0, [&] {
    if (r2c) return trunc_axis(W0, W1, Mp0, N1, M1, Nf, Mf, 1.0);
    MFFT_TRY(trunc_axis(W0, W2, Mp0 * M1, N2, M2, 1, 1, 1.0, c2c_fold));
    return trunc_axis(W2, W1, Mp0, N1, M1, N2, N2, 1.0, c2c_fold);
  }));
  void* xin = W1;
  if (P > 1) {
    // pack (Mp0, P, Np1, Nf) -> (P, Mp0, Np1, Nf) and exchange
    MFFT_TRY(stage("pack", 0, [&] {
      for (int p = 0; p < P; ++p)
        MFFT_TRY(box(static_cast<char*>(W1) + (size_t)(p * Np1 * Nf) * es,
                     static_cast<char*>(W0) + (size_t)p * (Mp0 * Np1 * Nf) * es, Mp0, 1, Np1 * Nf, N1 * Nf, 0, Np1 * Nf, 0));
      return 0;
    }));
    MFFT_TRY(stage("fwd_a2a", 0, [&] { return xchg(0, true, true, W0, W2); }));
    xin = W2;
  }
  MFFT_TRY(stage("fwd_x", 0, [&] { return col(xin, xin, M0, false, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf)); }));
  // R2C folds the x Nyquist plane (slab.py:480-482); C2C copies the two halves (slab.py:796-797)
  MFFT_TRY(stage("trunc_x", 0, [&] { return trunc_axis(xin, fu, 1, N0, M0, Np1 * Nf, Np1 * Nf, isc3, r2c); }));
  return 0;
}

// ===========================================================================
// Round 6: the nonlinear term of a pseudo-spectral step, out = fftn(ifftn(a) x ifftn(b)), as ONE plan-level operation
// (what demo/spectral_dns_solver.py:53-71 composes from six ifftn, a cross product in real space and three fftn).
// ===========================================================================
extern "C" int mfft_ew_cross(mfft_plan_t plan, const void* a, const void* b, void* out, size_t n, int precision);

int64_t mfft_plan_s::local_real_count(bool padded) const {
  if (d.line2d) return 0;
  if (d.decomp == MFFT_SLAB) return padded ? (int64_t)(d.padsize * Np0) * M1 * M2 : Np0 * N1 * N2;
  return padded ? (int64_t)(d.padsize * N1_0) * (int64_t)(d.padsize * N2_1) * M2 : N1_0 * N2_1 * N2;
}

// Composed route (every decomposition and length): the transforms the caller would run, on nine work arrays of the plan.
int mfft_plan_s::nonlinear_cross_composed(const void* a, const void* b, void* out, int dealias) {
  const bool pad = dealias == MFFT_DEALIAS_3_2, masked = dealias == MFFT_DEALIAS_2_3;
  const int64_t nr = local_real_count(pad), nc = local_complex_alloc();       // (pitched arrays: a component is that much larger)
  if (nr <= 0 || !r2c) return set_error(MFFT_ERR_UNSUPPORTED, "nonlinear_cross needs a 3-D real-to-complex plan");
  MFFT_TRY(ensure_buf(&nlr, &nlr_bytes, (size_t)(9 * nr) * rs));
  char* R = static_cast<char*>(nlr);
  auto back = [&](const void* in, void* o) { return exec(false, in, o, dealias); };
  auto fwd = [&](const void* in, void* o) { return exec(true, in, o, masked ? (int)MFFT_DEALIAS_NONE : dealias); };
  for (int f = 0; f < 3; ++f) {
    MFFT_TRY(back(static_cast<const char*>(a) + (size_t)(f * nc) * es, R + (size_t)(f * nr) * rs));
    MFFT_TRY(back(static_cast<const char*>(b) + (size_t)(f * nc) * es, R + (size_t)((3 + f) * nr) * rs));
  }
  MFFT_TRY(stage("nl_cross", 9.0 * (double)nr * rs, [&] {
    return mfft_ew_cross(this, R, R + (size_t)(3 * nr) * rs, R + (size_t)(6 * nr) * rs, (size_t)nr, prec);
  }));
  for (int f = 0; f < 3; ++f) MFFT_TRY(fwd(R + (size_t)((6 + f) * nr) * rs, static_cast<char*>(out) + (size_t)(f * nc) * es));
  return 0;
}

// Fused route: one rank, slab, real data, radix kernels on every axis.
bool mfft_plan_s::nonlinear_fusable(int dealias) const {
  static const bool off = getenv("MFFT_NO_NLZ") && atoi(getenv("MFFT_NO_NLZ")) != 0;
  static const bool ranks_off = getenv("MFFT_NO_NLZ_RANKS") && atoi(getenv("MFFT_NO_NLZ_RANKS")) != 0;
  if (off || d.decomp != MFFT_SLAB || !r2c || d.line2d || d.drop_nyquist || N0 < 2 || N1 < 2 || N2 < 2) return false;
  if (P > 1 && (ranks_off || pitched() || (dealias == MFFT_DEALIAS_3_2 && P > N0 / 2))) return false;
  if (dealias == MFFT_DEALIAS_3_2) return can_fuse_pad() && nlz_supported(M2, prec);
  auto plain_ok = [&](int64_t n) {
    return n < 65536 && find_kernel(FAM_COL, (int)n, prec, 0) && find_kernel(FAM_COL, (int)n, prec, 1);
  };
  if (!plain_ok(N0) || !plain_ok(N1) || !nlz_supported(N2, prec)) return false;
  if (dealias == MFFT_DEALIAS_2_3) return mask && mask_count == (size_t)local_complex_count() && mask_fusable(N0, prec);
  return dealias == MFFT_DEALIAS_NONE;
}

// The six spectra go through their inverse x passes into (L0, N1, Za) buffers of the plan (L = the padded mesh under the
// 3/2-rule; Za = the row pitch, whole cache lines where rows are long).  Everything after that is local to an x plane, so
// batches of x planes then run: inverse y pass of the six fields -> NlzFft (six z rows in, the three rows of the cross
// product out, in place on the first three) -> forward y pass of the three results back into the x-pass buffers, whose
// planes of that batch are free by then.  Three forward x passes finish.  The real-space arrays never exist; the batch
// buffers are at most 16 GiB (1024^3 with the 3/2-rule: 6 x 13.1 GB of x-pass buffers + 14.7 GB of batch buffers, where the
// composed route needs 9 x 29 GB of real work arrays).
int mfft_plan_s::nonlinear_cross_fused(const void* a, const void* b, void* out, int dealias) {
  const bool pad = dealias == MFFT_DEALIAS_3_2, masked = dealias == MFFT_DEALIAS_2_3;
  const int64_t L0 = pad ? M0 : N0, L1 = pad ? M1 : N1, L2 = pad ? M2 : N2;
  const int64_t line = (int64_t)(128 / es);
  static const int align_mode = getenv("MFFT_NLZ_ALIGN") ? atoi(getenv("MFFT_NLZ_ALIGN")) : -1;      // 0 compact rows, 1 aligned, unset: rows of 2 KiB and more
  const bool aligned = align_mode > 0 || (align_mode < 0 && Nf * (int64_t)es >= 2048);
  const int64_t Zi = Zc();                         // row pitch of the caller's arrays
  const int64_t Za = nat_pitch() ? Zp : aligned ? (Nf + line - 1) / line * line : Nf;
  const int64_t C = N0 * N1 * Zi;                  // elements of one component of the caller's arrays
  const size_t xelems = (size_t)(L0 * N1 * Za);    // ... of one x-pass buffer
  // Batch of x planes.  Large batches win (512^3 with the 3/2-rule, ms per Runge-Kutta step against the MiB of a batch's six
  // y-pass outputs: 80: 181, 160: 157, 320: 143, 640: 130, 1536: 117, 6000: 111 -- batches that would fit the 256 MB Infinity Cache
  // gain nothing from it and pay for their short launches: profiles/r06_dns_batch.txt), so: batches of 16 GiB -- ONE batch up to
  // 512^3 with the 3/2-rule (14.9 GB), four at 768^3, eight at 1024^3 (14.7 GB of batch buffers beside 78.5 GB of x-pass
  // buffers).  MFFT_NLZ_BATCH_MB overrides.
  static const long batch_mb = getenv("MFFT_NLZ_BATCH_MB") ? atol(getenv("MFFT_NLZ_BATCH_MB")) : 16384;
  const size_t plane6 = (size_t)(6 * L1 * Za) * es;
  const int64_t nbat = (int64_t)((plane6 * (size_t)L0 + ((size_t)batch_mb << 20) - 1) / ((size_t)batch_mb << 20));
  int64_t mb = (L0 + std::max<int64_t>(nbat, 1) - 1) / std::max<int64_t>(nbat, 1);
  if (mb < 1) mb = 1;
  if (mb > L0) mb = L0;
  MFFT_TRY(ensure_buf(&nlx, &nlx_bytes, 6 * xelems * es));
  MFFT_TRY(ensure_buf(&nly, &nly_bytes, (size_t)mb * plane6));
  char* X = static_cast<char*>(nlx);
  char* Y = static_cast<char*>(nly);
  const size_t yelems = (size_t)(mb * L1 * Za);
  const double sc3 = pad ? padscale() : 1.0;
  const double Cb = (double)C * es, Xb = (double)(L0 * N1 * Nf) * es, Yb = (double)(L0 * L1 * Nf) * es;
  struct MaskScope {            // the mask belongs to this call only
    mfft_plan_s* p;
    ~MaskScope() { p->mask_src = nullptr; p->lband_use = false; }
  } mask_scope{this};
  lband_use = false;
  // 2/3-rule with the reference's own filter (detect_band): the six inverse transforms are PRUNED as in slab_backward -- the x
  // pass neither loads the removed kx rows nor touches the removed ky and kz columns, the y pass works on the kept kz columns
  // and does not load the removed ky rows, the fused z kernel reads ba2 bins per row (NlzParams::valid_in) and stores all Nf.
  // 512^3, per Taylor-Green step: nl_x_inv 12.9 -> 5.9 ms, nl_y_inv 9.3 -> 5.2, nl_z 8.2 -> 7.1; step 60.3 -> 45.5 ms
  // (profiles/r06_dns_23rule.txt).  MFFT_NO_PRUNE=1: the masked loads below.
  const bool pruned = masked && band_ok && !(getenv("MFFT_NO_PRUNE") && atoi(getenv("MFFT_NO_PRUNE")) != 0);
  const double keep0 = pruned ? 1.0 - (double)(bb0 - ba0) / (double)N0 : 1.0, keep1 = pruned ? 1.0 - (double)(bb1 - ba1) / (double)N1 : 1.0,
               keep2 = pruned ? (double)ba2 / (double)Nf : 1.0;
  MFFT_TRY(stage("nl_x_inv", 6 * (Cb * keep0 + Xb) * keep1 * keep2, [&] {
    for (int f = 0; f < 6; ++f) {
      const void* src = static_cast<const char*>(f < 3 ? a : b) + (size_t)((f % 3) * C) * es;
      void* dst = X + (size_t)f * xelems * es;
      if (pruned) {                                // one outer batch per ky, the kept kz columns of it
        ColArgs::Band bx;
        bx.row_lo = ba0; bx.row_hi = bb0; bx.g_off = 0; bx.g_step = 1; bx.g_lo = ba1; bx.g_hi = bb1;
        MFFT_TRY(col_band(src, dst, N0, N1, ba2, Zi, plain(N1 * Zi), Za, plain(N1 * Za), bx));
      } else if (masked) {                                // `fu * dealias` (slab.py:237-245) applied while the spectrum is loaded
        mask_src = src;
        MFFT_TRY(col(src, dst, N0, true, N1, Nf, Zi, plain(N1 * Zi), Za, plain(N1 * Za)));
        mask_src = nullptr;
      } else if (Zi == Za && Za != Nf) {           // pitched caller rows: whole planes of N1 * Za columns
        MFFT_TRY(col_pad(src, dst, L0, true, pad ? 1 : 0, false, 1, N1 * Za, 0, plain(N1 * Za), 0, plain(N1 * Za), sc3 / (double)L0));
      } else if (Za != Nf) {                       // one outer batch per y row: compact rows in, pitched rows out
        MFFT_TRY(col_pad(src, dst, L0, true, pad ? 1 : 0, false, N1, Nf, Nf, plain(N1 * Nf), Za, plain(N1 * Za), sc3 / (double)L0,
                         0, 0, 1));
      } else {
        MFFT_TRY(col_pad(src, dst, L0, true, pad ? 1 : 0, false, 1, N1 * Nf, 0, plain(N1 * Nf), 0, plain(N1 * Nf), sc3 / (double)L0));
      }
    }
    return 0;
  }));
  for (int64_t i0 = 0; i0 < L0; i0 += mb) {
    const int64_t m = std::min(mb, L0 - i0);
    const double frac = (double)m / (double)L0;
    MFFT_TRY(stage("nl_y_inv", 6 * (Xb * keep1 + Yb) * keep2 * frac, [&] {
      for (int f = 0; f < 6; ++f) {
        const void* src = X + ((size_t)f * xelems + (size_t)(i0 * N1 * Za)) * es;
        void* dst = Y + (size_t)f * yelems * es;
        if (pruned) {
          ColArgs::Band by;
          by.row_lo = ba1; by.row_hi = bb1; by.c_lim = ba2;
          MFFT_TRY(col_band(src, dst, N1, m, ba2, N1 * Za, plain(Za), L1 * Za, plain(Za), by));
        } else {
          MFFT_TRY(col_pad(src, dst, L1, true, pad ? 1 : 0, false, m, Nf, N1 * Za, plain(Za), L1 * Za, plain(Za), 1.0 / (double)L1));
        }
      }
      return 0;
    }));
    MFFT_TRY(stage("nl_z", (6 * keep2 + 3) * Yb * frac, [&] {
      NlzArgs z;
      for (int f = 0; f < 3; ++f) {
        z.a[f] = Y + (size_t)f * yelems * es;
        z.b[f] = Y + (size_t)(3 + f) * yelems * es;
        z.out[f] = Y + (size_t)f * yelems * es;
      }
      z.n = (int)L2; z.prec = prec; z.in_stride = Za; z.out_stride = Za; z.nrows = m * L1; z.valid = (int)Nf;
      z.valid_in = pruned ? ba2 : 0;
      z.scale = 1.0 / ((double)L2 * (double)L2);
      return launch_nlz(z, stream);
    }));
    MFFT_TRY(stage("nl_y_fwd", 3 * (Xb + Yb) * frac, [&] {
      for (int f = 0; f < 3; ++f)
        MFFT_TRY(col_pad(Y + (size_t)f * yelems * es, X + ((size_t)f * xelems + (size_t)(i0 * N1 * Za)) * es, L1, false, pad ? 2 : 0,
                         pad, m, Nf, L1 * Za, plain(Za), N1 * Za, plain(Za), 1.0));
      return 0;
    }));
  }
  MFFT_TRY(stage("nl_x_fwd", 3 * (Cb + Xb), [&] {
    for (int f = 0; f < 3; ++f) {
      const void* src = X + (size_t)f * xelems * es;
      void* dst = static_cast<char*>(out) + (size_t)(f * C) * es;
      if (Zi == Za && Za != Nf)                    // pitched result
        MFFT_TRY(col_pad(src, dst, L0, false, pad ? 2 : 0, pad, 1, N1 * Za, 0, plain(N1 * Za), 0, plain(N1 * Za), 1.0 / sc3));
      else if (Za != Nf)                           // tiles of the compact result; input column (y, z) sits at y * Za + z
        MFFT_TRY(col_pad(src, dst, L0, false, pad ? 2 : 0, pad, 1, N1 * Nf, 0, plain(N1 * Za), 0, plain(N1 * Nf), 1.0 / sc3, Nf, Za - Nf));
      else
        MFFT_TRY(col_pad(src, dst, L0, false, pad ? 2 : 0, pad, 1, N1 * Nf, 0, plain(N1 * Nf), 0, plain(N1 * Nf), 1.0 / sc3));
    }
    return 0;
  }));
  return 0;
}

// The same over several ranks of a slab plan (blocking exchanges, whatever pipeline the plan's transforms use): six inverse x
// passes, six all-to-alls, then batches of the rank's x planes -- inverse y passes reading the receive layout through the
// two-level row map (transpose_Uc fused, maths.pyx:21-31), the fused z kernel, forward y passes writing the packed send
// layout (slab.py:403) -- three all-to-alls, three forward x passes.  Nine exchanges as in the composition, no real arrays.
int mfft_plan_s::nonlinear_cross_fused_ranks(const void* a, const void* b, void* out, int dealias) {
  const bool pad = dealias == MFFT_DEALIAS_3_2, masked = dealias == MFFT_DEALIAS_2_3;
  const int64_t L0 = pad ? M0 : N0, L1 = pad ? M1 : N1, L2 = pad ? M2 : N2, Lp0 = L0 / P;
  const int64_t line = (int64_t)(128 / es);
  const int64_t Za = Nf * (int64_t)es >= 2048 ? (Nf + line - 1) / line * line : Nf;        // batch buffers: line-aligned rows
  const int64_t S = Np1 * Nf + (pad ? 0 : xplane_pad(true));      // x-row pitch of the forward exchange's layout (sched())
  const int64_t C = N0 * Np1 * Nf;                                 // one component of the caller's arrays
  const size_t xelems = (size_t)(L0 * S);                          // one field in any of the exchanged layouts
  static const long batch_mb = getenv("MFFT_NLZ_BATCH_MB") ? atol(getenv("MFFT_NLZ_BATCH_MB")) : 16384;
  const size_t plane6 = (size_t)(6 * L1 * Za) * es;
  const int64_t nbat = (int64_t)((plane6 * (size_t)Lp0 + ((size_t)batch_mb << 20) - 1) / ((size_t)batch_mb << 20));
  const int64_t mb = (Lp0 + std::max<int64_t>(nbat, 1) - 1) / std::max<int64_t>(nbat, 1);
  MFFT_TRY(ensure_xbuf(0, 6 * xelems * es));
  MFFT_TRY(ensure_xbuf(1, 6 * xelems * es));
  MFFT_TRY(ensure_buf(&nly, &nly_bytes, (size_t)mb * plane6));
  char *X = static_cast<char*>(nlw[0]), *R = static_cast<char*>(nlw[1]), *Y = static_cast<char*>(nly);
  const size_t yelems = (size_t)(mb * L1 * Za);
  const double sc3 = pad ? padscale() : 1.0;
  struct MaskScope {
    mfft_plan_s* p;
    ~MaskScope() { p->mask_src = nullptr; p->lband_use = false; }
  } mask_scope{this};
  lband_use = false;
  // 2/3-rule with the reference's own filter: the six inverse transforms pruned as in slab_backward's blocking route -- the x
  // pass reads the kept kx rows and writes (N0, Np1, ap) with the kept kz only (zeros for the ky this rank's mask removes), the
  // six inverse exchanges carry ap / Nf of the bytes, the y pass and the fused z kernel work on a2 bins per row
  const int64_t a2 = ba2, ap = (a2 + line - 1) / line * line;      // rows of the pruned layout start on cache lines
  const bool pruned = masked && band_ok && ap <= Nf && !(getenv("MFFT_NO_PRUNE") && atoi(getenv("MFFT_NO_PRUNE")) != 0);
  MFFT_TRY(stage("nl_x_inv", 0, [&] {
    for (int f = 0; f < 6; ++f) {
      const void* src = static_cast<const char*>(f < 3 ? a : b) + (size_t)((f % 3) * C) * es;
      void* dst = X + (size_t)f * xelems * es;
      if (pruned) {
        if (band_allzero) {                        // nothing of this rank's spectrum survives the mask
          MFFT_TRY(zero(dst, (size_t)(N0 * Np1 * ap) * es));
        } else {
          ColArgs::Band bx;
          bx.row_lo = ba0; bx.row_hi = bb0; bx.g_off = 0; bx.g_step = 1; bx.g_lo = ba1; bx.g_hi = bb1; bx.g_zero = 1;
          MFFT_TRY(col_band(src, dst, N0, Np1, a2, Nf, plain(Np1 * Nf), ap, plain(Np1 * ap), bx));
        }
      } else if (masked) {
        mask_src = src;
        MFFT_TRY(col(src, dst, N0, true, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf)));
        mask_src = nullptr;
      } else {
        MFFT_TRY(col_pad(src, dst, L0, true, pad ? 1 : 0, false, 1, Np1 * Nf, 0, plain(Np1 * Nf), 0, plain(Np1 * Nf), sc3 / (double)L0));
      }
    }
    return 0;
  }));
  MFFT_TRY(stage("nl_a2a_inv", 0, [&] {
    for (int f = 0; f < 6; ++f) {
      if (pruned) MFFT_TRY(exchange_equal(world, X + (size_t)f * xelems * es, R + (size_t)f * xelems * es, (size_t)(Np0 * Np1 * ap) * es));
      else MFFT_TRY(xchg(0, false, pad, X + (size_t)f * xelems * es, R + (size_t)f * xelems * es));
    }
    return 0;
  }));
  for (int64_t i0 = 0; i0 < Lp0; i0 += mb) {
    const int64_t m = std::min(mb, Lp0 - i0);
    MFFT_TRY(stage("nl_y_inv", 0, [&] {
      for (int f = 0; f < 6; ++f) {
        if (pruned)
          MFFT_TRY(col(R + ((size_t)f * xelems + (size_t)(i0 * Np1 * ap)) * es, Y + (size_t)f * yelems * es, N1, true, m, a2, Np1 * ap,
                       two_level(Np1, Np0 * Np1 * ap, ap), L1 * Za, plain(Za)));
        else
          MFFT_TRY(col_pad(R + ((size_t)f * xelems + (size_t)(i0 * Np1 * Nf)) * es, Y + (size_t)f * yelems * es, L1, true, pad ? 1 : 0,
                           false, m, Nf, Np1 * Nf, two_level(Np1, Lp0 * Np1 * Nf, Nf), L1 * Za, plain(Za), 1.0 / (double)L1));
      }
      return 0;
    }));
    MFFT_TRY(stage("nl_z", 0, [&] {
      NlzArgs z;
      for (int f = 0; f < 3; ++f) {
        z.a[f] = Y + (size_t)f * yelems * es;
        z.b[f] = Y + (size_t)(3 + f) * yelems * es;
        z.out[f] = Y + (size_t)f * yelems * es;
      }
      z.n = (int)L2; z.prec = prec; z.in_stride = Za; z.out_stride = Za; z.nrows = m * L1; z.valid = (int)Nf;
      z.valid_in = pruned ? (int)a2 : 0;
      z.scale = 1.0 / ((double)L2 * (double)L2);
      return launch_nlz(z, stream);
    }));
    MFFT_TRY(stage("nl_y_fwd", 0, [&] {      // truncate + fold in y, straight into the packed (P, Lp0, S) send layout
      for (int f = 0; f < 3; ++f)
        MFFT_TRY(col_pad(Y + (size_t)f * yelems * es, X + ((size_t)f * xelems + (size_t)(i0 * S)) * es, L1, false, pad ? 2 : 0, pad, m,
                         Nf, L1 * Za, plain(Za), S, two_level(Np1, Lp0 * S, Nf), 1.0));
      return 0;
    }));
  }
  MFFT_TRY(stage("nl_a2a_fwd", 0, [&] {
    for (int f = 0; f < 3; ++f) MFFT_TRY(xchg(0, true, pad, X + (size_t)f * xelems * es, R + (size_t)f * xelems * es));
    return 0;
  }));
  MFFT_TRY(stage("nl_x_fwd", 0, [&] {
    for (int f = 0; f < 3; ++f)
      MFFT_TRY(col_pad(R + (size_t)f * xelems * es, static_cast<char*>(out) + (size_t)(f * C) * es, L0, false, pad ? 2 : 0, pad, 1,
                       Np1 * Nf, 0, plain(S), 0, plain(Np1 * Nf), 1.0 / sc3));
    return 0;
  }));
  return 0;
}

int mfft_plan_s::nonlinear_cross(const void* a, const void* b, void* out, int dealias) {
  if (dealias == MFFT_DEALIAS_2_3 && (!mask || mask_count != (size_t)local_complex_count()))
    return set_error(MFFT_ERR_INVALID, "2/3-rule requested but no dealias mask was set");
  if (nonlinear_fusable(dealias)) return P == 1 ? nonlinear_cross_fused(a, b, out, dealias) : nonlinear_cross_fused_ranks(a, b, out, dealias);
  return nonlinear_cross_composed(a, b, out, dealias);
}

// ===========================================================================
// pencil
// ===========================================================================
// pack the z chunks of Z (rows, Nf) into consecutive (rows, len_l) blocks / the reverse
static int pack_z(mfft_plan_s* p, const void* Z, void* S, int64_t rows, int64_t nf, const std::vector<Chunk>& zc, bool unpack) {
  size_t off = 0;
  for (const Chunk& c : zc) {
    const char* zp = static_cast<const char*>(Z) + (size_t)c.start * p->es;
    char* sp = static_cast<char*>(S) + off;
    if (!unpack) MFFT_TRY(p->box(zp, sp, 1, rows, c.len, 0, nf, 0, c.len));
    else MFFT_TRY(p->box(sp, const_cast<char*>(zp), 1, rows, c.len, 0, c.len, 0, nf));
    off += (size_t)(rows * c.len) * p->es;
  }
  return 0;
}

// ---- pencil, X alignment: batches of local x rows pipelined through BOTH exchanges -------------------
// Everything up to the final x transform (forward) / after the first x transform (inverse) is independent per
// local x row i, and the blocks a row batch contributes to either exchange are contiguous in the packed layouts,
// so batch b's exchanges run on the communication stream while batch b+1 is transformed on the compute stream.
// Sub-schedules of one batch [i0, i0+mb) of the m local rows (bytes):
int mfft_plan_s::sched_rows(int which, bool forward, int64_t i0, int64_t mb, Sched* o) const {
  const int64_t m = N1_0, n = N2_1;
  const bool X = d.decomp == MFFT_PENCIL_X;
  fill_part(which == 0 ? !X : X, &o->part);
  if (which == 0) {            // z-splitting exchange (X: group1, Y: group0): uneven chunks <-> (m, n, q) blocks, rows [i0, i0+mb) of m
    const std::vector<int>& gz = X ? group1 : group0;
    const int Pz = (int)gz.size();
    o->peers = gz;
    o->sc.resize(Pz); o->sd.resize(Pz); o->rc.resize(Pz); o->rd.resize(Pz);
    size_t base = 0;
    const int64_t pq = zrow_pitch(q, forward);
    for (int l = 0; l < Pz; ++l) {
      const int64_t pl = zrow_pitch(zc[l].len, forward);
      const size_t usz = (size_t)(mb * n * pl) * es, uoff = base + (size_t)(i0 * n * pl) * es;
      const size_t esz = (size_t)(mb * n * pq) * es, eoff = (size_t)(l * m * n * pq + i0 * n * pq) * es;
      if (forward) { o->sc[l] = usz; o->sd[l] = uoff; o->rc[l] = esz; o->rd[l] = eoff; }
      else         { o->sc[l] = esz; o->sd[l] = eoff; o->rc[l] = usz; o->rd[l] = uoff; }
      base += (size_t)(m * n * pl) * es;
    }
    return 0;
  }
  if (!X) {
    // Y alignment, x-chunk exchange over group1 (P2 ranks): rows [i0, i0+mb) of the N2_0 rows of every block
    // [c][x'][j][k] (N2_0, n, q); the send block c is rows c*N2_0.. of (N0, n, q), the receive block c' the same shape
    const int Pg = (int)group1.size();
    // x-row pitch (inverse: padded where the compact one reads slowly; forward: rows at the z kernel's pitch)
    const int64_t SY = n * zrow_pitch(q, forward) + xplane_pad(forward);
    o->peers = group1;
    o->sc.assign(Pg, (size_t)(mb * SY) * es);
    o->rc = o->sc;
    o->sd.resize(Pg); o->rd.resize(Pg);
    for (int g = 0; g < Pg; ++g) o->sd[g] = o->rd[g] = (size_t)((g * N2_0 + i0) * SY) * es;
    return 0;
  }
  // X alignment, y-chunk exchange over group0 (P1 ranks): P1 blocks (m, N1_1, q) <-> rows of (N0, N1_1, q)
  const int Pg = (int)group0.size();
  const int64_t SX = N1_1 * q + xplane_pad(forward);        // x-row pitch (forward: padded where the compact one reads slowly)
  o->peers = group0;
  o->sc.assign(Pg, (size_t)(mb * SX) * es);
  o->rc = o->sc;
  o->sd.resize(Pg); o->rd.resize(Pg);
  for (int g = 0; g < Pg; ++g) {
    const size_t blk = (size_t)((g * m + i0) * SX) * es;                     // [g][i][j'][k]
    const size_t row = (size_t)((g * m + i0) * SX) * es;                     // x = g*m + i
    if (forward) { o->sd[g] = blk; o->rd[g] = row; }
    else         { o->sd[g] = row; o->rd[g] = blk; }
  }
  return 0;
}

// z chunks of rows [r0, r0+nr) of Z (rows_total, nf) <-> the matching sub-blocks of the packed chunk blocks
static int pack_z_rows(mfft_plan_s* p, const void* Z, void* S, int64_t rows_total, int64_t r0, int64_t nr, int64_t nf,
                       const std::vector<Chunk>& zc, bool unpack) {
  size_t base = 0;
  for (const Chunk& c : zc) {
    const char* zp = static_cast<const char*>(Z) + (size_t)(r0 * nf + c.start) * p->es;
    char* sp = static_cast<char*>(S) + base + (size_t)(r0 * c.len) * p->es;
    if (!unpack) MFFT_TRY(p->box(zp, sp, 1, nr, c.len, 0, nf, 0, c.len));
    else MFFT_TRY(p->box(sp, const_cast<char*>(zp), 1, nr, c.len, 0, c.len, 0, nf));
    base += (size_t)(rows_total * c.len) * p->es;
  }
  return 0;
}

int mfft_plan_s::pencil_forward_pipelined_x(const void* u, void* fu) {
  const int64_t m = N1_0, n = N2_1;
  const double Cb = (double)(m * n * Nf) * es, Rb = (double)(m * n * N2) * rs;
  const bool zsolo = P2 == 1, g2solo = P1 == 1;
  // x-row pitch of the blocks of the second exchange: N1_1 * q, plus a cache line where that pitch reads slowly
  // (xplane_pad); then the chunks land in a work buffer and the x transform runs out of place into the result
  const int64_t SX = N1_1 * q + xplane_pad(true);
  const bool xoop = SX != N1_1 * q;
  const int64_t PQ = zrow_pitch(q, true);        // row pitch of the received z blocks (zrow_pitch: q, or whole cache lines)
  const size_t wb = (size_t)std::max(std::max(std::max(m * n * Nf, zsend_elems(m * n)), m * N1 * PQ), xoop ? N0 * SX : (int64_t)0) * es;
  for (int i = 0; i < 3; ++i) MFFT_TRY(ensure_work(i, wb));
  char *W0 = static_cast<char*>(work[0]), *W1 = static_cast<char*>(work[1]), *W2 = static_cast<char*>(work[2]);
  const char* in = static_cast<const char*>(u);
  char* out = static_cast<char*>(fu);
  const int B = nbatch;
  auto rows = [&](int b, int64_t* i0, int64_t* mb) { *i0 = m * b / B; *mb = m * (b + 1) / B - *i0; };
  // z transform + z-chunk pack of a batch on the compute stream, its exchange on the communication stream
  for (int b = 0; b < B; ++b) {
    int64_t i0, mb;
    rows(b, &i0, &mb);
    if (!zsolo && zfuse) {          // the z transform writes the batch's rows of the send blocks itself
      MFFT_TRY(stage("fwd_z", (Rb + Cb) / B, [&] {
        return z_forward_chunked(in + (size_t)(i0 * n * N2) * rs, W1, mb * n, i0 * n, m * n);
      }));
    } else {
      MFFT_TRY(stage("fwd_z", (Rb + Cb) / B, [&] {
        return z_forward(in + (size_t)(i0 * n * N2) * rs, W0 + (size_t)(i0 * n * Nf) * es, mb * n, N2, Nf);
      }));
      if (zsolo) continue;
      MFFT_TRY(stage("fwd_packz", 0, [&] { return pack_z_rows(this, W0, W1, m * n, i0 * n, mb * n, Nf, zc, false); }));
    }
    MFFT_HIP(hipEventRecord(ev_compute[b], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[b], 0));
    MFFT_TRY(stage_on(cstream, "fwd_a2a_1", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, true, b, &sc));
      return run_sched(sc, W1, W2, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[b], cstream));
  }
  // y transform of a batch as soon as its z chunks have arrived; W0 is free again (all packs are behind us on
  // this stream) and takes the P1 blocks (m, N1_1, q) that feed the second exchange
  const char* ysrc = zsolo ? W0 : W2;
  char* ydst = g2solo ? out : (zsolo ? W1 : W0);
  // padded pitch: where the second exchange delivers.  W1 was the send buffer of the z exchanges, all of which are ahead
  // of every second exchange on the communication stream; on a P1 x 1 grid (no z exchange) W1 is ydst and W2 is unused
  char* xrecv = xoop ? (zsolo ? W2 : W1) : out;
  for (int b = 0; b < B; ++b) {
    int64_t i0, mb;
    rows(b, &i0, &mb);
    if (!zsolo) MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[b], 0));
    MFFT_TRY(stage("fwd_y", 2 * Cb / B, [&] {
      return col(ysrc + (size_t)(i0 * n * PQ) * es, ydst + (size_t)(i0 * SX) * es, N1, false, mb, q, n * PQ,
                 two_level(n, m * n * PQ, PQ), SX, two_level(N1_1, m * SX, q));
    }));
    if (g2solo) continue;
    MFFT_HIP(hipEventRecord(ev2_compute[b], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev2_compute[b], 0));
    MFFT_TRY(stage_on(cstream, "fwd_a2a_2", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(1, true, b, &sc));
      return run_sched(sc, ydst, xrecv, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev2_comm[b], cstream));
  }
  if (!g2solo) MFFT_HIP(hipStreamWaitEvent(stream, ev2_comm[B - 1], 0));     // in order on the comm stream: all batches
  if (xoop) {
    MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(xrecv, fu, N0, false, 1, N1_1 * q, 0, plain(SX), 0, plain(N1_1 * q)); }));
    return 0;
  }
  MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(fu, fu, N0, false, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q)); }));
  return 0;
}

int mfft_plan_s::pencil_backward_pipelined_x(const void* src, void* u) {
  const int64_t m = N1_0, n = N2_1;
  const double Cb = (double)(m * n * Nf) * es, Rb = (double)(m * n * N2) * rs;
  const bool zsolo = P2 == 1, g2solo = P1 == 1;
  const size_t wb = (size_t)std::max(m * n * Nf, m * N1 * q) * es;
  for (int i = 0; i < 2; ++i) MFFT_TRY(ensure_work(i, wb));
  // third buffer: work[2], unless it holds the masked copy of the spectrum (src)
  const bool src_in_work2 = work[2] != nullptr && src == work[2];
  if (src_in_work2) MFFT_TRY(ensure_work3(wb));
  else MFFT_TRY(ensure_work(2, wb));
  char *W0 = static_cast<char*>(work[0]), *W1 = static_cast<char*>(work[1]);
  char* W2 = static_cast<char*>(src_in_work2 ? work3 : work[2]);
  char* out = static_cast<char*>(u);
  const int B = nbatch;
  auto rows = [&](int b, int64_t* i0, int64_t* mb) { *i0 = m * b / B; *mb = m * (b + 1) / B - *i0; };
  MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(src, W0, N0, true, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q)); }));
  if (!g2solo) {
    MFFT_HIP(hipEventRecord(ev2_compute[0], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev2_compute[0], 0));
    for (int b = 0; b < B; ++b) {
      int64_t i0, mb;
      rows(b, &i0, &mb);
      MFFT_TRY(stage_on(cstream, "bwd_a2a_2", 0, [&] {
        Sched sc;
        MFFT_TRY(piece_sched(1, false, b, &sc));
        return run_sched(sc, W0, W1, cstream);
      }));
      MFFT_HIP(hipEventRecord(ev2_comm[b], cstream));
    }
  }
  // y transform of a batch (P1 blocks gathered through the row map) -> P2 blocks (m, n, q) in W2, then its z exchange
  const char* ysrc = g2solo ? W0 : W1;
  // Where the z exchange delivers its chunks.  With a second exchange (P1 > 1) every bwd_a2a_2 is ahead of it on the
  // communication stream and the y transforms read W1, so W0 is free.  On a 1 x P2 grid there is no second exchange and
  // the y transforms of LATER batches still read W0 on the compute stream: the chunks go to W1 (unused there) and are
  // unpacked into W0 once every y transform is behind the unpack on the compute stream.
  char* zrecv = g2solo ? W1 : W0;
  char* zfull = g2solo ? W0 : W1;
  for (int b = 0; b < B; ++b) {
    int64_t i0, mb;
    rows(b, &i0, &mb);
    if (!g2solo) MFFT_HIP(hipStreamWaitEvent(stream, ev2_comm[b], 0));
    MFFT_TRY(stage("bwd_y", 2 * Cb / B, [&] {
      return col(ysrc + (size_t)(i0 * N1_1 * q) * es, W2 + (size_t)(i0 * n * q) * es, N1, true, mb, q, N1_1 * q,
                 two_level(N1_1, m * N1_1 * q, q), n * q, two_level(n, m * n * q, q));
    }));
    if (zsolo) continue;
    MFFT_HIP(hipEventRecord(ev_compute[b], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[b], 0));
    MFFT_TRY(stage_on(cstream, "bwd_a2a_1", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, false, b, &sc));
      return run_sched(sc, W2, zrecv, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[b], cstream));
  }
  // z chunks of a batch back into full rows (zfull: every y transform that read it is behind us on this stream), c2r
  for (int b = 0; b < B; ++b) {
    int64_t i0, mb;
    rows(b, &i0, &mb);
    const char* zin = W2 + (size_t)(i0 * n * Nf) * es;
    if (!zsolo) {
      MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[b], 0));
      if (zfuse) {                    // the z transform reads the batch's rows out of the received blocks itself
        MFFT_TRY(stage("bwd_z", (Rb + Cb) / B, [&] {
          return z_backward_chunked(zrecv, out + (size_t)(i0 * n * N2) * rs, mb * n, i0 * n, m * n);
        }));
        continue;
      }
      MFFT_TRY(stage("bwd_unpackz", 0, [&] { return pack_z_rows(this, zfull, zrecv, m * n, i0 * n, mb * n, Nf, zc, true); }));
      zin = zfull + (size_t)(i0 * n * Nf) * es;
    }
    MFFT_TRY(stage("bwd_z", (Rb + Cb) / B, [&] {
      return z_backward(zin, out + (size_t)(i0 * n * N2) * rs, mb * n, N2, Nf);
    }));
  }
  return 0;
}

// ---- pencil, Y alignment: exchange pipeline in two halves ------------------------------------------------------
// (pencil.py:730-754 forward, 483-507 inverse).  The x transform between the two exchanges needs every row of both,
// so the pipeline is cut there: the z stage runs in batches of the local x rows, each batch's z-splitting exchange on
// the communication stream while the next batch is transformed (the z transform writes the send blocks itself: zfuse);
// after the x transform the x-chunk exchange goes out in batches of the rows a rank owns afterwards, and the y
// transform of a batch starts as soon as that batch has landed.  Compute order z0 z1 .. x y0 y1 ..; only z0, the x
// transform and the last y batch are not overlapped.  Needs P1 > 1, P2 > 1 and the fused z-chunk kernels.
int mfft_plan_s::pencil_forward_pipelined_y(const void* u, void* fu) {
  const int64_t m = N1_0, n = N2_1;
  const double Cb = (double)(m * n * Nf) * es, Rb = (double)(m * n * N2) * rs;
  const int64_t PQ = zrow_pitch(q, true);        // row pitch of the z blocks: it stays through the x pass and the second exchange
  const size_t wb = (size_t)std::max(std::max(m * n * Nf, zsend_elems(m * n)), N0 * n * PQ) * es;
  for (int i = 0; i < 2; ++i) MFFT_TRY(ensure_work(i, wb));
  char *W0 = static_cast<char*>(work[0]), *W1 = static_cast<char*>(work[1]);
  const char* in = static_cast<const char*>(u);
  char* out = static_cast<char*>(fu);
  const int B = nbatch;
  for (int b = 0; b < B; ++b) {
    const int64_t i0 = m * b / B, mb = m * (b + 1) / B - i0;
    MFFT_TRY(stage("fwd_z", (Rb + Cb) / B, [&] {
      return z_forward_chunked(in + (size_t)(i0 * n * N2) * rs, W1, mb * n, i0 * n, m * n);
    }));
    MFFT_HIP(hipEventRecord(ev_compute[b], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[b], 0));
    MFFT_TRY(stage_on(cstream, "fwd_a2a_1", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, true, b, &sc));
      return run_sched(sc, W1, W0, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[b], cstream));
  }
  MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[B - 1], 0));                    // in order on the comm stream: all batches
  MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(W0, W0, N0, false, 1, n * PQ, 0, plain(n * PQ), 0, plain(n * PQ)); }));
  MFFT_HIP(hipEventRecord(ev2_compute[0], stream));
  MFFT_HIP(hipStreamWaitEvent(cstream, ev2_compute[0], 0));
  for (int b = 0; b < B; ++b) {
    MFFT_TRY(stage_on(cstream, "fwd_a2a_2", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(1, true, b, &sc));
      return run_sched(sc, W0, W1, cstream);                                  // W1: every z exchange read it long ago
    }));
    MFFT_HIP(hipEventRecord(ev2_comm[b], cstream));
  }
  for (int b = 0; b < B; ++b) {
    const int64_t x0 = N2_0 * b / B, xb = N2_0 * (b + 1) / B - x0;
    MFFT_HIP(hipStreamWaitEvent(stream, ev2_comm[b], 0));
    MFFT_TRY(stage("fwd_y", 2 * Cb / B, [&] {
      return col(W1 + (size_t)(x0 * n * PQ) * es, out + (size_t)(x0 * N1 * q) * es, N1, false, xb, q, n * PQ,
                 two_level(n, N2_0 * n * PQ, PQ), N1 * q, plain(q));
    }));
  }
  return 0;
}

int mfft_plan_s::pencil_backward_pipelined_y(const void* src, void* u) {
  const int64_t m = N1_0, n = N2_1;
  const double Cb = (double)(m * n * Nf) * es, Rb = (double)(m * n * N2) * rs;
  const int64_t SY = n * q + xplane_pad(false);        // x-row pitch of the blocks of the x-chunk exchange (see xplane_pad)
  const bool xoop = SY != n * q;
  const size_t wb = (size_t)std::max(m * n * Nf, N0 * SY) * es;
  for (int i = 0; i < 2; ++i) MFFT_TRY(ensure_work(i, wb));
  char *W0 = static_cast<char*>(work[0]), *W1 = static_cast<char*>(work[1]);
  const char* in = static_cast<const char*>(src);
  char* out = static_cast<char*>(u);
  const int B = nbatch;
  // y transform of a batch of my rows (written as P2 blocks (N2_0, n, q)), its x-chunk exchange behind it
  for (int b = 0; b < B; ++b) {
    const int64_t x0 = N2_0 * b / B, xb = N2_0 * (b + 1) / B - x0;
    MFFT_TRY(stage("bwd_y", 2 * Cb / B, [&] {
      return col(in + (size_t)(x0 * N1 * q) * es, W0 + (size_t)(x0 * SY) * es, N1, true, xb, q, N1 * q, plain(q), SY,
                 two_level(n, N2_0 * SY, q));
    }));
    MFFT_HIP(hipEventRecord(ev2_compute[b], stream));
    MFFT_HIP(hipStreamWaitEvent(cstream, ev2_compute[b], 0));
    MFFT_TRY(stage_on(cstream, "bwd_a2a_2", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(1, false, b, &sc));
      return run_sched(sc, W0, W1, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev2_comm[b], cstream));
  }
  MFFT_HIP(hipStreamWaitEvent(stream, ev2_comm[B - 1], 0));
  // x transform: in place on the received (N0, n, q), or -- rows SY apart -- out of place into W0, which every x-chunk
  // exchange has read by now; the z-gathering exchange then goes the other way round
  char *xsend = W1, *zrecv = W0;
  if (xoop) {
    MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(W1, W0, N0, true, 1, n * q, 0, plain(SY), 0, plain(n * q)); }));
    xsend = W0;
    zrecv = W1;
  } else {
    MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(W1, W1, N0, true, 1, n * q, 0, plain(n * q), 0, plain(n * q)); }));
  }
  MFFT_HIP(hipEventRecord(ev_compute[0], stream));
  MFFT_HIP(hipStreamWaitEvent(cstream, ev_compute[0], 0));
  for (int b = 0; b < B; ++b) {
    MFFT_TRY(stage_on(cstream, "bwd_a2a_1", 0, [&] {
      Sched sc;
      MFFT_TRY(piece_sched(0, false, b, &sc));
      return run_sched(sc, xsend, zrecv, cstream);
    }));
    MFFT_HIP(hipEventRecord(ev_comm[b], cstream));
  }
  for (int b = 0; b < B; ++b) {
    const int64_t i0 = m * b / B, mb = m * (b + 1) / B - i0;
    MFFT_HIP(hipStreamWaitEvent(stream, ev_comm[b], 0));
    MFFT_TRY(stage("bwd_z", (Rb + Cb) / B, [&] {
      return z_backward_chunked(zrecv, out + (size_t)(i0 * n * N2) * rs, mb * n, i0 * n, m * n);
    }));
  }
  return 0;
}

int mfft_plan_s::pencil_forward(const void* u, void* fu) {
  if (nbatch > 1) return d.decomp == MFFT_PENCIL_X ? pencil_forward_pipelined_x(u, fu) : pencil_forward_pipelined_y(u, fu);
  const int64_t m = N1_0, n = N2_1;                 // local real rows in x, y
  const double Cb = (double)(m * n * Nf) * es, Rb = (double)(m * n * N2) * rs;
  const bool X = d.decomp == MFFT_PENCIL_X;
  // a group of one rank exchanges nothing: its pack / copy steps are skipped altogether
  const bool zsolo = (X ? P2 : P1) == 1 && !d.drop_nyquist, g2solo = (X ? P1 : P2) == 1;
  // largest intermediate of this alignment: X: (m, N1, q) after the z exchange; Y: (N0, n, q) after it
  const int64_t SX = N1_1 * q + (X ? xplane_pad(true) : 0);     // x-row pitch of the blocks of the second exchange (X)
  // row pitch of the received z blocks: q, or whole cache lines where the fused z kernel wrote them so (zrow_pitch)
  const int64_t PQ = (!zsolo && zfuse) ? zrow_pitch(q, true) : q;
  const size_t wb = (size_t)std::max(std::max(m * n * Nf, zsend_elems(m * n)), X ? std::max(m * N1 * PQ, N0 * SX) : N0 * n * PQ) * es;
  MFFT_TRY(ensure_work(0, wb));
  MFFT_TRY(ensure_work(1, wb));
  void *W0 = work[0], *W1 = work[1];
  if (!zsolo && zfuse) {            // z transform straight into the Pz send blocks (pencil.py:218-246 fused)
    MFFT_TRY(stage("fwd_z", Rb + Cb, [&] { return z_forward_chunked(u, W1, m * n, 0, m * n); }));
    MFFT_TRY(stage("fwd_a2a_1", 0, [&] { return xchg(0, true, false, W1, W0); }));
  } else {
    MFFT_TRY(stage("fwd_z", Rb + Cb, [&] { return z_forward(u, W0, m * n, N2, Nf); }));
    if (!zsolo) {
      MFFT_TRY(stage("fwd_packz", 0, [&] { return pack_z(this, W0, W1, m * n, Nf, zc, false); }));
      MFFT_TRY(stage("fwd_a2a_1", 0, [&] { return xchg(0, true, false, W1, W0); }));
    }
  }
  if (X) {
    // W0 = P2 blocks (m, n, q) -> y transform (gathers y through two-level rows) -> P1 blocks (m, N1_1, q)
    void* ydst = g2solo ? fu : W1;
    MFFT_TRY(stage("fwd_y", 2 * Cb, [&] {
      return col(W0, ydst, N1, false, m, q, n * PQ, two_level(n, m * n * PQ, PQ), SX, two_level(N1_1, m * SX, q));
    }));
    if (g2solo || (xpass_inplace && SX == N1_1 * q)) {
      if (!g2solo) MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, false, W1, fu); }));
      MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(fu, fu, N0, false, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q)); }));
    } else {
      // the chunks land in W0 (free: the y transform has read it), rows SX apart; x transform out of place into the result
      MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, false, W1, W0); }));
      MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(W0, fu, N0, false, 1, N1_1 * q, 0, plain(SX), 0, plain(N1_1 * q)); }));
    }
  } else {
    // W0 = (N0, n, q): x transform in place, x chunks are contiguous -> exchange -> y transform gathers
    MFFT_TRY(stage("fwd_x", 2 * Cb, [&] { return col(W0, W0, N0, false, 1, n * PQ, 0, plain(n * PQ), 0, plain(n * PQ)); }));
    void* ysrc = W0;
    if (!g2solo) {
      MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, false, W0, W1); }));
      ysrc = W1;
    }
    MFFT_TRY(stage("fwd_y", 2 * Cb, [&] {
      return col(ysrc, fu, N1, false, N2_0, q, n * PQ, two_level(n, N2_0 * n * PQ, PQ), N1 * q, plain(q));
    }));
  }
  return 0;
}

int mfft_plan_s::pencil_backward(const void* fu, void* u, bool masked) {
  const int64_t m = N1_0, n = N2_1;
  const double Cb = (double)(m * n * Nf) * es, Rb = (double)(m * n * N2) * rs;
  const bool X = d.decomp == MFFT_PENCIL_X;
  const bool zsolo = (X ? P2 : P1) == 1 && !d.drop_nyquist, g2solo = (X ? P1 : P2) == 1;
  const void* src = fu;
  struct MaskScope {
    mfft_plan_s* p;
    ~MaskScope() { p->mask_src = nullptr; p->lband_use = false; }
  } mask_scope{this};
  if (masked) {
    bool fused = false;
    MFFT_TRY(fuse_mask(fu, X ? N0 : N1, &fused));
    if (!fused) {
      void* mm = nullptr;
      MFFT_TRY(stage("bwd_mask", 2 * Cb, [&] { return apply_mask_copy(fu, &mm); }));
      src = mm;
    }
  }
  if (nbatch > 1) return d.decomp == MFFT_PENCIL_X ? pencil_backward_pipelined_x(src, u) : pencil_backward_pipelined_y(src, u);
  // largest intermediate of this alignment: X: (m, N1, q) after the z exchange; Y: (N0, n, q) after it
  const int64_t SY = n * q + (X ? 0 : xplane_pad(false));       // x-row pitch of the blocks of the second exchange (Y)
  const size_t wb = (size_t)std::max(m * n * Nf, X ? m * N1 * q : N0 * SY) * es;
  MFFT_TRY(ensure_work(0, wb));
  MFFT_TRY(ensure_work(1, wb));
  void *W0 = work[0], *W1 = work[1];
  void* cur = nullptr;     // buffer holding the Pz blocks (m, n, q) that enter the z-gathering exchange
  if (X) {
    MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(src, W0, N0, true, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q)); }));
    void* ysrc = W0;
    if (!g2solo) {
      MFFT_TRY(stage("bwd_a2a_2", 0, [&] { return xchg(1, false, false, W0, W1); }));
      ysrc = W1;
    }
    cur = ysrc == W0 ? W1 : W0;
    MFFT_TRY(stage("bwd_y", 2 * Cb, [&] {
      return col(ysrc, cur, N1, true, m, q, N1_1 * q, two_level(N1_1, m * N1_1 * q, q), n * q, two_level(n, m * n * q, q));
    }));
  } else {
    MFFT_TRY(stage("bwd_y", 2 * Cb, [&] {
      return col(src, W0, N1, true, N2_0, q, N1 * q, plain(q), SY, two_level(n, N2_0 * SY, q));
    }));
    cur = W0;
    if (!g2solo) {
      MFFT_TRY(stage("bwd_a2a_2", 0, [&] { return xchg(1, false, false, W0, W1); }));
      cur = W1;
    }
    if (g2solo || (xpass_inplace && SY == n * q)) {
      // (N0, n, q): x transform in place; its x chunks are the contiguous blocks of the next exchange
      MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(cur, cur, N0, true, 1, n * q, 0, plain(n * q), 0, plain(n * q)); }));
    } else {
      // rows SY apart in W1 -> compact (N0, n, q) in W0 (free: the exchange has sent it), out of place
      MFFT_TRY(stage("bwd_x", 2 * Cb, [&] { return col(W1, W0, N0, true, 1, n * q, 0, plain(SY), 0, plain(n * q)); }));
      cur = W0;
    }
  }
  if (!zsolo) {
    void* other = cur == W0 ? W1 : W0;
    MFFT_TRY(stage("bwd_a2a_1", 0, [&] { return xchg(0, false, false, cur, other); }));
    if (zfuse) {      // the z transform reads the received Pz blocks itself (a dropped Nyquist column reads as zero)
      MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward_chunked(other, u, m * n, 0, m * n); }));
      return 0;
    }
    MFFT_TRY(stage("bwd_unpackz", 0, [&] { return pack_z(this, cur, other, m * n, Nf, zc, true); }));
  }
  if (d.drop_nyquist)   // the neglected Nyquist column counts as zero (pencil.py:430, 1045)
    MFFT_HIP(hipMemset2DAsync(static_cast<char*>(cur) + (size_t)(Nf - 1) * es, (size_t)Nf * es, 0, es, (size_t)(m * n), stream));
  MFFT_TRY(stage("bwd_z", Rb + Cb, [&] { return z_backward(cur, u, m * n, N2, Nf); }));
  return 0;
}

// ---- fused 3/2-rule, pencil: same idea as the slab (see can_fuse_pad): pad-on-load / truncate-on-store
// column kernels whose two-level row maps also do the y-chunk pack / unpack, column-limited real kernels.
// Only the z-chunk pack / unpack around the z-splitting exchange remain as copies.
int mfft_plan_s::pencil_backward_padded_fused(const void* fu, void* u) {
  const double sc3 = padscale();
  const bool X = d.decomp == MFFT_PENCIL_X;
  const int64_t mp = M0 / P1, np = M1 / P2;          // padded local real rows in x, y
  const size_t wb = (size_t)std::max(std::max(M0 * N1_1 * q, mp * M1 * q), std::max(std::max(N2_0 * M1 * q, M0 * np * q), mp * np * Nf)) * es;
  for (int i = 0; i < 3; ++i) MFFT_TRY(ensure_work(i, wb));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  // a group of one rank exchanges nothing: with the fused z kernels its exchange is skipped altogether (the transform
  // on the far side reads the buffer the near side wrote)
  const bool fz = zfuse_pad();
  const bool zsolo = fz && (X ? P2 : P1) == 1, g2solo = fz && (X ? P1 : P2) == 1;
  void* cur = W0;                                    // what the next stage reads
  auto other = [&](void* b) { return b == W0 ? W1 : W0; };
  if (X) {
    // fu (N0, N1_1, q) -> ifft x over M0 rows, the zero band never read
    MFFT_TRY(stage("bwd_x", 0, [&] {
      return col_pad(fu, W0, M0, true, 1, false, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q), sc3 / (double)M0);
    }));
    if (!g2solo) {
      MFFT_TRY(stage("bwd_a2a_2", 0, [&] { return xchg(1, false, true, W0, W1); }));
      cur = W1;
    }
    // cur = P1 blocks (mp, N1_1, q): gather y through the input row map, pad on load, write P2 blocks (mp, np, q)
    void* dst = other(cur);
    MFFT_TRY(stage("bwd_y", 0, [&] {
      return col_pad(cur, dst, M1, true, 1, false, mp, q, N1_1 * q, two_level(N1_1, mp * N1_1 * q, q), np * q,
                     two_level(np, mp * np * q, q), 1.0 / (double)M1);
    }));
    cur = dst;
  } else {
    // fu (N2_0, N1, q) -> ifft y over M1 rows, written as P2 blocks (N2_0, np, q)
    MFFT_TRY(stage("bwd_y", 0, [&] {
      return col_pad(fu, W0, M1, true, 1, false, N2_0, q, N1 * q, plain(q), np * q, two_level(np, N2_0 * np * q, q),
                     sc3 / (double)M1);
    }));
    if (!g2solo) {
      MFFT_TRY(stage("bwd_a2a_2", 0, [&] { return xchg(1, false, true, W0, W1); }));
      cur = W1;
    }
    // cur = (N0, np, q) -> ifft x over M0 rows; its x chunks (mp rows) are the blocks of the next exchange
    void* dst = other(cur);
    MFFT_TRY(stage("bwd_x", 0, [&] {
      return col_pad(cur, dst, M0, true, 1, false, 1, np * q, 0, plain(np * q), 0, plain(np * q), 1.0 / (double)M0);
    }));
    cur = dst;
  }
  if (!zsolo) {
    void* dst = other(cur);
    MFFT_TRY(stage("bwd_a2a_1", 0, [&] { return xchg(0, false, true, cur, dst); }));
    cur = dst;
  }
  // only the Nf kept columns exist: c2r reads the others as zeros
  if (fz) {                     // ... and reads the kept ones out of the received z-chunk blocks itself
    MFFT_TRY(stage("bwd_z", 0, [&] {
      RealArgs a;
      a.in = cur; a.out = u; a.n = (int)M2; a.prec = prec; a.in_stride = Nf; a.out_stride = M2; a.nrows = mp * np;
      a.scale = 1.0 / (double)M2; a.valid = (int)Nf; a.zs = zsplit(mp * np, 0);
      return launch_c2r(a, stream);
    }));
    return 0;
  }
  MFFT_TRY(stage("bwd_unpackz", 0, [&] { return pack_z(this, W2, cur, mp * np, Nf, zc, true); }));
  MFFT_TRY(stage("bwd_z", 0, [&] { return c2r_rows(W2, u, mp * np, M2, Nf, M2, 1.0 / (double)M2, (int)Nf); }));
  return 0;
}

int mfft_plan_s::pencil_forward_padded_fused(const void* u, void* fu) {
  const double isc3 = 1.0 / padscale();
  const bool X = d.decomp == MFFT_PENCIL_X;
  const int64_t mp = M0 / P1, np = M1 / P2;
  const size_t wb = (size_t)std::max(std::max(M0 * N1_1 * q, mp * M1 * q), std::max(std::max(N2_0 * M1 * q, M0 * np * q), mp * np * Nf)) * es;
  for (int i = 0; i < 3; ++i) MFFT_TRY(ensure_work(i, wb));
  void *W0 = work[0], *W1 = work[1];
  const bool fz = zfuse_pad();
  const bool zsolo = fz && (X ? P2 : P1) == 1, g2solo = fz && (X ? P1 : P2) == 1;
  auto other = [&](void* b) { return b == W0 ? W1 : W0; };
  if (fz) {                     // r2c stores the kept columns straight into the z-chunk send blocks
    MFFT_TRY(stage("fwd_z", 0, [&] {
      RealArgs a;
      a.in = u; a.out = W1; a.n = (int)M2; a.prec = prec; a.in_stride = M2; a.out_stride = Nf; a.nrows = mp * np;
      a.scale = 1.0; a.valid = (int)Nf; a.zs = zsplit(mp * np, 0);
      return launch_r2c(a, stream);
    }));
  } else {
    MFFT_TRY(stage("fwd_z", 0, [&] { return r2c_rows(u, W0, mp * np, M2, M2, Nf, 1.0, (int)Nf); }));
    MFFT_TRY(stage("fwd_packz", 0, [&] { return pack_z(this, W0, W1, mp * np, Nf, zc, false); }));
  }
  void* cur = W1;
  if (!zsolo) {
    MFFT_TRY(stage("fwd_a2a_1", 0, [&] { return xchg(0, true, true, W1, W0); }));
    cur = W0;
  }
  if (X) {
    // cur = P2 blocks (mp, np, q): fft y gathering over M1 rows, truncate + fold on store, straight into
    // the P1 blocks (mp, N1_1, q) of the next exchange
    void* dst = other(cur);
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col_pad(cur, dst, M1, false, 2, true, mp, q, np * q, two_level(np, mp * np * q, q), N1_1 * q,
                     two_level(N1_1, mp * N1_1 * q, q), 1.0);
    }));
    cur = dst;
    if (!g2solo) {
      dst = other(cur);
      MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, true, cur, dst); }));
      cur = dst;
    }
    MFFT_TRY(stage("fwd_x", 0, [&] {
      return col_pad(cur, fu, M0, false, 2, true, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q), isc3);
    }));
  } else {
    // cur = (M0, np, q): fft x, truncate + fold to (N0, np, q)
    void* dst = other(cur);
    MFFT_TRY(stage("fwd_x", 0, [&] {
      return col_pad(cur, dst, M0, false, 2, true, 1, np * q, 0, plain(np * q), 0, plain(np * q), 1.0);
    }));
    cur = dst;
    if (!g2solo) {
      dst = other(cur);
      MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, true, cur, dst); }));
      cur = dst;
    }
    // cur = P2 blocks (N2_0, np, q): fft y gathering over M1 rows, truncate + fold into fu (N2_0, N1, q)
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col_pad(cur, fu, M1, false, 2, true, N2_0, q, np * q, two_level(np, N2_0 * np * q, q), N1 * q, plain(q), isc3);
    }));
  }
  return 0;
}

// ---- 3/2-rule, pencil (Alltoallw branches; padding of an axis happens right
// before the transform along it, when the axis is locally complete) -------------
int mfft_plan_s::pencil_backward_padded(const void* fu, void* u) {
  if (!r2c) return set_error(MFFT_ERR_UNSUPPORTED, "3/2-rule is implemented for R2C plans");
  if (d.drop_nyquist) return set_error(MFFT_ERR_UNSUPPORTED, "3/2-rule with communication='AlltoallN' is not implemented");
  if (can_fuse_pad()) return pencil_backward_padded_fused(fu, u);
  const double sc3 = padscale();
  const bool X = d.decomp == MFFT_PENCIL_X;
  const int64_t mp = M0 / P1, np = M1 / P2;          // padded local real rows in x, y
  const size_t wb = (size_t)std::max(std::max(M0 * N1_1 * q, mp * M1 * q), std::max(std::max(N2_0 * M1 * q, M0 * np * q), mp * np * Mf)) * es;
  for (int i = 0; i < 3; ++i) MFFT_TRY(ensure_work(i, wb));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  if (X) {
    // fu (N0, N1_1, q): pad x -> (M0, N1_1, q), ifft x
    MFFT_TRY(stage("pad_x", 0, [&] { return pad_axis(fu, W0, 1, N0, M0, N1_1 * q, sc3); }));
    MFFT_TRY(stage("bwd_x", 0, [&] { return col(W0, W0, M0, true, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q)); }));
    MFFT_TRY(stage("bwd_a2a_2", 0, [&] { return xchg(1, false, true, W0, W1); }));
    // W1 = P1 blocks (mp, N1_1, q) -> gather to (mp, N1, q)
    MFFT_TRY(stage("unpack", 0, [&] {
      for (int g = 0; g < P1; ++g)
        MFFT_TRY(box(static_cast<char*>(W1) + (size_t)g * (mp * N1_1 * q) * es, static_cast<char*>(W2) + (size_t)(g * N1_1 * q) * es,
                     mp, 1, N1_1 * q, N1_1 * q, 0, N1 * q, 0));
      return 0;
    }));
    MFFT_TRY(stage("pad_y", 0, [&] { return pad_axis(W2, W0, mp, N1, M1, q, 1.0); }));
    // ifft y on (mp, M1, q), writing P2 blocks (mp, np, q)
    MFFT_TRY(stage("bwd_y", 0, [&] {
      return col(W0, W1, M1, true, mp, q, M1 * q, plain(q), np * q, two_level(np, mp * np * q, q));
    }));
  } else {
    // fu (N2_0, N1, q): pad y -> (N2_0, M1, q), ifft y writing P2 blocks (N2_0, np, q)
    MFFT_TRY(stage("pad_y", 0, [&] { return pad_axis(fu, W0, N2_0, N1, M1, q, sc3); }));
    MFFT_TRY(stage("bwd_y", 0, [&] {
      return col(W0, W1, M1, true, N2_0, q, M1 * q, plain(q), np * q, two_level(np, N2_0 * np * q, q));
    }));
    MFFT_TRY(stage("bwd_a2a_2", 0, [&] { return xchg(1, false, true, W1, W0); }));
    // W0 = (N0, np, q): pad x -> (M0, np, q), ifft x
    MFFT_TRY(stage("pad_x", 0, [&] { return pad_axis(W0, W1, 1, N0, M0, np * q, 1.0); }));
    MFFT_TRY(stage("bwd_x", 0, [&] { return col(W1, W1, M0, true, 1, np * q, 0, plain(np * q), 0, plain(np * q)); }));
  }
  // W1 holds Pz blocks (mp, np, q) (X: y chunks; Y: contiguous x chunks)
  MFFT_TRY(stage("bwd_a2a_1", 0, [&] { return xchg(0, false, true, W1, W0); }));
  // unpack z into the zero-padded (mp*np, Mf) rows
  MFFT_TRY(stage("bwd_unpackz", 0, [&] {
    MFFT_TRY(zero(W2, (size_t)(mp * np * Mf) * es));
    return pack_z(this, W2, W0, mp * np, Mf, zc, true);
  }));
  MFFT_TRY(stage("bwd_z", 0, [&] { return c2r_rows(W2, u, mp * np, M2, Mf, M2, 1.0 / (double)M2); }));
  return 0;
}

int mfft_plan_s::pencil_forward_padded(const void* u, void* fu) {
  if (!r2c) return set_error(MFFT_ERR_UNSUPPORTED, "3/2-rule is implemented for R2C plans");
  if (d.drop_nyquist) return set_error(MFFT_ERR_UNSUPPORTED, "3/2-rule with communication='AlltoallN' is not implemented");
  if (can_fuse_pad()) return pencil_forward_padded_fused(u, fu);
  const double isc3 = 1.0 / padscale();
  const bool X = d.decomp == MFFT_PENCIL_X;
  const int64_t mp = M0 / P1, np = M1 / P2;
  const size_t wb = (size_t)std::max(std::max(M0 * N1_1 * q, mp * M1 * q), std::max(std::max(N2_0 * M1 * q, M0 * np * q), mp * np * Mf)) * es;
  for (int i = 0; i < 3; ++i) MFFT_TRY(ensure_work(i, wb));
  void *W0 = work[0], *W1 = work[1], *W2 = work[2];
  MFFT_TRY(stage("fwd_z", 0, [&] { return r2c_rows(u, W0, mp * np, M2, M2, Mf); }));
  if (d.line2d && P > 1)   // line.py:231 + swap_Nq: c0 <- Re c0 - Im cN, cN <- Re cN (cN = column Nf-1, not real here)
    MFFT_TRY(stage("fwd_nyq", 0, [&] { return launch_line_nyquist(W0, mp * np, Mf, Nf - 1, prec, stream); }));
  // only the first Nf modes travel (truncation in z); pack z chunks
  MFFT_TRY(stage("fwd_packz", 0, [&] { return pack_z(this, W0, W1, mp * np, Mf, zc, false); }));
  MFFT_TRY(stage("fwd_a2a_1", 0, [&] { return xchg(0, true, true, W1, W0); }));
  if (X) {
    // W0 = P2 blocks (mp, np, q): fft y over M1 = P2*np (gather), out (mp, M1, q)
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col(W0, W1, M1, false, mp, q, np * q, two_level(np, mp * np * q, q), M1 * q, plain(q));
    }));
    // (the 2-D class truncates without the Nyquist fold on one rank: line.py:185 `fu_padded[ks, :Nf]`)
    MFFT_TRY(stage("trunc_y", 0, [&] { return trunc_axis(W1, W0, mp, N1, M1, q, q, 1.0, !(d.line2d && P == 1)); }));
    // pack y chunks (mp, P1, N1_1, q) -> (P1, mp, N1_1, q)
    MFFT_TRY(stage("pack", 0, [&] {
      for (int g = 0; g < P1; ++g)
        MFFT_TRY(box(static_cast<char*>(W0) + (size_t)(g * N1_1 * q) * es, static_cast<char*>(W1) + (size_t)g * (mp * N1_1 * q) * es,
                     mp, 1, N1_1 * q, N1 * q, 0, N1_1 * q, 0));
      return 0;
    }));
    MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, true, W1, W2); }));
    MFFT_TRY(stage("fwd_x", 0, [&] { return col(W2, W2, M0, false, 1, N1_1 * q, 0, plain(N1_1 * q), 0, plain(N1_1 * q)); }));
    MFFT_TRY(stage("trunc_x", 0, [&] { return trunc_axis(W2, fu, 1, N0, M0, N1_1 * q, N1_1 * q, isc3); }));
  } else {
    // W0 = (M0, np, q): fft x, truncate to (N0, np, q)
    MFFT_TRY(stage("fwd_x", 0, [&] { return col(W0, W0, M0, false, 1, np * q, 0, plain(np * q), 0, plain(np * q)); }));
    MFFT_TRY(stage("trunc_x", 0, [&] { return trunc_axis(W0, W1, 1, N0, M0, np * q, np * q, 1.0); }));
    MFFT_TRY(stage("fwd_a2a_2", 0, [&] { return xchg(1, true, true, W1, W0); }));
    // W0 = P2 blocks (N2_0, np, q): fft y over M1 gathering, out (N2_0, M1, q)
    MFFT_TRY(stage("fwd_y", 0, [&] {
      return col(W0, W1, M1, false, N2_0, q, np * q, two_level(np, N2_0 * np * q, q), M1 * q, plain(q));
    }));
    MFFT_TRY(stage("trunc_y", 0, [&] { return trunc_axis(W1, fu, N2_0, N1, M1, q, q, isc3); }));
  }
  return 0;
}

namespace mfft {
hipStream_t plan_stream(mfft_plan_t plan) { return plan ? plan->stream : nullptr; }
}  // namespace mfft

// ===========================================================================
// C ABI
// ===========================================================================
extern "C" {

// host-only part of plan construction: decomposition bookkeeping (no HIP call)
static int decomp_init(mfft_plan_s* p, const mfft_plan_desc* desc, int nranks, int rank) {
  p->d = *desc;
  p->P = nranks;
  p->rank = rank;
  p->prec = desc->precision;
  p->r2c = desc->kind == MFFT_R2C;
  p->N0 = desc->n[0];
  p->N1 = desc->n[1];
  p->N2 = desc->n[2];
  if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(MFFT_ERR_INVALID, "bad rank %d of %d", rank, nranks);
  if (p->N0 < 1 || p->N1 < 1 || p->N2 < 1) return set_error(MFFT_ERR_INVALID, "bad mesh");
  if (desc->precision != MFFT_SINGLE && desc->precision != MFFT_DOUBLE) return set_error(MFFT_ERR_INVALID, "bad precision");
  p->Nf = p->r2c ? p->N2 / 2 + 1 : p->N2;
  p->es = elem_bytes(p->prec, true);
  p->rs = p->r2c ? elem_bytes(p->prec, false) : p->es;
  const double ps = desc->padsize > 0 ? desc->padsize : 1.5;
  p->d.padsize = ps;
  p->M0 = (int64_t)(ps * p->N0);
  p->M1 = (int64_t)(ps * p->N1);
  p->M2 = (int64_t)(ps * p->N2);
  p->Mf = p->r2c ? (int64_t)(ps * p->N2) / 2 + 1 : p->M2;
  p->world.resize(p->P);
  for (int i = 0; i < p->P; ++i) p->world[i] = i;
  p->xpad_on = !(getenv("MFFT_NO_XPAD") && atoi(getenv("MFFT_NO_XPAD")) != 0);
  p->xpass_inplace = getenv("MFFT_XPASS_INPLACE") && atoi(getenv("MFFT_XPASS_INPLACE")) != 0;
  if (getenv("MFFT_P1_XPAD")) p->p1_xpad_lines = atoi(getenv("MFFT_P1_XPAD"));
  p->zpitch_on = !(getenv("MFFT_NO_ZPITCH") && atoi(getenv("MFFT_NO_ZPITCH")) != 0);
  if (getenv("MFFT_PAD_ALIGN")) p->pad_align = atoi(getenv("MFFT_PAD_ALIGN")) != 0 ? 1 : 0;
  if (getenv("MFFT_PAD_ALIGN_INV")) p->pad_align_inv = atoi(getenv("MFFT_PAD_ALIGN_INV"));
  const int P = p->P;
  if (p->r2c && p->N2 % 2) return set_error(MFFT_ERR_UNSUPPORTED, "odd N[2]=%lld is not supported for R2C", (long long)p->N2);
  p->Zp = 0;                     // complex_pitch: resolved below, once the local z extent is known
  if (desc->decomp == MFFT_SLAB) {
    if (p->N0 % P || p->N1 % P) return set_error(MFFT_ERR_INVALID, "N[0]=%lld and N[1]=%lld must be divisible by the number of ranks %d", (long long)p->N0, (long long)p->N1, P);
    p->Np0 = p->N0 / P;
    p->Np1 = p->N1 / P;
    if (P > 1 && desc->pipeline < 0) {
      p->nbatch = (int)std::min<int64_t>(-(int64_t)desc->pipeline, p->Np0);      // batches of local x rows
    } else if (P > 1) {
      // kz slices for the exchange pipeline: boundaries on 16-column (tile) multiples
      const int want = desc->pipeline > 0 ? desc->pipeline : 4;
      const int64_t unit = 16;
      const int64_t per = (p->Nf / want) / unit * unit;
      if (want > 1 && per >= unit) {
        for (int s = 0; s < want; ++s) {
          const int64_t st = per * s;
          p->kslice.push_back(Chunk{s == want - 1 ? p->Nf - st : per, st});
        }
        p->nslice = want;
      }
    }
  } else if (desc->decomp == MFFT_PENCIL_X || desc->decomp == MFFT_PENCIL_Y) {
    int P1 = desc->p1, P2;
    if (P1 <= 0) compute_dims(P, &P1, &P2);
    else {
      if (P % P1) return set_error(MFFT_ERR_INVALID, "P1=%d does not divide %d ranks", P1, P);
      P2 = P / P1;
    }
    p->P1 = P1;
    p->P2 = P2;
    p->c0 = p->rank % P1;       // comm0 = consecutive ranks (pencil.py:192-195)
    p->c1 = p->rank / P1;
    // real (N0/P1, N1/P2, N2); X: complex (N0, N1/P1, N2/P2-chunk); Y: complex (N0/P2, N1, N2/P1-chunk)
    const bool alignX = desc->decomp == MFFT_PENCIL_X;
    if (p->N0 % P1 || p->N1 % P2 || (alignX ? (p->N1 % P1 || p->N2 % P2) : (p->N0 % P2 || p->N2 % P1)))
      return set_error(MFFT_ERR_INVALID, "mesh not divisible by the %dx%d process grid", P1, P2);
    if (desc->line2d && !(alignX && P1 == 1 && p->N0 == 1 && p->r2c))
      return set_error(MFFT_ERR_INVALID, "line2d is an x-aligned R2C pencil plan of a (1, Nx, Ny) mesh on a 1 x P grid");
    p->N1_0 = p->N0 / P1;
    p->N1_1 = p->N1 / P1;
    p->N2_0 = p->N0 / P2;
    p->N2_1 = p->N1 / P2;
    for (int i = 0; i < P1; ++i) p->group0.push_back(p->c1 * P1 + i);
    for (int i = 0; i < P2; ++i) p->group1.push_back(p->c0 + i * P1);
    const int Pz = desc->decomp == MFFT_PENCIL_X ? P2 : P1;
    const int cz = desc->decomp == MFFT_PENCIL_X ? p->c1 : p->c0;
    if (p->r2c && Pz > 1 && ((p->N2 / Pz) % 2)) return set_error(MFFT_ERR_UNSUPPORTED, "N[2]/%d must be even for the pencil z split", Pz);
    if (p->Nf % Pz > 1) return set_error(MFFT_ERR_UNSUPPORTED, "Nf=%lld cannot be split over %d ranks", (long long)p->Nf, Pz);
    p->zc = pencil_chunks(p->Nf, Pz);
    if (desc->drop_nyquist) {      // 'AlltoallN': equal chunks of the N2/2 non-Nyquist columns
      if (!p->r2c) return set_error(MFFT_ERR_INVALID, "drop_nyquist is an R2C mode");
      p->zc = pencil_chunks(p->N2 / 2, Pz);
    }
    p->q = p->zc[cz].len;
    p->zstart = p->zc[cz].start;
    // z-chunk pack / unpack fused into the z transform when a radix kernel with chunked stores / loads exists
    // (MFFT_NO_ZFUSE: the copy-based path, kept for A/B runs and for the lengths that go through chirp-z)
    p->zfuse = !desc->line2d && getenv("MFFT_NO_ZFUSE") == nullptr &&
               zsplit_supported(p->N2, p->prec, p->r2c) &&
               (!p->r2c || p->N2 % 2 == 0) && p->zc[0].len < 65536;
    // exchange pipeline of the x-aligned pencil: batches of local x rows (`pipeline`, default 4 like the slab's)
    const int want = desc->pipeline > 0 ? desc->pipeline : desc->pipeline < 0 ? -desc->pipeline : 4;
    if (desc->decomp == MFFT_PENCIL_X && want > 1 && P > 1 && !desc->drop_nyquist && !desc->line2d)
      p->nbatch = (int)std::min<int64_t>(want, p->N1_0);
    // ... and of the y-aligned one (pencil_forward_pipelined_y): two halves around the x transform
    if (desc->decomp == MFFT_PENCIL_Y && want > 1 && P1 > 1 && P2 > 1 && p->zfuse && !desc->drop_nyquist)
      p->nbatch = (int)std::min<int64_t>(want, std::min(p->N1_0, p->N2_0));
  } else {
    return set_error(MFFT_ERR_INVALID, "bad decomposition %d", desc->decomp);
  }
  // pitched spectrum (complex_pitch): -1 = whole cache lines, > 0 = that many elements (at least the local z extent)
  if (desc->complex_pitch != 0) {
    int64_t a_, b_, zloc = 0;
    p->cdims(&a_, &b_, &zloc);
    const int64_t line = (int64_t)(128 / p->es);
    const int64_t want = desc->complex_pitch < 0 ? (zloc + line - 1) / line * line : (int64_t)desc->complex_pitch;
    if (want < zloc) return set_error(MFFT_ERR_INVALID, "complex_pitch %lld is shorter than the local z extent %lld", (long long)want, (long long)zloc);
    p->Zp = want;
  }
  // every transform length must have a kernel
  auto need = [&](int64_t n, bool real) -> int {
    if (n == 1 && !real) return 0;
    if (!length_supported(n, real)) return set_error(MFFT_ERR_UNSUPPORTED, "transform length %lld%s is not supported (lengths run from 1 to 2^20: include/mpifft4py_amd.h mfft_length_route)", (long long)n, real ? " (real)" : "");
    return 0;
  };
  MFFT_TRY(need(p->N0, false));
  MFFT_TRY(need(p->N1, false));
  MFFT_TRY(need(p->N2, p->r2c));
  return 0;
}

int mfft_plan_create(mfft_comm_t comm, const mfft_plan_desc* desc, mfft_plan_t* out) {
  if (!comm || !desc || !out) return set_error(MFFT_ERR_INVALID, "null argument");
  std::unique_ptr<mfft_plan_s> p(new mfft_plan_s());
  p->comm = comm;
  MFFT_TRY(decomp_init(p.get(), desc, comm->size, comm->rank));
  comm->plan_refs++;                       // released by mfft_plan_destroy / the error path below
  struct Unref {
    mfft_comm_s* c; bool armed;
    ~Unref() { if (armed) c->plan_refs--; }
  } unref{comm, true};
  MFFT_HIP(hipGetDevice(&p->dev));
  const int nev = std::max(p->nslice > 1 ? p->nslice : 0, p->nbatch > 1 ? p->nbatch : 0);
  // Pipelined plans: the exchange runs on a stream of its own, and that stream gets CUs of its own.  The transform
  // kernels fill every CU (two 1024-thread workgroups with 80 KB of LDS each), so whatever the exchange launches -- the
  // IPC transport's pull kernel, RCCL's send / recv kernels -- would otherwise wait for transform workgroups to retire
  // and then take a whole CU away from them, workgroup by workgroup.  With comm_cus = K > 0 the communication stream
  // is confined to K CUs and the compute stream to the other ones (hipExtStreamCreateWithCUMask; KFD deals the bits of
  // a mask round-robin over the XCDs, so the last K = 8k bits are k CUs of every XCD); measured in
  // profiles/r03_cu_mask_probe.txt.  desc.comm_cus: 0 = $MFFT_COMM_CUS or the default, < 0 = no masks.
  int comm_cus = desc->comm_cus;
  if (comm_cus == 0) comm_cus = getenv("MFFT_COMM_CUS") ? atoi(getenv("MFFT_COMM_CUS")) : MFFT_DEFAULT_COMM_CUS;
  int ncu = 0;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, p->dev);
  auto masked = [&](int lo, int hi, hipStream_t* st) {
    std::vector<uint32_t> m((size_t)(ncu + 31) / 32, 0u);
    for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32);
    if (hipExtStreamCreateWithCUMask(st, (uint32_t)m.size(), m.data()) == hipSuccess) return true;
    (void)hipGetLastError();
    *st = nullptr;
    return false;
  };
  if (nev > 0 && comm_cus > 0 && ncu >= 16 && comm_cus <= ncu / 2) {
    if (masked(ncu - comm_cus, ncu, &p->cstream)) {
      if (masked(0, ncu - comm_cus, &p->stream)) {
        p->comm_cus = comm_cus;
      } else {
        (void)hipStreamDestroy(p->cstream);
        p->cstream = nullptr;
      }
    }
  }
  if (!p->stream) MFFT_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  {
    // Opt-in (MFFT_GRAPH=1), single rank only (no host-side rendezvous inside the sequence).  Measured:
    // a transform is only three kernels, and replaying a 3-node graph (~16 us) costs more than three
    // direct launches (64^3: 41.5k pairs/s direct vs 30k replayed); it pays when the caller's own
    // kernels ride on the same stream between transforms (32^3 Taylor-Green step 1.29 -> 1.02 ms).
    const char* e = getenv("MFFT_GRAPH");
    p->use_graphs = p->P == 1 && e && atoi(e) != 0;
  }
  if (nev > 0) {
    if (!p->cstream) {
      // no CU masks.  The communication stream is created at the LOWEST priority of the device's range (HIP: least = 1 =
      // low, 0 = normal, greatest = -1 = high; the compute stream is a plain non-blocking stream = normal) -- this is
      // what every round-3 profile was taken with (its notes call it "normal": ADVICE r03), and it is kept because it is
      // the measured configuration.  MFFT_COMM_PRIORITY=0 asks for normal, 1 for the highest priority: measured
      // (profiles/r03_cu_mask_probe.txt) the highest changes nothing for the pull kernel -- workgroups of a second queue
      // are admitted as transform workgroups retire, with or without it -- and with several ranks on ONE device it
      // inverts priorities (a high-priority queue polling for a flag that a lower-priority queue of another process has
      // yet to write: 0.8 -> 45 ms per pair at two processes, profiles/r03_shared_gpu_pipeline_latency.txt).
      int prio_least = 0, prio_greatest = 0;
      MFFT_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
      int prio = prio_least;
      if (const char* e = getenv("MFFT_COMM_PRIORITY")) prio = atoi(e) > 0 ? prio_greatest : atoi(e) == 0 ? 0 : prio_least;
      prio = std::max(prio_greatest, std::min(prio_least, prio));
      MFFT_HIP(hipStreamCreateWithPriority(&p->cstream, hipStreamNonBlocking, prio));
    }
    for (std::vector<hipEvent_t>* v : {&p->ev_compute, &p->ev_comm, &p->ev2_compute, &p->ev2_comm}) {
      if ((v == &p->ev2_compute || v == &p->ev2_comm) && p->nbatch <= 1) continue;
      v->resize(nev);
      for (int s = 0; s < nev; ++s) MFFT_HIP(hipEventCreateWithFlags(&(*v)[s], hipEventDisableTiming));
    }
  }
  unref.armed = false;
  *out = p.release();
  return 0;
}

// Exchange schedule of a transform, computed WITHOUT a device: which peers a rank
// exchanges with and the byte counts / displacements of every chunk.  This is the
// same code the executor uses (mfft_plan_s::sched); tests drive it from CPU-only
// multi-process runs (gloo) to validate the distributed bookkeeping.
int mfft_plan_exchange_schedule(const mfft_plan_desc* desc, int nranks, int rank, int which, int forward, int padded,
                                int max_peers, int* npeers, int* peers, size_t* scount, size_t* sdisp, size_t* rcount,
                                size_t* rdisp) {
  if (!desc || !npeers) return set_error(MFFT_ERR_INVALID, "null argument");
  mfft_plan_s p;
  MFFT_TRY(decomp_init(&p, desc, nranks, rank));
  Sched sc;
  MFFT_TRY(p.sched(which, forward != 0, padded != 0, &sc));
  *npeers = (int)sc.peers.size();
  if (*npeers > max_peers) return set_error(MFFT_ERR_INVALID, "schedule has %d peers, room for %d", *npeers, max_peers);
  for (int i = 0; i < *npeers; ++i) {
    if (peers) peers[i] = sc.peers[i];
    if (scount) scount[i] = sc.sc[i];
    if (sdisp) sdisp[i] = sc.sd[i];
    if (rcount) rcount[i] = sc.rc[i];
    if (rdisp) rdisp[i] = sc.rd[i];
  }
  return 0;
}

int mfft_plan_exchange_pieces(const mfft_plan_desc* desc, int nranks, int rank, int which, int forward, int piece,
                              int max_peers, int* npieces, int* npeers, int* peers, size_t* scount, size_t* sdisp,
                              size_t* rcount, size_t* rdisp) {
  if (!desc || !npeers || !npieces) return set_error(MFFT_ERR_INVALID, "null argument");
  mfft_plan_s p;
  MFFT_TRY(decomp_init(&p, desc, nranks, rank));
  *npieces = p.npieces();
  Sched sc;
  MFFT_TRY(p.piece_sched(which, forward != 0, piece, &sc));
  *npeers = (int)sc.peers.size();
  if (*npeers > max_peers) return set_error(MFFT_ERR_INVALID, "schedule has %d peers, room for %d", *npeers, max_peers);
  for (int i = 0; i < *npeers; ++i) {
    if (peers) peers[i] = sc.peers[i];
    if (scount) scount[i] = sc.sc[i];
    if (sdisp) sdisp[i] = sc.sd[i];
    if (rcount) rcount[i] = sc.rc[i];
    if (rdisp) rdisp[i] = sc.rd[i];
  }
  return 0;
}

// What `rank` pulls, phase by phase, when exchange `which` of a pencil plan runs over the IPC transport with relay
// striping (relay_plan.h) -- device-free, the same enumeration the transport executes (relay_moves): tests replay it on
// host buffers and count the bytes per link.
int mfft_plan_relay_schedule(const mfft_plan_desc* desc, int nranks, int rank, int which, int forward, int max_moves,
                             int* nmoves, int* phase, int* kind, int* from, int* msg_src, int* msg_dst, size_t* msg_off,
                             size_t* bytes) {
  if (!desc || !nmoves) return set_error(MFFT_ERR_INVALID, "null argument");
  if (desc->decomp == MFFT_SLAB) return set_error(MFFT_ERR_INVALID, "slab plans exchange over all ranks: nothing to relay");
  std::vector<size_t> B((size_t)nranks * nranks, 0);
  std::vector<int> part;
  for (int s = 0; s < nranks; ++s) {
    mfft_plan_s p;
    MFFT_TRY(decomp_init(&p, desc, nranks, s));
    Sched sc;
    MFFT_TRY(p.sched(which, forward != 0, false, &sc));
    for (size_t i = 0; i < sc.peers.size(); ++i) B[(size_t)s * nranks + sc.peers[i]] = sc.sc[i];
    if (s == rank) part = sc.part;
  }
  if ((int)part.size() != nranks) return set_error(MFFT_ERR_INTERNAL, "no partition for this exchange");
  std::vector<RelayMove> mv;
  relay_moves(nranks, rank, part.data(), [&](int s, int d) { return B[(size_t)s * nranks + d]; }, &mv);
  *nmoves = (int)mv.size();
  if (*nmoves > max_moves) return set_error(MFFT_ERR_INVALID, "relay schedule has %d moves, room for %d", *nmoves, max_moves);
  for (int i = 0; i < *nmoves; ++i) {
    if (phase) phase[i] = mv[i].phase;
    if (kind) kind[i] = mv[i].kind;
    if (from) from[i] = mv[i].from;
    if (msg_src) msg_src[i] = mv[i].msg_src;
    if (msg_dst) msg_dst[i] = mv[i].msg_dst;
    if (msg_off) msg_off[i] = mv[i].msg_off;
    if (bytes) bytes[i] = mv[i].bytes;
  }
  return 0;
}

int mfft_plan_destroy(mfft_plan_t plan) {
  if (!plan) return 0;
  (void)hipStreamSynchronize(plan->stream);
  mfft_comm_s* c = plan->comm;
  delete plan;                             // returns its work buffers to the communicator
  if (c && --c->plan_refs == 0 && c->destroy_requested) delete c;
  return 0;
}

static int layout_of(const mfft_plan_s* p, int64_t rshape[3], int64_t cshape[3], int64_t rstart[3], int64_t cstart[3],
                     int64_t rshape_pad[3], int64_t grid[2], int64_t sub[2]) {
  int64_t rs_[3], cs_[3], r0[3], c0_[3], rp[3];
  if (p->d.decomp == MFFT_SLAB) {
    rs_[0] = p->Np0; rs_[1] = p->N1; rs_[2] = p->N2;
    cs_[0] = p->N0; cs_[1] = p->Np1; cs_[2] = p->Nf;
    r0[0] = p->rank * p->Np0; r0[1] = 0; r0[2] = 0;
    c0_[0] = 0; c0_[1] = p->rank * p->Np1; c0_[2] = 0;
    rp[0] = p->M0 / p->P; rp[1] = p->M1; rp[2] = p->M2;
    if (grid) { grid[0] = p->P; grid[1] = 1; }
    if (sub) { sub[0] = p->rank; sub[1] = 0; }
  } else {
    rs_[0] = p->N1_0; rs_[1] = p->N2_1; rs_[2] = p->N2;
    r0[0] = p->c0 * p->N1_0; r0[1] = p->c1 * p->N2_1; r0[2] = 0;
    rp[0] = p->M0 / p->P1; rp[1] = p->M1 / p->P2; rp[2] = p->M2;
    if (p->d.decomp == MFFT_PENCIL_X) {
      cs_[0] = p->N0; cs_[1] = p->N1_1; cs_[2] = p->q;
      c0_[0] = 0; c0_[1] = p->c0 * p->N1_1; c0_[2] = p->zstart;
    } else {
      cs_[0] = p->N2_0; cs_[1] = p->N1; cs_[2] = p->q;
      c0_[0] = p->c1 * p->N2_0; c0_[1] = 0; c0_[2] = p->zstart;
    }
    if (grid) { grid[0] = p->P1; grid[1] = p->P2; }
    if (sub) { sub[0] = p->c0; sub[1] = p->c1; }
  }
  for (int i = 0; i < 3; ++i) {
    if (rshape) rshape[i] = rs_[i];
    if (cshape) cshape[i] = cs_[i];
    if (rstart) rstart[i] = r0[i];
    if (cstart) cstart[i] = c0_[i];
    if (rshape_pad) rshape_pad[i] = rp[i];
  }
  return 0;
}

int mfft_plan_layout(mfft_plan_t p, int64_t rshape[3], int64_t cshape[3], int64_t rstart[3], int64_t cstart[3],
                     int64_t rshape_pad[3], int64_t grid[2], int64_t sub[2]) {
  if (!p) return set_error(MFFT_ERR_INVALID, "null plan");
  return layout_of(p, rshape, cshape, rstart, cstart, rshape_pad, grid, sub);
}

// The same answers for rank `rank` of `nranks` WITHOUT a plan, a communicator or a device: the decomposition
// bookkeeping of the constructors (slab.py:82-96, pencil.py:187-216, 903-913) is host arithmetic.
int mfft_layout_query(const mfft_plan_desc* desc, int nranks, int rank, int64_t rshape[3], int64_t cshape[3],
                      int64_t rstart[3], int64_t cstart[3], int64_t rshape_pad[3], int64_t grid[2], int64_t sub[2]) {
  if (!desc) return set_error(MFFT_ERR_INVALID, "null argument");
  mfft_plan_s p;
  MFFT_TRY(decomp_init(&p, desc, nranks, rank));
  return layout_of(&p, rshape, cshape, rstart, cstart, rshape_pad, grid, sub);
}

// Row pitch (complex elements) and allocation size (elements) of this rank's spectrum under desc->complex_pitch: what a
// caller must allocate -- shape (d0, d1, pitch) physically, (d0, d1, d2) logically.  Device-free.
int mfft_layout_complex_pitch(const mfft_plan_desc* desc, int nranks, int rank, int64_t* pitch, int64_t* alloc_elems) {
  if (!desc || !pitch || !alloc_elems) return set_error(MFFT_ERR_INVALID, "null argument");
  mfft_plan_s p;
  MFFT_TRY(decomp_init(&p, desc, nranks, rank));
  int64_t a, b, c;
  p.cdims(&a, &b, &c);
  *pitch = p.pitched() ? p.Zp : c;
  *alloc_elems = p.local_complex_alloc();
  return 0;
}

int mfft_plan_workspace_bytes(mfft_plan_t p, size_t* bytes) {
  if (!p || !bytes) return set_error(MFFT_ERR_INVALID, "null argument");
  *bytes = p->work_bytes[0] + p->work_bytes[1] + p->work_bytes[2] + p->work3_bytes + p->pcomp_bytes;
  return 0;
}

static int check_ready(mfft_plan_t p, const void* a, const void* b) {
  if (!p || !a || !b) return set_error(MFFT_ERR_INVALID, "null argument");
  int dev = -1;
  MFFT_HIP(hipGetDevice(&dev));
  if (dev != p->dev) MFFT_HIP(hipSetDevice(p->dev));
  return 0;
}

static int run_route(mfft_plan_t p, bool forward, const void* in, void* out, int dealias) {
  const bool pad = dealias == MFFT_DEALIAS_3_2, masked = dealias == MFFT_DEALIAS_2_3;
  if (forward) {
    if (p->d.decomp == MFFT_SLAB) return pad ? p->slab_forward_padded(in, out) : p->slab_forward(in, out);
    return pad ? p->pencil_forward_padded(in, out) : p->pencil_forward(in, out);
  }
  if (p->d.decomp == MFFT_SLAB) return pad ? p->slab_backward_padded(in, out) : p->slab_backward(in, out, masked);
  return pad ? p->pencil_backward_padded(in, out) : p->pencil_backward(in, out, masked);
}

// A pitched spectrum on a plan whose routes want compact rows (several ranks, pencils, complex data): converted at the
// boundary through the plan's compact copy -- one more pass over the spectrum, every decomposition served.
int mfft_plan_s::exec(bool forward, const void* in, void* out, int dealias) {
  // (a one-rank plan whose 3/2-rule route is the copy-based one -- lengths without pad / truncate kernels -- converts too)
  const bool native = nat_pitch() && (dealias != MFFT_DEALIAS_3_2 || can_fuse_pad());
  if (!pitched() || native) return run_route(this, forward, in, out, dealias);
  MFFT_TRY(ensure_buf(&pcomp, &pcomp_bytes, (size_t)local_complex_count() * es));
  struct Scope {
    mfft_plan_s* p;
    ~Scope() { p->conv_now = false; }
  } scope{this};
  if (forward) {
    conv_now = true;
    MFFT_TRY(run_route(this, true, in, pcomp, dealias));
    conv_now = false;
    return stage("fwd_pitch", 0, [&] { return repitch(pcomp, out, true); });
  }
  MFFT_TRY(stage("bwd_pitch", 0, [&] { return repitch(in, pcomp, false); }));
  conv_now = true;
  return run_route(this, false, pcomp, out, dealias);
}

static int run_direct(mfft_plan_t p, bool forward, const void* in, void* out, int dealias) { return p->exec(forward, in, out, dealias); }

// Small single-rank transforms are launch-bound (six ~10 us kernels per pair): the kernel sequence of
// a (direction, in, out, dealias) combination is captured into a hipGraph the second time it is seen
// and replayed afterwards.  The first execution runs directly so that work buffers, twiddle tables
// and function attributes exist before anything is captured.
static int run_graphed(mfft_plan_t p, bool forward, const void* in, void* out, int dealias) {
  for (auto& g : p->graphs)
    if (g.forward == forward && g.in == in && g.out == out && g.dealias == dealias) {
      if (g.exec) {
        MFFT_HIP(hipGraphLaunch(g.exec, p->stream));
        return 0;
      }
      // second sighting: capture
      hipGraph_t graph = nullptr;
      MFFT_HIP(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
      const int rc = run_direct(p, forward, in, out, dealias);
      const hipError_t e = hipStreamEndCapture(p->stream, &graph);
      if (rc != 0) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
      }
      if (e != hipSuccess || !graph) {      // capture not possible: fall back to direct launches for good
        (void)hipGetLastError();
        p->use_graphs = false;
        return run_direct(p, forward, in, out, dealias);
      }
      hipGraphExec_t exec = nullptr;
      const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) {
        (void)hipGetLastError();
        p->use_graphs = false;
        return run_direct(p, forward, in, out, dealias);
      }
      g.exec = exec;
      MFFT_HIP(hipGraphLaunch(g.exec, p->stream));
      return 0;
    }
  if (p->graphs.size() >= 64) {              // bounded cache (callers that cycle through many buffers)
    for (auto& g : p->graphs)
      if (g.exec) (void)hipGraphExecDestroy(g.exec);
    p->graphs.clear();
  }
  p->graphs.push_back(GraphEntry{forward, in, out, dealias, nullptr});
  return run_direct(p, forward, in, out, dealias);
}

int mfft_forward(mfft_plan_t p, const void* u, void* fu, int dealias) {
  MFFT_TRY(check_ready(p, u, fu));
  if (p->use_graphs && !p->timing) return run_graphed(p, true, u, fu, dealias);
  return run_direct(p, true, u, fu, dealias);
}

int mfft_backward(mfft_plan_t p, const void* fu, void* u, int dealias) {
  MFFT_TRY(check_ready(p, fu, u));
  if (p->use_graphs && !p->timing) return run_graphed(p, false, fu, u, dealias);
  return run_direct(p, false, fu, u, dealias);
}

// out = fftn(ifftn(a) x ifftn(b)) with the plan's transforms under the given dealias mode: a, b, out are vector fields of
// shape (3,) + the local complex shape, component-major; out may be a or b.  See include/mpifft4py_amd.h.
int mfft_nonlinear_cross(mfft_plan_t p, const void* a_hat, const void* b_hat, void* out_hat, int dealias) {
  MFFT_TRY(check_ready(p, a_hat, b_hat));
  if (!out_hat) return set_error(MFFT_ERR_INVALID, "null argument");
  if (dealias != MFFT_DEALIAS_NONE && dealias != MFFT_DEALIAS_2_3 && dealias != MFFT_DEALIAS_3_2)
    return set_error(MFFT_ERR_INVALID, "unknown dealias mode %d", dealias);
  return p->nonlinear_cross(a_hat, b_hat, out_hat, dealias);
}

int mfft_plan_sync(mfft_plan_t p) {
  if (!p) return set_error(MFFT_ERR_INVALID, "null plan");
  if (p->cstream) MFFT_HIP(hipStreamSynchronize(p->cstream));
  MFFT_HIP(hipStreamSynchronize(p->stream));
  return 0;
}

int mfft_plan_set_dealias_mask(mfft_plan_t p, const uint8_t* mask_host, size_t count) {
  if (!p || !mask_host) return set_error(MFFT_ERR_INVALID, "null argument");
  if ((int64_t)count != p->local_complex_count()) return set_error(MFFT_ERR_INVALID, "mask has %zu entries, local spectrum has %lld", count, (long long)p->local_complex_count());
  p->drop_graphs();              // captured sequences hold the old mask pointer
  if (p->mask) MFFT_HIP(hipFree(p->mask));
  p->mask = nullptr;
  if (p->nat_pitch()) {          // the masked-load kernels index the mask like the spectrum: same row pitch, zeros between
    const size_t rows = (size_t)(p->N0 * p->Np1);
    MFFT_HIP(hipMalloc(reinterpret_cast<void**>(&p->mask), rows * (size_t)p->Zp));
    MFFT_HIP(hipMemset(p->mask, 0, rows * (size_t)p->Zp));
    MFFT_HIP(hipMemcpy2D(p->mask, (size_t)p->Zp, mask_host, (size_t)p->Nf, (size_t)p->Nf, rows, hipMemcpyHostToDevice));
  } else {
    MFFT_HIP(hipMalloc(reinterpret_cast<void**>(&p->mask), count));
    MFFT_HIP(hipMemcpy(p->mask, mask_host, count, hipMemcpyHostToDevice));
  }
  p->mask_count = count;
  p->detect_band(mask_host);
  p->detect_band_local(mask_host);
  return 0;
}

int mfft_plan_get_info(mfft_plan_t p, const char* key, int64_t* value) {
  if (!p || !key || !value) return set_error(MFFT_ERR_INVALID, "null argument");
  const std::string k(key);
  if (k == "pruned_route") *value = p->band_ok ? (p->band_allzero ? 2 : 1) : 0;
  else if (k == "local_band") *value = p->lband_ok ? 1 : 0;
  else if (k == "comm_cus") *value = p->comm_cus;
  else if (k == "kz_slices") *value = p->nslice;
  else if (k == "row_batches") *value = p->nbatch;
  else if (k == "zfuse") *value = p->zfuse ? 1 : 0;
  else if (k == "ranks") *value = p->P;
  else if (k == "complex_pitch") *value = p->pitched() ? p->Zp : 0;        // row pitch of the caller's spectrum (elements), 0 = compact
  else if (k == "complex_pitch_native") *value = p->nat_pitch() ? 1 : 0;  // 1: the routes run on the pitched rows themselves
  else if (k == "nonlinear_fused_none") *value = p->nonlinear_fusable(MFFT_DEALIAS_NONE) ? 1 : 0;
  else if (k == "nonlinear_fused_2_3") *value = p->nonlinear_fusable(MFFT_DEALIAS_2_3) ? 1 : 0;
  else if (k == "nonlinear_fused_3_2") *value = p->nonlinear_fusable(MFFT_DEALIAS_3_2) ? 1 : 0;
  else if (k == "nonlinear_bytes") *value = (int64_t)(p->nlx_bytes + p->nly_bytes + p->nlr_bytes + p->nlw_bytes[0] + p->nlw_bytes[1]);
  else if (k == "plane_pad") *value = (p->P == 1 && p->d.decomp == MFFT_SLAB) ? p->p1_plane_pad() : 0;   // elements added to the intermediate's plane pitch
  else return set_error(MFFT_ERR_INVALID, "mfft_plan_get_info: unknown key '%s'", key);
  return 0;
}

int mfft_plan_timing(mfft_plan_t p, int enable) {
  if (!p) return set_error(MFFT_ERR_INVALID, "null plan");
  p->timing = enable != 0;
  return 0;
}

int mfft_plan_timing_reset(mfft_plan_t p) {
  if (!p) return set_error(MFFT_ERR_INVALID, "null plan");
  MFFT_TRY(p->collect_timing());
  for (auto& t : p->timers) {
    t.total_ms = 0;
    t.calls = 0;
  }
  return 0;
}

int mfft_plan_timing_get(mfft_plan_t p, int max_stages, char names[][32], double* total_ms, int64_t* calls, double* alg_bytes) {
  if (!p) return set_error(MFFT_ERR_INVALID, "null plan");
  MFFT_TRY(p->collect_timing());
  const int n = (int)p->timers.size();
  for (int i = 0; i < n && i < max_stages; ++i) {
    if (names) {
      strncpy(names[i], p->timers[i].name.c_str(), 31);
      names[i][31] = 0;
    }
    if (total_ms) total_ms[i] = p->timers[i].total_ms;
    if (calls) calls[i] = p->timers[i].calls;
    if (alg_bytes) alg_bytes[i] = p->timers[i].alg_bytes;
  }
  return n;
}

}  // extern "C"
