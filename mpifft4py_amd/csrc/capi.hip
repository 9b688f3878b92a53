// capi.hip -- the extern "C" surface of libmpifft4py_amd.so that is not the plan
// executor (plan.hip): device/memory helpers, communicators, the stage-level
// "serialFFT seam" entry points and HIP-event timers.
#include <cstring>
#include "comm.h"
#include "mfft_internal.h"

using namespace mfft;

extern "C" {

int mfft_version(void) { return 100; }   // 0.1.0
const char* mfft_last_error(void) { return last_error(); }

int mfft_device_count(int* count) {
  if (!count) return set_error(MFFT_ERR_INVALID, "null argument");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return set_error(MFFT_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  return 0;
}
int mfft_set_device(int device) {
  MFFT_HIP(hipSetDevice(device));
  return 0;
}
int mfft_get_device(int* device) {
  MFFT_HIP(hipGetDevice(device));
  return 0;
}
int mfft_device_name(char* buf, size_t buflen) {
  int dev = 0;
  MFFT_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MFFT_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}
int mfft_device_sync(void) {
  MFFT_HIP(hipDeviceSynchronize());
  return 0;
}

int mfft_malloc(void** dptr, size_t bytes) {
  if (!dptr) return set_error(MFFT_ERR_INVALID, "null argument");
  *dptr = nullptr;
  return dev_alloc(dptr, bytes);      // registered: the buffer may be the source or target of an exchange
}
int mfft_free(void* dptr) { return dev_free(dptr); }
// The transforms run asynchronously on their plan's own non-blocking stream, which the null stream does not order
// against.  These helpers therefore wait for ALL work of the device (every plan's streams) before they touch memory,
// and return when the copy is complete: a copy issued after mfft_forward sees its result, a transform issued after
// a copy sees the copied data.
int mfft_memset(void* dptr, int value, size_t bytes) {
  MFFT_HIP(hipDeviceSynchronize());
  MFFT_HIP(hipMemset(dptr, value, bytes));
  MFFT_HIP(hipDeviceSynchronize());
  return 0;
}
int mfft_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  MFFT_HIP(hipDeviceSynchronize());
  MFFT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int mfft_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  MFFT_HIP(hipDeviceSynchronize());
  MFFT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
// rows of `width_bytes` bytes, `rows` of them: device rows dev_pitch_bytes apart <-> packed host rows (pitched spectra)
int mfft_memcpy_rows_h2d(void* dst, size_t dev_pitch_bytes, const void* src_host, size_t width_bytes, size_t rows) {
  if (width_bytes > dev_pitch_bytes) return set_error(MFFT_ERR_INVALID, "row wider than its pitch");
  MFFT_HIP(hipDeviceSynchronize());
  MFFT_HIP(hipMemcpy2D(dst, dev_pitch_bytes, src_host, width_bytes, width_bytes, rows, hipMemcpyHostToDevice));
  return 0;
}
int mfft_memcpy_rows_d2h(void* dst_host, const void* src, size_t dev_pitch_bytes, size_t width_bytes, size_t rows) {
  if (width_bytes > dev_pitch_bytes) return set_error(MFFT_ERR_INVALID, "row wider than its pitch");
  MFFT_HIP(hipDeviceSynchronize());
  MFFT_HIP(hipMemcpy2D(dst_host, width_bytes, src, dev_pitch_bytes, width_bytes, rows, hipMemcpyDeviceToHost));
  return 0;
}
int mfft_memcpy_d2d(void* dst, const void* src, size_t bytes) {
  MFFT_HIP(hipDeviceSynchronize());
  MFFT_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
  MFFT_HIP(hipDeviceSynchronize());
  return 0;
}
int mfft_fill_uniform(void* dptr, size_t count, int precision, uint64_t seed) {
  MFFT_TRY(launch_fill_uniform(dptr, count, precision, seed, nullptr));
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

// ---- communicators -------------------------------------------------------------
int mfft_comm_create_self(mfft_comm_t* comm) {
  if (!comm) return set_error(MFFT_ERR_INVALID, "null argument");
  return comm_create_self(comm);
}
int mfft_get_unique_id(void* id128) { return comm_get_unique_id(id128); }
int mfft_comm_create_rccl(int nranks, int rank, const void* id128, mfft_comm_t* comm) {
  if (!comm || !id128) return set_error(MFFT_ERR_INVALID, "null argument");
  return comm_create_rccl(nranks, rank, id128, comm);
}
int mfft_comm_create_local(int nranks, const int* devices, mfft_comm_t* comms_out) {
  if (!comms_out) return set_error(MFFT_ERR_INVALID, "null argument");
  return comm_create_local(nranks, devices, comms_out);
}
int mfft_comm_size(mfft_comm_t c, int* size) {
  if (!c || !size) return set_error(MFFT_ERR_INVALID, "null argument");
  *size = c->size;
  return 0;
}
int mfft_comm_rank(mfft_comm_t c, int* rank) {
  if (!c || !rank) return set_error(MFFT_ERR_INVALID, "null argument");
  *rank = c->rank;
  return 0;
}
int mfft_comm_barrier(mfft_comm_t c) { return c ? c->barrier() : set_error(MFFT_ERR_INVALID, "null comm"); }
int mfft_comm_bcast_host(mfft_comm_t c, void* buf, size_t bytes, int root) {
  return c ? c->bcast_host(buf, bytes, root) : set_error(MFFT_ERR_INVALID, "null comm");
}
int mfft_comm_allreduce_sum_host(mfft_comm_t c, double* v, int n) {
  return c ? c->allreduce_host(v, n, 0) : set_error(MFFT_ERR_INVALID, "null comm");
}
int mfft_comm_allreduce_max_host(mfft_comm_t c, double* v, int n) {
  return c ? c->allreduce_host(v, n, 1) : set_error(MFFT_ERR_INVALID, "null comm");
}
int mfft_comm_selftest(mfft_comm_t c, size_t bytes_per_peer, int timeout_ms) {
  if (!c || timeout_ms <= 0) return set_error(MFFT_ERR_INVALID, "bad argument");
  return c->selftest(bytes_per_peer ? bytes_per_peer : 4096, timeout_ms);
}
int mfft_comm_set_option(mfft_comm_t c, const char* key, int64_t value) {
  if (!c || !key) return set_error(MFFT_ERR_INVALID, "null argument");
  return c->set_option(key, (long long)value);
}
// PCI bus id of a device ("0000:05:00.0"): names the physical GPU behind a rank in a bench line
int mfft_device_pci_bus_id(int device, char* buf, size_t buflen) {
  if (!buf || buflen < 16) return set_error(MFFT_ERR_INVALID, "buffer of at least 16 bytes needed");
  MFFT_HIP(hipDeviceGetPCIBusId(buf, (int)buflen, device));
  return 0;
}

int mfft_comm_get_option(mfft_comm_t c, const char* key, int64_t* value) {
  if (!c || !key || !value) return set_error(MFFT_ERR_INVALID, "null argument");
  *value = (int64_t)c->get_option(key);
  return 0;
}
int mfft_comm_abort(mfft_comm_t c) {
  if (c) c->abort();
  return 0;
}
int mfft_comm_destroy(mfft_comm_t c) {
  if (!c) return 0;
  if (c->plan_refs > 0) {                  // plans still use it (their work buffers live in it): the last one frees it
    c->destroy_requested = true;
    return 0;
  }
  delete c;
  return 0;
}

// ---- stage level ------------------------------------------------------------------
int mfft_length_supported(int64_t n, int real_transform) { return length_supported(n, real_transform != 0) ? 1 : 0; }
int mfft_length_route(int64_t n, int real_transform) { return length_route(n, real_transform != 0); }
int mfft_length_route_precision(int64_t n, int real_transform, int precision) {
  if (precision != MFFT_DOUBLE && precision != MFFT_SINGLE) return 0;
  return length_route(n, real_transform != 0, precision);
}

int mfft_kernel_name(int family, int64_t n, int precision, int inverse, int nt, char* buf, size_t buflen) {
  if (!buf || buflen == 0 || family < FAM_COL || family > FAM_C2R) return set_error(MFFT_ERR_INVALID, "bad argument");
  const KernelEntry* e = find_kernel(family, (int)n, precision, family == FAM_C2R ? 1 : (inverse ? 1 : 0), nt ? 1 : 0);
  if (!e) return set_error(MFFT_ERR_UNSUPPORTED, "no radix kernel of family %d for length %lld", family, (long long)n);
  snprintf(buf, buflen, "%s tile=%d threads=%d lds=%d%s", e->name, e->tile, e->threads, e->lds_bytes, e->nt ? " nt" : "");
  return 0;
}

int mfft_c2c_axis(const void* in, void* out, const int64_t shape[3], int axis, int inverse, int precision) {
  if (!in || !out || !shape) return set_error(MFFT_ERR_INVALID, "null argument");
  const int64_t s0 = shape[0], s1 = shape[1], s2 = shape[2];
  if (s0 < 1 || s1 < 1 || s2 < 1) return set_error(MFFT_ERR_INVALID, "bad shape");
  const int64_t n = shape[axis];
  const size_t es = elem_bytes(precision, true);
  if (n == 1) {
    if (in != out) MFFT_HIP(hipMemcpy(out, in, (size_t)(s0 * s1 * s2) * es, hipMemcpyDeviceToDevice));
    return 0;
  }
  if (axis == 2) {
    RowArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = precision; a.inverse = inverse != 0;
    a.in_stride = a.out_stride = s2; a.nrows = s0 * s1; a.scale = inverse ? 1.0 / (double)n : 1.0;
    MFFT_TRY(launch_row(a, nullptr));
  } else if (axis == 0 || axis == 1) {
    ColArgs a;
    a.in = in; a.out = out; a.n = (int)n; a.prec = precision; a.inverse = inverse != 0;
    a.scale = inverse ? 1.0 / (double)n : 1.0;
    if (axis == 1) {
      a.nouter = s0; a.ncols = s2; a.in_outer = a.out_outer = s1 * s2;
      a.in_rows.lo = a.out_rows.lo = s2;
    } else {
      a.nouter = 1; a.ncols = s1 * s2; a.in_outer = a.out_outer = 0;
      a.in_rows.lo = a.out_rows.lo = s1 * s2;
    }
    MFFT_TRY(launch_col(a, nullptr));
  } else {
    return set_error(MFFT_ERR_INVALID, "bad axis %d", axis);
  }
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

// The strided transform with every stride spelled out (the "advanced" layout of a stage-level call: what numpy does for a
// non-contiguous view, numpy_fft.py:25-37): nouter batches in_outer / out_outer elements apart, each ncols contiguous
// columns wide, rows in_pitch / out_pitch elements apart.
int mfft_c2c_strided(const void* in, void* out, int64_t n, int64_t nouter, int64_t ncols, int64_t in_outer, int64_t in_pitch,
                     int64_t out_outer, int64_t out_pitch, int inverse, int precision) {
  if (!in || !out) return set_error(MFFT_ERR_INVALID, "null argument");
  if (n < 1 || nouter < 1 || ncols < 1 || in_pitch < ncols || out_pitch < ncols)
    return set_error(MFFT_ERR_INVALID, "bad extents: n %lld, %lld batches of %lld columns, pitches %lld / %lld", (long long)n,
                     (long long)nouter, (long long)ncols, (long long)in_pitch, (long long)out_pitch);
  ColArgs a;
  a.in = in; a.out = out; a.n = (int)n; a.prec = precision; a.inverse = inverse != 0;
  a.scale = inverse ? 1.0 / (double)n : 1.0;
  a.nouter = nouter; a.ncols = ncols; a.in_outer = in_outer; a.out_outer = out_outer;
  a.in_rows.lo = in_pitch; a.out_rows.lo = out_pitch;
  MFFT_TRY(launch_col(a, nullptr));
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

int mfft_r2c_last(const void* in, void* out, const int64_t rshape[3], int precision) {
  if (!in || !out || !rshape) return set_error(MFFT_ERR_INVALID, "null argument");
  RealArgs a;
  a.in = in; a.out = out; a.n = (int)rshape[2]; a.prec = precision;
  a.in_stride = rshape[2]; a.out_stride = rshape[2] / 2 + 1; a.nrows = rshape[0] * rshape[1]; a.scale = 1.0;
  MFFT_TRY(launch_r2c(a, nullptr));
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

int mfft_c2r_last(const void* in, void* out, const int64_t rshape[3], int precision) {
  if (!in || !out || !rshape) return set_error(MFFT_ERR_INVALID, "null argument");
  RealArgs a;
  a.in = in; a.out = out; a.n = (int)rshape[2]; a.prec = precision;
  a.in_stride = rshape[2] / 2 + 1; a.out_stride = rshape[2]; a.nrows = rshape[0] * rshape[1];
  a.scale = 1.0 / (double)rshape[2];
  MFFT_TRY(launch_c2r(a, nullptr));
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

// fused nonlinear z stage on rows (csrc/fft_nlz.h): a, b, out are (3, nrows, pitch) complex, `valid` bins per row exist
int mfft_nlz_rows(const void* a, const void* b, void* out, int64_t nrows, int64_t n, int64_t pitch, int64_t valid, int precision,
                  int sync) {
  if (!a || !b || !out || nrows < 1 || n < 2 || pitch < valid || valid < 1) return set_error(MFFT_ERR_INVALID, "bad argument");
  const size_t es = elem_bytes(precision, true);
  NlzArgs z;
  for (int f = 0; f < 3; ++f) {
    z.a[f] = static_cast<const char*>(a) + (size_t)(f * nrows * pitch) * es;
    z.b[f] = static_cast<const char*>(b) + (size_t)(f * nrows * pitch) * es;
    z.out[f] = static_cast<char*>(out) + (size_t)(f * nrows * pitch) * es;
  }
  z.n = (int)n; z.prec = precision; z.in_stride = pitch; z.out_stride = pitch; z.nrows = nrows; z.valid = (int)valid;
  z.scale = 1.0 / ((double)n * (double)n);
  MFFT_TRY(launch_nlz(z, nullptr));
  if (sync) MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

// U_mpi[p, i, j, k] = Uc_hatT[i, p*Np1 + j, k]   (slab.py:403)
int mfft_slab_pack(const void* uc_hatT, void* u_mpi, int P, int64_t np0, int64_t np1, int64_t nf, int precision) {
  if (!uc_hatT || !u_mpi || P < 1) return set_error(MFFT_ERR_INVALID, "bad argument");
  const size_t es = elem_bytes(precision, true);
  for (int p = 0; p < P; ++p) {
    BoxArgs b;
    b.src = static_cast<const char*>(uc_hatT) + (size_t)(p * np1 * nf) * es;
    b.dst = static_cast<char*>(u_mpi) + (size_t)p * (np0 * np1 * nf) * es;
    b.e0 = np0; b.e1 = 1; b.e2 = np1 * nf; b.s0 = (int64_t)P * np1 * nf; b.d0 = np1 * nf;
    b.elem = (int)es; b.prec = precision;
    MFFT_TRY(launch_box_copy(b, nullptr));
  }
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

// Uc_hatT[i, p*Np1 + j, k] = U_mpi[p, i, j, k]   (cython/maths.pyx:21-31)
int mfft_slab_unpack(const void* u_mpi, void* uc_hatT, int P, int64_t np0, int64_t np1, int64_t nf, int precision) {
  if (!uc_hatT || !u_mpi || P < 1) return set_error(MFFT_ERR_INVALID, "bad argument");
  const size_t es = elem_bytes(precision, true);
  for (int p = 0; p < P; ++p) {
    BoxArgs b;
    b.src = static_cast<const char*>(u_mpi) + (size_t)p * (np0 * np1 * nf) * es;
    b.dst = static_cast<char*>(uc_hatT) + (size_t)(p * np1 * nf) * es;
    b.e0 = np0; b.e1 = 1; b.e2 = np1 * nf; b.s0 = np1 * nf; b.d0 = (int64_t)P * np1 * nf;
    b.elem = (int)es; b.prec = precision;
    MFFT_TRY(launch_box_copy(b, nullptr));
  }
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

int mfft_dealias_filter(void* fu, const uint8_t* mask_dev, size_t count, int precision) {
  if (!fu || !mask_dev) return set_error(MFFT_ERR_INVALID, "null argument");
  MFFT_TRY(launch_mask(fu, mask_dev, count, precision, nullptr));
  MFFT_HIP(hipStreamSynchronize(nullptr));
  return 0;
}

// ---- timers ---------------------------------------------------------------------
struct mfft_timer_s {
  hipEvent_t a = nullptr, b = nullptr;
};
int mfft_timer_create(mfft_timer_t* t) {
  if (!t) return set_error(MFFT_ERR_INVALID, "null argument");
  mfft_timer_s* x = new mfft_timer_s();
  hipError_t e1 = hipEventCreate(&x->a), e2 = hipEventCreate(&x->b);
  if (e1 != hipSuccess || e2 != hipSuccess) {
    delete x;
    return set_error(MFFT_ERR_HIP, "hipEventCreate failed");
  }
  *t = x;
  return 0;
}
int mfft_timer_start(mfft_timer_t t) {
  MFFT_HIP(hipEventRecord(t->a, nullptr));
  return 0;
}
int mfft_timer_stop(mfft_timer_t t, float* ms) {
  MFFT_HIP(hipEventRecord(t->b, nullptr));
  MFFT_HIP(hipEventSynchronize(t->b));
  MFFT_HIP(hipEventElapsedTime(ms, t->a, t->b));
  return 0;
}
int mfft_timer_destroy(mfft_timer_t t) {
  if (t) {
    (void)hipEventDestroy(t->a);
    (void)hipEventDestroy(t->b);
    delete t;
  }
  return 0;
}

}  // extern "C"
