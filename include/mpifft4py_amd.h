/* mpifft4py_amd.h -- C ABI of libmpifft4py_amd.so
 *
 * MI355X-native (gfx950) distributed 3-D FFT: the drop-in boundary for the hot
 * path of spectralDNS/mpiFFT4py (slab / pencil R2C.fftn / ifftn).  The
 * reference has no C ABI of its own; its backend seam is the Python module
 * `mpiFFT4py/serialFFT` (serialFFT/__init__.py:1-6) plus the mpi4py
 * communicator handed to the constructors.  Every entry point below names the
 * reference interface it replaces.  The Python package `mpifft4py_amd` binds
 * this file with ctypes (mpifft4py_amd/_lib.py); INTEGRATION.md shows the stub
 * a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers, sizes, ints.  No C++/torch types cross the boundary.
 *   - every function returns 0 on success or a negative mfft_status; the text of
 *     the last failure on the calling thread is mfft_last_error().
 *   - all data pointers are DEVICE pointers (HBM) unless named *_host.
 *   - arrays are C-order; complex = interleaved (re, im).
 *   - a plan is bound to the device current at creation and to one comm.
 */
#ifndef MPIFFT4PY_AMD_H
#define MPIFFT4PY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFFT_API __attribute__((visibility("default")))

typedef enum {
  MFFT_OK = 0,
  MFFT_ERR_INVALID = -1,      /* bad argument */
  MFFT_ERR_UNSUPPORTED = -2,  /* e.g. a transform length beyond mfft_length_supported */
  MFFT_ERR_HIP = -3,          /* HIP runtime failure */
  MFFT_ERR_RCCL = -4,         /* RCCL failure / library not loadable */
  MFFT_ERR_NOMEM = -5,
  MFFT_ERR_INTERNAL = -6
} mfft_status;

typedef enum { MFFT_SINGLE = 0, MFFT_DOUBLE = 1 } mfft_precision;     /* mpibase.py:133-137 */
typedef enum { MFFT_R2C = 0, MFFT_C2C = 1 } mfft_kind;                 /* slab.R2C / slab.C2C */
typedef enum { MFFT_SLAB = 0, MFFT_PENCIL_X = 1, MFFT_PENCIL_Y = 2 } mfft_decomp;
typedef enum { MFFT_DEALIAS_NONE = 0, MFFT_DEALIAS_2_3 = 1, MFFT_DEALIAS_3_2 = 2 } mfft_dealias;

typedef struct mfft_comm_s* mfft_comm_t;
typedef struct mfft_plan_s* mfft_plan_t;

/* ---- library / device ------------------------------------------------- */
MFFT_API int mfft_version(void);
MFFT_API const char* mfft_last_error(void);
MFFT_API int mfft_device_count(int* count);
MFFT_API int mfft_set_device(int device);
MFFT_API int mfft_get_device(int* device);
MFFT_API int mfft_device_name(char* buf, size_t buflen);
MFFT_API int mfft_device_sync(void);
MFFT_API int mfft_device_pci_bus_id(int device, char* buf, size_t buflen);   /* "0000:05:00.0": which physical GPU a rank ran on */

/* ---- device memory (replaces mpibase.py:38-51 empty/zeros and the
 *      work_arrays cache, mpibase.py:53-131, for device-resident buffers) ---
 * mfft_memset / mfft_memcpy_* wait for ALL work of the current device (the transforms run asynchronously on their
 * plan's own stream, see mfft_forward) before they touch memory and return when the copy is complete. */
MFFT_API int mfft_malloc(void** dptr, size_t bytes);
MFFT_API int mfft_free(void* dptr);
MFFT_API int mfft_memset(void* dptr, int value, size_t bytes);
MFFT_API int mfft_memcpy_h2d(void* dst, const void* src_host, size_t bytes);
MFFT_API int mfft_memcpy_d2h(void* dst_host, const void* src, size_t bytes);
MFFT_API int mfft_memcpy_d2d(void* dst, const void* src, size_t bytes);
/* rows of width_bytes, device rows dev_pitch_bytes apart <-> packed host rows: how numpy arrays of the reference's shapes
 * (slab.py:102-104) go into / come out of a pitched spectrum (mfft_plan_desc::complex_pitch) */
MFFT_API int mfft_memcpy_rows_h2d(void* dst, size_t dev_pitch_bytes, const void* src_host, size_t width_bytes, size_t rows);
MFFT_API int mfft_memcpy_rows_d2h(void* dst_host, const void* src, size_t dev_pitch_bytes, size_t width_bytes, size_t rows);
MFFT_API int mfft_fill_uniform(void* dptr, size_t count, int precision, uint64_t seed); /* U[0,1) synthetic input */

/* ---- communicators (replace the mpi4py Comm injected at slab.py:77-81,
 *      pencil.py:173-195: Get_size/Get_rank/Split + Alltoall(w)) ----------- */
#define MFFT_UNIQUE_ID_BYTES 128
MFFT_API int mfft_comm_create_self(mfft_comm_t* comm);                 /* P = 1, no RCCL */
MFFT_API int mfft_get_unique_id(void* id128);                           /* rank 0; ncclGetUniqueId */
MFFT_API int mfft_comm_create_rccl(int nranks, int rank, const void* id128, mfft_comm_t* comm);
/* in-process group: nranks virtual ranks, each driven by its own host thread,
 * rank r living on devices[r] (NULL = all on the current device); exchange is
 * peer-to-peer device copies. */
MFFT_API int mfft_comm_create_local(int nranks, const int* devices, mfft_comm_t* comms_out);
MFFT_API int mfft_comm_size(mfft_comm_t comm, int* size);
MFFT_API int mfft_comm_rank(mfft_comm_t comm, int* rank);
MFFT_API int mfft_comm_barrier(mfft_comm_t comm);
/* One small all-to-all of a known byte pattern over all ranks (collective), verified on the host, waiting at most
 * timeout_ms for the device: 0 = the transport moves data correctly here.  A hung IPC exchange is released from the
 * host and reported as an error instead of blocking forever. */
MFFT_API int mfft_comm_selftest(mfft_comm_t comm, size_t bytes_per_peer, int timeout_ms);
/* Transport knobs, per communicator.  The IPC transport (the stand-in for the MPI library's intra-node all-to-all,
 * slab.py:406/281, pencil.py:741-750) knows "ipc_pull" = how a rank fetches its chunks from the peers' buffers: 1 one
 * pull kernel over all peers at once (default), 2 one copy-engine transfer per peer on per-peer streams (EXPERIMENTAL: the one
 * mode that ever stalled -- twice, with ranks sharing a device, round 4 -- and that the default test run therefore does not sweep:
 * MP_WORKER_STREAMS=1 / MFFT_BENCH_PULL_STREAMS=1 add it; do not rely on it before it has run on a node with a GPU per rank), 0 copy-engine
 * transfers one after the other; and "ipc_pull_wgs" = workgroups per peer of that kernel.  A rank-local choice (the
 * flag protocol is the same).  "ipc_relay" = 1 / 0: sub-group exchanges of pencil plans use the links to the ranks
 * outside the group as well (mfft_plan_relay_schedule) -- every rank must set the same value.  Other transports reject
 * every key; get returns -1 in *value for an unknown key. */
MFFT_API int mfft_comm_set_option(mfft_comm_t comm, const char* key, int64_t value);
MFFT_API int mfft_comm_get_option(mfft_comm_t comm, const char* key, int64_t* value);
/* host-buffer helpers for tests/demos (tests/test_FFT.py:77-78 Bcast; demo:103 reduce) */
MFFT_API int mfft_comm_bcast_host(mfft_comm_t comm, void* buf_host, size_t bytes, int root);
MFFT_API int mfft_comm_allreduce_sum_host(mfft_comm_t comm, double* vals_host, int count);
MFFT_API int mfft_comm_allreduce_max_host(mfft_comm_t comm, double* vals_host, int count);
/* mark an in-process group broken so that ranks blocked in its barriers return an error
 * (called when one rank's thread fails); no-op for the other transports */
MFFT_API int mfft_comm_abort(mfft_comm_t comm);
MFFT_API int mfft_comm_destroy(mfft_comm_t comm);

/* ---- plans: slab.R2C / slab.C2C / pencil.R2CX / pencil.R2CY --------------
 * (constructors slab.py:67-96, 556-574; pencil.py:167-216, 903-913) */
typedef struct {
  int64_t n[3];       /* global real mesh N0,N1,N2 */
  int precision;      /* mfft_precision */
  int kind;           /* mfft_kind */
  int decomp;         /* mfft_decomp */
  int p1;             /* pencil: ranks along the first axis, 0 = Compute_dims default.  Any p1 that divides the number
                         of ranks is accepted (also 1 x P, P x 1 and odd grids); the reference, and the Python classes
                         by default, insist on even P1 and P2 (pencil.py:204-208) */
  double padsize;     /* 3/2-rule pad factor (1.5) */
  int pipeline;       /* exchange pipeline: slab: n > 1 = n kz slices, n < -1 = |n| batches of local x rows; x-aligned
                         pencil: |n| batches of local x rows; each piece is exchanged on a second stream while the
                         next one is transformed; 0 = default (4), 1 = off */
  int drop_nyquist;   /* pencil 'AlltoallN' mode (pencil.py:410-432, 647-668): the kz = N2/2 column is
                         neither exchanged nor returned; the inverse treats it as zero */
  int line2d;         /* 1: the 2-D class of line.py:41-340, expressed as an x-aligned pencil plan of the mesh
                         (1, Nx, Ny) on a 1 x P grid: padsize^2 scaling, no Nyquist fold on one rank (line.py:185)
                         and, for P > 1, the Nyquist packing of line.py:231 in the padded forward transform */
  int comm_cus;       /* pipelined plans: compute units set aside for the communication stream (the compute stream gets
                         the others): > 0 that many, 0 = $MFFT_COMM_CUS or the library default, < 0 = no CU masks.
                         CU-masked streams are BLOCKING streams (hipExtStreamCreateWithCUMask takes no flags): work
                         on the legacy null stream of the device then orders against the plan's streams, which the
                         plain (non-blocking) plan streams do not.  The library itself never relies on either. */
  int complex_pitch;  /* PITCHED SPECTRUM (round 6; SURVEY 7 "padded device pitch internally, exact Nf at the API boundary"): 0 =
                         the caller's complex array is compact, rows of the local z extent (Nf = N2/2 + 1 bins: 8208 bytes
                         at 1024^3, never a whole cache line); -1 = rows a whole number of 128-byte lines apart (513 -> 520
                         bins); n > 0 = rows n elements apart.  The logical shape stays the reference's (slab.py:102-104);
                         mfft_layout_complex_pitch says what to allocate.  One-rank slab R2C plans run every pass on the
                         pitched rows; all other plans convert at the boundary (one more pass, same results). */
  int reserved[3];
} mfft_plan_desc;

MFFT_API int mfft_plan_create(mfft_comm_t comm, const mfft_plan_desc* desc, mfft_plan_t* plan);
MFFT_API int mfft_plan_destroy(mfft_plan_t plan);
/* local shapes and global start offsets (slab.py:98-144, pencil.py:248-287, 915-943) */
MFFT_API int mfft_plan_layout(mfft_plan_t plan, int64_t real_shape[3], int64_t complex_shape[3],
                              int64_t real_start[3], int64_t complex_start[3],
                              int64_t real_shape_padded[3], int64_t grid[2], int64_t subranks[2]);
/* The same layout for rank `rank` of `nranks`, computed on the host WITHOUT a plan, a communicator or a device (the
 * decomposition bookkeeping of slab.py:82-144, pencil.py:187-287, 903-943 is integer arithmetic); it also rejects what
 * mfft_plan_create would reject (indivisible meshes, lengths without a kernel).  The Python classes build their shape,
 * slice and mesh helpers (slab.py:146-197, pencil.py:289-349) on it, so those work on a machine without a GPU. */
MFFT_API int mfft_layout_query(const mfft_plan_desc* desc, int nranks, int rank, int64_t real_shape[3],
                               int64_t complex_shape[3], int64_t real_start[3], int64_t complex_start[3],
                               int64_t real_shape_padded[3], int64_t grid[2], int64_t subranks[2]);
MFFT_API int mfft_layout_complex_pitch(const mfft_plan_desc* desc, int nranks, int rank, int64_t* pitch_elems, int64_t* alloc_elems);
MFFT_API int mfft_plan_workspace_bytes(mfft_plan_t plan, size_t* bytes);
/* The all-to-all-v a rank performs, computed on the host WITHOUT a device: the peer
 * list (world ranks, in group order) and byte counts / displacements of every chunk.
 * slab: which = 0.  pencil: which = 0 is the z-splitting exchange (uneven last chunk;
 * comm1 for X, comm0 for Y), which = 1 the other one.  It is the schedule the executor
 * itself uses; it describes what the reference expresses with Alltoall counts and the
 * Alltoallw sub-array types (slab.py:199-211, 406, 281; pencil.py:218-246, 971-999). */
MFFT_API int mfft_plan_exchange_schedule(const mfft_plan_desc* desc, int nranks, int rank, int which,
                                         int forward, int padded, int max_peers, int* npeers, int* peers,
                                         size_t* scount, size_t* sdisp, size_t* rcount, size_t* rdisp);
/* The same for ONE piece of the pipelined exchange that `desc->pipeline` selects (slab: kz slice or batch of local
 * x rows; x-aligned pencil: batch of local x rows of exchange `which`): displacements are relative to the whole send /
 * receive buffers, *npieces returns how many pieces the exchange has (1 = not pipelined: the whole exchange). */
MFFT_API int mfft_plan_exchange_pieces(const mfft_plan_desc* desc, int nranks, int rank, int which, int forward,
                                       int piece, int max_peers, int* npieces, int* npeers, int* peers,
                                       size_t* scount, size_t* sdisp, size_t* rcount, size_t* rdisp);

/* Relay striping of the pencils' sub-group exchanges over the IPC transport (csrc/relay_plan.h): on a fully connected
 * xGMI node a rank that exchanges with the g - 1 peers of its comm0 / comm1 group (pencil.py:741-750, 1324-1333) leaves
 * its links to the other P - g ranks idle; a message is therefore cut into a direct part and P - g stripes that travel
 * through those ranks in two hops.  This query lists, device-free, what `rank` PULLS in exchange `which`: phase (1, 2),
 * kind (0 own chunk, 1 direct part read from msg_src's send buffer, 2 first hop: rank is the relay and stages the stripe
 * of msg_src -> msg_dst, 3 second hop: read from relay `from`'s staging area), the offset inside the message and the
 * byte count.  It is the enumeration the transport executes when the "ipc_relay" option of mfft_comm_set_option (or
 * MFFT_IPC_RELAY=1) is on; the default is off. */
MFFT_API int mfft_plan_relay_schedule(const mfft_plan_desc* desc, int nranks, int rank, int which, int forward,
                                      int max_moves, int* nmoves, int* phase, int* kind, int* from, int* msg_src,
                                      int* msg_dst, size_t* msg_off, size_t* bytes);

/* mfft_forward / mfft_backward ENQUEUE the transform on the plan's own (non-blocking) HIP stream and return; calls on
 * one plan execute in order.  Before the host or another stream reads the result (or reuses the input), call
 * mfft_plan_sync -- or one of the mfft_memcpy_* helpers, which synchronise the device themselves.
 * fftn: slab.py:349-485 / pencil.py:634-883, 1228-1475.  `u` is never written. */
MFFT_API int mfft_forward(mfft_plan_t plan, const void* u, void* fu, int dealias);
/* ifftn: slab.py:214-346 / pencil.py:386-632, 1001-1224.  `fu` is never written. */
MFFT_API int mfft_backward(mfft_plan_t plan, const void* fu, void* u, int dealias);
/* block the host until everything the plan enqueued has finished */
MFFT_API int mfft_plan_sync(mfft_plan_t plan);

/* 2/3-rule mask (slab.py:191-197; applied by cython/maths.pyx:9-19): uint8 per local complex element, copied from
 * the host array.  mfft_backward(..., MFFT_DEALIAS_2_3) transforms `fu * mask` without writing fu: the first inverse
 * pass applies the mask while it loads.  A mask of the form get_dealias_filter builds (the product of three 1-D band
 * conditions) is recognised here and lets slab R2C plans run pruned passes and, over P ranks, a smaller exchange; the
 * ranks agree on that with a host all-reduce, so for plans over more than one rank this call is COLLECTIVE. */
MFFT_API int mfft_plan_set_dealias_mask(mfft_plan_t plan, const uint8_t* mask_host, size_t count);

/* What a plan decided, by key: "pruned_route" (after mfft_plan_set_dealias_mask: 0 = the mask is applied on load,
 * 1 = pruned passes and a smaller exchange, 2 = the same with every ky of THIS rank removed), "comm_cus" (CUs set aside
 * for the communication stream, 0 = no masks), "kz_slices", "row_batches" (pieces of the exchange pipeline), "zfuse"
 * (pencils: z-chunk pack fused into the z transform), "ranks", "plane_pad" (one-rank slab plans: elements added to the
 * plane pitch of the intermediate because the mesh's own plane pitch is one the strided x pass reads slowly).  Unknown keys are MFFT_ERR_INVALID. */
MFFT_API int mfft_plan_get_info(mfft_plan_t plan, const char* key, int64_t* value);

/* per-stage timing with HIP events on the plan's own streams (bench roofline) */
MFFT_API int mfft_plan_timing(mfft_plan_t plan, int enable);
MFFT_API int mfft_plan_timing_reset(mfft_plan_t plan);
/* returns number of stages; arrays may be NULL to query the count */
MFFT_API int mfft_plan_timing_get(mfft_plan_t plan, int max_stages, char names[][32],
                                  double* total_ms, int64_t* calls, double* alg_bytes_per_call);

/* ---- stage level: the serialFFT seam (numpy_fft.py:25-107 / pyfftw_fft.py:26-203)
 * batched transforms of one axis of a contiguous C-order 3-D array on the
 * current device, default stream; synchronous w.r.t. the host on return. */
MFFT_API int mfft_c2c_axis(const void* in, void* out, const int64_t shape[3], int axis,
                           int inverse, int precision);                 /* fft / ifft  (ifft scaled 1/n) */
/* the same along a strided axis with every stride spelled out (a non-contiguous view in numpy's terms): `nouter` batches
 * in_outer / out_outer elements apart, each `ncols` contiguous columns wide, rows in_pitch / out_pitch elements apart */
MFFT_API int mfft_c2c_strided(const void* in, void* out, int64_t n, int64_t nouter, int64_t ncols, int64_t in_outer,
                              int64_t in_pitch, int64_t out_outer, int64_t out_pitch, int inverse, int precision);
MFFT_API int mfft_r2c_last(const void* in, void* out, const int64_t real_shape[3], int precision);    /* rfft axis=2 */
MFFT_API int mfft_c2r_last(const void* in, void* out, const int64_t real_shape[3], int precision);    /* irfft axis=2, scaled 1/n */
/* The fused nonlinear z stage on its own (csrc/fft_nlz.h; the z stages of mfft_nonlinear_cross): a, b, out are
 * (3, nrows, pitch) complex arrays of half-spectra rows of real length n of which the first `valid` bins exist
 * (n/2 + 1, or fewer: 3/2-rule); out_f[row] = rfft((irfft(a[:, row]) x irfft(b[:, row]))_f)[:valid], irfft as numpy's.
 * out may be a or b.  sync = 0: enqueued on the default stream and left running (timing loops).  Replaces the z stages
 * of six FFT.ifftn + three FFT.fftn around the demo's cross product (demo/spectral_dns_solver.py:53-71). */
MFFT_API int mfft_nlz_rows(const void* a, const void* b, void* out, int64_t nrows, int64_t n, int64_t pitch, int64_t valid,
                           int precision, int sync);
/* slab pack / unpack (slab.py:403; cython/maths.pyx:21-31 transpose_Uc) */
MFFT_API int mfft_slab_pack(const void* uc_hatT, void* u_mpi, int P, int64_t np0, int64_t np1, int64_t nf, int precision);
MFFT_API int mfft_slab_unpack(const void* u_mpi, void* uc_hatT, int P, int64_t np0, int64_t np1, int64_t nf, int precision);
/* fu[i] *= mask[i] (cython/maths.pyx:9-19 dealias_filter) */
MFFT_API int mfft_dealias_filter(void* fu, const uint8_t* mask_dev, size_t count, int precision);
/* 1 if a transform of length n along an axis is supported -- every n from 1 to 2^20, as numpy.fft / FFTW, the reference's
 * backends, take every n (numpy_fft.py:25-46).  mfft_length_route tells HOW: 1 = a radix plan (one kernel, register-resident:
 * 2^a <= 8192, 3*2^a <= 6144, 5*2^a <= 5120, 7*2^a <= 7168, 9*2^a <= 4608, 21*2^a <= 2688, 63*2^a <= 2016, 25*2^a <= 1600, 27*2^a <= 3456, 81*2^a <= 2592,
 * 125*2^a <= 2000, 15*2^a <= 3840, 45*2^a <= 1440, 75*2^a <= 2400, 135*2^a <= 2160, 225*2^a <= 1800, 375*2^a <= 3000, 675*2^a <= 2700,
 * 1125*2^a <= 2250, in single precision also 35*2^a <= 2240; real: twice that), 2 = the one-workgroup chirp-z kernels
 * (every other length up to 4096, even real lengths up to 8192; ~0.17 of the roofline), 3 = Bluestein's convolution over a
 * four-step power-of-two transform in a scratch buffer (csrc/bigfft.hip: every other length; plain transforms only -- the
 * 3/2-rule and 2/3-rule then take their copy-based routes; ~0.05 - 0.1 of the roofline), 0 = not supported (n > 2^20). */
MFFT_API int mfft_length_supported(int64_t n, int real_transform);
MFFT_API int mfft_length_route(int64_t n, int real_transform);
/* ... for one precision: the 35 * 2^a lengths have radix plans in single precision only (mfft_length_route answers for double) */
MFFT_API int mfft_length_route_precision(int64_t n, int real_transform, int precision);
/* Which compiled kernel a strided-axis (family 0), contiguous-axis c2c (1), r2c (2) or c2r (3) transform of length n
 * runs: "<plan name> tile=<columns or rows> threads=<n> lds=<bytes>[ nt]", e.g.
 * "cols n1024(8, 8, 4, 4)double tile=8 threads=1024 lds=81792".  Device-free.  bench.py uses it to check that a
 * committed rocprof profile belongs to the kernel it has just timed.  Returns MFFT_ERR_UNSUPPORTED for lengths that
 * go through the chirp-z kernels. */
MFFT_API int mfft_kernel_name(int family, int64_t n, int precision, int inverse, int nt, char* buf, size_t buflen);

/* ---- element-wise pieces of a pseudo-spectral Navier-Stokes step on device-resident
 * fields (what the reference demo does with numpy on the host,
 * demo/spectral_dns_solver.py:53-80).  Vector fields are (3, n) component-major; kx/ky/kz
 * are 1-D device vectors of the local spectral extents shape[0..2].  Enqueued on the
 * plan's stream (in order with its transforms); plan may be NULL (default stream). */
MFFT_API int mfft_ew_cross(mfft_plan_t plan, const void* a, const void* b, void* out, size_t n, int precision);          /* demo:53-58 */
MFFT_API int mfft_ew_curl_hat(mfft_plan_t plan, const void* U_hat, void* out, const void* kx, const void* ky, const void* kz,
                              const int64_t shape[3], int precision);                                   /* demo:60-64 */
MFFT_API int mfft_ew_ns_rhs(mfft_plan_t plan, void* dU, const void* U_hat, const void* kx, const void* ky, const void* kz,
                            const int64_t shape[3], double nu, int precision);                          /* demo:73-77 */
MFFT_API int mfft_ew_axpbz(mfft_plan_t plan, void* y, const void* x, const void* z, double alpha, double beta, size_t n_real,
                           int precision);                                                              /* demo:94-97 */
/* One Runge-Kutta stage of the demo's loop in one sweep (demo:73-77 compute_rhs' projection and viscous term, :94-97 the two
 * updates, :60-64 the curl the NEXT stage transforms): on entry N_hat holds the nonlinear term (mfft_nonlinear_cross);
 *   dU = N_hat - K (K . N_hat)/|K|^2 - nu |K|^2 U_hat;   U_hat1 += a_dt dU;
 *   last == 0: U_hat = U_hat0 + b_dt dU        last != 0: U_hat = U_hat0 = U_hat1   (the next step's `U_hat1[:] = U_hat0[:] = U_hat`)
 *   N_hat = i K x U_hat (new).
 * All four fields (3,) + shape, component-major; a_dt = a[rk] dt, b_dt = b[rk] dt. */
MFFT_API int mfft_ew_ns_rk_stage(mfft_plan_t plan, void* N_hat, void* U_hat, void* U_hat0, void* U_hat1, const void* kx,
                                 const void* ky, const void* kz, const int64_t shape[3], double nu, double a_dt, double b_dt,
                                 int last, int precision);
MFFT_API int mfft_ew_sumsq(mfft_plan_t plan, const void* x, size_t n_real, int precision, double* result_host);           /* demo:103 */

/* The nonlinear term of a pseudo-spectral step as ONE operation:
 *     out_hat = fftn(ifftn(a_hat) x ifftn(b_hat))        (cross product in real space, component by component)
 * with the plan's own transforms under `dealias` -- what demo/spectral_dns_solver.py:53-71 composes from six
 * FFT.ifftn(.., dealias), numpy products and three FFT.fftn(.., dealias).  a_hat, b_hat, out_hat: (3,) + local complex
 * shape, component-major, device-resident; out_hat may be a_hat or b_hat.  On one rank with radix kernels on every axis
 * (slab R2C plans; mfft_plan_get_info "nonlinear_fused_3_2" / "_none" / "_2_3") the z stages are ONE kernel per batch of x
 * planes (csrc/fft_nlz.h): six half-spectra rows in, the cross product formed in registers, three rows out -- the nine
 * real-space work arrays of the composition (9 x 1536^3 x 8 B at 1024^3 with the 3/2-rule) never exist.  Every other plan
 * runs the composition on work arrays of its own.  Enqueued on the plan's stream like mfft_forward. */
MFFT_API int mfft_nonlinear_cross(mfft_plan_t plan, const void* a_hat, const void* b_hat, void* out_hat, int dealias);

/* Direct evaluation of up to 16 DFT bins of a distributed field, for checking transforms of meshes that no host
 * transform can hold (BASELINE config 5, 2048^3): result[2b], result[2b+1] = Re, Im of
 *   sum over this rank's block u[x][y][z] * exp(-/+ 2 pi i (k0 (start0 + x) / n0 + k1 (start1 + y) / n1 + k2 (start2 + z) / n2)),
 * bins[3b .. 3b+2] = (k0, k1, k2), sign - for inverse = 0; products and sums in double precision, phase tables exact
 * (integer reduction mod n, long double).  The global bin is the sum of the ranks' results.  Synchronous.  What the
 * reference's tests do by comparing with a serial transform of the same data (tests/test_FFT.py:66-91). */
MFFT_API int mfft_ew_dft_bins(mfft_plan_t plan, const void* u, int is_complex, const int64_t shape[3], const int64_t start[3],
                              const int64_t n[3], int inverse, int nbins, const int64_t* bins_host, int precision,
                              double* result_host);

/* ---- HIP-event timers on the default stream (bench.py) ------------------ */
typedef struct mfft_timer_s* mfft_timer_t;
MFFT_API int mfft_timer_create(mfft_timer_t* t);
MFFT_API int mfft_timer_start(mfft_timer_t t);
MFFT_API int mfft_timer_stop(mfft_timer_t t, float* elapsed_ms);   /* synchronises on the stop event */
MFFT_API int mfft_timer_destroy(mfft_timer_t t);

#ifdef __cplusplus
}
#endif
#endif /* MPIFFT4PY_AMD_H */
