// ipc_pull.h -- the receiver side of the IPC transport's all-to-all as ONE kernel: every job is a chunk that lies in
// a peer's send buffer (reached through its IPC mapping, i.e. over that peer's own xGMI link) and has to land in this
// rank's receive buffer.  One launch serves all peers of the exchange at once, a few workgroups per peer, so all links
// of the fully connected node carry data at the same time -- which a stream of copy commands (one hipMemcpyAsync after
// the other, ipc_comm.hip "copy" mode) cannot do.  It replaces, like the rest of the transport, the MPI library behind
// comm.Alltoall / Alltoallw (slab.py:406, 281; pencil.py:741-750, 1324-1333).
//
// Shape of the launch: grid = njobs * wgs workgroups of 256 threads; workgroup b serves job b % njobs (neighbouring
// workgroups read from DIFFERENT peers: whatever subset of the grid is resident, the links are loaded evenly) and is the
// (b / njobs)-th of the wgs workgroups of that job.  The workgroups of a job walk the chunk in interleaved 4 KiB
// pieces (256 lanes x 16 B), eight pieces in flight per workgroup (128 B per lane): with wgs = 8 that is 256 KiB in
// flight per link, more than the bandwidth-delay product of an xGMI link (~77 GB/s x 2-3 us).  Loads and stores are
// non-temporal: the data is touched once on either side.  No LDS, < 48 VGPRs: the workgroups fit beside anything.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mfft {

constexpr int PULL_MAX_JOBS = 64;        // a relayed exchange of 16 ranks in groups of 4: 3 direct + 1 self + 36 first hops
constexpr int PULL_THREADS = 256;
constexpr int PULL_UNROLL = 8;

struct PullJob {
  const void* src;            // in the peer's (or, for the self chunk, this rank's) send buffer
  void* dst;                  // in this rank's receive buffer
  unsigned long long bytes;
};
struct PullArgs {
  PullJob job[PULL_MAX_JOBS];
  int njobs;
  int wgs;                    // workgroups per job
};

// Batched flag operations of the transport (one launch instead of one stream memory operation per peer -- which on
// this stack are kernels of their own anyway, __amd_rocclr_streamOpsWait / Write): lane i of a single wave either
// stores value[i] to *addr[i] with system-scope release (a peer's flag word, through its IPC mapping) or polls *addr[i]
// (a word of this rank's own fine-grained flag array) until it is >= value[i] in wrap-around arithmetic.
constexpr int FLAG_MAX_OPS = 64;
struct FlagOps {
  unsigned int* addr[FLAG_MAX_OPS];
  unsigned int value[FLAG_MAX_OPS];
  int n;
};

#if defined(__HIPCC__)
static __global__ __launch_bounds__(64) void ipc_signal_kernel(FlagOps f) {
  const int i = (int)threadIdx.x;
  if (i < f.n) __hip_atomic_store(f.addr[i], f.value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static __global__ __launch_bounds__(64) void ipc_wait_kernel(FlagOps f) {
  const int i = (int)threadIdx.x;
  if (i < f.n) {
    while ((int)(__hip_atomic_load(f.addr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - f.value[i]) < 0) __builtin_amdgcn_s_sleep(8);
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (system scope; the kernels behind this one start with their own acquire as well)
}
static inline hipError_t launch_flags(bool wait, const FlagOps& f, hipStream_t s) {
  if (f.n <= 0) return hipSuccess;
  if (wait) hipLaunchKernelGGL(ipc_wait_kernel, dim3(1), dim3(64), 0, s, f);
  else hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(64), 0, s, f);
  return hipGetLastError();
}

typedef unsigned int pull_v4 __attribute__((ext_vector_type(4)));
typedef unsigned int pull_v2 __attribute__((ext_vector_type(2)));

template <typename V>
__device__ __forceinline__ void pull_stream(const char* __restrict__ src, char* __restrict__ dst, unsigned long long bytes,
                                            int w, int wgs, int tid) {
  const V* __restrict__ s = reinterpret_cast<const V*>(src);
  V* __restrict__ d = reinterpret_cast<V*>(dst);
  const unsigned long long n = bytes / sizeof(V);
  const unsigned long long stride = (unsigned long long)wgs * PULL_THREADS;
  unsigned long long i = (unsigned long long)w * PULL_THREADS + (unsigned)tid;
  for (; i + (PULL_UNROLL - 1) * stride < n; i += PULL_UNROLL * stride) {
    V v[PULL_UNROLL];
#pragma unroll
    for (int u = 0; u < PULL_UNROLL; ++u) v[u] = __builtin_nontemporal_load(s + i + u * stride);
#pragma unroll
    for (int u = 0; u < PULL_UNROLL; ++u) __builtin_nontemporal_store(v[u], d + i + u * stride);
  }
  for (; i < n; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
  // bytes beyond the last whole vector (only when a count is not a multiple of the vector size)
  if (w == 0) {
    const unsigned long long done = n * sizeof(V);
    for (unsigned long long b = done + (unsigned)tid; b < bytes; b += PULL_THREADS) dst[b] = src[b];
  }
}

static __global__ __launch_bounds__(PULL_THREADS) void ipc_pull_kernel(PullArgs a) {
  // The sources were written by OTHER devices' kernels and released to system scope before their "ready" words were set
  // (ipc_comm.hip); this launch sits behind the wait for those words.  An explicit system-scope acquire on top of the
  // one the launch itself carries: no line of a peer's memory cached by an earlier exchange may be served again.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  const int j = (int)blockIdx.x % a.njobs, w = (int)blockIdx.x / a.njobs;
  const PullJob jb = a.job[j];
  const char* src = static_cast<const char*>(jb.src);
  char* dst = static_cast<char*>(jb.dst);
  const uintptr_t al = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst);
  if ((al & 15) == 0) pull_stream<pull_v4>(src, dst, jb.bytes, w, a.wgs, (int)threadIdx.x);
  else if ((al & 7) == 0) pull_stream<pull_v2>(src, dst, jb.bytes, w, a.wgs, (int)threadIdx.x);   // complex64 chunks at odd element offsets
  else pull_stream<unsigned char>(src, dst, jb.bytes, w, a.wgs, (int)threadIdx.x);
}

static inline hipError_t launch_pull(const PullArgs& a, hipStream_t s) {
  if (a.njobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(ipc_pull_kernel, dim3((unsigned)(a.njobs * a.wgs)), dim3(PULL_THREADS), 0, s, a);
  return hipGetLastError();
}
#endif

}  // namespace mfft
