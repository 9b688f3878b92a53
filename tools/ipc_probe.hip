// ipc_probe.hip -- what works between two PROCESSES on this pool's GPUs (developer tool): IPC memory handles,
// interprocess events, stream memory operations (hipStreamWriteValue32 / hipStreamWaitValue32) on IPC-mapped memory,
// and the rate of a copy-engine push into the peer's buffer.  Forks before the first HIP call.
//   make -C tools ipc_probe && tools/build/ipc_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

struct Shared {
  std::atomic<int> arrived[256];
  hipIpcMemHandle_t mem[2], flag[2];
  hipIpcEventHandle_t ev[2];
  int ev_ok[2];
};
static Shared* sh;
static void barrier(int idx) {
  sh->arrived[idx].fetch_add(1);
  while (sh->arrived[idx].load() < 2) usleep(50);
}
#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("[rank %d] %s -> %s (line %d)\n", rank, #x, hipGetErrorString(e_), __LINE__);     \
      fflush(stdout);                                                                          \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

__global__ void fill(unsigned* p, size_t n, unsigned v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i;
}

static int run(int rank) {
  const int peer = 1 - rank;
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  CK(hipSetDevice(rank % ndev));
  const size_t bytes = (size_t)256 << 20;
  unsigned *buf = nullptr, *flags = nullptr;
  CK(hipMalloc(&buf, 2 * bytes));                  // [send | recv]
  CK(hipMalloc(&flags, 4096));
  CK(hipMemset(flags, 0, 4096));
  CK(hipIpcGetMemHandle(&sh->mem[rank], buf));
  CK(hipIpcGetMemHandle(&sh->flag[rank], flags));
  hipEvent_t ev = nullptr;
  hipError_t ee = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess);
  if (ee == hipSuccess) ee = hipIpcGetEventHandle(&sh->ev[rank], ev);
  sh->ev_ok[rank] = ee == hipSuccess;
  printf("[rank %d] interprocess event create+export: %s\n", rank, hipGetErrorString(ee));
  (void)hipGetLastError();
  barrier(0);
  unsigned *pbuf = nullptr, *pflags = nullptr;
  CK(hipIpcOpenMemHandle((void**)&pbuf, sh->mem[peer], hipIpcMemLazyEnablePeerAccess));
  CK(hipIpcOpenMemHandle((void**)&pflags, sh->flag[peer], hipIpcMemLazyEnablePeerAccess));
  hipEvent_t pev = nullptr;
  if (sh->ev_ok[peer]) {
    hipError_t e2 = hipIpcOpenEventHandle(&pev, sh->ev[peer]);
    printf("[rank %d] hipIpcOpenEventHandle: %s\n", rank, hipGetErrorString(e2));
    (void)hipGetLastError();
    if (e2 != hipSuccess) pev = nullptr;
  }
  int can = 0;
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, rank % ndev);
  printf("[rank %d] hipDeviceAttributeCanUseStreamWaitValue = %d\n", rank, can);
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t n = bytes / 4;
  // round r: fill send half, push it into the peer's recv half, then signal the peer's flag[0] = r; wait for own flag
  for (unsigned r = 1; r <= 3; ++r) {
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, s, buf, n, r * 1000u + (unsigned)rank);
    auto t0 = std::chrono::steady_clock::now();
    CK(hipMemcpyAsync(pbuf + n, buf, bytes, hipMemcpyDeviceToDevice, s));
    CK(hipStreamWriteValue32(s, pflags, r, 0));
    CK(hipStreamWaitValue32(s, flags, r, hipStreamWaitValueGte, 0xFFFFFFFFu));
    CK(hipStreamSynchronize(s));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::vector<unsigned> h(16);
    CK(hipMemcpy(h.data(), buf + n, 64, hipMemcpyDeviceToHost));
    unsigned tail = 0;
    CK(hipMemcpy(&tail, buf + n + n - 1, 4, hipMemcpyDeviceToHost));
    const unsigned want0 = r * 1000u + (unsigned)peer, wantt = want0 + (unsigned)(n - 1);
    printf("[rank %d] round %u: push %zu MB + flag handshake %.3f ms (%.1f GB/s); recv[0] = %u (want %u), recv[last] = %u (want %u) %s\n", rank,
           r, bytes >> 20, ms, bytes / ms / 1e6, h[0], want0, tail, wantt, (h[0] == want0 && tail == wantt) ? "OK" : "MISMATCH");
    fflush(stdout);
    barrier(r);          // both have checked before the next round overwrites
  }
  if (pev) {             // event flavour: record own, wait peer's
    CK(hipEventRecord(ev, s));
    barrier(4);
    hipError_t e3 = hipStreamWaitEvent(s, pev, 0);
    printf("[rank %d] hipStreamWaitEvent on the peer's interprocess event: %s\n", rank, hipGetErrorString(e3));
    CK(hipStreamSynchronize(s));
  } else {
    barrier(4);
  }
  barrier(5);
  // ---- re-export after free: does a new allocation at a recycled address export, with the peer's old mapping
  // still open (case A) or closed first (case B)?
  for (int cas = 0; cas < 2; ++cas) {
    for (size_t sz : {(size_t)532480, (size_t)798720, (size_t)(3u << 20)}) {
      void* a = nullptr;
      CK(hipMalloc(&a, sz));
      hipIpcMemHandle_t h1;
      hipError_t e1 = hipIpcGetMemHandle(&h1, a);
      memcpy(&sh->mem[rank], &h1, sizeof h1);
      barrier(10 + cas * 20 + (int)(sz % 7));
      void* pm = nullptr;
      hipError_t eo = hipIpcOpenMemHandle(&pm, sh->mem[peer], hipIpcMemLazyEnablePeerAccess);
      barrier(40 + cas * 20 + (int)(sz % 7));
      if (cas == 1 && pm) (void)hipIpcCloseMemHandle(pm);
      barrier(70 + cas * 20 + (int)(sz % 7));
      CK(hipFree(a));
      void* b2 = nullptr;
      CK(hipMalloc(&b2, sz + 4096));
      hipIpcMemHandle_t h2;
      hipError_t e2 = hipIpcGetMemHandle(&h2, b2);
      printf("[rank %d] case %c size %zu: first export %s, peer open %s, new allocation %s address, export of it: %s\n", rank, cas ? 'B' : 'A', sz,
             hipGetErrorString(e1), hipGetErrorString(eo), b2 == a ? "SAME" : "other", hipGetErrorString(e2));
      (void)hipGetLastError();
      fflush(stdout);
      barrier(100 + cas * 20 + (int)(sz % 7));
      if (cas == 0 && pm) (void)hipIpcCloseMemHandle(pm);
      CK(hipFree(b2));
      barrier(130 + cas * 20 + (int)(sz % 7));
    }
  }
  CK(hipIpcCloseMemHandle(pbuf));
  CK(hipIpcCloseMemHandle(pflags));
  barrier(6);
  CK(hipFree(buf));
  CK(hipFree(flags));
  printf("[rank %d] done\n", rank);
  return 0;
}

int main() {
  sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset((void*)sh, 0, sizeof(Shared));
  pid_t pid = fork();
  if (pid == 0) _exit(run(1));
  int rc = run(0), st = 0;
  waitpid(pid, &st, 0);
  printf("exit codes: %d %d\n", rc, WEXITSTATUS(st));
  return rc || WEXITSTATUS(st);
}
