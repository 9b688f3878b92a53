// ipc_probe2.hip -- interprocess events used the way ipc_comm.hip uses them (developer tool)
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <chrono>
struct Shared { std::atomic<int> arrived[2048]; hipIpcEventHandle_t ev[2][4]; };
static Shared* sh;
static int bidx = 0;
static void barrier() { int i = bidx++; sh->arrived[i].fetch_add(1); while (sh->arrived[i].load() < 2) usleep(20); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[rank %d] %s -> %s (line %d)\n", rank, #x, hipGetErrorString(e_), __LINE__); fflush(stdout); (void)hipGetLastError(); } } while (0)
__global__ void spin_kernel(long long cycles, unsigned* out) {
  const long long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (out) *out = 12345u;
}
static int run(int rank) {
  const int peer = 1 - rank;
  CK(hipSetDevice(0));
  hipEvent_t mine[4], theirs[4];
  for (int i = 0; i < 4; ++i) {
    CK(hipEventCreateWithFlags(&mine[i], hipEventDisableTiming | hipEventInterprocess));
    CK(hipIpcGetEventHandle(&sh->ev[rank][i], mine[i]));
  }
  barrier();
  for (int i = 0; i < 4; ++i) CK(hipIpcOpenEventHandle(&theirs[i], sh->ev[peer][i]));
  barrier();
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  void* d = nullptr;
  CK(hipMalloc(&d, 1 << 20));
  for (int round = 0; round < 3; ++round) {
    CK(hipMemsetAsync(d, round, 1 << 20, s));
    CK(hipEventRecord(mine[0], s));
    barrier();
    hipError_t e = hipStreamWaitEvent(s, theirs[0], 0);
    printf("[rank %d] round %d wait ready: %s\n", rank, round, hipGetErrorString(e));
    (void)hipGetLastError();
    CK(hipEventRecord(mine[1], s));
    barrier();
    e = hipStreamWaitEvent(s, theirs[1], 0);
    printf("[rank %d] round %d wait done: %s\n", rank, round, hipGetErrorString(e));
    (void)hipGetLastError();
    barrier();
  }
  {  // a record BEHIND pending work: is it visible to the peer's wait, and does the peer really wait for it?
    unsigned* flag = (unsigned*)d;
    CK(hipMemsetAsync(d, 0, 64, s));
    CK(hipStreamSynchronize(s));
    barrier();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, s, 200000000LL, flag);     // ~0.1 s
    CK(hipEventRecord(mine[0], s));
    barrier();
    hipStream_t s3;
    CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
    hipError_t e = hipStreamWaitEvent(s3, theirs[0], 0);
    CK(hipStreamSynchronize(s3));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("[rank %d] wait on a record that sits behind a ~100 ms kernel: %s, returned after %.1f ms\n", rank, hipGetErrorString(e), ms);
    (void)hipGetLastError();
    CK(hipStreamSynchronize(s));
    barrier();
  }
  {  // how many record / wait rounds does ONE interprocess event survive?
    int first_bad = -1;
    for (int round = 0; round < 100; ++round) {
      CK(hipEventRecord(mine[1], s));
      barrier();
      hipError_t e = hipStreamWaitEvent(s, theirs[1], 0);
      (void)hipGetLastError();
      if (e != hipSuccess && first_bad < 0) first_bad = round;
      barrier();
      CK(hipStreamSynchronize(s));
      barrier();
    }
    printf("[rank %d] 100 record/wait rounds on one event: first failing round %d (3 earlier rounds on this event)\n", rank, first_bad);
  }
  hipStream_t s2;
  int pl = 0, pg = 0;
  CK(hipDeviceGetStreamPriorityRange(&pl, &pg));
  printf("[rank %d] stream priority range: least %d greatest %d\n", rank, pl, pg);
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, pg));
  for (int round = 0; round < 2; ++round) {
    // as the pipelined plans do: the second stream first waits for a plain event of the first stream
    hipEvent_t plain;
    CK(hipEventCreateWithFlags(&plain, hipEventDisableTiming));
    CK(hipMemsetAsync(d, round, 1 << 20, s));
    CK(hipEventRecord(plain, s));
    CK(hipStreamWaitEvent(s2, plain, 0));
    CK(hipEventRecord(mine[2], s2));
    barrier();
    hipError_t e = hipStreamWaitEvent(s2, theirs[2], 0);
    printf("[rank %d] round %d (events 2/3, second stream) wait ready: %s\n", rank, round, hipGetErrorString(e));
    (void)hipGetLastError();
    CK(hipEventRecord(mine[3], s2));
    barrier();
    e = hipStreamWaitEvent(s2, theirs[3], 0);
    printf("[rank %d] round %d (events 2/3, second stream) wait done: %s\n", rank, round, hipGetErrorString(e));
    (void)hipGetLastError();
    barrier();
  }
  CK(hipStreamSynchronize(s));
  CK(hipStreamSynchronize(s2));
  // variant: wait BEFORE any record of that event ever happened
  hipError_t e = hipSuccess;
  printf("[rank %d] (skipped) wait on a never-recorded peer event: %s\n", rank, hipGetErrorString(e));
  (void)hipGetLastError();
  barrier();
  fflush(stdout);
  return 0;
}
int main() {
  sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset((void*)sh, 0, sizeof(Shared));
  pid_t pid = fork();
  if (pid == 0) { bidx = 0; _exit(run(1)); }
  int rc = run(0), st = 0;
  waitpid(pid, &st, 0);
  return rc;
}
