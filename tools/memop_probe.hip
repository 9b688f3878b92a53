// memop_probe.hip -- do HIP events see stream memory operations?  (developer tool)
// Stream s: hipStreamWaitValue32(flag >= 1) with the flag still 0, then hipEventRecord(ev, s).  If hipEventQuery(ev)
// reports completion before the host sets the flag, an event recorded behind a wait does not wait for it, and every
// cross-stream dependency built on such an event (the plan's ev_comm behind IpcComm's done-flag waits) is void.
//   make -C tools memop_probe && tools/build/memop_probe
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);        \
      return 1;                                                                   \
    }                                                                             \
  } while (0)
__global__ void mark(unsigned* p, unsigned v) { *p = v; }
__global__ void nop() {}

static int round_(const char* what, bool fence_kernel, bool write_first) {
  unsigned *flag = nullptr, *out = nullptr;
  CK(hipMalloc(&flag, 4096));
  CK(hipMalloc(&out, 4096));
  CK(hipMemset(flag, 0, 4096));
  CK(hipMemset(out, 0, 4096));
  hipStream_t s, s2;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev;
  CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s);          // a tracked command ahead of the memory operations
  if (write_first) CK(hipStreamWriteValue32(s, flag + 16, 7, 0));
  CK(hipStreamWaitValue32(s, flag, 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
  if (fence_kernel) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, s);
  CK(hipEventRecord(ev, s));
  CK(hipStreamWaitEvent(s2, ev, 0));
  hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, s2, out, 42u);
  usleep(300000);
  hipError_t q = hipEventQuery(ev);
  unsigned h = 0;
  hipStream_t s3;
  CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  CK(hipMemcpyAsync(&h, out, 4, hipMemcpyDeviceToHost, s3));
  CK(hipStreamSynchronize(s3));
  printf("%-62s before the flag is set: event %s, dependent kernel on another stream %s\n", what,
         q == hipSuccess ? "COMPLETE (ignores the wait)" : "not ready", h == 42 ? "HAS RUN (ordering broken)" : "has not run");
  CK(hipStreamWriteValue32(s3, flag, 1, 0));                 // release
  CK(hipStreamSynchronize(s));
  CK(hipStreamSynchronize(s2));
  CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
  printf("%-62s after: dependent kernel %s\n", "", h == 42 ? "ran" : "DID NOT RUN");
  (void)hipGetLastError();
  CK(hipFree(flag));
  CK(hipFree(out));
  return 0;
}

int main() {
  if (round_("wait, record", false, false)) return 1;
  if (round_("wait, empty kernel, record", true, false)) return 1;
  if (round_("write, wait, record", false, true)) return 1;
  return 0;
}
