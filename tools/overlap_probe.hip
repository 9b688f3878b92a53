// overlap_probe.hip -- can the exchange's pull kernel run BESIDE the strided FFT kernel, and what does it cost?
// (developer tool, not part of the library; links the library's object files for the real ColFft launches.)
//   make -C tools overlap_probe && tools/build/overlap_probe [comm_cus ...]
//
// Stream A runs the library's strided-axis transform of length 1024 over a (1024, 1024*513) fp64 array, `reps` launches
// back to back (the kernel that owns every CU: two 1024-thread workgroups with 80 KB of LDS each per CU).  Stream B
// runs ONE launch of the IPC transport's pull kernel (mpifft4py_amd/csrc/ipc_pull.h) copying 7 chunks of 134.5 MB
// (the slab exchange of a 1024^3 cube over 8 ranks) between two local buffers -- no xGMI in a one-GPU box, so the
// number is about scheduling, not about links.  B is launched right behind A's first kernel.
// For every stream configuration it prints:  A alone, B alone, then both together: A's time, B's time, and how long
// after A's start B finished.  B overlapped if B-done-after-A-start is close to "B alone"; it queued if it is close
// to "A alone + B alone".
//   configurations of (A, B):  plain/plain, plain/high priority, plain/CU-masked(K), CU-masked(256-K)/CU-masked(K)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ipc_pull.h"
#include "mfft_internal.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static int g_ncu = 256;

// CU mask with bits [lo, hi) set.  KFD deals the bits of a queue's mask round-robin over the XCDs first (bit i -> XCD
// i % 8), so a contiguous range of K = 8k bits is k CUs on every XCD.
static std::vector<uint32_t> cu_mask(int lo, int hi) {
  std::vector<uint32_t> m((g_ncu + 31) / 32, 0u);
  for (int i = lo; i < hi; ++i) m[i / 32] |= 1u << (i % 32);
  return m;
}
static hipStream_t masked_stream(int lo, int hi) {
  hipStream_t s = nullptr;
  std::vector<uint32_t> m = cu_mask(lo, hi);
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data());
  if (e != hipSuccess) {
    fprintf(stderr, "hipExtStreamCreateWithCUMask([%d,%d)): %s\n", lo, hi, hipGetErrorString(e));
    (void)hipGetLastError();
    return nullptr;
  }
  return s;
}

struct Work {
  void* fft_buf;
  int64_t N, NF;
  char *src, *dst;
  size_t chunk;
  int njobs, wgs, reps;
};

static void run_fft(const Work& w, hipStream_t s) {
  mfft::ColArgs a;
  a.in = w.fft_buf; a.out = w.fft_buf; a.n = (int)w.N; a.prec = MFFT_DOUBLE; a.inverse = false;
  a.nouter = 1; a.ncols = w.N * w.NF; a.in_outer = 0; a.out_outer = 0;
  a.in_rows.lo = w.N * w.NF; a.out_rows.lo = w.N * w.NF;
  a.scale = 1.0 / 1024.0;              // keeps the data bounded over many repetitions
  for (int r = 0; r < w.reps; ++r)
    if (mfft::launch_col(a, s) != 0) { fprintf(stderr, "launch_col: %s\n", mfft::last_error()); exit(1); }
}
static void run_pull(const Work& w, hipStream_t s) {
  mfft::PullArgs a;
  memset(&a, 0, sizeof a);
  a.njobs = w.njobs; a.wgs = w.wgs;
  for (int j = 0; j < w.njobs; ++j) { a.job[j].src = w.src + (size_t)j * w.chunk; a.job[j].dst = w.dst + (size_t)j * w.chunk; a.job[j].bytes = w.chunk; }
  CK(mfft::launch_pull(a, s));
}

static void measure(const char* name, const Work& w, hipStream_t A, hipStream_t B) {
  if (!A || !B) { printf("%-46s  (stream not available)\n", name); return; }
  hipEvent_t a0, a1, b0, b1;
  CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
  float fa = 0, fb = 0, ca = 0, cb = 0, cbd = 0;
  auto once = [&](bool doA, bool doB, float* ta, float* tb, float* tbd) {
    CK(hipDeviceSynchronize());
    if (doA) { CK(hipEventRecord(a0, A)); run_fft(w, A); CK(hipEventRecord(a1, A)); }
    if (doB) { CK(hipEventRecord(b0, B)); run_pull(w, B); CK(hipEventRecord(b1, B)); }
    CK(hipDeviceSynchronize());
    if (doA) CK(hipEventElapsedTime(ta, a0, a1));
    if (doB) CK(hipEventElapsedTime(tb, b0, b1));
    if (doA && doB) CK(hipEventElapsedTime(tbd, a0, b1));
  };
  float best_a = 1e9f, best_b = 1e9f;
  for (int it = 0; it < 4; ++it) {      // first round warms up
    once(true, false, &fa, nullptr, nullptr);
    once(false, true, nullptr, &fb, nullptr);
    if (it) { best_a = fa < best_a ? fa : best_a; best_b = fb < best_b ? fb : best_b; }
  }
  float sa = 0, sb = 0, sbd = 0;
  const int rounds = 5;
  for (int it = 0; it < rounds; ++it) {
    once(true, true, &ca, &cb, &cbd);
    sa += ca; sb += cb; sbd += cbd;
  }
  const double copied = 2.0 * (double)w.chunk * w.njobs;
  printf("%-46s  A alone %6.2f ms | B alone %6.2f ms (%5.0f GB/s) | together: A %6.2f ms, B %6.2f ms, B done %6.2f ms after A's start\n",
         name, best_a, best_b, copied / (best_b * 1e-3) / 1e9, sa / rounds, sb / rounds, sbd / rounds);
  CK(hipEventDestroy(a0)); CK(hipEventDestroy(a1)); CK(hipEventDestroy(b0)); CK(hipEventDestroy(b1));
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_ncu = prop.multiProcessorCount;
  printf("# %s, %d CUs\n", prop.name, g_ncu);
  Work w;
  w.N = 1024; w.NF = 513; w.reps = 4; w.njobs = 7; w.wgs = 8;
  w.chunk = (size_t)128 * 128 * 513 * 16;          // Np0 * Np1 * Nf * 16 B: one peer chunk of the 1024^3 slab exchange over 8 ranks
  if (getenv("PROBE_WGS")) w.wgs = atoi(getenv("PROBE_WGS"));
  if (getenv("PROBE_REPS")) w.reps = atoi(getenv("PROBE_REPS"));
  CK(hipMalloc(&w.fft_buf, (size_t)w.N * w.N * w.NF * 16));
  CK(hipMemset(w.fft_buf, 0, (size_t)w.N * w.N * w.NF * 16));
  CK(hipMalloc((void**)&w.src, w.chunk * w.njobs));
  CK(hipMalloc((void**)&w.dst, w.chunk * w.njobs));
  CK(hipMemset(w.src, 1, w.chunk * w.njobs));
  printf("# A = %d x ColFft n=1024 fp64 over (1024, 1024*513) in place; B = pull kernel, %d jobs x %.1f MB, %d workgroups per job\n",
         w.reps, w.njobs, w.chunk / 1e6, w.wgs);
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t plainA, plainB, prioB;
  CK(hipStreamCreateWithFlags(&plainA, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&plainB, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&prioB, hipStreamNonBlocking, hi));
  measure("A plain, B plain", w, plainA, plainB);
  measure("A plain, B highest priority", w, plainA, prioB);
  std::vector<int> ks;
  for (int i = 1; i < argc; ++i) ks.push_back(atoi(argv[i]));
  if (ks.empty()) ks = {8, 16, 32};
  for (int k : ks) {
    char nm[96];
    hipStream_t mb = masked_stream(g_ncu - k, g_ncu), ma = masked_stream(0, g_ncu - k);
    snprintf(nm, sizeof nm, "A plain, B masked to %d CUs", k);
    measure(nm, w, plainA, mb);
    snprintf(nm, sizeof nm, "A masked to %d CUs, B masked to the other %d", g_ncu - k, k);
    measure(nm, w, ma, mb);
    if (ma) CK(hipStreamDestroy(ma));
    if (mb) CK(hipStreamDestroy(mb));
  }
  // verify the copy
  std::vector<unsigned char> h(4096);
  CK(hipMemcpy(h.data(), w.dst + w.chunk * (w.njobs - 1) + w.chunk - 4096, 4096, hipMemcpyDeviceToHost));
  for (unsigned char c : h) if (c != 1) { printf("COPY WRONG\n"); return 1; }
  printf("# copy verified\n");
  return 0;
}
