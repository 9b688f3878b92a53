// occ_probe.hip -- how many workgroups does a CU of this GPU really hold at once, as a function of the workgroup's dynamic
// LDS and its threads?  (developer tool, not part of the library)
//
// Round 5 found the strided kernel of length 1200 (320 threads, 76 800 bytes of LDS, capped at 168 registers) running ONE
// workgroup per CU although hipOccupancyMaxActiveBlocksPerMultiprocessor answers 2 (tools/membench stamp1200: no two
// lifetimes on a CU overlap).  Every workgroup here does nothing but wait a fixed number of shader cycles, so the run time
// of a grid of G x 256 workgroups is  G / (resident per CU)  waits: the residency is read off the time.
//   make -C tools occ_probe && tools/build/occ_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);  \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int THREADS, int WPS>
__global__ __launch_bounds__(THREADS, WPS) void k_wait(int* sink, long long cycles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && cycles < 0) sink[blockIdx.x] = lds[0];
}

// the same wait in a kernel that HOLDS a given number of vector registers (the highest one is written: the allocation is
// what counts): does the second workgroup of 5 waves find room when a SIMD takes three waves of 168 registers?
template <int THREADS, int WPS, int VTOP>
__global__ __launch_bounds__(THREADS, WPS) void k_wait_regs(int* sink, long long cycles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  if constexpr (VTOP == 167) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  if constexpr (VTOP == 127) asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if constexpr (VTOP == 95) asm volatile("v_mov_b32 v95, 0" ::: "v95");
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (sink && threadIdx.x == 0 && cycles < 0) sink[blockIdx.x] = lds[0];
}
template <int THREADS, int WPS, int VTOP>
static void run_regs(int lds) {
  CK(hipFuncSetAttribute((const void*)k_wait_regs<THREADS, WPS, VTOP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds > 65536 ? lds : 65536));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_wait_regs<THREADS, WPS, VTOP>, THREADS, lds));
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)k_wait_regs<THREADS, WPS, VTOP>));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long long cycles = 100000;
  double t[2];
  for (int per_cu : {1, 12}) {
    hipLaunchKernelGGL((k_wait_regs<THREADS, WPS, VTOP>), dim3(256 * per_cu), dim3(THREADS), lds, 0, nullptr, cycles);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_wait_regs<THREADS, WPS, VTOP>), dim3(256 * per_cu), dim3(THREADS), lds, 0, nullptr, cycles);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    t[per_cu == 1 ? 0 : 1] = ms;
  }
  printf("%4d threads (%4.1f waves per SIMD), %3d registers (compiler: %d), lds %6d B: %.2f resident per CU   (API: %d)\n", THREADS, THREADS / 256.0,
         VTOP + 1, fa.numRegs, lds, 12 * t[0] / t[1], occ);
  fflush(stdout);
}

// where do the waves of a workgroup go?  every wave records its SIMD (HW_ID bits 5:4) and its CU
template <int THREADS, int WPS, int VTOP>
__global__ __launch_bounds__(THREADS, WPS) void k_where(unsigned* out, long long cycles) {
  if constexpr (VTOP == 167) asm volatile("v_mov_b32 v167, 0" ::: "v167");
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (THREADS / 64) + threadIdx.x / 64] = hwid;
}
template <int THREADS, int WPS, int VTOP>
static void where() {
  const int W = THREADS / 64, grid = 256 * 6;
  unsigned* d = nullptr;
  CK(hipMalloc(&d, sizeof(unsigned) * grid * W));
  hipLaunchKernelGGL((k_where<THREADS, WPS, VTOP>), dim3(grid), dim3(THREADS), 0, 0, d, 20000);
  CK(hipDeviceSynchronize());
  unsigned* h = (unsigned*)malloc(sizeof(unsigned) * grid * W);
  CK(hipMemcpy(h, d, sizeof(unsigned) * grid * W, hipMemcpyDeviceToHost));
  long pat[256] = {0};      // key: waves on SIMD0..3 as four base-4... digits (up to 3 each here; W <= 8: base 9)
  long per_simd[4] = {0, 0, 0, 0};
  for (int b = 0; b < grid; ++b) {
    int cnt[4] = {0, 0, 0, 0};
    for (int w = 0; w < W; ++w) ++cnt[(h[b * W + w] >> 4) & 3];
    for (int i = 0; i < 4; ++i) per_simd[i] += cnt[i];
    int key = 0;
    for (int i = 0; i < 4; ++i) key = key * 4 + (cnt[i] > 3 ? 3 : cnt[i]);
    ++pat[key];
  }
  printf("%4d threads = %d waves, %3d registers: waves per SIMD over %d workgroups: %ld %ld %ld %ld; patterns (SIMD0..3 -> workgroups):", THREADS, W, VTOP + 1,
         grid, per_simd[0], per_simd[1], per_simd[2], per_simd[3]);
  for (int k = 0; k < 256; ++k)
    if (pat[k]) printf("  %d%d%d%d -> %ld", k / 64, (k / 16) % 4, (k / 4) % 4, k % 4, pat[k]);
  printf("\n");
  fflush(stdout);
  free(h);
  CK(hipFree(d));
}

template <int THREADS, int WPS>
static double run(int lds, int per_cu, long long cycles, int* api_occ) {
  CK(hipFuncSetAttribute((const void*)k_wait<THREADS, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds > 65536 ? lds : 65536));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(api_occ, k_wait<THREADS, WPS>, THREADS, lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grid = 256 * per_cu;
  hipLaunchKernelGGL((k_wait<THREADS, WPS>), dim3(grid), dim3(THREADS), lds, 0, nullptr, cycles);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_wait<THREADS, WPS>), dim3(grid), dim3(THREADS), lds, 0, nullptr, cycles);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms;
}

template <int THREADS, int WPS>
static void sweep(const char* what) {
  const long long cycles = 100 * 1000 * 100 / 100;      // readcyclecounter: 100 MHz constant clock -> 1 ms?  measured below
  // calibrate: one workgroup per CU
  int occ = 0;
  const double one = run<THREADS, WPS>(1024, 1, cycles, &occ);
  printf("== %s: %d threads, register cap for %d waves per SIMD; one wait = %.3f ms\n", what, THREADS, WPS, one);
  const int per_cu = 12;
  for (int kb : {16, 32, 40, 48, 52, 56, 60, 64, 68, 70, 72, 73, 74, 75, 76, 78, 80, 96, 128, 160}) {
    const int lds = kb * 1024 < 163840 ? kb * 1024 : 163840;
    const double ms = run<THREADS, WPS>(lds, per_cu, cycles, &occ);
    printf("   lds %6d B (%3d KB): %7.3f ms for %d per CU -> %.2f resident per CU   (API: %d)\n", lds, kb, ms, per_cu, per_cu * one / ms, occ);
  }
  fflush(stdout);
}

int main() {
  printf("== where the waves of a workgroup go\n");
  where<64, 1, 0>();
  where<128, 1, 0>();
  where<192, 1, 0>();
  where<256, 1, 0>();
  where<320, 3, 167>();
  where<320, 1, 0>();
  where<384, 1, 0>();
  where<640, 1, 0>();
  printf("== workgroups that hold registers\n");
  run_regs<320, 3, 167>(76800);      // the 1200 kernel under its cap
  run_regs<320, 3, 167>(16384);
  run_regs<320, 4, 127>(76800);
  run_regs<320, 5, 95>(76800);
  run_regs<384, 3, 167>(76800);      // 6 waves
  run_regs<256, 3, 167>(76800);      // 4 waves
  run_regs<256, 3, 167>(16384);
  run_regs<512, 3, 167>(16384);      // 8 waves: 2 per SIMD, so one workgroup of three-per-SIMD registers
  run_regs<192, 3, 167>(16384);      // 3 waves
  run_regs<640, 5, 95>(76800);       // 10 waves
  run_regs<1024, 8, 0>(65536);
  sweep<320, 3>("the 1200 kernel's shape");
  sweep<320, 1>("the same, no register cap asked");
  sweep<256, 2>("256 threads");
  sweep<384, 3>("the 1440 kernel's shape");
  sweep<512, 2>("512 threads");
  sweep<1024, 8>("1024 threads (the 1024 kernel's shape)");
  // exact byte counts around the 1200 kernel's
  int occ = 0;
  const long long cycles = 100000;
  const double one = run<320, 3>(1024, 1, cycles, &occ);
  for (int lds : {73728, 74752, 75776, 76800, 77824, 78848, 79872, 80896, 81920}) {
    const double ms = run<320, 3>(lds, 12, cycles, &occ);
    printf("320 threads, lds %6d B: %.2f resident per CU (API: %d)\n", lds, 12 * one / ms, occ);
  }
  return 0;
}
