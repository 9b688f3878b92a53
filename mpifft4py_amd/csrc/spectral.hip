// spectral.hip -- element-wise kernels of a pseudo-spectral Navier-Stokes step on
// device-resident fields, so that the 36 transforms of an RK4 step are not
// bracketed by PCIe copies (the reference demo does these in numpy on the host:
// demo/spectral_dns_solver.py:53-80).  All HBM-bound, 16 B per lane where the
// layout allows.  Fields are (3, n) component-major; wavenumbers are passed as
// three 1-D device vectors of the local spectral extents.
#include <math.h>
#include <vector>
#include "mfft_internal.h"

using namespace mfft;

namespace {

constexpr int EW_BLOCK = 256;
inline unsigned ew_grid(size_t n) {
  size_t g = (n + EW_BLOCK - 1) / EW_BLOCK;
  return (unsigned)(g > 8192 ? 8192 : (g ? g : 1));
}

// out = a x b   (real space, demo:53-58)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void cross_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                        T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T a0 = a[i], a1 = a[n + i], a2 = a[2 * n + i];
    const T b0 = b[i], b1 = b[n + i], b2 = b[2 * n + i];
    out[i] = a1 * b2 - a2 * b1;
    out[n + i] = a2 * b0 - a0 * b2;
    out[2 * n + i] = a0 * b1 - a1 * b0;
  }
}

// out = i K x U   (spectral space, demo:60-64)
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void curl_kernel(const cx<T>* __restrict__ U, cx<T>* __restrict__ out,
                                                       const T* __restrict__ kx, const T* __restrict__ ky,
                                                       const T* __restrict__ kz, int64_t s1, int64_t s2, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int64_t k = (int64_t)(i % s2), j = (int64_t)((i / s2) % s1), l = (int64_t)(i / (s2 * s1));
    const T K0 = kx[l], K1 = ky[j], K2 = kz[k];
    const cx<T> u0 = U[i], u1 = U[n + i], u2 = U[2 * n + i];
    // i * (a) = (-a.y, a.x)
    const cx<T> c0 = mk<T>(K1 * u2.x - K2 * u1.x, K1 * u2.y - K2 * u1.y);
    const cx<T> c1 = mk<T>(K2 * u0.x - K0 * u2.x, K2 * u0.y - K0 * u2.y);
    const cx<T> c2 = mk<T>(K0 * u1.x - K1 * u0.x, K0 * u1.y - K1 * u0.y);
    out[i] = mk<T>(-c0.y, c0.x);
    out[n + i] = mk<T>(-c1.y, c1.x);
    out[2 * n + i] = mk<T>(-c2.y, c2.x);
  }
}

// pressure projection and viscous term (demo:73-77):
//   P = sum_i dU_i K_i / |K|^2 ;  dU_i -= P K_i ;  dU_i -= nu |K|^2 U_i
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void rhs_kernel(cx<T>* __restrict__ dU, const cx<T>* __restrict__ U,
                                                      const T* __restrict__ kx, const T* __restrict__ ky,
                                                      const T* __restrict__ kz, int64_t s1, int64_t s2, size_t n, T nu) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int64_t k = (int64_t)(i % s2), j = (int64_t)((i / s2) % s1), l = (int64_t)(i / (s2 * s1));
    const T K0 = kx[l], K1 = ky[j], K2 = kz[k];
    const T k2 = K0 * K0 + K1 * K1 + K2 * K2;
    const T inv = k2 == (T)0 ? (T)1 : (T)1 / k2;
    cx<T> d0 = dU[i], d1 = dU[n + i], d2 = dU[2 * n + i];
    const cx<T> P = mk<T>((d0.x * K0 + d1.x * K1 + d2.x * K2) * inv, (d0.y * K0 + d1.y * K1 + d2.y * K2) * inv);
    const cx<T> u0 = U[i], u1 = U[n + i], u2 = U[2 * n + i];
    const T v = nu * k2;
    dU[i] = mk<T>(d0.x - P.x * K0 - v * u0.x, d0.y - P.y * K0 - v * u0.y);
    dU[n + i] = mk<T>(d1.x - P.x * K1 - v * u1.x, d1.y - P.y * K1 - v * u1.y);
    dU[2 * n + i] = mk<T>(d2.x - P.x * K2 - v * u2.x, d2.y - P.y * K2 - v * u2.y);
  }
}

// One Runge-Kutta stage of the demo's time loop (demo/spectral_dns_solver.py:73-77, 91-98) in ONE sweep: N holds the
// nonlinear term (mfft_nonlinear_cross); per element
//     dU = N - K (K . N) / |K|^2 - nu |K|^2 U          pressure projection + viscous term
//     U1 += a_dt dU
//     U   = U0 + b_dt dU                  (stages 0 - 2)     |    U = U0 = U1   (last stage: the next step's starting values)
//     N   = i K x U                       the curl the next stage transforms
// instead of the rhs kernel, two or three axpbz sweeps and the curl kernel over the same arrays (15 field passes -> 7).
template <typename T, bool LAST>
__global__ __launch_bounds__(EW_BLOCK) void rk_stage_kernel(cx<T>* __restrict__ N, cx<T>* __restrict__ U, cx<T>* __restrict__ U0,
                                                           cx<T>* __restrict__ U1, const T* __restrict__ kx,
                                                           const T* __restrict__ ky, const T* __restrict__ kz, int64_t s1,
                                                           int64_t s2, size_t n, T nu, T a_dt, T b_dt) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int64_t k = (int64_t)(i % s2), j = (int64_t)((i / s2) % s1), l = (int64_t)(i / (s2 * s1));
    const T K[3] = {kx[l], ky[j], kz[k]};
    const T k2 = K[0] * K[0] + K[1] * K[1] + K[2] * K[2];
    const T inv = k2 == (T)0 ? (T)1 : (T)1 / k2;
    cx<T> d[3], u[3], w[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { d[c] = N[c * n + i]; u[c] = U[c * n + i]; w[c] = U1[c * n + i]; }
    const cx<T> P = mk<T>((d[0].x * K[0] + d[1].x * K[1] + d[2].x * K[2]) * inv, (d[0].y * K[0] + d[1].y * K[1] + d[2].y * K[2]) * inv);
    const T v = nu * k2;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      d[c] = mk<T>(d[c].x - P.x * K[c] - v * u[c].x, d[c].y - P.y * K[c] - v * u[c].y);
      w[c] = mk<T>(w[c].x + a_dt * d[c].x, w[c].y + a_dt * d[c].y);
      if constexpr (LAST) {
        u[c] = w[c];
        U0[c * n + i] = w[c];
      } else {
        const cx<T> u0 = U0[c * n + i];
        u[c] = mk<T>(u0.x + b_dt * d[c].x, u0.y + b_dt * d[c].y);
      }
      U1[c * n + i] = w[c];
      U[c * n + i] = u[c];
    }
    const cx<T> c0 = mk<T>(K[1] * u[2].x - K[2] * u[1].x, K[1] * u[2].y - K[2] * u[1].y);
    const cx<T> c1 = mk<T>(K[2] * u[0].x - K[0] * u[2].x, K[2] * u[0].y - K[0] * u[2].y);
    const cx<T> c2 = mk<T>(K[0] * u[1].x - K[1] * u[0].x, K[0] * u[1].y - K[1] * u[0].y);
    N[i] = mk<T>(-c0.y, c0.x);
    N[n + i] = mk<T>(-c1.y, c1.x);
    N[2 * n + i] = mk<T>(-c2.y, c2.x);
  }
}

// y = alpha * x + beta * z   (any of the pointers may alias); counts in reals
template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void axpbz_kernel(T* y, const T* x, const T* z, T alpha, T beta, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    y[i] = alpha * x[i] + beta * z[i];
}

template <typename T>
__global__ __launch_bounds__(EW_BLOCK) void sumsq_kernel(const T* __restrict__ x, size_t n, double* out) {
  __shared__ double part[EW_BLOCK / 64];
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += (double)x[i] * (double)x[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < EW_BLOCK / 64; ++w) t += part[w];
    atomicAdd(out, t);
  }
}

// Direct evaluation of a few DFT bins of a distributed field (a checker for meshes too large for a host transform):
//   S[b] = sum over this rank's block of u[x, y, z] * wx_b[x] * wy_b[y] * wz_b[z],  w_b[.] = exp(-/+ 2 pi i k_b g / N)
// with the three phase tables of every bin evaluated on the host (exact integer reduction of k g mod N, long double)
// and the products / sums in double precision whatever the field's.  One pass over the block for up to DFT_MAX_BINS bins.
constexpr int DFT_MAX_BINS = 16;
template <typename T, bool COMPLEX>
__global__ __launch_bounds__(EW_BLOCK) void dft_bins_kernel(const void* __restrict__ u, int64_t s0, int64_t s1, int64_t s2,
                                                           const cx<double>* __restrict__ tab, int nbins, double* out) {
  __shared__ double part[EW_BLOCK / 64][2 * DFT_MAX_BINS];
  double ar[DFT_MAX_BINS], ai[DFT_MAX_BINS];
#pragma unroll
  for (int b = 0; b < DFT_MAX_BINS; ++b) ar[b] = ai[b] = 0;
  const size_t n = (size_t)(s0 * s1 * s2);
  const int64_t per = s0 + s1 + s2;                  // entries of one bin's three tables
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const int64_t z = (int64_t)(i % (size_t)s2), y = (int64_t)((i / (size_t)s2) % (size_t)s1), x = (int64_t)(i / (size_t)(s2 * s1));
    double vr, vi;
    if constexpr (COMPLEX) { const cx<T> v = static_cast<const cx<T>*>(u)[i]; vr = (double)v.x; vi = (double)v.y; }
    else { vr = (double)static_cast<const T*>(u)[i]; vi = 0; }
#pragma unroll
    for (int b = 0; b < DFT_MAX_BINS; ++b) {
      if (b >= nbins) break;
      const cx<double>* t = tab + (size_t)b * per;
      const cx<double> w = t[x] * t[s0 + y] * t[s0 + s1 + z];
      ar[b] += vr * w.x - vi * w.y;
      ai[b] += vr * w.y + vi * w.x;
    }
  }
  for (int b = 0; b < nbins; ++b) {
    double r = ar[b], m = ai[b];
    for (int o = 32; o > 0; o >>= 1) { r += __shfl_down(r, o, 64); m += __shfl_down(m, o, 64); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][2 * b] = r; part[threadIdx.x >> 6][2 * b + 1] = m; }
  }
  __syncthreads();
  if ((int)threadIdx.x < 2 * nbins) {
    double t = 0;
    for (int w = 0; w < EW_BLOCK / 64; ++w) t += part[w][threadIdx.x];
    atomicAdd(out + threadIdx.x, t);
  }
}

}  // namespace

extern "C" {

int mfft_ew_dft_bins(mfft_plan_t plan, const void* u, int is_complex, const int64_t shape[3], const int64_t start[3],
                     const int64_t n[3], int inverse, int nbins, const int64_t* bins, int precision, double* result_host) {
  hipStream_t st = plan_stream(plan);
  if (!u || !shape || !start || !n || !bins || !result_host) return set_error(MFFT_ERR_INVALID, "null argument");
  if (nbins < 1 || nbins > DFT_MAX_BINS) return set_error(MFFT_ERR_INVALID, "1 .. %d bins per call", DFT_MAX_BINS);
  for (int a = 0; a < 3; ++a)
    if (shape[a] < 1 || n[a] < 1 || start[a] < 0 || start[a] + shape[a] > n[a]) return set_error(MFFT_ERR_INVALID, "block outside the mesh");
  const int64_t per = shape[0] + shape[1] + shape[2];
  std::vector<cx<double>> tab((size_t)per * nbins);
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int b = 0; b < nbins; ++b) {
    cx<double>* t = tab.data() + (size_t)b * per;
    for (int a = 0; a < 3; ++a) {
      const int64_t k = ((bins[3 * b + a] % n[a]) + n[a]) % n[a];
      for (int64_t i = 0; i < shape[a]; ++i) {
        const int64_t r = (int64_t)(((__int128)k * (__int128)(start[a] + i)) % (__int128)n[a]);
        const long double ang = two_pi * (long double)r / (long double)n[a];
        t[i] = mk<double>((double)cosl(ang), (double)(inverse ? sinl(ang) : -sinl(ang)));
      }
      t += shape[a];
    }
  }
  cx<double>* dtab = nullptr;
  double* dout = nullptr;
  MFFT_HIP(hipMalloc(reinterpret_cast<void**>(&dtab), tab.size() * sizeof(cx<double>)));
  if (hipMalloc(reinterpret_cast<void**>(&dout), 2 * DFT_MAX_BINS * sizeof(double)) != hipSuccess) {
    (void)hipFree(dtab);
    return set_error(MFFT_ERR_NOMEM, "hipMalloc failed");
  }
  hipError_t e = hipMemcpyAsync(dtab, tab.data(), tab.size() * sizeof(cx<double>), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(dout, 0, 2 * DFT_MAX_BINS * sizeof(double), st);
  if (e == hipSuccess) {
    const size_t cnt = (size_t)(shape[0] * shape[1] * shape[2]);
    const dim3 grid(ew_grid(cnt)), block(EW_BLOCK);
    if (precision == MFFT_DOUBLE) {
      if (is_complex) hipLaunchKernelGGL((dft_bins_kernel<double, true>), grid, block, 0, st, u, shape[0], shape[1], shape[2], dtab, nbins, dout);
      else hipLaunchKernelGGL((dft_bins_kernel<double, false>), grid, block, 0, st, u, shape[0], shape[1], shape[2], dtab, nbins, dout);
    } else {
      if (is_complex) hipLaunchKernelGGL((dft_bins_kernel<float, true>), grid, block, 0, st, u, shape[0], shape[1], shape[2], dtab, nbins, dout);
      else hipLaunchKernelGGL((dft_bins_kernel<float, false>), grid, block, 0, st, u, shape[0], shape[1], shape[2], dtab, nbins, dout);
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(result_host, dout, 2 * (size_t)nbins * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);        // (tab must outlive the copy out of it)
  (void)hipFree(dtab);
  (void)hipFree(dout);
  if (e != hipSuccess) return set_error(MFFT_ERR_HIP, "mfft_ew_dft_bins: %s", hipGetErrorString(e));
  return 0;
}

int mfft_ew_cross(mfft_plan_t plan, const void* a, const void* b, void* out, size_t n, int precision) {
  hipStream_t st = plan_stream(plan);
  if (!a || !b || !out) return set_error(MFFT_ERR_INVALID, "null argument");
  if (precision == MFFT_DOUBLE)
    hipLaunchKernelGGL(cross_kernel<double>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, (const double*)a, (const double*)b, (double*)out, n);
  else
    hipLaunchKernelGGL(cross_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, (const float*)a, (const float*)b, (float*)out, n);
  MFFT_HIP(hipGetLastError());
  return 0;
}

int mfft_ew_curl_hat(mfft_plan_t plan, const void* U_hat, void* out, const void* kx, const void* ky, const void* kz,
                     const int64_t shape[3], int precision) {
  hipStream_t st = plan_stream(plan);
  if (!U_hat || !out || !kx || !ky || !kz || !shape) return set_error(MFFT_ERR_INVALID, "null argument");
  const size_t n = (size_t)(shape[0] * shape[1] * shape[2]);
  if (precision == MFFT_DOUBLE)
    hipLaunchKernelGGL(curl_kernel<double>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, (const cx<double>*)U_hat,
                       (cx<double>*)out, (const double*)kx, (const double*)ky, (const double*)kz, shape[1], shape[2], n);
  else
    hipLaunchKernelGGL(curl_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, (const cx<float>*)U_hat,
                       (cx<float>*)out, (const float*)kx, (const float*)ky, (const float*)kz, shape[1], shape[2], n);
  MFFT_HIP(hipGetLastError());
  return 0;
}

int mfft_ew_ns_rhs(mfft_plan_t plan, void* dU, const void* U_hat, const void* kx, const void* ky, const void* kz,
                   const int64_t shape[3], double nu, int precision) {
  hipStream_t st = plan_stream(plan);
  if (!dU || !U_hat || !kx || !ky || !kz || !shape) return set_error(MFFT_ERR_INVALID, "null argument");
  const size_t n = (size_t)(shape[0] * shape[1] * shape[2]);
  if (precision == MFFT_DOUBLE)
    hipLaunchKernelGGL(rhs_kernel<double>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, (cx<double>*)dU,
                       (const cx<double>*)U_hat, (const double*)kx, (const double*)ky, (const double*)kz, shape[1], shape[2], n, nu);
  else
    hipLaunchKernelGGL(rhs_kernel<float>, dim3(ew_grid(n)), dim3(EW_BLOCK), 0, st, (cx<float>*)dU,
                       (const cx<float>*)U_hat, (const float*)kx, (const float*)ky, (const float*)kz, shape[1], shape[2], n, (float)nu);
  MFFT_HIP(hipGetLastError());
  return 0;
}

int mfft_ew_ns_rk_stage(mfft_plan_t plan, void* N_hat, void* U_hat, void* U_hat0, void* U_hat1, const void* kx, const void* ky,
                        const void* kz, const int64_t shape[3], double nu, double a_dt, double b_dt, int last, int precision) {
  hipStream_t st = plan_stream(plan);
  if (!N_hat || !U_hat || !U_hat0 || !U_hat1 || !kx || !ky || !kz || !shape) return set_error(MFFT_ERR_INVALID, "null argument");
  const size_t n = (size_t)(shape[0] * shape[1] * shape[2]);
  const dim3 grid(ew_grid(n)), block(EW_BLOCK);
  if (precision == MFFT_DOUBLE) {
    typedef cx<double> C;
    if (last) hipLaunchKernelGGL((rk_stage_kernel<double, true>), grid, block, 0, st, (C*)N_hat, (C*)U_hat, (C*)U_hat0, (C*)U_hat1, (const double*)kx, (const double*)ky, (const double*)kz, shape[1], shape[2], n, nu, a_dt, b_dt);
    else hipLaunchKernelGGL((rk_stage_kernel<double, false>), grid, block, 0, st, (C*)N_hat, (C*)U_hat, (C*)U_hat0, (C*)U_hat1, (const double*)kx, (const double*)ky, (const double*)kz, shape[1], shape[2], n, nu, a_dt, b_dt);
  } else {
    typedef cx<float> C;
    if (last) hipLaunchKernelGGL((rk_stage_kernel<float, true>), grid, block, 0, st, (C*)N_hat, (C*)U_hat, (C*)U_hat0, (C*)U_hat1, (const float*)kx, (const float*)ky, (const float*)kz, shape[1], shape[2], n, (float)nu, (float)a_dt, (float)b_dt);
    else hipLaunchKernelGGL((rk_stage_kernel<float, false>), grid, block, 0, st, (C*)N_hat, (C*)U_hat, (C*)U_hat0, (C*)U_hat1, (const float*)kx, (const float*)ky, (const float*)kz, shape[1], shape[2], n, (float)nu, (float)a_dt, (float)b_dt);
  }
  MFFT_HIP(hipGetLastError());
  return 0;
}

int mfft_ew_axpbz(mfft_plan_t plan, void* y, const void* x, const void* z, double alpha, double beta, size_t n_real, int precision) {
  hipStream_t st = plan_stream(plan);
  if (!y || !x || !z) return set_error(MFFT_ERR_INVALID, "null argument");
  if (precision == MFFT_DOUBLE)
    hipLaunchKernelGGL(axpbz_kernel<double>, dim3(ew_grid(n_real)), dim3(EW_BLOCK), 0, st, (double*)y, (const double*)x, (const double*)z, alpha, beta, n_real);
  else
    hipLaunchKernelGGL(axpbz_kernel<float>, dim3(ew_grid(n_real)), dim3(EW_BLOCK), 0, st, (float*)y, (const float*)x, (const float*)z, (float)alpha, (float)beta, n_real);
  MFFT_HIP(hipGetLastError());
  return 0;
}

int mfft_ew_sumsq(mfft_plan_t plan, const void* x, size_t n_real, int precision, double* result_host) {
  hipStream_t st = plan_stream(plan);
  if (!x || !result_host) return set_error(MFFT_ERR_INVALID, "null argument");
  double* d = nullptr;
  MFFT_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(double)));
  MFFT_HIP(hipMemsetAsync(d, 0, sizeof(double), st));
  if (precision == MFFT_DOUBLE)
    hipLaunchKernelGGL(sumsq_kernel<double>, dim3(ew_grid(n_real)), dim3(EW_BLOCK), 0, st, (const double*)x, n_real, d);
  else
    hipLaunchKernelGGL(sumsq_kernel<float>, dim3(ew_grid(n_real)), dim3(EW_BLOCK), 0, st, (const float*)x, n_real, d);
  MFFT_HIP(hipGetLastError());
  MFFT_HIP(hipMemcpyAsync(result_host, d, sizeof(double), hipMemcpyDeviceToHost, st));
  MFFT_HIP(hipStreamSynchronize(st));
  MFFT_HIP(hipFree(d));
  return 0;
}

}  // extern "C"
