/* roundtrip.c -- the C ABI used from plain C (no Python, no C++): what a binding in any host language sees.
 *
 *   gcc -O2 -I include tests/cabi/roundtrip.c -o /tmp/roundtrip -L mpifft4py_amd -lmpifft4py_amd -lm \
 *       -Wl,-rpath,$PWD/mpifft4py_amd && /tmp/roundtrip 64 48 32
 *
 * Creates a one-rank slab R2C plan (mpiFFT4py slab.R2C, slab.py:67-96), fills a real field with a known
 * superposition of plane waves, runs fftn / ifftn (slab.py:349-443 / 214-308) and checks
 *   - the spectrum against the analytic coefficients of the plane waves (numpy's unnormalised convention),
 *   - the round trip against the input,
 *   - that the input array was not modified,
 *   - the error path (unsupported length, beyond 2^20 -> negative status + message).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "mpifft4py_amd.h"

#define CK(call)                                                                 \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ < 0) {                                                               \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mfft_last_error());          \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char** argv) {
  const int64_t n0 = argc > 1 ? atoll(argv[1]) : 32, n1 = argc > 2 ? atoll(argv[2]) : 48, n2 = argc > 3 ? atoll(argv[3]) : 64;
  const int64_t nf = n2 / 2 + 1;
  const double pi = 3.14159265358979323846;
  int ndev = 0;
  CK(mfft_device_count(&ndev));
  if (ndev < 1) { fprintf(stderr, "no GPU\n"); return 2; }
  CK(mfft_set_device(0));

  mfft_comm_t comm = NULL;
  CK(mfft_comm_create_self(&comm));
  mfft_plan_desc d;
  memset(&d, 0, sizeof d);
  d.n[0] = n0; d.n[1] = n1; d.n[2] = n2;
  d.precision = MFFT_DOUBLE; d.kind = MFFT_R2C; d.decomp = MFFT_SLAB; d.padsize = 1.5;
  mfft_plan_t plan = NULL;
  CK(mfft_plan_create(comm, &d, &plan));
  int64_t rs[3], cs[3], r0[3], c0[3], rp[3], grid[2], sub[2];
  CK(mfft_plan_layout(plan, rs, cs, r0, c0, rp, grid, sub));
  if (rs[0] != n0 || rs[1] != n1 || rs[2] != n2 || cs[0] != n0 || cs[1] != n1 || cs[2] != nf) {
    fprintf(stderr, "unexpected layout\n");
    return 1;
  }

  /* u = 1.5 + 2 cos(2 pi (3x/n0 + 2y/n1 + 5z/n2)) + sin(2 pi (x/n0 - 4z/n2 ... )) kept simple: two waves */
  const int64_t kx = 3 % n0, ky = 2 % n1, kz = 5 % nf;
  const size_t nr = (size_t)(n0 * n1 * n2), nc = (size_t)(n0 * n1 * nf);
  double* u = malloc(nr * sizeof(double));
  double* back = malloc(nr * sizeof(double));
  double* fu = malloc(nc * 2 * sizeof(double));
  for (int64_t i = 0; i < n0; ++i)
    for (int64_t j = 0; j < n1; ++j)
      for (int64_t k = 0; k < n2; ++k)
        u[(i * n1 + j) * n2 + k] = 1.5 + 2.0 * cos(2 * pi * ((double)(kx * i) / n0 + (double)(ky * j) / n1 + (double)(kz * k) / n2));

  void *du = NULL, *dfu = NULL, *dback = NULL;
  CK(mfft_malloc(&du, nr * sizeof(double)));
  CK(mfft_malloc(&dback, nr * sizeof(double)));
  CK(mfft_malloc(&dfu, nc * 2 * sizeof(double)));
  CK(mfft_memcpy_h2d(du, u, nr * sizeof(double)));
  CK(mfft_forward(plan, du, dfu, MFFT_DEALIAS_NONE));
  CK(mfft_backward(plan, dfu, dback, MFFT_DEALIAS_NONE));
  CK(mfft_plan_sync(plan));
  CK(mfft_memcpy_d2h(fu, dfu, nc * 2 * sizeof(double)));
  CK(mfft_memcpy_d2h(back, dback, nr * sizeof(double)));

  /* analytic spectrum: N*1.5 at k = 0; N at (kx, ky, kz) [and its mirror, which lies outside the stored half unless
   * kz == 0 or kz == n2/2]; zero elsewhere */
  const double ntot = (double)nr;
  double worst = 0.0;
  for (int64_t i = 0; i < n0; ++i)
    for (int64_t j = 0; j < n1; ++j)
      for (int64_t k = 0; k < nf; ++k) {
        double er = 0.0, ei = 0.0;
        if (i == 0 && j == 0 && k == 0) er += 1.5 * ntot;
        if (i == kx && j == ky && k == kz) er += ntot;
        if ((kz == 0 || 2 * kz == n2) && i == (n0 - kx) % n0 && j == (n1 - ky) % n1 && k == kz) er += ntot;
        const double gr = fu[2 * ((i * n1 + j) * nf + k)], gi = fu[2 * ((i * n1 + j) * nf + k) + 1];
        const double e = fabs(gr - er) + fabs(gi - ei);
        if (e > worst) worst = e;
      }
  double rt = 0.0;
  for (size_t i = 0; i < nr; ++i) {
    const double e = fabs(back[i] - u[i]);
    if (e > rt) rt = e;
  }
  double* u_after = malloc(nr * sizeof(double));
  CK(mfft_memcpy_d2h(u_after, du, nr * sizeof(double)));
  const int untouched = memcmp(u_after, u, nr * sizeof(double)) == 0;
  printf("mesh %lld x %lld x %lld: max |spectrum - analytic| / N = %.3e, max |roundtrip - u| = %.3e, input untouched: %s\n",
         (long long)n0, (long long)n1, (long long)n2, worst / ntot, rt, untouched ? "yes" : "NO");

  /* error path: a length beyond the supported range (1 ... 2^20, mfft_length_route) must fail with a message, not crash */
  mfft_plan_desc bad = d;
  bad.n[0] = (1 << 20) + 2;
  mfft_plan_t p2 = NULL;
  const int rc = mfft_plan_create(comm, &bad, &p2);
  const int err_ok = rc < 0 && strlen(mfft_last_error()) > 0 && p2 == NULL;
  printf("unsupported mesh -> status %d (%s)\n", rc, mfft_last_error());

  CK(mfft_free(du)); CK(mfft_free(dfu)); CK(mfft_free(dback));
  CK(mfft_plan_destroy(plan));
  CK(mfft_comm_destroy(comm));
  free(u); free(back); free(fu); free(u_after);
  const int ok = worst / ntot < 1e-12 && rt < 1e-12 && untouched && err_ok;
  printf("%s\n", ok ? "C_ABI_ROUNDTRIP_OK" : "C_ABI_ROUNDTRIP_FAILED");
  return ok ? 0 : 1;
}
