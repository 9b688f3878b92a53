// gfx950 instantiations: plan group N, double precision
#define MFFT_TU_PLANS MFFT_PLANS_N
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_N
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
