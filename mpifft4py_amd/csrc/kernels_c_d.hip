// gfx950 instantiations: plan group C, double precision
#define MFFT_TU_PLANS MFFT_PLANS_C
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_C
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
