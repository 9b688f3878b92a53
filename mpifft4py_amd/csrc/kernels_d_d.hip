// gfx950 instantiations: plan group D, double precision
#define MFFT_TU_PLANS MFFT_PLANS_D
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_D
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
