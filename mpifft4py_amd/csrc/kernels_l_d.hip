// gfx950 instantiations: plan group L, double precision
#define MFFT_TU_PLANS MFFT_PLANS_L
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_L
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
