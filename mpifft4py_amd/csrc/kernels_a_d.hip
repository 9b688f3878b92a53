// gfx950 instantiations: plan group A, double precision
#define MFFT_TU_PLANS MFFT_PLANS_A
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_A
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
