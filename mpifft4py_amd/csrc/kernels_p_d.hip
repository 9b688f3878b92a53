// gfx950 instantiations: plan group P, double precision
#define MFFT_TU_PLANS MFFT_PLANS_P
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_P
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
