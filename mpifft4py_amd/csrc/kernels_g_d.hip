// gfx950 instantiations: plan group G, double precision
#define MFFT_TU_PLANS MFFT_PLANS_G
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_G
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
