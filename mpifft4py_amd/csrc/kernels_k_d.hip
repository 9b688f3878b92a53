// gfx950 instantiations: plan group K, double precision
#define MFFT_TU_PLANS MFFT_PLANS_K
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_F64_K
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
