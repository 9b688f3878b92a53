// gfx950 instantiations: plan group D, float precision
#define MFFT_TU_PLANS MFFT_PLANS_D
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_D
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
