// gfx950 instantiations: plan group M, float precision
#define MFFT_TU_PLANS MFFT_PLANS_M
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_M
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
