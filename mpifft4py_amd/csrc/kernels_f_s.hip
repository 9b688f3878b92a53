// gfx950 instantiations: plan group F, float precision
#define MFFT_TU_PLANS MFFT_PLANS_F
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_F
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
