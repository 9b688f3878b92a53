// gfx950 instantiations: plan group C, float precision
#define MFFT_TU_PLANS MFFT_PLANS_C
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_C
#define MFFT_TU_COLPLANS MFFT_COLPLANS_F32_C
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
