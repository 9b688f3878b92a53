// gfx950 instantiations: plan group G, float precision
#define MFFT_TU_PLANS MFFT_PLANS_G
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_G
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
