// gfx950 instantiations: plan group B, float precision
#define MFFT_TU_PLANS MFFT_PLANS_B
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_B
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
