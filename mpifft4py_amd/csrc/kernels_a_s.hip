// gfx950 instantiations: plan group A, float precision
#define MFFT_TU_PLANS MFFT_PLANS_A
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_A
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
