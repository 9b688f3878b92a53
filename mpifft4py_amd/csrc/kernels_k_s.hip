// gfx950 instantiations: plan group K, single precision
#define MFFT_TU_PLANS MFFT_PLANS_K
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_K
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
