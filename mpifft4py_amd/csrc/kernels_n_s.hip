// gfx950 instantiations: plan group N, single precision
#define MFFT_TU_PLANS MFFT_PLANS_N
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_N
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
