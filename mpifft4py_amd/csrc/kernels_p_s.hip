// gfx950 instantiations: plan group P, single precision
#define MFFT_TU_PLANS MFFT_PLANS_P
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_P
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
