// gfx950 instantiations: plan group H, single precision
#define MFFT_TU_PLANS MFFT_PLANS_H
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_H
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
