// gfx950 instantiations: plan group H, double precision
#define MFFT_TU_PLANS MFFT_PLANS_H
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_H
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
