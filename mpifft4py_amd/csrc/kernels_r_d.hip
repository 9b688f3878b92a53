// gfx950 instantiations: plan group R, double precision
#define MFFT_TU_PLANS MFFT_PLANS_R
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_R
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
