// gfx950 instantiations: plan group R, single precision
#define MFFT_TU_PLANS MFFT_PLANS_R
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_R
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
