// gfx950 instantiations: plan group J, double precision
#define MFFT_TU_PLANS MFFT_PLANS_J
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_J
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
