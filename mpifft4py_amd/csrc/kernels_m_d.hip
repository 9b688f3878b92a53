// gfx950 instantiations: plan group M, double precision
#define MFFT_TU_PLANS MFFT_PLANS_M
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_M
#define MFFT_TU_COLPLANS MFFT_COLPLANS_F64_M
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
