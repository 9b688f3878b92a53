// gfx950 instantiations: plan group B, double precision
#define MFFT_TU_PLANS MFFT_PLANS_B
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_B
#define MFFT_TU_COLPLANS MFFT_COLPLANS_F64_B
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
