// gfx950 instantiations: plan group I, double precision
#define MFFT_TU_PLANS MFFT_PLANS_I
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_I
#define MFFT_TU_COLPLANS MFFT_COLPLANS_F64_I
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
