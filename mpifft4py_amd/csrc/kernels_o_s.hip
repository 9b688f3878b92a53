// gfx950 instantiations: plan group O, single precision
#define MFFT_TU_PLANS MFFT_PLANS_O
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_O
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
