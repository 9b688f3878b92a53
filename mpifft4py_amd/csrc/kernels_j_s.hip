// gfx950 instantiations: plan group J, single precision
#define MFFT_TU_PLANS MFFT_PLANS_J
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_J
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
