// gfx950 instantiations: plan group Q, single precision
#define MFFT_TU_PLANS MFFT_PLANS_Q
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_Q
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
