// gfx950 instantiations: plan group T (27 * 2^a), single precision
#define MFFT_TU_PLANS MFFT_PLANS_T
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_T
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
