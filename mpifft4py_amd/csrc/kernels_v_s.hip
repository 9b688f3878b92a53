// gfx950 instantiations: plan group V (81 * 2^a), single precision
#define MFFT_TU_PLANS MFFT_PLANS_V
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_V
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
