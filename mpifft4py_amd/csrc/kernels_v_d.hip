// gfx950 instantiations: plan group V (81 * 2^a), double precision
#define MFFT_TU_PLANS MFFT_PLANS_V
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_V
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
