// gfx950 instantiations: plan group W (63 * 2^a), double precision
#define MFFT_TU_PLANS MFFT_PLANS_W
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_W
#define MFFT_TU_REAL double
#include "kernels_tu.inc"
