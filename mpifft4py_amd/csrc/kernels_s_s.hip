// gfx950 instantiations: plan group S (35 * 2^a), single precision only
#define MFFT_TU_PLANS MFFT_PLANS_S
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_S
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
