// gfx950 instantiations: plan group E, float precision
#define MFFT_TU_PLANS MFFT_PLANS_E
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_E
#define MFFT_TU_COL3PLANS MFFT_COL3PLANS_E
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
