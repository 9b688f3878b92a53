// gfx950 instantiations: plan group U (135 * 2^a and the images of 900 / 1500 / 1800), single precision
#define MFFT_TU_PLANS MFFT_PLANS_U
#define MFFT_TU_ROWPLANS MFFT_ROWPLANS_U
#define MFFT_TU_REAL float
#include "kernels_tu.inc"
