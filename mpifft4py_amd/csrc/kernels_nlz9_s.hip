// gfx950 instantiations: fused nonlinear z stage (fft_nlz.h), float precision
#include "registry_nlz.h"
#include "plans.h"
namespace {
#define MFFT_REG_NLZ(N, ...) mfft::register_nlz<mfft::Spec<N, __VA_ARGS__>, float>("nlz n" #N "(" #__VA_ARGS__ ")float");
#define MFFT_REG_NLZ3(N, ...) mfft::register_nlz3<mfft::Spec<N, __VA_ARGS__>, float>("nlz3 l" #N "(" #__VA_ARGS__ ")float");
mfft::PlanRegistrar registrar([] { MFFT_NLZPLANS_9(MFFT_REG_NLZ) });
}
