// gfx950 instantiations: fused nonlinear z stage (fft_nlz.h), double precision
#include "registry_nlz.h"
#include "plans.h"
namespace {
#define MFFT_REG_NLZ(N, ...) mfft::register_nlz<mfft::Spec<N, __VA_ARGS__>, double>("nlz n" #N "(" #__VA_ARGS__ ")double");
#define MFFT_REG_NLZ3(N, ...) mfft::register_nlz3<mfft::Spec<N, __VA_ARGS__>, double>("nlz3 l" #N "(" #__VA_ARGS__ ")double");
mfft::PlanRegistrar registrar([] { MFFT_NLZPLANS_P2(MFFT_REG_NLZ) MFFT_NLZPLANS_3(MFFT_REG_NLZ) MFFT_NLZ3PLANS(MFFT_REG_NLZ3) });
#ifdef MFFT_NLZ_EXPERIMENTS
mfft::PlanRegistrar registrar4([] {
  typedef mfft::Spec<256, 4, 4, 4, 4> A256;
  typedef mfft::Spec<256, 8, 8, 4> B256;
  typedef mfft::Spec<512, 4, 4, 4, 4, 2> A512;
  typedef mfft::Spec<512, 8, 8, 8> B512;
  mfft::register_nlz3_var<A256, double, 1, 0, 11>("nlz3 v11");
  mfft::register_nlz3_var<A256, double, 1, 4, 12>("nlz3 v12");
  mfft::register_nlz3_var<A256, double, 2, 4, 13>("nlz3 v13");
  mfft::register_nlz3_var<B256, double, 2, 3, 14>("nlz3 v14");
  mfft::register_nlz3_var<B256, double, 2, 2, 15>("nlz3 v15");
  mfft::register_nlz3_var<B256, double, 4, 3, 16>("nlz3 v16");
  mfft::register_nlz3_var<A512, double, 1, 0, 11>("nlz3 v11");
  mfft::register_nlz3_var<A512, double, 1, 4, 12>("nlz3 v12");
  mfft::register_nlz3_var<B512, double, 1, 3, 14>("nlz3 v14");
  mfft::register_nlz3_var<B512, double, 1, 2, 15>("nlz3 v15");
  mfft::register_nlz3_var<B512, double, 2, 3, 16>("nlz3 v16");
  mfft::register_nlz_var<mfft::Spec<1024, 8, 8, 4, 4>, double, 2, true, 3, 21>("nlz v21");
  mfft::register_nlz_var<mfft::Spec<512, 8, 8, 8>, double, 4, true, 3, 21>("nlz v21");
  mfft::register_nlz_var<mfft::Spec<1024, 8, 8, 4, 4>, double, 1, true, 3, 22>("nlz v22");
  mfft::register_nlz_var<mfft::Spec<512, 8, 8, 8>, double, 2, true, 3, 22>("nlz v22");
});
#endif
}
