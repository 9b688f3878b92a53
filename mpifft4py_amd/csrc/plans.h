// plans.h -- the compile-time radix plans (length, radix sequence) that get
// instantiated as gfx950 kernels.  First pass first; the first pass needs no
// twiddles, so the largest radix goes there.  E = lcm(radices) values per
// thread, TPT = N/E threads per transform.
//
// Lengths covered: 2^a (2..4096), 3*2^a (6..3072), 5*2^a (10..2560): these are
// what the reference's power-of-two meshes and their 3/2-rule padded
// counterparts (slab.py:75-76, 487-489) produce.  The groups only exist so the
// instantiations can be compiled in parallel translation units.
#pragma once

#define MFFT_PLANS_A(X) X(2, 2) X(4, 4) X(8, 8) X(16, 16) X(32, 8, 4) X(64, 8, 8) X(128, 16, 8) X(256, 16, 16)
#define MFFT_PLANS_B(X) X(512, 8, 8, 8) X(1024, 16, 8, 8)
#define MFFT_PLANS_C(X) X(2048, 16, 16, 8) X(4096, 16, 16, 16)
#define MFFT_PLANS_D(X) X(6, 3, 2) X(12, 4, 3) X(24, 8, 3) X(48, 4, 4, 3) X(96, 8, 4, 3) X(192, 8, 8, 3)
#define MFFT_PLANS_E(X) X(384, 8, 8, 3, 2) X(768, 8, 8, 4, 3) X(1536, 8, 8, 8, 3) X(3072, 8, 8, 4, 4, 3)
#define MFFT_PLANS_F(X) X(10, 5, 2) X(20, 5, 4) X(40, 5, 4, 2) X(80, 5, 4, 4) X(160, 8, 4, 5)
#define MFFT_PLANS_G(X) X(320, 8, 8, 5) X(640, 8, 4, 4, 5) X(1280, 8, 8, 4, 5) X(2560, 8, 8, 8, 5)

#define MFFT_FOR_EACH_PLAN(X) MFFT_PLANS_A(X) MFFT_PLANS_B(X) MFFT_PLANS_C(X) MFFT_PLANS_D(X) MFFT_PLANS_E(X) MFFT_PLANS_F(X) MFFT_PLANS_G(X)
