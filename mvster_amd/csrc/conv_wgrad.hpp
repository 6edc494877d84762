// Argument record shared by the weight-gradient kernels (conv_wgrad.hip, conv_wgrad_pers.hip).
#pragma once
#include "common.hpp"

namespace mvwgrad {

struct WgradArgs {
    const float* x;     // [B, Di, Hi, Wi, CI]
    const float* gy;    // [B, Do, Ho, Wo, CO]
    float* partial;     // [nblk, taps, COT*16, CIT*16]
    int B, Di, Hi, Wi, CI;
    int Do, Ho, Wo, CO;
    int kd, kh, kw, sd, sh, sw, pd, ph, pw;
};

// conv_wgrad_pers.hip: persistent LDS-DMA form for the (1|3)x3x3 stride-1 layers with 16 / 32 / 64 channels on both sides;
// MVSTER_ERR_UNSUPPORTED for everything else (the caller falls through to the other kernels)
int try_wgrad_pers(const WgradArgs& a, int nblk, int cot, int cit, hipStream_t s);
// slots of `partial` that kernel fills by itself (the caller may allocate exactly that many), 0 if it does not apply
int wgrad_pers_slots(const WgradArgs& a, int cot, int cit);

}  // namespace mvwgrad
