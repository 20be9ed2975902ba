// ng_common.h — the float32 net gradient of picasso/localize.py:202-244, 279-286 in pieces shared by the packed scan's
// exact stage (identify_fast.hip) and the start-value kernel that takes that stage over on the fused path (gaussmle_g8.hip).
#pragma once
#include "pmi_common.h"

namespace pmi {

// float32 ops that must round exactly like the reference's unfused arithmetic: the pragma inside each body keeps the
// instruction free of the `contract` flag whatever the including file's setting is (the __f*_rn helpers of the HIP
// headers are compiled with contraction allowed).
static __device__ __forceinline__ float mul_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
static __device__ __forceinline__ float add_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a + b;
}
static __device__ __forceinline__ float sub_rn(float a, float b)
{
#pragma clang fp contract(off)
    return a - b;
}

// float32 sqrt of 0..128 (the squared lengths that occur for box <= 17), correctly rounded; the host
// checks them against sqrtf before the first launch (unit_vectors_match).
constexpr float SQRT_F32[129] = {
    0x0.0p+0f, 0x1.0p+0f, 0x1.6a09e6p+0f, 0x1.bb67aep+0f, 0x1.0p+1f, 0x1.1e377ap+1f, 0x1.3988e2p+1f,
    0x1.52a7fap+1f, 0x1.6a09e6p+1f, 0x1.8p+1f, 0x1.94c584p+1f, 0x1.a8872ap+1f, 0x1.bb67aep+1f, 0x1.cd82b4p+1f,
    0x1.deeea2p+1f, 0x1.efbdecp+1f, 0x1.0p+2f, 0x1.07e0f6p+2f, 0x1.0f876cp+2f, 0x1.16f834p+2f, 0x1.1e377ap+2f,
    0x1.2548ecp+2f, 0x1.2c2fc6p+2f, 0x1.32eee8p+2f, 0x1.3988e2p+2f, 0x1.4p+2f, 0x1.465656p+2f, 0x1.4c8dc2p+2f,
    0x1.52a7fap+2f, 0x1.58a68ap+2f, 0x1.5e8adep+2f, 0x1.64564p+2f, 0x1.6a09e6p+2f, 0x1.6fa6eap+2f, 0x1.752e5p+2f,
    0x1.7aa10ep+2f, 0x1.8p+2f, 0x1.854bfcp+2f, 0x1.8a85c2p+2f, 0x1.8fae0cp+2f, 0x1.94c584p+2f, 0x1.99cccap+2f,
    0x1.9ec474p+2f, 0x1.a3ad12p+2f, 0x1.a8872ap+2f, 0x1.ad5336p+2f, 0x1.b211b2p+2f, 0x1.b6c30cp+2f, 0x1.bb67aep+2f,
    0x1.cp+2f, 0x1.c48c6p+2f, 0x1.c90d2ap+2f, 0x1.cd82b4p+2f, 0x1.d1ed52p+2f, 0x1.d64d52p+2f, 0x1.daa2fep+2f,
    0x1.deeea2p+2f, 0x1.e3307cp+2f, 0x1.e768d4p+2f, 0x1.eb97e4p+2f, 0x1.efbdecp+2f, 0x1.f3db22p+2f, 0x1.f7efbep+2f,
    0x1.fbfbf8p+2f, 0x1.0p+3f, 0x1.01fe04p+3f, 0x1.03f82p+3f, 0x1.05ee68p+3f, 0x1.07e0f6p+3f, 0x1.09cfdcp+3f,
    0x1.0bbb3p+3f, 0x1.0da304p+3f, 0x1.0f876cp+3f, 0x1.11687ap+3f, 0x1.13464p+3f, 0x1.1520cep+3f, 0x1.16f834p+3f,
    0x1.18cc82p+3f, 0x1.1a9dc8p+3f, 0x1.1c6c16p+3f, 0x1.1e377ap+3f, 0x1.2p+3f, 0x1.21c5b8p+3f, 0x1.2388acp+3f,
    0x1.2548ecp+3f, 0x1.270682p+3f, 0x1.28c17cp+3f, 0x1.2a79e4p+3f, 0x1.2c2fc6p+3f, 0x1.2de32cp+3f, 0x1.2f9422p+3f,
    0x1.3142b4p+3f, 0x1.32eee8p+3f, 0x1.3498cap+3f, 0x1.364064p+3f, 0x1.37e5bep+3f, 0x1.3988e2p+3f, 0x1.3b29d8p+3f,
    0x1.3cc8aap+3f, 0x1.3e655ep+3f, 0x1.4p+3f, 0x1.419894p+3f, 0x1.432f24p+3f, 0x1.44c3b8p+3f, 0x1.465656p+3f,
    0x1.47e706p+3f, 0x1.4975cep+3f, 0x1.4b02b4p+3f, 0x1.4c8dc2p+3f, 0x1.4e16fep+3f, 0x1.4f9e6cp+3f, 0x1.512414p+3f,
    0x1.52a7fap+3f, 0x1.542a28p+3f, 0x1.55aaap+3f, 0x1.57296ap+3f, 0x1.58a68ap+3f, 0x1.5a2208p+3f, 0x1.5b9be6p+3f,
    0x1.5d142cp+3f, 0x1.5e8adep+3f, 0x1.6p+3f, 0x1.617398p+3f, 0x1.62e5acp+3f, 0x1.64564p+3f, 0x1.65c558p+3f,
    0x1.6732f8p+3f, 0x1.689f26p+3f, 0x1.6a09e6p+3f};
// unit vectors of picasso/localize.py:279-286 as compile-time float32 constants:
// ux[k][l] = (H - l) / |(H - l, H - k)|, uy[k][l] = (H - k) / |...|  (float32 sqrt and divide)
template <int H> constexpr float unit_x(int k, int l)
{
    const int vx = H - l, vy = H - k;
    return (vx == 0 && vy == 0) ? 0.0f : (float)vx / SQRT_F32[vx * vx + vy * vy];
}
template <int H> constexpr float unit_y(int k, int l) { return unit_x<H>(l, k); }


}  // namespace pmi
