// Pieces shared by the forward/dgrad convolution (conv.hip) and the weight-gradient kernel (wgrad.hip).
#pragma once
#include "common.h"

#define PITCH 80  // bytes per LDS row (64 payload + 16 pad)

template <typename T>
struct ET;
template <>
struct ET<bf16_raw> {
    static constexpr int CK = 32;  // channels per 64-byte chunk
    static constexpr int PE = 8;   // elements per 16-byte piece
};
template <>
struct ET<float> {
    static constexpr int CK = 16;
    static constexpr int PE = 4;
};

// y = act(a*x+b) on one 16-byte piece
template <typename T>
__device__ __forceinline__ uint4 apply_pre(uint4 v, const float* __restrict__ a, const float* __restrict__ b, int silu);

template <>
__device__ __forceinline__ uint4 apply_pre<bf16_raw>(uint4 v, const float* __restrict__ a, const float* __restrict__ b, int silu) {
    const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
    float f[8];
    f[0] = fmaf(a0.x, __uint_as_float(v.x << 16), b0.x);
    f[1] = fmaf(a0.y, __uint_as_float(v.x & 0xFFFF0000u), b0.y);
    f[2] = fmaf(a0.z, __uint_as_float(v.y << 16), b0.z);
    f[3] = fmaf(a0.w, __uint_as_float(v.y & 0xFFFF0000u), b0.w);
    f[4] = fmaf(a1.x, __uint_as_float(v.z << 16), b1.x);
    f[5] = fmaf(a1.y, __uint_as_float(v.z & 0xFFFF0000u), b1.y);
    f[6] = fmaf(a1.z, __uint_as_float(v.w << 16), b1.z);
    f[7] = fmaf(a1.w, __uint_as_float(v.w & 0xFFFF0000u), b1.w);
    if (silu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
    }
    uint4 r;
    r.x = pack_bf16x2(f[0], f[1]);
    r.y = pack_bf16x2(f[2], f[3]);
    r.z = pack_bf16x2(f[4], f[5]);
    r.w = pack_bf16x2(f[6], f[7]);
    return r;
}

template <>
__device__ __forceinline__ uint4 apply_pre<float>(uint4 v, const float* __restrict__ a, const float* __restrict__ b, int silu) {
    const float4 a0 = *reinterpret_cast<const float4*>(a);
    const float4 b0 = *reinterpret_cast<const float4*>(b);
    float f[4];
    f[0] = fmaf(a0.x, __uint_as_float(v.x), b0.x);
    f[1] = fmaf(a0.y, __uint_as_float(v.y), b0.y);
    f[2] = fmaf(a0.z, __uint_as_float(v.z), b0.z);
    f[3] = fmaf(a0.w, __uint_as_float(v.w), b0.w);
    if (silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = silu_f(f[j]);
    }
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}

template <typename T>
__device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x16_t& acc);

template <>
__device__ __forceinline__ void mma_step<bf16_raw>(const uint4& a, const uint4& b, f32x16_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(const uint4& a, const uint4& b, f32x16_t& acc) {
    // lanes 0-31 carry channels {0,1,2,3} of the 8-channel group, lanes 32-63 channels {4,5,6,7};
    // MFMA #q contracts the channel pair (q, 4+q): any K permutation is valid as A and B agree.
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}


// ------------------------------------------------------------------------------------------ host
namespace rho_conv {

struct TileChoice {
    int TD, TH, TW, ID, IH, IW, NP;
    long long tiles;
    bool ok;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// pick the 256-position output tile: fewest tiles first, then the smallest halo, then the widest W
inline TileChoice choose_tile(const rho_conv_desc& d, int Dm, int Do, int Ho, int Wo, int np_cap) {
    TileChoice best{};
    best.ok = false;
    double best_cost = 1e300;
    for (int TD = 1; TD <= 256; TD *= 2)
        for (int TH = 1; TH * TD <= 256; TH *= 2) {
            const int TW = 256 / (TD * TH);
            if (d.up_h && TH < 2) continue;
            if (d.up_w && TW < 2) continue;
            const int ID = TD + (d.kd - 1);
            const int IH = d.up_h ? TH / 2 + 2 : (TH - 1) * d.sh + d.kh;
            const int IW = d.up_w ? TW / 2 + 2 : (TW - 1) * d.sw + d.kw;
            const int NP = ID * IH * IW;
            if (NP > np_cap) continue;
            const long long tiles = (long long)cdiv(Do, TD) * cdiv(Ho, TH) * cdiv(Wo, TW);
            // cost model: per tile, staging ~ NP rows and taps*256 MFMA columns
            const double cost = (double)tiles * (NP * 1.5 + 256.0 * d.kd * d.kh * d.kw) - 1e-3 * TW;
            if (cost < best_cost) {
                best_cost = cost;
                best = TileChoice{TD, TH, TW, ID, IH, IW, NP, tiles, true};
            }
        }
    (void)Dm;
    return best;
}


}  // namespace rho_conv
