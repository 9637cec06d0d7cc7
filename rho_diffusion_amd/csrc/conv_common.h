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

// bf16: two channels per operation with the packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, full
// rate on CDNA3/4); the same operation sequence per element as the scalar form (fma; then x * rcp(1 + exp2(-log2e * x)) as
// silu_f), so results are bit-identical to k_gn_apply.  44 instead of 60 VALU instructions per 16-byte piece: the loader's
// prologue competes with the MFMA issue of the other wave on the SIMD.
template <>
__device__ __forceinline__ uint4 apply_pre<bf16_raw>(uint4 v, const float* __restrict__ a, const float* __restrict__ b, int silu) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
    const f32x2_t av[4] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}};
    const f32x2_t bv[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t r[4];
    // one wave-uniform branch around the whole piece: with the test inside the element loop the compiler speculates the sigmoid
    // and selects (8 v_cndmask per piece on top of it)
    if (silu) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2_t x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xFFFF0000u)};
            f32x2_t y = __builtin_elementwise_fma(av[j], x, bv[j]);
            const f32x2_t t = y * -1.4426950408889634f;
            const f32x2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            const f32x2_t d = e + 1.0f;                      // 1 + exp2(.) == exp2(.) + 1 (commutative, one rounding)
            const f32x2_t q = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
            y = y * q;
            const bf16x2_t h = {(__bf16)y.x, (__bf16)y.y};
            r[j] = __builtin_bit_cast(uint32_t, h);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x2_t x = {__uint_as_float(w[j] << 16), __uint_as_float(w[j] & 0xFFFF0000u)};
            const f32x2_t y = __builtin_elementwise_fma(av[j], x, bv[j]);
            const bf16x2_t h = {(__bf16)y.x, (__bf16)y.y};
            r[j] = __builtin_bit_cast(uint32_t, h);
        }
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
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
#ifdef RHO_PROBE_M16   /* TIMING-ONLY probe build (wrong numerics): the same FLOPs as two 16x16x32 MFMAs */
    f32x4_t v0 = {acc[0], acc[1], acc[2], acc[3]}, v1 = {acc[4], acc[5], acc[6], acc[7]};
    v0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), v0, 0, 0, 0);
    v1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), v1, 0, 0, 0);
    acc[0] = v0[0]; acc[1] = v0[1]; acc[2] = v0[2]; acc[3] = v0[3];
    acc[4] = v1[0]; acc[5] = v1[1]; acc[6] = v1[2]; acc[7] = v1[3];
#else
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
#endif
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
