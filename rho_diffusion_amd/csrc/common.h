// Shared device helpers for librho_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rho_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef unsigned short bf16_raw;  // storage type of a bfloat16

#define RHO_LAUNCH_CHECK()                    \
    do {                                      \
        hipError_t e_ = hipGetLastError();    \
        if (e_ != hipSuccess) return (int)e_; \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; a plain cast keeps NaNs NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;
    return *reinterpret_cast<bf16_raw*>(&h);
}

// (as a 2-vector cast this is ONE v_cvt_pk_bf16_f32; two scalar casts + shift + or came out as 2 cvt + lshl + or_sdwa per pair)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_pack_t;
    const bf16x2_pack_t h = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, h);
}

// x * sigmoid(x) with the hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: ~1 ulp each) instead of an IEEE
// division sequence: this sits in the conv loaders, where every VALU instruction competes with the MFMA issue.
__device__ __forceinline__ float silu_f(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
// d silu(u) / du
__device__ __forceinline__ float dsilu_f(float u) {
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * u));
    return s * (1.0f + u * (1.0f - s));
}

// Activation codes of the C ABI wherever an `act` / `pre_silu` / `act_in` flag appears: 0 identity, 1 SiLU (each kernel keeps its own
// SiLU expression - the conv loaders and the GroupNorm passes share silu_f / dsilu_f above), and the registry's other elementwise,
// parameter-free activations (rho_diffusion/registry.py:162-170): 2 ReLU, 3 GELU (erf form = nn.GELU() default), 4 Tanh, 5 Sigmoid,
// 6 ELU (alpha = 1).  These run in the HBM-bound passes only (k_gn_apply / k_gn_bwd_*, the embedding linears): a UNetv2 built with one
// of them materialises its activated conv inputs, the conv loaders never see them.
__device__ __forceinline__ float act_other_f(float u, int act) {
    switch (act) {
        case 2: return u > 0.0f ? u : 0.0f;
        case 3: return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
        case 4: return tanhf(u);
        case 5: return 1.0f / (1.0f + expf(-u));
        case 6: return u > 0.0f ? u : expm1f(u);
        default: return u;
    }
}
__device__ __forceinline__ float dact_other_f(float u, int act) {
    switch (act) {
        case 2: return u > 0.0f ? 1.0f : 0.0f;
        case 3: return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
        case 4: { const float t = tanhf(u); return 1.0f - t * t; }
        case 5: { const float s = 1.0f / (1.0f + expf(-u)); return s * (1.0f - s); }
        case 6: return u > 0.0f ? 1.0f : expf(u);
        default: return 1.0f;
    }
}

// ---- counter-based RNG shared by rho_philox_normal and the dropout masks of the GroupNorm passes
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// Dropout (nn.Dropout(p) of ResBlock.out_layers, unet_v2.py:239) as a Philox mask over the activated tensor [N, S, C]: element e keeps
// its value (scaled by 1 / (1 - p)) iff word (e & 3) of Philox(counter = *off_dev + (e >> 2), seed) >= p * 2^32.  The same (seed,
// counter) regenerates the mask in the backward passes: nothing is stored.
struct DropK {
    uint32_t thr;            // p * 2^32 (p < 1)
    float inv_keep;          // 1 / (1 - p)
    uint64_t seed;
    const uint64_t* off_dev; // uint64[1] on the device: advanced by the engine once per training forward
};
// multipliers of the 8 consecutive elements e0 .. e0 + 7 (e0 a multiple of 8)
__device__ __forceinline__ void drop_mult8(const DropK& dk, uint64_t off, int64_t e0, float (&m)[8]) {
    uint32_t r0[4], r1[4];
    philox4x32_10(off + ((uint64_t)e0 >> 2), dk.seed, r0);
    philox4x32_10(off + ((uint64_t)e0 >> 2) + 1, dk.seed, r1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = r0[j] >= dk.thr ? dk.inv_keep : 0.0f;
        m[4 + j] = r1[j] >= dk.thr ? dk.inv_keep : 0.0f;
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2).  The tiles of one (batch, head) pair all stream the same
// K / V (or Q / dO) - 2 MB at T = 4096, ch = 128: with the plain grid its 32 tiles sat on all 8 XCDs, every L2 held 8 pairs' worth
// and thrashed.  Launch slot k = L / 8 of XCD e = L % 8 takes tile k % tiles of pair (k / tiles) * 8 + e, so one XCD's 32 CUs walk one
// pair at a time (needs heads * batch to be a multiple of 8; else the plain grid).
__device__ __forceinline__ void attn_block(int& tile, int& h, int& b) {
    tile = blockIdx.x; h = blockIdx.y; b = blockIdx.z;
    const int pairs = gridDim.y * gridDim.z;
    if ((pairs & 7) == 0) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int e = L & 7, k = L >> 3;
        tile = k % (int)gridDim.x;
        const int pr = (k / (int)gridDim.x) * 8 + e;
        h = pr % (int)gridDim.y;
        b = pr / (int)gridDim.y;
    }
}

