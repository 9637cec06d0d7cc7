// Backward of the UNet self-attention (autograd of QKVAttention*, unet_v2.py:365-436), flash-style:
// P is recomputed from Q, K and the forward's per-query log-sum-exp, the [T, T] matrices never exist.
// The reference always recomputes attention in backward too (checkpoint(..., True), unet_v2.py:334).
//
//   delta[q]   = sum_c dO[q,c] * O[q,c]
//   P[q,k]     = exp2(c * q.k - lse[q]),        dP[q,k] = dO[q,:] . V[k,:]
//   dS[q,k]    = P * (dP - delta[q]) * ch^-0.5
//   dQ = dS K,   dK = dS^T Q,   dV = P^T dO
//
// Two MFMA kernels instead of one with atomics (deterministic, no cross-workgroup traffic):
//   k_attn_dq  : query-stationary (128 queries / workgroup), sweeps key tiles, accumulates dQ^T[c][q]
//   k_attn_dkv : key-stationary   (128 keys / workgroup),    sweeps query tiles, accumulates dK^T, dV^T[c][key]
// Both keep the "stationary" index on the lane, so softmax terms are lane-local, the score tile is
// used directly as the next MFMA's B operand (accumulator-as-operand, rows permuted by pi), and every
// operand that needs the contraction index contiguous is fetched from the row-major LDS tiles with the
// transposing read ds_read_b64_tr_b16.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ uint4 tr_frag2(const char* lds, int r0, int r1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + r0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + r1));
    uint4 f;
    f.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    f.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    f.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    f.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return f;
}

__device__ __forceinline__ f32x16_t mma_bf16(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ int pi_row(int r) { return (r & 0x13) | ((r & 4) << 1) | ((r & 8) >> 1); }

// accumulator registers 8s..8s+7 -> bf16 B-operand fragment of k-step s
__device__ __forceinline__ uint4 acc_to_frag(const f32x16_t& v, int st) {
    uint4 f;
    f.x = pack_bf16x2(v[8 * st + 0], v[8 * st + 1]);
    f.y = pack_bf16x2(v[8 * st + 2], v[8 * st + 3]);
    f.z = pack_bf16x2(v[8 * st + 4], v[8 * st + 5]);
    f.w = pack_bf16x2(v[8 * st + 6], v[8 * st + 7]);
    return f;
}

// ------------------------------------------------------------------------------------------------ delta
template <typename T>
__global__ __launch_bounds__(256) void k_attn_delta(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta,
                                                    int64_t BT, int heads, int ch, int64_t T_) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over (b*T + q, h)
    if (i >= BT * heads) return;
    const int h = (int)(i % heads);
    const int64_t bq = i / heads;
    const T* po = o + (bq * heads + h) * ch;
    const T* pd = dout + (bq * heads + h) * ch;
    float acc = 0.0f;
    for (int c = 0; c < ch; ++c) {
        float a, b;
        if constexpr (sizeof(T) == 2) { a = bf16_to_f32(po[c]); b = bf16_to_f32(pd[c]); } else { a = po[c]; b = pd[c]; }
        acc = fmaf(a, b, acc);
    }
    const int64_t b_ = bq / T_, q = bq % T_;
    delta[(b_ * heads + h) * T_ + q] = acc;
}

// ------------------------------------------------------------------------------------------------ dQ (bf16)
template <int CH, int KT>
__global__ __launch_bounds__(256) void k_attn_dq(const bf16_raw* __restrict__ qk, const bf16_raw* __restrict__ vt,
                                                 const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
                                                 const float* __restrict__ delta, bf16_raw* __restrict__ dqk, int T, int C,
                                                 float scale_log2e, float scale, int dqk_rs) {
    constexpr int KP = CH * 2 + 16;
    constexpr int VP = KT * 2 + 16;
    constexpr int NKK = CH / 16;
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = KT / 32;
    __shared__ __attribute__((aligned(16))) char k_lds[KT * KP];            // [key][ch]
    __shared__ __attribute__((aligned(16))) char v_lds[NCT * 32 * VP];      // [ch][key]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int qi = blockIdx.x * 128 + wave * 32 + col;
    const int qc = qi < T ? qi : T - 1;
    const size_t row2c = (size_t)2 * C;

    uint4 qf[NKK], dof[NKK];
    {
        const bf16_raw* qp = qk + ((size_t)b * T + qc) * row2c + (size_t)h * CH + 8 * half;
        const bf16_raw* dp = dout + ((size_t)b * T + qc) * C + (size_t)h * CH + 8 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            qf[kk] = *reinterpret_cast<const uint4*>(qp + 16 * kk);
            dof[kk] = *reinterpret_cast<const uint4*>(dp + 16 * kk);
        }
    }
    const float lse_q = lse[((size_t)b * heads + h) * T + qc];
    const float del_q = delta[((size_t)b * heads + h) * T + qc];

    f32x16_t dq[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[ct][r] = 0.0f;

    const int prow = pi_row(col);
    const int pswap = (pp == 1) ? 2 : (pp == 2 ? 1 : pp);     // pi applied to the 4-column chunk a lane addresses
    const bool vec_v = ((T & 7) == 0);

    for (int kt0 = 0; kt0 < T; kt0 += KT) {
        __syncthreads();
        for (int pc = tid; pc < KT * (CH / 8); pc += 256) {
            const int key = pc / (CH / 8), piece = pc % (CH / 8);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (kt0 + key < T)
                v = *reinterpret_cast<const uint4*>(qk + ((size_t)b * T + kt0 + key) * row2c + C + (size_t)h * CH + piece * 8);
            *reinterpret_cast<uint4*>(k_lds + key * KP + piece * 16) = v;
        }
        if (vec_v) {
            for (int pc = tid; pc < NCT * 32 * (KT / 8); pc += 256) {
                const int c = pc / (KT / 8), piece = pc % (KT / 8);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (c < CH && kt0 + piece * 8 < T)
                    v = *reinterpret_cast<const uint4*>(vt + ((size_t)b * C + (size_t)h * CH + c) * T + kt0 + piece * 8);
                *reinterpret_cast<uint4*>(v_lds + c * VP + piece * 16) = v;
            }
        } else {
            for (int e = tid; e < NCT * 32 * KT; e += 256) {
                const int c = e / KT, key = e % KT;
                bf16_raw v = 0;
                if (c < CH && kt0 + key < T) v = vt[((size_t)b * C + (size_t)h * CH + c) * T + kt0 + key];
                *reinterpret_cast<bf16_raw*>(v_lds + c * VP + key * 2) = v;
            }
        }
        __syncthreads();

#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const char* kp = k_lds + (32 * u + prow) * KP + 16 * half;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const uint4 a = *reinterpret_cast<const uint4*>(kp + 32 * kk);
                s = mma_bf16(a, qf[kk], s);
                // dP^T[key][q]: A = V[key][ch] read transposed out of the [ch][key] tile, rows in pi order
                const int r0 = (16 * kk + 8 * (grp >> 1) + qq) * VP + (32 * u + 16 * (grp & 1) + 4 * pswap) * 2;
                const uint4 av = tr_frag2(v_lds, r0, r0 + 4 * VP);
                dp = mma_bf16(av, dof[kk], dp);
            }
            // dS^T = P^T * (dP^T - delta) * scale, masked keys -> 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int key = kt0 + 32 * u + pi_row(row);
                const float pv = (key < T) ? exp2f(s[r] * scale_log2e - lse_q) : 0.0f;
                s[r] = pv * (dp[r] - del_q) * scale;
            }
            // dQ^T[c][q] += K^T[c][key] * dS^T[key][q]
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 bs = acc_to_frag(s, st);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int r0 = (32 * u + 16 * st + 8 * (grp >> 1) + qq) * KP + (32 * ct + 16 * (grp & 1) + 4 * pp) * 2;
                    const uint4 ak = tr_frag2(k_lds, r0, r0 + 4 * KP);
                    dq[ct] = mma_bf16(ak, bs, dq[ct]);
                }
            }
        }
    }
    if (qi < T) {
        bf16_raw* op = dqk + ((size_t)b * T + qi) * dqk_rs + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH)
                    *reinterpret_cast<uint2*>(op + c) = make_uint2(pack_bf16x2(dq[ct][4 * rg + 0], dq[ct][4 * rg + 1]),
                                                                   pack_bf16x2(dq[ct][4 * rg + 2], dq[ct][4 * rg + 3]));
            }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV (bf16)
template <int CH, int QT>
__global__ __launch_bounds__(256) void k_attn_dkv(const bf16_raw* __restrict__ qk, const bf16_raw* __restrict__ vt,
                                                  const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
                                                  const float* __restrict__ delta, bf16_raw* __restrict__ dqk,
                                                  bf16_raw* __restrict__ dv, int T, int C, float scale_log2e, float scale,
                                                  int dqk_rs, int dv_rs) {
    constexpr int KP = CH * 2 + 16;
    constexpr int NKK = CH / 16;
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = QT / 32;
    __shared__ __attribute__((aligned(16))) char q_lds[QT * KP];     // Q  [query][ch]
    __shared__ __attribute__((aligned(16))) char d_lds[QT * KP];     // dO [query][ch]
    __shared__ float lse_s[QT], del_s[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int k0 = blockIdx.x * 128;
    const int kj = k0 + wave * 32 + col;
    const int kc = kj < T ? kj : T - 1;
    const size_t row2c = (size_t)2 * C;

    // K fragments (B operand: lane = key, 8 consecutive channels)
    uint4 kf[NKK], vf[NKK];
    {
        const bf16_raw* kp = qk + ((size_t)b * T + kc) * row2c + C + (size_t)h * CH + 8 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) kf[kk] = *reinterpret_cast<const uint4*>(kp + 16 * kk);
    }
    // V fragments (B operand: lane = key, k = channel): V^T is channel-major, so each element is its own
    // 2-byte load -- once per workgroup, outside the query sweep
    {
        const bf16_raw* vp = vt + ((size_t)b * C + (size_t)h * CH + 8 * half) * T + kc;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t lo = vp[(size_t)(16 * kk + 2 * j) * T];
                const uint32_t hi = vp[(size_t)(16 * kk + 2 * j + 1) * T];
                w[j] = lo | (hi << 16);
            }
            vf[kk] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }

    f32x16_t dkacc[NCT], dvacc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dkacc[ct][r] = 0.0f; dvacc[ct][r] = 0.0f; }
    const int prow = pi_row(col);

    for (int qt0 = 0; qt0 < T; qt0 += QT) {
        __syncthreads();
        for (int pc = tid; pc < QT * (CH / 8); pc += 256) {
            const int q = pc / (CH / 8), piece = pc % (CH / 8);
            uint4 vq = make_uint4(0u, 0u, 0u, 0u), vd = vq;
            if (qt0 + q < T) {
                vq = *reinterpret_cast<const uint4*>(qk + ((size_t)b * T + qt0 + q) * row2c + (size_t)h * CH + piece * 8);
                vd = *reinterpret_cast<const uint4*>(dout + ((size_t)b * T + qt0 + q) * C + (size_t)h * CH + piece * 8);
            }
            *reinterpret_cast<uint4*>(q_lds + q * KP + piece * 16) = vq;
            *reinterpret_cast<uint4*>(d_lds + q * KP + piece * 16) = vd;
        }
        if (tid < QT) {
            const bool ok = qt0 + tid < T;
            lse_s[tid] = ok ? lse[((size_t)b * heads + h) * T + qt0 + tid] : INFINITY;   // exp2(-inf) = 0 masks the row
            del_s[tid] = ok ? delta[((size_t)b * heads + h) * T + qt0 + tid] : 0.0f;
        }
        __syncthreads();

#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const char* qp = q_lds + (32 * u + prow) * KP + 16 * half;
            const char* dp_ = d_lds + (32 * u + prow) * KP + 16 * half;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const uint4 aq = *reinterpret_cast<const uint4*>(qp + 32 * kk);
                const uint4 ad = *reinterpret_cast<const uint4*>(dp_ + 32 * kk);
                s = mma_bf16(aq, kf[kk], s);        // S[q][key]
                dp = mma_bf16(ad, vf[kk], dp);      // dP[q][key]
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int q = 32 * u + pi_row(row);
                const float pv = exp2f(s[r] * scale_log2e - lse_s[q]);
                dp[r] = pv * (dp[r] - del_s[q]) * scale;     // dS
                s[r] = pv;                                   // P
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 bp = acc_to_frag(s, st);
                const uint4 bs = acc_to_frag(dp, st);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int r0 = (32 * u + 16 * st + 8 * (grp >> 1) + qq) * KP + (32 * ct + 16 * (grp & 1) + 4 * pp) * 2;
                    const uint4 ado = tr_frag2(d_lds, r0, r0 + 4 * KP);     // dO^T[c][q]
                    const uint4 aq = tr_frag2(q_lds, r0, r0 + 4 * KP);      // Q^T[c][q]
                    dvacc[ct] = mma_bf16(ado, bp, dvacc[ct]);
                    dkacc[ct] = mma_bf16(aq, bs, dkacc[ct]);
                }
            }
        }
    }
    if (kj < T) {
        bf16_raw* okp = dqk + ((size_t)b * T + kj) * dqk_rs + C + (size_t)h * CH;
        bf16_raw* ovp = dv + ((size_t)b * T + kj) * dv_rs + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH) {
                    *reinterpret_cast<uint2*>(okp + c) = make_uint2(pack_bf16x2(dkacc[ct][4 * rg + 0], dkacc[ct][4 * rg + 1]),
                                                                    pack_bf16x2(dkacc[ct][4 * rg + 2], dkacc[ct][4 * rg + 3]));
                    *reinterpret_cast<uint2*>(ovp + c) = make_uint2(pack_bf16x2(dvacc[ct][4 * rg + 0], dvacc[ct][4 * rg + 1]),
                                                                    pack_bf16x2(dvacc[ct][4 * rg + 2], dvacc[ct][4 * rg + 3]));
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ exact-f32 (VALU) variants
template <int CH, int KT>
__global__ __launch_bounds__(256) void k_attn_dq_f32(const float* __restrict__ qk, const float* __restrict__ vt,
                                                     const float* __restrict__ dout, const float* __restrict__ lse,
                                                     const float* __restrict__ delta, float* __restrict__ dqk, int T, int C,
                                                     float scale_log2e, float scale, int dqk_rs) {
    constexpr int CP = CH / 4;
    __shared__ float k_lds[KT][CH + 1];
    __shared__ float v_lds[KT][CH + 1];
    const int tid = threadIdx.x, part = tid & 3;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int qi = blockIdx.x * 64 + (tid >> 2);
    const int qc = qi < T ? qi : T - 1;
    const size_t row2c = (size_t)2 * C;
    float q[CP], dO[CP], dq[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        q[c] = qk[((size_t)b * T + qc) * row2c + (size_t)h * CH + part * CP + c];
        dO[c] = dout[((size_t)b * T + qc) * C + (size_t)h * CH + part * CP + c];
        dq[c] = 0.0f;
    }
    const float lse_q = lse[((size_t)b * heads + h) * T + qc], del_q = delta[((size_t)b * heads + h) * T + qc];
    for (int kt0 = 0; kt0 < T; kt0 += KT) {
        __syncthreads();
        for (int e = tid; e < KT * CH; e += 256) {
            const int key = e / CH, c = e % CH;
            k_lds[key][c] = (kt0 + key < T) ? qk[((size_t)b * T + kt0 + key) * row2c + C + (size_t)h * CH + c] : 0.0f;
        }
        for (int e = tid; e < KT * CH; e += 256) {
            const int c = e / KT, key = e % KT;
            v_lds[key][c] = (kt0 + key < T) ? vt[((size_t)b * C + (size_t)h * CH + c) * T + kt0 + key] : 0.0f;
        }
        __syncthreads();
        for (int k = 0; k < KT; ++k) {
            float s = 0.0f, dp = 0.0f;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                s = fmaf(q[c], k_lds[k][part * CP + c], s);
                dp = fmaf(dO[c], v_lds[k][part * CP + c], dp);
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64);
            const float pv = (kt0 + k < T) ? exp2f(s * scale_log2e - lse_q) : 0.0f;
            const float ds = pv * (dp - del_q) * scale;
#pragma unroll
            for (int c = 0; c < CP; ++c) dq[c] = fmaf(ds, k_lds[k][part * CP + c], dq[c]);
        }
    }
    if (qi < T) {
#pragma unroll
        for (int c = 0; c < CP; ++c) dqk[((size_t)b * T + qi) * dqk_rs + (size_t)h * CH + part * CP + c] = dq[c];
    }
}

template <int CH, int QT>
__global__ __launch_bounds__(256) void k_attn_dkv_f32(const float* __restrict__ qk, const float* __restrict__ vt,
                                                      const float* __restrict__ dout, const float* __restrict__ lse,
                                                      const float* __restrict__ delta, float* __restrict__ dqk,
                                                      float* __restrict__ dv, int T, int C, float scale_log2e, float scale,
                                                      int dqk_rs, int dv_rs) {
    constexpr int CP = CH / 4;
    __shared__ float q_lds[QT][CH + 1];
    __shared__ float d_lds[QT][CH + 1];
    __shared__ float lse_s[QT], del_s[QT];
    const int tid = threadIdx.x, part = tid & 3;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int kj = blockIdx.x * 64 + (tid >> 2);
    const int kc = kj < T ? kj : T - 1;
    const size_t row2c = (size_t)2 * C;
    float kk[CP], vv[CP], dk[CP], dvv[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        kk[c] = qk[((size_t)b * T + kc) * row2c + C + (size_t)h * CH + part * CP + c];
        vv[c] = vt[((size_t)b * C + (size_t)h * CH + part * CP + c) * T + kc];
        dk[c] = 0.0f;
        dvv[c] = 0.0f;
    }
    for (int qt0 = 0; qt0 < T; qt0 += QT) {
        __syncthreads();
        for (int e = tid; e < QT * CH; e += 256) {
            const int q = e / CH, c = e % CH;
            const bool ok = qt0 + q < T;
            q_lds[q][c] = ok ? qk[((size_t)b * T + qt0 + q) * row2c + (size_t)h * CH + c] : 0.0f;
            d_lds[q][c] = ok ? dout[((size_t)b * T + qt0 + q) * C + (size_t)h * CH + c] : 0.0f;
        }
        if (tid < QT) {
            const bool ok = qt0 + tid < T;
            lse_s[tid] = ok ? lse[((size_t)b * heads + h) * T + qt0 + tid] : INFINITY;
            del_s[tid] = ok ? delta[((size_t)b * heads + h) * T + qt0 + tid] : 0.0f;
        }
        __syncthreads();
        for (int q = 0; q < QT; ++q) {
            float s = 0.0f, dp = 0.0f;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                s = fmaf(kk[c], q_lds[q][part * CP + c], s);
                dp = fmaf(vv[c], d_lds[q][part * CP + c], dp);
            }
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
            dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64);
            const float pv = exp2f(s * scale_log2e - lse_s[q]);
            const float ds = pv * (dp - del_s[q]) * scale;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                dvv[c] = fmaf(pv, d_lds[q][part * CP + c], dvv[c]);
                dk[c] = fmaf(ds, q_lds[q][part * CP + c], dk[c]);
            }
        }
    }
    if (kj < T) {
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            dqk[((size_t)b * T + kj) * dqk_rs + C + (size_t)h * CH + part * CP + c] = dk[c];
            dv[((size_t)b * T + kj) * dv_rs + (size_t)h * CH + part * CP + c] = dvv[c];
        }
    }
}

// qk [B,T,2C], vt [B,C,T], o / dout [B,T,C] channels-last; lse, delta_ws float32 [B,heads,T];
// outputs: dqk [B,T,2C] (dq | dk), dv [B,T,C], both channels-last in `dtype`.
extern "C" int rho_attention_bwd(const void* qk, const void* vt, const void* o, const void* dout, const float* lse,
                                 float* delta_ws, void* dqk, int64_t dqk_row_stride, void* dv, int64_t dv_row_stride, int dtype,
                                 int64_t batch, int64_t t, int64_t heads, int64_t ch, void* stream) {
    if (!qk || !vt || !o || !dout || !lse || !delta_ws || !dqk || !dv || batch <= 0 || t <= 0 || heads <= 0) return RHO_E_ARG;
    if (dtype != RHO_BF16 && dtype != RHO_F32) return RHO_E_ARG;
    const int C = (int)(heads * ch);
    const int qrs = (int)(dqk_row_stride > 0 ? dqk_row_stride : 2 * C), vrs = (int)(dv_row_stride > 0 ? dv_row_stride : C);
    const float scale = (float)(1.0 / sqrt((double)ch));
    const float sl2 = (float)(1.4426950408889634 / sqrt((double)ch));
    hipStream_t st = as_stream(stream);
    const int64_t nd = batch * t * heads;
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_attn_delta<bf16_raw>, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, (const bf16_raw*)o,
                           (const bf16_raw*)dout, delta_ws, batch * t, (int)heads, (int)ch, t);
    else
        hipLaunchKernelGGL(k_attn_delta<float>, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, (const float*)o,
                           (const float*)dout, delta_ws, batch * t, (int)heads, (int)ch, t);
    RHO_LAUNCH_CHECK();
    if (dtype == RHO_BF16) {
        dim3 grid((unsigned)((t + 127) / 128), (unsigned)heads, (unsigned)batch), block(256);
#define RHO_ATTB(chv, ktv)                                                                                                     \
    case chv:                                                                                                                  \
        hipLaunchKernelGGL((k_attn_dq<chv, ktv>), grid, block, 0, st, (const bf16_raw*)qk, (const bf16_raw*)vt,                  \
                           (const bf16_raw*)dout, lse, delta_ws, (bf16_raw*)dqk, (int)t, C, sl2, scale, qrs);                   \
        hipLaunchKernelGGL((k_attn_dkv<chv, ktv>), grid, block, 0, st, (const bf16_raw*)qk, (const bf16_raw*)vt,                 \
                           (const bf16_raw*)dout, lse, delta_ws, (bf16_raw*)dqk, (bf16_raw*)dv, (int)t, C, sl2, scale, qrs, vrs); \
        break;
        switch (ch) {
            RHO_ATTB(16, 64)
            RHO_ATTB(32, 64)
            RHO_ATTB(64, 64)
            RHO_ATTB(128, 64)
            RHO_ATTB(256, 32)
            default:
                return RHO_E_SHAPE;
        }
#undef RHO_ATTB
    } else {
        dim3 grid((unsigned)((t + 63) / 64), (unsigned)heads, (unsigned)batch), block(256);
#define RHO_ATTB32(chv, ktv)                                                                                                  \
    case chv:                                                                                                                 \
        hipLaunchKernelGGL((k_attn_dq_f32<chv, ktv>), grid, block, 0, st, (const float*)qk, (const float*)vt,                  \
                           (const float*)dout, lse, delta_ws, (float*)dqk, (int)t, C, sl2, scale, qrs);                        \
        hipLaunchKernelGGL((k_attn_dkv_f32<chv, ktv>), grid, block, 0, st, (const float*)qk, (const float*)vt,                 \
                           (const float*)dout, lse, delta_ws, (float*)dqk, (float*)dv, (int)t, C, sl2, scale, qrs, vrs);       \
        break;
        switch (ch) {
            RHO_ATTB32(16, 64)
            RHO_ATTB32(32, 64)
            RHO_ATTB32(64, 64)
            RHO_ATTB32(128, 32)
            RHO_ATTB32(256, 16)
            default:
                return RHO_E_SHAPE;
        }
#undef RHO_ATTB32
    }
    RHO_LAUNCH_CHECK();
    return 0;
}
