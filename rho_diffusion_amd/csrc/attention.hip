// Self-attention of the UNet's AttentionBlock (rho_diffusion/models/unet_v2.py:365-436) as a
// flash-style kernel: the [T, T] logits (4.3 GB per head at T = 32768 in the reference) never
// exist; fp32 online softmax, MFMA only for the QK^T and PV contractions.
//
// Data: qk channels-last [B, T, 2C] (q of head h at h*ch, k at C + h*ch), vt channel-major
// [B, C, T] -- both written by the qkv 1x1 projection's epilogue -- and out channels-last [B, T, C].
//
// One workgroup = 4 waves = 128 queries of one (batch, head); a wave owns 32 queries.
//   S^T[key][query] = K * Q^T   (v_mfma_f32_32x32x16_bf16: A = K rows from LDS, B = Q in registers)
// puts the query on the lane and the 32 keys of a tile in the 16 accumulator registers of the two
// half-waves, so the row max / row sum are in-register reductions plus ONE lane^32 exchange.
// K rows are fetched in the permuted order pi (bits 2 and 3 of the row swapped): the accumulator
// of S^T, converted pairwise to bf16, then IS the B operand of
//   O^T[c][query] += V^T[c][key] * P^T[key][query]
// with V^T read as plain 16-byte rows of 8 consecutive keys (A operand) -- no transposes, no LDS
// round trip for P (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").
#include <cstdlib>

#include "common.h"

// max / sum of a value over the two half-waves of a column (lanes l and l ^ 32), in every lane: one v_permlane32_swap (gfx950: the
// operand pair comes back as {x[l & 31], x[(l & 31) + 32]}) instead of a ds_bpermute round trip through the LDS crossbar and the
// lgkmcnt(0) it drags behind it, which also waits for every LDS read in flight.  Both operations commute: the results are the bits
// the exchange gave.
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float halves_max(float x) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
__device__ __forceinline__ float halves_sum(float x) {
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}

template <int CH, int KT>  // KT keys per LDS tile (64; 32 for CH = 256 to stay inside static LDS)
__global__ __launch_bounds__(256, CH <= 128 ? 2 : 1) void k_attn_bf16(const bf16_raw* __restrict__ qk, const bf16_raw* __restrict__ vt,
                                                   bf16_raw* __restrict__ out, int T, int C, float scale_log2e,
                                                   float* __restrict__ lse) {
    constexpr int KP = CH * 2 + 16;   // K tile row pitch (bytes): odd number of 16-B slots => conflict-free
    constexpr int VP = KT * 2 + 16;   // V^T tile row pitch
    constexpr int NKK = CH / 16;      // k-steps of the QK^T contraction
    constexpr int NCT = (CH + 31) / 32;  // 32-row channel tiles of O^T
    constexpr int NU = KT / 32;          // 32-key sub-tiles per LDS tile
    __shared__ __attribute__((aligned(16))) char k_lds[KT * KP];
    __shared__ __attribute__((aligned(16))) char v_lds[NCT * 32 * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    int tile_, h, b;
    attn_block(tile_, h, b);
    const int q0 = tile_ * 128 + wave * 32;
    const int qi = q0 + col;
    const size_t row2c = (size_t)2 * C;

    // ---- Q fragments (B operand): 8 consecutive channels of this lane's query per k-step
    uint4 qf[NKK];
    {
        const int qc = qi < T ? qi : T - 1;
        const bf16_raw* qp = qk + ((size_t)b * T + qc) * row2c + (size_t)h * CH + 8 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = *reinterpret_cast<const uint4*>(qp + 16 * kk);
    }

    f32x16_t o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // permuted K row for the A operand: swap bits 2 and 3 of the tile row
    const int prow = (col & 0x13) | ((col & 4) << 1) | ((col & 8) >> 1);
    const bool vec_v = ((T & 7) == 0);

    // staging addresses: thread -> (row, 16-byte piece), fixed for the whole key walk; only the tile origin moves
    constexpr int KPC = CH / 8;                   // 16-byte pieces per K row
    constexpr int KIT = (KT * KPC + 255) / 256;   // K pieces per thread
    constexpr bool KEXACT = (KIT * 256 == KT * KPC);
    constexpr int VPC = KT / 8;                   // 16-byte pieces per V^T row
    constexpr int VIT = (NCT * 32 * VPC + 255) / 256;
    const bf16_raw* const kbase = qk + (size_t)b * T * row2c + C + (size_t)h * CH;
    const bf16_raw* const vbase = vt + ((size_t)b * C + (size_t)h * CH) * T;
    // m_run is kept in the scaled base-2 domain; the logits stay raw and are scaled inside the exponent's FMA
    const float sc = scale_log2e;

    // K / V^T tiles: issue-early / write-late (cdna_hip_programming.md T14).  The global loads of tile t+1 are issued into
    // registers BEFORE the MFMA / softmax work of tile t and written to LDS after it, so their latency (HBM or L2) runs under
    // ~2 us of compute instead of in front of it; r01 loaded, waited and wrote each tile between two barriers with only the
    // other workgroup of the CU to cover the wait.
    uint4 kv[KIT], vv[VIT];
    auto load_tile = [&](int kt0) {                  // branch-free (clamped addresses, zeroed at write time): counted vmcnt
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int pc = tid + 256 * i;
            const int key = pc / KPC, piece = pc % KPC;
            const int kc = min(kt0 + (KEXACT ? key : min(key, KT - 1)), T - 1);
            kv[i] = *reinterpret_cast<const uint4*>(kbase + (size_t)kc * row2c + piece * 8);
        }
        if (vec_v) {
#pragma unroll
            for (int i = 0; i < VIT; ++i) {
                const int pc = tid + 256 * i;
                const int c = pc / VPC, piece = pc % VPC;
                const bool ok = (VIT * 256 == NCT * 32 * VPC || pc < NCT * 32 * VPC) && c < CH && kt0 + piece * 8 < T;
                vv[i] = *reinterpret_cast<const uint4*>(vbase + (size_t)(ok ? c : 0) * T + (ok ? kt0 + piece * 8 : 0));
            }
        }
    };
    auto store_tile = [&](int kt0) {
        if (vec_v) {
#pragma unroll
            for (int i = 0; i < VIT; ++i) {
                const int pc = tid + 256 * i;
                const int c = pc / VPC, piece = pc % VPC;
                const bool ok = (VIT * 256 == NCT * 32 * VPC || pc < NCT * 32 * VPC) && c < CH && kt0 + piece * 8 < T;
                if (VIT * 256 == NCT * 32 * VPC || pc < NCT * 32 * VPC)
                    *reinterpret_cast<uint4*>(v_lds + c * VP + piece * 16) = ok ? vv[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        } else {                                     // T not a multiple of 8 (tiny test shapes): element-wise fill
            for (int e = tid; e < NCT * 32 * KT; e += 256) {
                const int c = e / KT, key = e % KT;
                bf16_raw v = 0;
                if (c < CH && kt0 + key < T) v = vbase[(size_t)c * T + kt0 + key];
                *reinterpret_cast<bf16_raw*>(v_lds + c * VP + key * 2) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int pc = tid + 256 * i;
            const int key = pc / KPC, piece = pc % KPC;
            if (KEXACT || pc < KT * KPC)
                *reinterpret_cast<uint4*>(k_lds + key * KP + piece * 16) = (kt0 + key < T) ? kv[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    // Whole tiles (every tile but a ragged last one) take a path without the clamps, selects and 64-bit multiplies above (~50 of
    // the ~210 vector instructions per tile and wave at CH = 64, where the softmax already out-issues the matrix cores): the
    // tile origin is a uniform (scalar) base, the thread's share a 32-bit offset computed once, the LDS addresses likewise.
    constexpr bool FAST = KEXACT && (VIT * 256 == NCT * 32 * VPC) && (CH % 32 == 0);
    unsigned koff[KIT], voff[VIT], klds[KIT], vlds[VIT];
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
        const int pc = tid + 256 * i, key = pc / KPC, piece = pc % KPC;
        koff[i] = (unsigned)(((size_t)key * row2c + piece * 8) * 2);         // bytes; 64 rows of <= 4 K channels
        klds[i] = (unsigned)(key * KP + piece * 16);
    }
#pragma unroll
    for (int i = 0; i < VIT; ++i) {
        const int pc = tid + 256 * i, c = pc / VPC, piece = pc % VPC;
        voff[i] = (unsigned)(((size_t)(c < CH ? c : 0) * T + piece * 8) * 2);  // bytes; CH rows of T <= 2^24 keys
        vlds[i] = (unsigned)(c * VP + piece * 16);
    }
    const bool fast_ok = FAST && vec_v && (size_t)CH * T * 2 < (1ull << 32);
    auto load_tile_fast = [&](int kt0) {
        const char* const kt_base = reinterpret_cast<const char*>(kbase) + (size_t)kt0 * row2c * 2;
        const char* const vt_base = reinterpret_cast<const char*>(vbase) + (size_t)kt0 * 2;
#pragma unroll
        for (int i = 0; i < KIT; ++i) kv[i] = *reinterpret_cast<const uint4*>(kt_base + koff[i]);
#pragma unroll
        for (int i = 0; i < VIT; ++i) vv[i] = *reinterpret_cast<const uint4*>(vt_base + voff[i]);
    };
    auto store_tile_fast = [&]() {
#pragma unroll
        for (int i = 0; i < VIT; ++i) *reinterpret_cast<uint4*>(v_lds + vlds[i]) = vv[i];
#pragma unroll
        for (int i = 0; i < KIT; ++i) *reinterpret_cast<uint4*>(k_lds + klds[i]) = kv[i];
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();

    for (int kt0 = 0; kt0 < T; kt0 += KT) {
        const bool more = kt0 + KT < T;
        const bool next_whole = fast_ok && kt0 + 2 * KT <= T;      // (uniform)
        if (more) {                                  // in flight under this tile's MFMAs
            if (next_whole) load_tile_fast(kt0 + KT);
            else load_tile(kt0 + KT);
        }

        // ---- S^T for the 32-key sub-tiles
        f32x16_t s[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = 0.0f;
            const char* kp = k_lds + (32 * u + prow) * KP + 16 * half;
            // K fragments in groups of 4 reads ahead of their MFMAs (one read per MFMA, waited for right before it,
            // leaves the LDS latency exposed 2 * NKK times per tile)
            constexpr int KG = NKK < 4 ? NKK : 4;
#pragma unroll
            for (int k0 = 0; k0 < NKK; k0 += KG) {
                uint4 a[KG];
#pragma unroll
                for (int g = 0; g < KG; ++g) a[g] = *reinterpret_cast<const uint4*>(kp + 32 * (k0 + g));
#pragma unroll
                for (int g = 0; g < KG; ++g)
                    s[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[g]),
                                                                  __builtin_bit_cast(bf16x8_t, qf[k0 + g]), s[u], 0, 0, 0);
            }
        }
        // ---- online softmax (base-2).  Only the last, partial tile needs the key mask (uniform branch).
        if (kt0 + KT > T) {
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;               // accumulator row
                    const int krow = (row & 0x13) | ((row & 4) << 1) | ((row & 8) >> 1);  // key held there (pi)
                    if (kt0 + 32 * u + krow >= T) s[u][r] = -INFINITY;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
        mx = halves_max(mx);
        const float m_new = fmaxf(m_run, mx * sc);          // sc > 0: max commutes with the scaling
        // the running maximum stops moving after the first few tiles: rescale only when some query of this wave
        // saw a new maximum (alpha == 1 exactly otherwise, so skipping is bit-identical)
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float alpha = exp2f(m_run - m_new);       // m_run = -inf on the first tile -> 0
            l_run *= alpha;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
            m_run = m_new;
        }
        float ps = 0.0f;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[u][r], sc, -m_run));
                s[u][r] = pv;
                ps += pv;
            }
        ps = halves_sum(ps);
        l_run += ps;

        // ---- O^T += V^T * P^T : P accumulator registers 8s..8s+7 are k-step s of the B operand
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uint4 pb;
                pb.x = pack_bf16x2(s[u][8 * st + 0], s[u][8 * st + 1]);
                pb.y = pack_bf16x2(s[u][8 * st + 2], s[u][8 * st + 3]);
                pb.z = pack_bf16x2(s[u][8 * st + 4], s[u][8 * st + 5]);
                pb.w = pack_bf16x2(s[u][8 * st + 6], s[u][8 * st + 7]);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const uint4 a = *reinterpret_cast<const uint4*>(v_lds + (32 * ct + col) * VP + (32 * u + 16 * st + 8 * half) * 2);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, pb),
                                                                   o[ct], 0, 0, 0);
                }
            }
        if (more) {
            __syncthreads();                         // this tile's LDS fragments are consumed by every wave
            if (next_whole) store_tile_fast();
            else store_tile(kt0 + KT);
            __syncthreads();
        }
    }

    // ---- normalise and store: lane holds channels 32*ct + 8*rg + 4*half + {0..3} of query qi
    if (qi < T) {
        const float inv = 1.0f / l_run;
        // base-2 log-sum-exp of the scaled logits, saved for the backward recompute
        if (lse != nullptr && half == 0) lse[((size_t)b * gridDim.y + h) * T + qi] = m_run + log2f(l_run);
        bf16_raw* op = out + ((size_t)b * T + qi) * C + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH) {
                    const uint2 w = make_uint2(pack_bf16x2(o[ct][4 * rg + 0] * inv, o[ct][4 * rg + 1] * inv),
                                               pack_bf16x2(o[ct][4 * rg + 2] * inv, o[ct][4 * rg + 3] * inv));
                    *reinterpret_cast<uint2*>(op + c) = w;
                }
            }
    }
}


// ------------------------------------------------------------------------------------------------
// Exact-fp32 variant (parity path, UNet compute_dtype = fp32): plain FMA flash attention.
// 64 queries per workgroup; 4 lanes share a query, each owning CH/4 channels of q and of the output.
template <int CH, int KT>
__global__ __launch_bounds__(256) void k_attn_f32(const float* __restrict__ qk, const float* __restrict__ vt,
                                                  float* __restrict__ out, int T, int C, float scale_log2e,
                                                  float* __restrict__ lse) {
    constexpr int CP = CH / 4;
    __shared__ float k_lds[KT][CH + 1];
    __shared__ float v_lds[KT][CH + 1];
    const int tid = threadIdx.x, part = tid & 3;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 64 + (tid >> 2);
    const size_t row2c = (size_t)2 * C;
    float q[CP], o[CP];
    {
        const int qc = qi < T ? qi : T - 1;
        const float* qp = qk + ((size_t)b * T + qc) * row2c + (size_t)h * CH + part * CP;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            q[c] = qp[c];
            o[c] = 0.0f;
        }
    }
    float m_run = -INFINITY, l_run = 0.0f;
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
        float sc[KT];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            float d = 0.0f;
#pragma unroll
            for (int c = 0; c < CP; ++c) d = fmaf(q[c], k_lds[k][part * CP + c], d);
            d += __shfl_xor(d, 1, 64);
            d += __shfl_xor(d, 2, 64);
            d = (kt0 + k < T) ? d * scale_log2e : -INFINITY;
            sc[k] = d;
            mx = fmaxf(mx, d);
        }
        const float m_new = fmaxf(m_run, mx);
        const float alpha = exp2f(m_run - m_new);
        float ps = 0.0f;
#pragma unroll
        for (int c = 0; c < CP; ++c) o[c] *= alpha;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const float pv = exp2f(sc[k] - m_new);
            ps += pv;
#pragma unroll
            for (int c = 0; c < CP; ++c) o[c] = fmaf(pv, v_lds[k][part * CP + c], o[c]);
        }
        l_run = l_run * alpha + ps;
        m_run = m_new;
    }
    if (qi < T) {
        const float inv = 1.0f / l_run;
        if (lse != nullptr && part == 0) lse[((size_t)b * gridDim.y + h) * T + qi] = m_run + log2f(l_run);
        float* op = out + ((size_t)b * T + qi) * C + (size_t)h * CH + part * CP;
#pragma unroll
        for (int c = 0; c < CP; ++c) op[c] = o[c] * inv;
    }
}

// ------------------------------------------------------------------------------------------------
// Exact-fp32 on the matrix cores (round 3): the same flash structure as k_attn_bf16 on v_mfma_f32_32x32x2_f32 (f32 in, f32
// accumulate: bitwise an fmaf chain per output, 64 FLOP / clk / SIMD = the f32 vector peak, MI355X_MICROARCH.md) - the VALU
// variant above spends its time on LDS-fed scalar FMAs and shuffles (c2, T = 256, ch = 128: 1.3 ms per block for 8.6 GFLOP).
//   S^T[key][q]  = K * Q^T : A = K[key = lane & 31][2 channels, one per half-wave], B = Q in registers.  Any pairing of channels
//                            into k-steps is valid as long as A and B agree: half 0 walks channels 0 .. CH/2-1, half 1 the other
//                            half, so both read whole float4 pieces (four k-steps per ds_read_b128).
//   O^T[c][q]   += V^T[c][key] * P^T[key][q] : accumulator register r of S^T holds key (r & 3) + 8 (r >> 2) + 4 half of the lane's
//                            query - exactly the two rows k-step r of the next MFMA contracts, so P never leaves its registers
//                            and is not rounded (the bf16 kernel rounds P to bf16 here); A = V^T rows read as float4 of 4 keys.
template <int CH, int KT>
__global__ __launch_bounds__(256) void k_attn_f32m(const float* __restrict__ qk, const float* __restrict__ vt, float* __restrict__ out,
                                                   int T, int C, float scale_log2e, float* __restrict__ lse) {
    constexpr int KP = CH * 4 + 16;      // K tile row pitch in bytes: odd number of 16-byte slots
    constexpr int VP = KT * 4 + 16;      // V^T tile row pitch
    constexpr int NJ = CH / 8;           // float4 pieces per half-wave of one K / Q row
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = KT / 32;
    static_assert(CH % 8 == 0 && KT % 32 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) char k_lds[KT * KP];
    __shared__ __attribute__((aligned(16))) char v_lds[NCT * 32 * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 128 + wave * 32 + col;
    const size_t row2c = (size_t)2 * C;

    float4 qf[NJ];
    {
        const int qc = qi < T ? qi : T - 1;
        const float* qp = qk + ((size_t)b * T + qc) * row2c + (size_t)h * CH + half * (CH / 2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) qf[j] = *reinterpret_cast<const float4*>(qp + 4 * j);
    }
    f32x16_t o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float sc = scale_log2e;
    const bool vec_v = ((T & 3) == 0);
    constexpr int KPC = CH / 4;           // float4 pieces per K row
    constexpr int VPC = KT / 4;           // float4 pieces per V^T row
    const float* const kbase = qk + (size_t)b * T * row2c + C + (size_t)h * CH;
    const float* const vbase = vt + ((size_t)b * C + (size_t)h * CH) * T;

    for (int kt0 = 0; kt0 < T; kt0 += KT) {
        __syncthreads();                  // the previous tile's fragments are consumed
        for (int pc = tid; pc < KT * KPC; pc += 256) {
            const int key = pc / KPC, piece = pc % KPC;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (kt0 + key < T) v = *reinterpret_cast<const float4*>(kbase + (size_t)(kt0 + key) * row2c + piece * 4);
            *reinterpret_cast<float4*>(k_lds + key * KP + piece * 16) = v;
        }
        if (vec_v) {
            for (int pc = tid; pc < NCT * 32 * VPC; pc += 256) {
                const int c = pc / VPC, piece = pc % VPC;
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (c < CH && kt0 + piece * 4 < T) v = *reinterpret_cast<const float4*>(vbase + (size_t)c * T + kt0 + piece * 4);
                *reinterpret_cast<float4*>(v_lds + c * VP + piece * 16) = v;
            }
        } else {                          // T not a multiple of 4 (tiny test shapes)
            for (int e = tid; e < NCT * 32 * KT; e += 256) {
                const int c = e / KT, key = e % KT;
                *reinterpret_cast<float*>(v_lds + c * VP + key * 4) = (c < CH && kt0 + key < T) ? vbase[(size_t)c * T + kt0 + key] : 0.0f;
            }
        }
        __syncthreads();

        f32x16_t s[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[u][r] = 0.0f;
            const char* kp = k_lds + (32 * u + col) * KP + half * (CH / 2) * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(kp + 16 * j);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[j].x, s[u], 0, 0, 0);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[j].y, s[u], 0, 0, 0);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, qf[j].z, s[u], 0, 0, 0);
                s[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, qf[j].w, s[u], 0, 0, 0);
            }
        }
        if (kt0 + KT > T) {               // partial last tile: mask the keys beyond T (uniform branch)
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kt0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * half >= T) s[u][r] = -INFINITY;
        }
        float mx = s[0][0];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
        mx = halves_max(mx);
        const float m_new = fmaxf(m_run, mx * sc);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
            const float alpha = exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
            m_run = m_new;
        }
        float ps = 0.0f;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = exp2f(fmaf(s[u][r], sc, -m_run));
                s[u][r] = pv;
                ps += pv;
            }
        ps = halves_sum(ps);
        l_run += ps;
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const char* vp = v_lds + (32 * ct + col) * VP + (32 * u + 4 * half) * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 a = *reinterpret_cast<const float4*>(vp + 32 * g);        // keys 32u + 8g + 4 half + {0..3}
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, s[u][4 * g + 0], o[ct], 0, 0, 0);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, s[u][4 * g + 1], o[ct], 0, 0, 0);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, s[u][4 * g + 2], o[ct], 0, 0, 0);
                    o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, s[u][4 * g + 3], o[ct], 0, 0, 0);
                }
            }
    }
    if (qi < T) {
        const float inv = 1.0f / l_run;
        if (lse != nullptr && half == 0) lse[((size_t)b * gridDim.y + h) * T + qi] = m_run + log2f(l_run);
        float* op = out + ((size_t)b * T + qi) * C + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH)
                    *reinterpret_cast<float4*>(op + c) = make_float4(o[ct][4 * rg + 0] * inv, o[ct][4 * rg + 1] * inv,
                                                                     o[ct][4 * rg + 2] * inv, o[ct][4 * rg + 3] * inv);
            }
    }
}

extern "C" int rho_attention_fwd(const void* qk, const void* vt, void* out, float* lse, int dtype, int64_t batch, int64_t t,
                                 int64_t heads, int64_t ch, void* stream) {
    if (!qk || !vt || !out || batch <= 0 || t <= 0 || heads <= 0) return RHO_E_ARG;
    if (dtype != RHO_BF16 && dtype != RHO_F32) return RHO_E_ARG;
    const int C = (int)(heads * ch);
    const float sl2 = (float)(1.4426950408889634 / sqrt((double)ch));
    hipStream_t st = as_stream(stream);
    if (dtype == RHO_F32) {
        // exact-f32 MFMA kernel (RHO_ATTN_F32_VALU=1 selects the round-1 VALU kernel: A/B and cross-check)
        static const bool valu_env = getenv("RHO_ATTN_F32_VALU") && atoi(getenv("RHO_ATTN_F32_VALU")) != 0;
        if (!valu_env) {
            dim3 grid((unsigned)((t + 127) / 128), (unsigned)heads, (unsigned)batch), block(256);
#define RHO_ATT32M(chv, ktv)                                                                                             \
    case chv:                                                                                                            \
        hipLaunchKernelGGL((k_attn_f32m<chv, ktv>), grid, block, 0, st, (const float*)qk, (const float*)vt, (float*)out, \
                           (int)t, C, sl2, lse);                                                                         \
        break;
            switch (ch) {
                RHO_ATT32M(16, 64)
                RHO_ATT32M(32, 64)
                RHO_ATT32M(64, 64)
                RHO_ATT32M(128, 32)
                RHO_ATT32M(256, 32)
                default:
                    return RHO_E_SHAPE;
            }
#undef RHO_ATT32M
            RHO_LAUNCH_CHECK();
            return 0;
        }
        dim3 grid((unsigned)((t + 63) / 64), (unsigned)heads, (unsigned)batch), block(256);
#define RHO_ATT32(chv, ktv)                                                                                             \
    case chv:                                                                                                           \
        hipLaunchKernelGGL((k_attn_f32<chv, ktv>), grid, block, 0, st, (const float*)qk, (const float*)vt, (float*)out, \
                           (int)t, C, sl2, lse);                                                                        \
        break;
        switch (ch) {
            RHO_ATT32(16, 64)
            RHO_ATT32(32, 64)
            RHO_ATT32(64, 64)
            RHO_ATT32(128, 32)
            RHO_ATT32(256, 16)
            default:
                return RHO_E_SHAPE;
        }
#undef RHO_ATT32
        RHO_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid((unsigned)((t + 127) / 128), (unsigned)heads, (unsigned)batch), block(256);
#define RHO_ATT(chv)                                                                                                      \
    case chv:                                                                                                             \
        hipLaunchKernelGGL((k_attn_bf16<chv, (chv >= 256 ? 32 : 64)>), grid, block, 0, st, (const bf16_raw*)qk,              \
                           (const bf16_raw*)vt, (bf16_raw*)out, (int)t, C, sl2, lse);                                     \
        break;
    switch (ch) {
        RHO_ATT(16)
        RHO_ATT(32)
        RHO_ATT(64)
        RHO_ATT(128)
        RHO_ATT(256)
        default:
            return RHO_E_SHAPE;
    }
#undef RHO_ATT
    RHO_LAUNCH_CHECK();
    return 0;
}
