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
#include <cstdlib>

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
// One 16-byte piece (8 bf16 / 4 f32 channels) of one (b, q) row per lane; the ch / PE lanes of a head are
// consecutive lanes of one wave (ch / PE is a power of two <= 64), reduced with butterfly shuffles: coalesced
// 16-byte loads at HBM rate instead of one thread walking a head's channels.
template <typename T>
__global__ __launch_bounds__(256) void k_attn_delta(const T* __restrict__ o, const T* __restrict__ dout, float* __restrict__ delta,
                                                    int64_t BT, int heads, int ch, int64_t T_) {
    constexpr int PE = 16 / (int)sizeof(T);
    const int lph = ch / PE;                       // lanes per head
    const int64_t npc = BT * heads * lph;          // pieces in total
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t ic = i < npc ? i : npc - 1;      // tail lanes shadow the last piece (uniform shuffles), no store
    const uint4 a = reinterpret_cast<const uint4*>(o)[ic];
    const uint4 b = reinterpret_cast<const uint4*>(dout)[ic];
    float acc;
    if constexpr (sizeof(T) == 2) {
        acc = __uint_as_float(a.x << 16) * __uint_as_float(b.x << 16);
        acc = fmaf(__uint_as_float(a.x & 0xFFFF0000u), __uint_as_float(b.x & 0xFFFF0000u), acc);
        acc = fmaf(__uint_as_float(a.y << 16), __uint_as_float(b.y << 16), acc);
        acc = fmaf(__uint_as_float(a.y & 0xFFFF0000u), __uint_as_float(b.y & 0xFFFF0000u), acc);
        acc = fmaf(__uint_as_float(a.z << 16), __uint_as_float(b.z << 16), acc);
        acc = fmaf(__uint_as_float(a.z & 0xFFFF0000u), __uint_as_float(b.z & 0xFFFF0000u), acc);
        acc = fmaf(__uint_as_float(a.w << 16), __uint_as_float(b.w << 16), acc);
        acc = fmaf(__uint_as_float(a.w & 0xFFFF0000u), __uint_as_float(b.w & 0xFFFF0000u), acc);
    } else {
        acc = __uint_as_float(a.x) * __uint_as_float(b.x);
        acc = fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
        acc = fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
        acc = fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
    }
    for (int m = 1; m < lph; m <<= 1) acc += __shfl_xor(acc, m, 64);
    if (i < npc && (i % lph) == 0) {
        const int64_t hq = i / lph;                // (b*T + q) * heads + h
        const int h = (int)(hq % heads);
        const int64_t bq = hq / heads;
        const int64_t b_ = bq / T_, q = bq % T_;
        delta[(b_ * heads + h) * T_ + q] = acc;
    }
}

// ------------------------------------------------------------------------------------------------ LDS tile layout
// A [row][CH] bf16 tile is read two ways: whole 16-byte row pieces (ds_read_b128, 16 different rows per lane group)
// and transposed 4-row x 64-byte windows (ds_read_b64_tr_b16, one half-wave = rows r..r+3 of one window).  A padded
// pitch serves only one of them (odd 16-byte-slot pitch: b128 conflict-free, the 4 windows overlap 4-fold - PMC:
// SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE in these kernels).  For CH = 128 the rows are therefore unpadded
// (256 B = all 64 banks) and the 16-byte slot is XORed with a bit-permutation of the row index: bits 0-1 of the row
// pick the 64-byte window (4 consecutive rows -> 4 windows: tr reads conflict-free), bits 2-3 the slot inside it
// (any 16 rows distinct mod 16 -> 16 distinct slots: b128 reads conflict-free).
template <int CH>
struct RowTile {
    static constexpr bool SWZ = (CH == 128);
    static constexpr int P = SWZ ? 256 : CH * 2 + 16;
    static __device__ __forceinline__ int swz(int row) { return SWZ ? ((((row & 3) << 2) | ((row >> 2) & 3)) << 4) : 0; }
    // byte offset of 16-byte-aligned `byte` (may carry an 8-byte sub-offset) in `row`
    static __device__ __forceinline__ int at(int row, int byte) { return row * P + (byte ^ swz(row)); }
};

// ------------------------------------------------------------------------------------------------ dQ (bf16)
template <int CH, int KT>
__global__ __launch_bounds__(256, CH <= 128 ? 2 : 1) void k_attn_dq(const bf16_raw* __restrict__ qk, const bf16_raw* __restrict__ vt,
                                                 const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
                                                 const float* __restrict__ delta, bf16_raw* __restrict__ dqk, int T, int C,
                                                 float scale_log2e, float scale, int dqk_rs) {
    using KL = RowTile<CH>;
    constexpr int KP = KL::P;
    constexpr int VP = (KT == 64) ? 192 : KT * 2 + 16;   // [ch][key] tile, transposing reads only: pitch = 64 (mod 256)
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
    const float nlse_q = -lse[((size_t)b * heads + h) * T + qc];
    const float ndel_q = -delta[((size_t)b * heads + h) * T + qc] * scale;

    f32x16_t dq[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[ct][r] = 0.0f;

    const int prow = pi_row(col);
    const int pswap = (pp == 1) ? 2 : (pp == 2 ? 1 : pp);     // pi applied to the 4-column chunk a lane addresses
    const bool vec_v = ((T & 7) == 0);

    constexpr int KPC = CH / 8;
    constexpr int KIT = (KT * KPC + 255) / 256;
    constexpr bool KEXACT = (KIT * 256 == KT * KPC);
    constexpr int VPC = KT / 8;
    constexpr int VIT = (NCT * 32 * VPC + 255) / 256;
    constexpr bool VEXACT = (VIT * 256 == NCT * 32 * VPC);
    const bf16_raw* const kbase = qk + (size_t)b * T * row2c + C + (size_t)h * CH;
    const bf16_raw* const vbase = vt + ((size_t)b * C + (size_t)h * CH) * T;

    // K / V^T tiles are loaded, waited for and written between the two barriers (PRE = false).  Fetching tile t + 1 under the MFMAs
    // of tile t (T14, as the forward kernel does) needs 16 more registers: CH = 64 then drops from three waves per SIMD to two and
    // measures 0.8 % slower (c5 shape), CH = 128 already uses all 256.  Whole tiles (all but a ragged last one) use a scalar tile
    // origin + per-thread 32-bit offsets computed once instead of clamps, selects and 64-bit multiplies per tile.
    constexpr bool PRE = false;
    constexpr bool FAST = KEXACT && VEXACT && (CH % 32 == 0);
    uint4 kv[KIT], vv[VIT];
    unsigned koff[KIT], voff[VIT], klds[KIT], vlds[VIT];
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
        const int pc = tid + 256 * i, key = pc / KPC, piece = pc % KPC;
        koff[i] = (unsigned)(((size_t)key * row2c + piece * 8) * 2);
        klds[i] = (unsigned)KL::at(key, piece * 16);
    }
#pragma unroll
    for (int i = 0; i < VIT; ++i) {
        const int pc = tid + 256 * i, c = pc / VPC, piece = pc % VPC;
        voff[i] = (unsigned)(((size_t)(c < CH ? c : 0) * T + piece * 8) * 2);
        vlds[i] = (unsigned)(c * VP + piece * 16);
    }
    const bool fast_ok = FAST && vec_v && (size_t)CH * T * 2 < (1ull << 32);
    auto load_kv = [&](int kt0) {
        if (fast_ok && kt0 + KT <= T) {                     // (uniform)
            const char* const kb = reinterpret_cast<const char*>(kbase) + (size_t)kt0 * row2c * 2;
            const char* const vb = reinterpret_cast<const char*>(vbase) + (size_t)kt0 * 2;
#pragma unroll
            for (int i = 0; i < KIT; ++i) kv[i] = *reinterpret_cast<const uint4*>(kb + koff[i]);
#pragma unroll
            for (int i = 0; i < VIT; ++i) vv[i] = *reinterpret_cast<const uint4*>(vb + voff[i]);
            return;
        }
#pragma unroll
        for (int i = 0; i < KIT; ++i) {
            const int pc = tid + 256 * i;
            const int key = KEXACT ? pc / KPC : min(pc / KPC, KT - 1), piece = pc % KPC;
            kv[i] = *reinterpret_cast<const uint4*>(kbase + (size_t)min(kt0 + key, T - 1) * row2c + piece * 8);
        }
        if (vec_v) {
#pragma unroll
            for (int i = 0; i < VIT; ++i) {
                const int pc = tid + 256 * i;
                const int c = pc / VPC, piece = pc % VPC;
                const bool ok = (VEXACT || pc < NCT * 32 * VPC) && c < CH && kt0 + piece * 8 < T;
                vv[i] = *reinterpret_cast<const uint4*>(vbase + (size_t)(ok ? c : 0) * T + (ok ? kt0 + piece * 8 : 0));
            }
        }
    };
    auto store_kv = [&](int kt0) {
        if (fast_ok && kt0 + KT <= T) {
#pragma unroll
            for (int i = 0; i < VIT; ++i) *reinterpret_cast<uint4*>(v_lds + vlds[i]) = vv[i];
#pragma unroll
            for (int i = 0; i < KIT; ++i) *reinterpret_cast<uint4*>(k_lds + klds[i]) = kv[i];
            return;
        }
        if (vec_v) {
#pragma unroll
            for (int i = 0; i < VIT; ++i) {
                const int pc = tid + 256 * i;
                const int c = pc / VPC, piece = pc % VPC;
                const bool ok = (VEXACT || pc < NCT * 32 * VPC) && c < CH && kt0 + piece * 8 < T;
                if (VEXACT || pc < NCT * 32 * VPC) *reinterpret_cast<uint4*>(v_lds + c * VP + piece * 16) = ok ? vv[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        } else {
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
                *reinterpret_cast<uint4*>(k_lds + KL::at(key, piece * 16)) = (kt0 + key < T) ? kv[i] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    if constexpr (PRE) load_kv(0);
    for (int kt0 = 0; kt0 < T; kt0 += KT) {
        __syncthreads();
        if constexpr (!PRE) load_kv(kt0);
        store_kv(kt0);
        __syncthreads();
        if constexpr (PRE) {
            if (kt0 + KT < T) load_kv(kt0 + KT);            // in flight under this tile's MFMAs
        }

#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const int krow = 32 * u + prow;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const uint4 a = *reinterpret_cast<const uint4*>(k_lds + KL::at(krow, 16 * half + 32 * kk));
                s = mma_bf16(a, qf[kk], s);
                // dP^T[key][q]: A = V[key][ch] read transposed out of the [ch][key] tile, rows in pi order
                const int r0 = (16 * kk + 8 * (grp >> 1) + qq) * VP + (32 * u + 16 * (grp & 1) + 4 * pswap) * 2;
                const uint4 av = tr_frag2(v_lds, r0, r0 + 4 * VP);
                dp = mma_bf16(av, dof[kk], dp);
            }
            // dS^T = P^T * (dP^T - delta) * scale; masked keys (partial last tile only) -> P = exp2(-inf) = 0
            if (kt0 + KT > T) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (kt0 + 32 * u + pi_row(row) >= T) s[r] = -INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(s[r], scale_log2e, nlse_q));
                s[r] = pv * fmaf(dp[r], scale, ndel_q);
            }
            // dQ^T[c][q] += K^T[c][key] * dS^T[key][q]
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 bs = acc_to_frag(s, st);
                const int rk = 32 * u + 16 * st + 8 * (grp >> 1) + qq;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int cb = (32 * ct + 16 * (grp & 1) + 4 * pp) * 2;
                    const uint4 ak = tr_frag2(k_lds, KL::at(rk, cb), KL::at(rk + 4, cb));
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

// ------------------------------------------------------------------------------------------------ dK / dV (bf16)
// Key-stationary (128 keys / workgroup, key on the lane), sweeping query tiles.  One kernel per gradient
// (WHAT = 0: dV = P^T dO, WHAT = 1: dK = dS^T Q): together they hold two 64-register accumulators plus the K and V
// operand fragments (another 64), which forces one wave per SIMD with every LDS / barrier wait exposed; apart they run
// two workgroups per CU, for one extra recompute of S.
template <int CH, int QT, int WHAT>
__global__ __launch_bounds__(256, CH <= 128 ? 2 : 1) void k_attn_dkv(const bf16_raw* __restrict__ qk, const bf16_raw* __restrict__ vt,
                                                  const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
                                                  const float* __restrict__ delta, bf16_raw* __restrict__ dst, int T, int C,
                                                  float scale_log2e, float scale, int dst_rs, int dst_off) {
    using QL = RowTile<CH>;
    constexpr int KP = QL::P;
    constexpr int NKK = CH / 16;
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = QT / 32;
    constexpr bool DK = (WHAT == 1);
    __shared__ __attribute__((aligned(16))) char q_lds[QT * KP];                 // Q  [query][ch]
    __shared__ __attribute__((aligned(16))) char d_lds[QT * KP];                 // dO [query][ch]
    __shared__ __attribute__((aligned(16))) float nlse_s[QT];                    // -lse           (+inf rows: masked)
    __shared__ __attribute__((aligned(16))) float ndel_s[DK ? QT : 4];           // -delta * scale
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int k0 = blockIdx.x * 128;
    const int kj = k0 + wave * 32 + col;
    const int kc = kj < T ? kj : T - 1;
    const size_t row2c = (size_t)2 * C;

    // K fragments (B operand: lane = key, 8 consecutive channels)
    uint4 kf[NKK], vf[DK ? NKK : 1];
    {
        const bf16_raw* kp = qk + ((size_t)b * T + kc) * row2c + C + (size_t)h * CH + 8 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) kf[kk] = *reinterpret_cast<const uint4*>(kp + 16 * kk);
    }
    // V fragments (B operand: lane = key, k = channel): V^T is channel-major, so each element is its own
    // 2-byte load -- once per workgroup, outside the query sweep
    if constexpr (DK) {
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

    f32x16_t acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.0f;
    const int prow = pi_row(col);

    constexpr int QPC = CH / 8;
    constexpr int QIT = (QT * QPC + 255) / 256;
    constexpr bool QEXACT = (QIT * 256 == QT * QPC);
    const bf16_raw* const qbase = qk + (size_t)b * T * row2c + (size_t)h * CH;
    const bf16_raw* const dbase = dout + (size_t)b * T * C + (size_t)h * CH;
    const float* const lbase = lse + ((size_t)b * heads + h) * T;
    const float* const ebase = delta + ((size_t)b * heads + h) * T;

    // Q / dO tiles.  PRE (CH <= 64): issue-early / write-late (T14), tile t + 1 fetched under the MFMAs of tile t; CH = 128 has no
    // registers for that (232 of 256) and loads between the barriers.  Whole tiles use a scalar tile origin + per-thread 32-bit
    // offsets computed once instead of clamps, selects and 64-bit multiplies per tile.
    constexpr bool PRE = (CH <= 64);
    uint4 vq[QIT], vd[QIT];
    float nl_r = 0.0f, nd_r = 0.0f;
    unsigned qoff[QIT], doff[QIT], qlds[QIT];
#pragma unroll
    for (int i = 0; i < QIT; ++i) {
        const int pc = tid + 256 * i, q = pc / QPC, piece = pc % QPC;
        qoff[i] = (unsigned)(((size_t)q * row2c + piece * 8) * 2);
        doff[i] = (unsigned)(((size_t)q * C + piece * 8) * 2);
        qlds[i] = (unsigned)QL::at(q, piece * 16);
    }
    auto load_q = [&](int qt0) {
        if (QEXACT && qt0 + QT <= T) {                       // (uniform)
            const char* const qb = reinterpret_cast<const char*>(qbase) + (size_t)qt0 * row2c * 2;
            const char* const db = reinterpret_cast<const char*>(dbase) + (size_t)qt0 * C * 2;
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                vq[i] = *reinterpret_cast<const uint4*>(qb + qoff[i]);
                vd[i] = *reinterpret_cast<const uint4*>(db + doff[i]);
            }
            if (tid < QT) {
                nl_r = -lbase[qt0 + tid];
                if constexpr (DK) nd_r = -ebase[qt0 + tid] * scale;
            }
        } else {
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                const int pc = tid + 256 * i;
                const int q = QEXACT ? pc / QPC : min(pc / QPC, QT - 1), piece = pc % QPC;
                const size_t qr = (size_t)min(qt0 + q, T - 1);
                vq[i] = *reinterpret_cast<const uint4*>(qbase + qr * row2c + piece * 8);
                vd[i] = *reinterpret_cast<const uint4*>(dbase + qr * C + piece * 8);
            }
            if (tid < QT) {
                const bool ok = qt0 + tid < T;
                nl_r = ok ? -lbase[qt0 + tid] : -INFINITY;          // exp2(-inf) = 0 masks the row
                if constexpr (DK) nd_r = ok ? -ebase[qt0 + tid] * scale : 0.0f;
            }
        }
    };
    auto store_q = [&](int qt0) {
        if (tid < QT) {
            nlse_s[tid] = nl_r;
            if constexpr (DK) ndel_s[tid] = nd_r;
        }
        if (QEXACT && qt0 + QT <= T) {
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                *reinterpret_cast<uint4*>(q_lds + qlds[i]) = vq[i];
                *reinterpret_cast<uint4*>(d_lds + qlds[i]) = vd[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                const int pc = tid + 256 * i;
                const int q = pc / QPC, piece = pc % QPC;
                if (QEXACT || pc < QT * QPC) {
                    const bool ok = qt0 + q < T;
                    *reinterpret_cast<uint4*>(q_lds + QL::at(q, piece * 16)) = ok ? vq[i] : make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(d_lds + QL::at(q, piece * 16)) = ok ? vd[i] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
    };
    if constexpr (PRE) load_q(0);
    for (int qt0 = 0; qt0 < T; qt0 += QT) {
        __syncthreads();
        if constexpr (!PRE) load_q(qt0);
        store_q(qt0);
        __syncthreads();
        if constexpr (PRE) {
            if (qt0 + QT < T) load_q(qt0 + QT);             // in flight under this tile's MFMAs
        }

#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const int qrow = 32 * u + prow;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const uint4 aq = *reinterpret_cast<const uint4*>(q_lds + QL::at(qrow, 16 * half + 32 * kk));
                s = mma_bf16(aq, kf[kk], s);                 // S[q][key]
                if constexpr (DK) {
                    const uint4 ad = *reinterpret_cast<const uint4*>(d_lds + QL::at(qrow, 16 * half + 32 * kk));
                    dp = mma_bf16(ad, vf[kk], dp);           // dP[q][key]
                }
            }
            // accumulator registers 4g..4g+3 hold the 4 consecutive queries 32u + pi(8g + 4 half) + {0..3}
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q4 = 32 * u + pi_row(8 * g + 4 * half);
                const float4 nl = *reinterpret_cast<const float4*>(&nlse_s[q4]);
                const float nlv[4] = {nl.x, nl.y, nl.z, nl.w};
                if constexpr (DK) {
                    const float4 nd = *reinterpret_cast<const float4*>(&ndel_s[q4]);
                    const float ndv[4] = {nd.x, nd.y, nd.z, nd.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float pv = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], scale_log2e, nlv[j]));
                        s[4 * g + j] = pv * fmaf(dp[4 * g + j], scale, ndv[j]);     // dS
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s[4 * g + j] = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], scale_log2e, nlv[j]));   // P
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 bf = acc_to_frag(s, st);
                const int rq = 32 * u + 16 * st + 8 * (grp >> 1) + qq;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int cb = (32 * ct + 16 * (grp & 1) + 4 * pp) * 2;
                    // dV: dO^T[c][q] * P[q][key];  dK: Q^T[c][q] * dS[q][key]
                    const uint4 a = tr_frag2(DK ? q_lds : d_lds, QL::at(rq, cb), QL::at(rq + 4, cb));
                    acc[ct] = mma_bf16(a, bf, acc[ct]);
                }
            }
        }
    }
    if (kj < T) {
        bf16_raw* op = dst + ((size_t)b * T + kj) * dst_rs + dst_off + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH)
                    *reinterpret_cast<uint2*>(op + c) = make_uint2(pack_bf16x2(acc[ct][4 * rg + 0], acc[ct][4 * rg + 1]),
                                                                   pack_bf16x2(acc[ct][4 * rg + 2], acc[ct][4 * rg + 3]));
            }
    }
}

// ------------------------------------------------------------------------------------------------ dK and dV in one kernel (bf16)
// Round 3: the key-stationary pair above recomputes S once per gradient (5 contractions: S, P^T dO | S, dP, dS^T Q).  Fused, S is
// computed once (4 contractions, -20 % MFMAs of the pair).  What kept r2 from fusing was registers: two 64-register accumulators
// plus the K and V B-operand fragments (64) left one wave per SIMD.  Here the V fragments (lane = key, 8 consecutive channels) stay
// in LDS - the workgroup's 128 V rows are staged once as [key][ch] (transposed out of the channel-major vt) and read per k-step -
// so a wave holds 128 (accumulators) + 32 (K fragments) + 32 (S, dP) + operands: two workgroups per CU as before.
template <int CH, int QT>
__global__ __launch_bounds__(256, 2) void k_attn_dkv_fused(const bf16_raw* __restrict__ qk, const bf16_raw* __restrict__ vt,
                                                           const bf16_raw* __restrict__ dout, const float* __restrict__ lse,
                                                           const float* __restrict__ delta, bf16_raw* __restrict__ dqk,
                                                           bf16_raw* __restrict__ dv, int T, int C, float scale_log2e, float scale,
                                                           int dqk_rs, int dv_rs) {
    using QL = RowTile<CH>;
    constexpr int KP = QL::P;
    constexpr int VBP = CH * 2 + 16;      // V rows [key][ch]: odd number of 16-byte slots, conflict-free ds_read_b128
    constexpr int NKK = CH / 16;
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = QT / 32;
    static_assert(CH % 32 == 0, "fused dK / dV: whole 32-channel tiles");
    __shared__ __attribute__((aligned(16))) char q_lds[QT * KP];                 // Q  [query][ch]
    __shared__ __attribute__((aligned(16))) char d_lds[QT * KP];                 // dO [query][ch]
    __shared__ __attribute__((aligned(16))) char vb_lds[128 * VBP];              // V  [key][ch] of this workgroup's keys
    __shared__ __attribute__((aligned(16))) float nlse_s[QT];
    __shared__ __attribute__((aligned(16))) float ndel_s[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int grp = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int k0 = blockIdx.x * 128;
    const int kj = k0 + wave * 32 + col;
    const int kc = kj < T ? kj : T - 1;
    const size_t row2c = (size_t)2 * C;

    uint4 kf[NKK];
    {
        const bf16_raw* kp = qk + ((size_t)b * T + kc) * row2c + C + (size_t)h * CH + 8 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) kf[kk] = *reinterpret_cast<const uint4*>(kp + 16 * kk);
    }
    // V rows of the workgroup's 128 keys: vt is channel-major [ch][T] - a thread reads 8 consecutive keys of one channel (16 bytes)
    // and scatters them into 8 rows; once per workgroup, outside the query sweep
    {
        const bf16_raw* vbase = vt + ((size_t)b * C + (size_t)h * CH) * T;
        const bool vec = ((T & 7) == 0);
        for (int pc = tid; pc < CH * 16; pc += 256) {
            const int c = pc >> 4, piece = pc & 15;
            const int key0 = k0 + piece * 8;
            bf16_raw e[8];
            if (vec && key0 + 8 <= T) {
                *reinterpret_cast<uint4*>(e) = *reinterpret_cast<const uint4*>(vbase + (size_t)c * T + key0);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) e[j] = (key0 + j < T) ? vbase[(size_t)c * T + key0 + j] : (bf16_raw)0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<bf16_raw*>(vb_lds + (piece * 8 + j) * VBP + c * 2) = e[j];
        }
    }
    const char* const vrow = vb_lds + (wave * 32 + col) * VBP + 16 * half;

    f32x16_t adv[NCT], adk[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { adv[ct][r] = 0.0f; adk[ct][r] = 0.0f; }
    const int prow = pi_row(col);

    constexpr int QPC = CH / 8;
    constexpr int QIT = (QT * QPC + 255) / 256;
    constexpr bool QEXACT = (QIT * 256 == QT * QPC);
    const bf16_raw* const qbase = qk + (size_t)b * T * row2c + (size_t)h * CH;
    const bf16_raw* const dbase = dout + (size_t)b * T * C + (size_t)h * CH;
    const float* const lbase = lse + ((size_t)b * heads + h) * T;
    const float* const ebase = delta + ((size_t)b * heads + h) * T;

    // Q / dO tiles: issue-early / write-late as in the forward kernel (cdna_hip_programming.md T14) - tile t + 1 is fetched into
    // registers under the MFMAs of tile t and written to LDS between the two barriers; whole tiles (all but a ragged last one) use a
    // scalar tile origin + per-thread 32-bit offsets computed once instead of clamps, selects and 64-bit multiplies per tile.
    uint4 vq[QIT], vd[QIT];
    float nl_r = 0.0f, nd_r = 0.0f;
    unsigned qoff[QIT], doff[QIT], qlds[QIT];
#pragma unroll
    for (int i = 0; i < QIT; ++i) {
        const int pc = tid + 256 * i, q = pc / QPC, piece = pc % QPC;
        qoff[i] = (unsigned)(((size_t)q * row2c + piece * 8) * 2);
        doff[i] = (unsigned)(((size_t)q * C + piece * 8) * 2);
        qlds[i] = (unsigned)QL::at(q, piece * 16);
    }
    auto load_q = [&](int qt0) {
        if (QEXACT && qt0 + QT <= T) {                       // (uniform)
            const char* const qb = reinterpret_cast<const char*>(qbase) + (size_t)qt0 * row2c * 2;
            const char* const db = reinterpret_cast<const char*>(dbase) + (size_t)qt0 * C * 2;
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                vq[i] = *reinterpret_cast<const uint4*>(qb + qoff[i]);
                vd[i] = *reinterpret_cast<const uint4*>(db + doff[i]);
            }
            if (tid < QT) {
                nl_r = -lbase[qt0 + tid];
                nd_r = -ebase[qt0 + tid] * scale;
            }
        } else {
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                const int pc = tid + 256 * i;
                const int q = QEXACT ? pc / QPC : min(pc / QPC, QT - 1), piece = pc % QPC;
                const size_t qr = (size_t)min(qt0 + q, T - 1);
                vq[i] = *reinterpret_cast<const uint4*>(qbase + qr * row2c + piece * 8);
                vd[i] = *reinterpret_cast<const uint4*>(dbase + qr * C + piece * 8);
            }
            if (tid < QT) {
                const bool ok = qt0 + tid < T;
                nl_r = ok ? -lbase[qt0 + tid] : -INFINITY;
                nd_r = ok ? -ebase[qt0 + tid] * scale : 0.0f;
            }
        }
    };
    auto store_q = [&](int qt0) {
        if (tid < QT) {
            nlse_s[tid] = nl_r;
            ndel_s[tid] = nd_r;
        }
        if (QEXACT && qt0 + QT <= T) {
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                *reinterpret_cast<uint4*>(q_lds + qlds[i]) = vq[i];
                *reinterpret_cast<uint4*>(d_lds + qlds[i]) = vd[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < QIT; ++i) {
                const int pc = tid + 256 * i;
                const int q = pc / QPC, piece = pc % QPC;
                if (QEXACT || pc < QT * QPC) {
                    const bool ok = qt0 + q < T;
                    *reinterpret_cast<uint4*>(q_lds + QL::at(q, piece * 16)) = ok ? vq[i] : make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(d_lds + QL::at(q, piece * 16)) = ok ? vd[i] : make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
    };
    load_q(0);
    for (int qt0 = 0; qt0 < T; qt0 += QT) {
        __syncthreads();                  // the previous tile is consumed (first pass: also publishes vb_lds)
        store_q(qt0);
        __syncthreads();
        if (qt0 + QT < T) load_q(qt0 + QT);      // in flight under this tile's MFMAs

#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const int qrow = 32 * u + prow;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const uint4 aq = *reinterpret_cast<const uint4*>(q_lds + QL::at(qrow, 16 * half + 32 * kk));
                const uint4 ad = *reinterpret_cast<const uint4*>(d_lds + QL::at(qrow, 16 * half + 32 * kk));
                const uint4 vf = *reinterpret_cast<const uint4*>(vrow + 32 * kk);
                s = mma_bf16(aq, kf[kk], s);                 // S[q][key]
                dp = mma_bf16(ad, vf, dp);                   // dP[q][key]
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q4 = 32 * u + pi_row(8 * g + 4 * half);
                const float4 nl = *reinterpret_cast<const float4*>(&nlse_s[q4]);
                const float4 nd = *reinterpret_cast<const float4*>(&ndel_s[q4]);
                const float nlv[4] = {nl.x, nl.y, nl.z, nl.w}, ndv[4] = {nd.x, nd.y, nd.z, nd.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(s[4 * g + j], scale_log2e, nlv[j]));
                    s[4 * g + j] = pv;                                             // P
                    dp[4 * g + j] = pv * fmaf(dp[4 * g + j], scale, ndv[j]);       // dS
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint4 bp = acc_to_frag(s, st), bs = acc_to_frag(dp, st);
                const int rq = 32 * u + 16 * st + 8 * (grp >> 1) + qq;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int cb = (32 * ct + 16 * (grp & 1) + 4 * pp) * 2;
                    const uint4 aD = tr_frag2(d_lds, QL::at(rq, cb), QL::at(rq + 4, cb));
                    adv[ct] = mma_bf16(aD, bp, adv[ct]);     // dV^T[c][key] += dO^T[c][q] P[q][key]
                    const uint4 aQ = tr_frag2(q_lds, QL::at(rq, cb), QL::at(rq + 4, cb));
                    adk[ct] = mma_bf16(aQ, bs, adk[ct]);     // dK^T[c][key] += Q^T[c][q] dS[q][key]
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
                *reinterpret_cast<uint2*>(okp + c) = make_uint2(pack_bf16x2(adk[ct][4 * rg + 0], adk[ct][4 * rg + 1]),
                                                                pack_bf16x2(adk[ct][4 * rg + 2], adk[ct][4 * rg + 3]));
                *reinterpret_cast<uint2*>(ovp + c) = make_uint2(pack_bf16x2(adv[ct][4 * rg + 0], adv[ct][4 * rg + 1]),
                                                                pack_bf16x2(adv[ct][4 * rg + 2], adv[ct][4 * rg + 3]));
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

// ------------------------------------------------------------------------------------------------ exact-f32 on the matrix cores
// Round 3: the two kernels above on v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate = an fmaf chain per output; 64 FLOP / clk / SIMD),
// same flash structure as the bf16 pair: the "stationary" index (query for dQ, key for dK / dV) sits on the lane, so the softmax
// terms are lane-local, and accumulator register r of a score tile holds row (r & 3) + 8 (r >> 2) + 4 half - exactly the two rows
// k-step r of the NEXT MFMA contracts, so P / dS feed it straight from their registers, unrounded.  (The VALU pair spent 19.6 of
// c2's 228 ms training step on 0.4 % of its FLOPs.)
//   k-step channel pairing of the QK^T / dO V^T contractions: half-wave h walks channels h * CH / 2 + {0 .. CH / 2 - 1}; any pairing
//   is valid as long as both operands agree, and this one reads whole float4 pieces.
__device__ __forceinline__ f32x16_t mma_f32(float a, float b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int CH, int KT>
__global__ __launch_bounds__(256) void k_attn_dq_f32m(const float* __restrict__ qk, const float* __restrict__ vt,
                                                      const float* __restrict__ dout, const float* __restrict__ lse,
                                                      const float* __restrict__ delta, float* __restrict__ dqk, int T, int C,
                                                      float scale_log2e, float scale, int dqk_rs) {
    constexpr int KP = CH * 4 + 16;      // K tile [key][ch], odd number of 16-byte slots per row
    constexpr int VP = KT * 4 + 16;      // V^T tile [ch][key]
    constexpr int NJ = CH / 8;
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = KT / 32;
    __shared__ __attribute__((aligned(16))) char k_lds[KT * KP];
    __shared__ __attribute__((aligned(16))) char v_lds[NCT * 32 * VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int qi = blockIdx.x * 128 + wave * 32 + col;
    const int qc = qi < T ? qi : T - 1;
    const size_t row2c = (size_t)2 * C;
    float4 qf[NJ], dof[NJ];
    {
        const float* qp = qk + ((size_t)b * T + qc) * row2c + (size_t)h * CH + half * (CH / 2);
        const float* dp = dout + ((size_t)b * T + qc) * C + (size_t)h * CH + half * (CH / 2);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            qf[j] = *reinterpret_cast<const float4*>(qp + 4 * j);
            dof[j] = *reinterpret_cast<const float4*>(dp + 4 * j);
        }
    }
    const float nlse_q = -lse[((size_t)b * heads + h) * T + qc];
    const float ndel_q = -delta[((size_t)b * heads + h) * T + qc] * scale;
    f32x16_t dq[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[ct][r] = 0.0f;
    constexpr int KPC = CH / 4, VPC = KT / 4;
    const bool vec_v = ((T & 3) == 0);
    const float* const kbase = qk + (size_t)b * T * row2c + C + (size_t)h * CH;
    const float* const vbase = vt + ((size_t)b * C + (size_t)h * CH) * T;

    for (int kt0 = 0; kt0 < T; kt0 += KT) {
        __syncthreads();
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
        } else {
            for (int e = tid; e < NCT * 32 * KT; e += 256) {
                const int c = e / KT, key = e % KT;
                *reinterpret_cast<float*>(v_lds + c * VP + key * 4) = (c < CH && kt0 + key < T) ? vbase[(size_t)c * T + kt0 + key] : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const char* kp = k_lds + (32 * u + col) * KP + half * (CH / 2) * 4;
            const char* vp = v_lds + (half * (CH / 2)) * VP + (32 * u + col) * 4;     // V^T[c][key = lane]: one float per k-step
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(kp + 16 * j);
                s = mma_f32(a.x, qf[j].x, s); s = mma_f32(a.y, qf[j].y, s); s = mma_f32(a.z, qf[j].z, s); s = mma_f32(a.w, qf[j].w, s);
                dp = mma_f32(*reinterpret_cast<const float*>(vp + (4 * j + 0) * VP), dof[j].x, dp);
                dp = mma_f32(*reinterpret_cast<const float*>(vp + (4 * j + 1) * VP), dof[j].y, dp);
                dp = mma_f32(*reinterpret_cast<const float*>(vp + (4 * j + 2) * VP), dof[j].z, dp);
                dp = mma_f32(*reinterpret_cast<const float*>(vp + (4 * j + 3) * VP), dof[j].w, dp);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt0 + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float pv = key < T ? exp2f(fmaf(s[r], scale_log2e, nlse_q)) : 0.0f;
                s[r] = pv * fmaf(dp[r], scale, ndel_q);            // dS^T[key][q]
            }
            // dQ^T[c][q] += K^T[c][key] * dS^T[key][q]: k-step r contracts the keys of accumulator register r
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const char* ka = k_lds + (32 * u + 4 * half) * KP + (32 * ct + col) * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dq[ct] = mma_f32(*reinterpret_cast<const float*>(ka + ((r & 3) + 8 * (r >> 2)) * KP), s[r], dq[ct]);
            }
        }
    }
    if (qi < T) {
        float* op = dqk + ((size_t)b * T + qi) * dqk_rs + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH) *reinterpret_cast<float4*>(op + c) = make_float4(dq[ct][4 * rg + 0], dq[ct][4 * rg + 1], dq[ct][4 * rg + 2], dq[ct][4 * rg + 3]);
            }
    }
}

template <int CH, int QT>
__global__ __launch_bounds__(256) void k_attn_dkv_f32m(const float* __restrict__ qk, const float* __restrict__ vt,
                                                       const float* __restrict__ dout, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, float* __restrict__ dqk, float* __restrict__ dv,
                                                       int T, int C, float scale_log2e, float scale, int dqk_rs, int dv_rs) {
    constexpr int QP = CH * 4 + 16;      // Q / dO tiles [query][ch]
    constexpr int NJ = CH / 8;
    constexpr int NCT = (CH + 31) / 32;
    constexpr int NU = QT / 32;
    __shared__ __attribute__((aligned(16))) char q_lds[QT * QP];
    __shared__ __attribute__((aligned(16))) char d_lds[QT * QP];
    __shared__ float nlse_s[QT], ndel_s[QT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int b = blockIdx.z, h = blockIdx.y, heads = gridDim.y;
    const int kj = blockIdx.x * 128 + wave * 32 + col;
    const int kc = kj < T ? kj : T - 1;
    const size_t row2c = (size_t)2 * C;
    float4 kf[NJ], vf[NJ];
    {
        const float* kp = qk + ((size_t)b * T + kc) * row2c + C + (size_t)h * CH + half * (CH / 2);
        const float* vp = vt + ((size_t)b * C + (size_t)h * CH + half * (CH / 2)) * T + kc;       // channel-major: one load per element
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            kf[j] = *reinterpret_cast<const float4*>(kp + 4 * j);
            vf[j] = make_float4(vp[(size_t)(4 * j + 0) * T], vp[(size_t)(4 * j + 1) * T], vp[(size_t)(4 * j + 2) * T], vp[(size_t)(4 * j + 3) * T]);
        }
    }
    f32x16_t dk[NCT], dvv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[ct][r] = 0.0f; dvv[ct][r] = 0.0f; }
    constexpr int QPC = CH / 4;
    const float* const qbase = qk + (size_t)b * T * row2c + (size_t)h * CH;
    const float* const dbase = dout + (size_t)b * T * C + (size_t)h * CH;
    const float* const lbase = lse + ((size_t)b * heads + h) * T;
    const float* const ebase = delta + ((size_t)b * heads + h) * T;

    for (int qt0 = 0; qt0 < T; qt0 += QT) {
        __syncthreads();
        for (int pc = tid; pc < QT * QPC; pc += 256) {
            const int q = pc / QPC, piece = pc % QPC;
            float4 vq = make_float4(0.0f, 0.0f, 0.0f, 0.0f), vd = vq;
            if (qt0 + q < T) {
                vq = *reinterpret_cast<const float4*>(qbase + (size_t)(qt0 + q) * row2c + piece * 4);
                vd = *reinterpret_cast<const float4*>(dbase + (size_t)(qt0 + q) * C + piece * 4);
            }
            *reinterpret_cast<float4*>(q_lds + q * QP + piece * 16) = vq;
            *reinterpret_cast<float4*>(d_lds + q * QP + piece * 16) = vd;
        }
        if (tid < QT) {
            const bool ok = qt0 + tid < T;
            nlse_s[tid] = ok ? -lbase[qt0 + tid] : -INFINITY;      // exp2(-inf) = 0 masks the row
            ndel_s[tid] = ok ? -ebase[qt0 + tid] * scale : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            f32x16_t s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.0f; dp[r] = 0.0f; }
            const char* qp = q_lds + (32 * u + col) * QP + half * (CH / 2) * 4;
            const char* dp_ = d_lds + (32 * u + col) * QP + half * (CH / 2) * 4;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float4 aq = *reinterpret_cast<const float4*>(qp + 16 * j);
                const float4 ad = *reinterpret_cast<const float4*>(dp_ + 16 * j);
                s = mma_f32(aq.x, kf[j].x, s); s = mma_f32(aq.y, kf[j].y, s); s = mma_f32(aq.z, kf[j].z, s); s = mma_f32(aq.w, kf[j].w, s);
                dp = mma_f32(ad.x, vf[j].x, dp); dp = mma_f32(ad.y, vf[j].y, dp); dp = mma_f32(ad.z, vf[j].z, dp); dp = mma_f32(ad.w, vf[j].w, dp);
            }
            // rows of the accumulators are queries: register r <-> query 32u + (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q4 = 32 * u + 8 * g + 4 * half;
                const float4 nl = *reinterpret_cast<const float4*>(&nlse_s[q4]);
                const float4 nd = *reinterpret_cast<const float4*>(&ndel_s[q4]);
                const float nlv[4] = {nl.x, nl.y, nl.z, nl.w}, ndv[4] = {nd.x, nd.y, nd.z, nd.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float pv = exp2f(fmaf(s[4 * g + i], scale_log2e, nlv[i]));
                    s[4 * g + i] = pv;                                             // P[q][key]
                    dp[4 * g + i] = pv * fmaf(dp[4 * g + i], scale, ndv[i]);       // dS[q][key]
                }
            }
            // dV^T[c][key] += dO^T[c][q] P[q][key];  dK^T[c][key] += Q^T[c][q] dS[q][key]
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const char* da = d_lds + (32 * u + 4 * half) * QP + (32 * ct + col) * 4;
                const char* qa = q_lds + (32 * u + 4 * half) * QP + (32 * ct + col) * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = ((r & 3) + 8 * (r >> 2)) * QP;
                    dvv[ct] = mma_f32(*reinterpret_cast<const float*>(da + ro), s[r], dvv[ct]);
                    dk[ct] = mma_f32(*reinterpret_cast<const float*>(qa + ro), dp[r], dk[ct]);
                }
            }
        }
    }
    if (kj < T) {
        float* ok_ = dqk + ((size_t)b * T + kj) * dqk_rs + C + (size_t)h * CH;
        float* ov = dv + ((size_t)b * T + kj) * dv_rs + (size_t)h * CH;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int c = 32 * ct + 8 * rg + 4 * half;
                if (c < CH) {
                    *reinterpret_cast<float4*>(ok_ + c) = make_float4(dk[ct][4 * rg + 0], dk[ct][4 * rg + 1], dk[ct][4 * rg + 2], dk[ct][4 * rg + 3]);
                    *reinterpret_cast<float4*>(ov + c) = make_float4(dvv[ct][4 * rg + 0], dvv[ct][4 * rg + 1], dvv[ct][4 * rg + 2], dvv[ct][4 * rg + 3]);
                }
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
    const int pe = dtype == RHO_BF16 ? 8 : 4;
    if (ch % pe != 0 || ch / pe > 64 || ((ch / pe) & (ch / pe - 1)) != 0) return RHO_E_SHAPE;
    const int64_t nd = batch * t * heads * (ch / pe);
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
        hipLaunchKernelGGL((k_attn_dkv<chv, ktv, 0>), grid, block, 0, st, (const bf16_raw*)qk, (const bf16_raw*)vt,              \
                           (const bf16_raw*)dout, lse, delta_ws, (bf16_raw*)dv, (int)t, C, sl2, scale, vrs, 0);                  \
        hipLaunchKernelGGL((k_attn_dkv<chv, ktv, 1>), grid, block, 0, st, (const bf16_raw*)qk, (const bf16_raw*)vt,              \
                           (const bf16_raw*)dout, lse, delta_ws, (bf16_raw*)dqk, (int)t, C, sl2, scale, qrs, C);                 \
        break;
        // ch = 64 (c5's heads): dQ + the fused dK / dV kernel; RHO_ATTN_DKV_SPLIT=1 keeps the r2 pair (A/B).  ch = 128 (c3) stays on the
        // pair: fused it needs 128 accumulator + 32 K-fragment + 32 score registers + operands > 256 per wave at two waves per SIMD
        // (hipcc: 276 bytes of scratch per lane), and one wave per SIMD is what r2 measured as slower than the extra S recompute.
        static const bool split_env = getenv("RHO_ATTN_DKV_SPLIT") && atoi(getenv("RHO_ATTN_DKV_SPLIT")) != 0;
#define RHO_ATTBF(chv, ktv)                                                                                                    \
    case chv:                                                                                                                  \
        hipLaunchKernelGGL((k_attn_dq<chv, ktv>), grid, block, 0, st, (const bf16_raw*)qk, (const bf16_raw*)vt,                  \
                           (const bf16_raw*)dout, lse, delta_ws, (bf16_raw*)dqk, (int)t, C, sl2, scale, qrs);                   \
        hipLaunchKernelGGL((k_attn_dkv_fused<chv, ktv>), grid, block, 0, st, (const bf16_raw*)qk, (const bf16_raw*)vt,           \
                           (const bf16_raw*)dout, lse, delta_ws, (bf16_raw*)dqk, (bf16_raw*)dv, (int)t, C, sl2, scale, qrs, vrs); \
        break;
        if (!split_env && ch == 64) {
            switch (ch) {
                RHO_ATTBF(64, 64)
            }
        } else {
            switch (ch) {
                RHO_ATTB(16, 64)
                RHO_ATTB(32, 64)
                RHO_ATTB(64, 64)
                RHO_ATTB(128, 64)
                RHO_ATTB(256, 32)
                default:
                    return RHO_E_SHAPE;
            }
        }
#undef RHO_ATTBF
#undef RHO_ATTB
    } else if (!(getenv("RHO_ATTN_F32_VALU") && atoi(getenv("RHO_ATTN_F32_VALU")) != 0) && ch <= 128) {
        // exact-f32 MFMA pair (ch = 256 keeps the VALU pair: two 128-register accumulators + fragments do not fit a wave)
        dim3 grid((unsigned)((t + 127) / 128), (unsigned)heads, (unsigned)batch), block(256);
#define RHO_ATTB32M(chv, ktv)                                                                                                 \
    case chv:                                                                                                                 \
        hipLaunchKernelGGL((k_attn_dq_f32m<chv, ktv>), grid, block, 0, st, (const float*)qk, (const float*)vt,                 \
                           (const float*)dout, lse, delta_ws, (float*)dqk, (int)t, C, sl2, scale, qrs);                        \
        hipLaunchKernelGGL((k_attn_dkv_f32m<chv, ktv>), grid, block, 0, st, (const float*)qk, (const float*)vt,                \
                           (const float*)dout, lse, delta_ws, (float*)dqk, (float*)dv, (int)t, C, sl2, scale, qrs, vrs);       \
        break;
        switch (ch) {
            RHO_ATTB32M(16, 64)
            RHO_ATTB32M(32, 64)
            RHO_ATTB32M(64, 32)
            RHO_ATTB32M(128, 32)
            default:
                return RHO_E_SHAPE;
        }
#undef RHO_ATTB32M
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
