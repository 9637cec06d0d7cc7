// The two 3x3x3 convolutions at the ends of the 3-D UNet have ONE channel on one side: the stem `conv_nd(dims, 1, mc, 3)`
// (rho_diffusion/models/unet_v2.py:535) and the head `conv_nd(dims, mc, 1, 3)` behind GroupNorm + SiLU (:679-683).  As implicit
// GEMMs over channels they pad that side to 32 (31/32 of the matrix work on zeros); round 2 ran them as 1x1x1 GEMMs with an HBM-rate
// helper on the other side (k_im2col_taps / k_tap_gather_sum: 2.1 ms of a 115 ms c3 sampling step, both launches bound by the fixed
// per-tile cost of a K = 32 / K = 64 GEMM and by a 0.5 GB intermediate).  Here each is ONE kernel whose contraction axis is the 27
// taps (stem) or the channels with the taps as output rows (head), and the intermediate lives in LDS:
//   stem:  y[p][co] = b[co] + sum_tap W[co][tap] x[p + off(tap)]            im2col in registers out of a 2.4 KB fp32 halo tile
//   head:  out[p]   = b + sum_tap T[p + off(tap)][tap],  T[q][tap] = sum_c W[tap][c] act(a_c x[q][c] + b_c)   T in LDS (bf16)
// Inference plans of the bf16 engine only (training keeps the 3x3x3 form: its backward needs that launch structure).
#include "common.h"

namespace {

constexpr int TD = 4, TH = 8, TW = 8;                 // output tile: 256 positions = 8 groups of 32 (MFMA columns)
constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;  // its halo: 600 positions

__device__ __forceinline__ f32x16_t mma32(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------ stem
// grid (tiles, 1, N), 256 threads.  MFMA column c of group g = tile position (pd = g >> 1, ph = 4 (g & 1) + (c >> 3), pw = c & 7).
template <int NCT>     // 32-row cout tiles (cout = 32 * NCT)
__global__ __launch_bounds__(256) void k_stem3d(const float* __restrict__ x, const bf16_raw* __restrict__ w, const float* __restrict__ bias,
                                                bf16_raw* __restrict__ y, float* __restrict__ stats, int D, int H, int W, int tiles_h,
                                                int tiles_w) {
    constexpr int COUT = 32 * NCT;
    constexpr int ROWB = COUT * 2 + 16;               // staged output row pitch (bytes): odd number of 16-byte slots
    __shared__ float halo[HD * HH * HW];
    __shared__ __attribute__((aligned(16))) char stg[256 * ROWB];      // (>= 20 KB: reused for the statistics reduction, 16 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int n = blockIdx.z, tiles = gridDim.x;
    const int bt = blockIdx.x;
    const int tw_i = bt % tiles_w, th_i = (bt / tiles_w) % tiles_h, td_i = bt / (tiles_w * tiles_h);
    const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
    const size_t S = (size_t)D * H * W;
    const float* const xs = x + (size_t)n * S;

    for (int i = tid; i < HD * HH * HW; i += 256) {
        const int iw = i % HW, ih = (i / HW) % HH, id = i / (HW * HH);
        const int gd = od0 + id - 1, gh = oh0 + ih - 1, gw = ow0 + iw - 1;
        const bool ok = (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        halo[i] = ok ? xs[((size_t)gd * H + gh) * W + gw] : 0.0f;
    }
    // A fragments: W[co][tap] (prepared [coutp][32], taps 27 .. 31 zero): row = 32 ct + col, k-piece = 16 s + 8 half
    uint4 wa[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int s = 0; s < 2; ++s) wa[ct][s] = *reinterpret_cast<const uint4*>(w + (size_t)(32 * ct + col) * 32 + 16 * s + 8 * half);
    // halo offsets of the 16 taps this lane gathers (tap = 16 s + 8 half + j), -1 = padding tap
    int toff[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tap = 16 * s + 8 * half + j;
            toff[s][j] = tap < 27 ? ((tap / 9) * HH + (tap / 3) % 3) * HW + tap % 3 : -1;
        }
    __syncthreads();

#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int g = wave * 2 + gi;
        const int pd = g >> 1, ph = 4 * (g & 1) + (col >> 3), pw = col & 7;
        const int base = (pd * HH + ph) * HW + pw;
        uint4 bf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = toff[s][j] >= 0 ? halo[base + toff[s][j]] : 0.0f;
            bf[s] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        }
        const int prow = g * 32 + col;                 // position index inside the tile
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x16_t acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            acc = mma32(wa[ct][0], bf[0], acc);
            acc = mma32(wa[ct][1], bf[1], acc);
            // accumulator register 4 q + e = cout 32 ct + 8 q + 4 half + e of this lane's position
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = 32 * ct + 8 * q + 4 * half;
                const float4 b4 = *reinterpret_cast<const float4*>(bias + co);
                *reinterpret_cast<uint2*>(stg + prow * ROWB + co * 2) =
                    make_uint2(pack_bf16x2(acc[4 * q + 0] + b4.x, acc[4 * q + 1] + b4.y), pack_bf16x2(acc[4 * q + 2] + b4.z, acc[4 * q + 3] + b4.w));
            }
        }
    }
    __syncthreads();
    // coalesced stores (16-byte pieces of a channels-last row) + the GroupNorm statistics of the stored values
    constexpr int PPR = COUT / 8;                      // pieces per row
    constexpr int NIT = 256 * PPR / 256;
    const int piece = tid % PPR;
    float ssum[8], ssq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) ssum[e] = ssq[e] = 0.0f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int r = tid / PPR + k * (256 / PPR);
        const int g = r >> 5, c = r & 31;
        const int od = od0 + (g >> 1), oh = oh0 + 4 * (g & 1) + (c >> 3), ow = ow0 + (c & 7);
        if (od < D && oh < H && ow < W) {
            const uint4 o = *reinterpret_cast<const uint4*>(stg + r * ROWB + piece * 16);
            *reinterpret_cast<uint4*>(y + ((size_t)n * S + ((size_t)od * H + oh) * W + ow) * COUT + piece * 8) = o;
            const uint32_t ws[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = __uint_as_float(ws[e] << 16), hi = __uint_as_float(ws[e] & 0xFFFF0000u);
                ssum[2 * e] += lo; ssq[2 * e] = fmaf(lo, lo, ssq[2 * e]);
                ssum[2 * e + 1] += hi; ssq[2 * e + 1] = fmaf(hi, hi, ssq[2 * e + 1]);
            }
        }
    }
    if (stats != nullptr) {
        // threads tid = piece (mod PPR) hold partials of the same 8 channels: combined in thread order (reproducible)
        float* const red = reinterpret_cast<float*>(stg);             // [256][16], over the staged tile every thread is done with
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[tid * 16 + e] = ssum[e];
            red[tid * 16 + 8 + e] = ssq[e];
        }
        __syncthreads();
        for (int item = tid; item < PPR * 16; item += 256) {
            const int pc = item / 16, e2 = item % 16;
            float a = 0.0f;
            for (int q = 0; q < 256 / PPR; ++q) a += red[(q * PPR + pc) * 16 + e2];
            const int stat = e2 / 8, ch = pc * 8 + (e2 % 8);
            stats[(((size_t)n * tiles + bt) * 2 + stat) * COUT + ch] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------------ head
// grid (tiles, 1, N), 256 threads.  Phase 1: T[q][tap] for the 600 halo positions q of the tile (19 groups of 32 columns, dealt to
// the 4 waves): B fragments straight from global memory (a lane = one position, 8 channels per k-piece) with GroupNorm + SiLU
// applied in registers and zeroed outside the volume (the conv pads the ACTIVATED tensor), A = W[tap][c] in registers; T is stored
// as bf16 (the rounding point of the round-2 pair of launches).  Phase 2: every thread sums the 27 taps of one output position.
template <int NKS>     // 16-channel k-steps (C = 16 * NKS)
__global__ __launch_bounds__(256) void k_head3d(const bf16_raw* __restrict__ x, const float* __restrict__ pre_a, const float* __restrict__ pre_b,
                                                int pre_silu, const bf16_raw* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ out, int D, int H, int W, int tiles_h, int tiles_w) {
    constexpr int C = 16 * NKS;
    constexpr int NQ = HD * HH * HW;                   // 600
    constexpr int NG = (NQ + 31) / 32;                 // 19
    constexpr int TP = 36;                             // T row pitch in bf16 elements (72 bytes: the 8-byte tap quads stay aligned)
    __shared__ bf16_raw T[NG * 32 * TP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, col = lane & 31;
    const int n = blockIdx.z;
    const int bt = blockIdx.x;
    const int tw_i = bt % tiles_w, th_i = (bt / tiles_w) % tiles_h, td_i = bt / (tiles_w * tiles_h);
    const int od0 = td_i * TD, oh0 = th_i * TH, ow0 = tw_i * TW;
    const size_t S = (size_t)D * H * W;
    const bf16_raw* const xs = x + (size_t)n * S * C;

    // A fragments: W[tap][c] (prepared [32][C], rows 27 .. 31 zero): row = col (tap), k-piece = 16 s + 8 half
    uint4 wa[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s) wa[s] = *reinterpret_cast<const uint4*>(w + (size_t)col * C + 16 * s + 8 * half);
    // folded GroupNorm affine of this lane's channels (one sample per workgroup)
    float ca[NKS][8], cb[NKS][8];
    if (pre_a != nullptr) {
#pragma unroll
        for (int s = 0; s < NKS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ca[s][j] = pre_a[(size_t)n * C + 16 * s + 8 * half + j];
                cb[s][j] = pre_b[(size_t)n * C + 16 * s + 8 * half + j];
            }
    }
    for (int g = wave; g < NG; g += 4) {
        const int q = g * 32 + col;
        const int iw = q % HW, ih = (q / HW) % HH, id = q / (HW * HH);
        const int gd = od0 + id - 1, gh = oh0 + ih - 1, gw = ow0 + iw - 1;
        const bool ok = q < NQ && (unsigned)gd < (unsigned)D && (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W;
        const size_t pos = ok ? ((size_t)gd * H + gh) * W + gw : 0;
        uint4 bf[NKS];
#pragma unroll
        for (int s = 0; s < NKS; ++s) bf[s] = *reinterpret_cast<const uint4*>(xs + pos * C + 16 * s + 8 * half);
        f32x16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            uint4 u = bf[s];
            if (pre_a != nullptr) {
                const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = fmaf(ca[s][2 * e], __uint_as_float(wv[e] << 16), cb[s][2 * e]);
                    v[2 * e + 1] = fmaf(ca[s][2 * e + 1], __uint_as_float(wv[e] & 0xFFFF0000u), cb[s][2 * e + 1]);
                }
                if (pre_silu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = silu_f(v[e]);
                }
                u = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            }
            if (!ok) u = make_uint4(0u, 0u, 0u, 0u);
            acc = mma32(wa[s], u, acc);
        }
        // accumulator register 4 r4 + e = tap 8 r4 + 4 half + e of halo position q
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<uint2*>(&T[q * TP + 8 * r4 + 4 * half]) =
                make_uint2(pack_bf16x2(acc[4 * r4 + 0], acc[4 * r4 + 1]), pack_bf16x2(acc[4 * r4 + 2], acc[4 * r4 + 3]));
    }
    __syncthreads();
    {
        const int pw = tid & 7, ph = (tid >> 3) & 7, pd = tid >> 6;
        const int od = od0 + pd, oh = oh0 + ph, ow = ow0 + pw;
        if (od < D && oh < H && ow < W) {
            float v[27];
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int qq = ((pd + tap / 9) * HH + ph + (tap / 3) % 3) * HW + pw + tap % 3;
                v[tap] = bf16_to_f32(T[qq * TP + tap]);
            }
            float a = bias ? bias[0] : 0.0f;
#pragma unroll
            for (int tap = 0; tap < 27; ++tap) a += v[tap];           // (the order of k_tap_gather_sum)
            out[(size_t)n * S + ((size_t)od * H + oh) * W + ow] = a;
        }
    }
}

}  // namespace

static inline int cdiv_i(int a, int b) { return (a + b - 1) / b; }

extern "C" int64_t rho_stem_conv3d_tiles(int64_t d, int64_t h, int64_t w) {
    return (int64_t)cdiv_i((int)d, TD) * cdiv_i((int)h, TH) * cdiv_i((int)w, TW);
}

extern "C" int rho_stem_conv3d(const float* x, const void* w, const float* bias, void* y, float* stats, int64_t n, int64_t d, int64_t h,
                               int64_t w_, int64_t cout, void* stream) {
    if (!x || !w || !bias || !y || n <= 0 || d <= 0 || h <= 0 || w_ <= 0) return RHO_E_ARG;
    if (cout != 32 && cout != 64) return RHO_E_SHAPE;          // (the staged output tile + the statistics scratch fit static LDS)
    if (n > 65535 || d * h * w_ >= (1LL << 31)) return RHO_E_SHAPE;
    const int th = cdiv_i((int)h, TH), tw = cdiv_i((int)w_, TW);
    dim3 grid((unsigned)rho_stem_conv3d_tiles(d, h, w_), 1, (unsigned)n), block(256);
    hipStream_t st = as_stream(stream);
#define RHO_STEM(NCT)                                                                                                           \
    hipLaunchKernelGGL(k_stem3d<NCT>, grid, block, 0, st, x, (const bf16_raw*)w, bias, (bf16_raw*)y, stats, (int)d, (int)h, (int)w_, th, tw)
    if (cout == 32) RHO_STEM(1);
    else RHO_STEM(2);
#undef RHO_STEM
    RHO_LAUNCH_CHECK();
    return 0;
}

extern "C" int rho_head_conv3d(const void* x, const float* pre_a, const float* pre_b, int pre_silu, const void* w, const float* bias,
                               float* out, int64_t n, int64_t d, int64_t h, int64_t w_, int64_t c, void* stream) {
    if (!x || !w || !out || n <= 0 || d <= 0 || h <= 0 || w_ <= 0) return RHO_E_ARG;
    if ((pre_a == nullptr) != (pre_b == nullptr)) return RHO_E_ARG;
    if (c != 32 && c != 64 && c != 96 && c != 128) return RHO_E_SHAPE;
    if (n > 65535 || d * h * w_ >= (1LL << 31)) return RHO_E_SHAPE;
    const int th = cdiv_i((int)h, TH), tw = cdiv_i((int)w_, TW);
    dim3 grid((unsigned)rho_stem_conv3d_tiles(d, h, w_), 1, (unsigned)n), block(256);
    hipStream_t st = as_stream(stream);
#define RHO_HEAD(NKS)                                                                                                           \
    hipLaunchKernelGGL(k_head3d<NKS>, grid, block, 0, st, (const bf16_raw*)x, pre_a, pre_b, pre_silu, (const bf16_raw*)w, bias, out, (int)d, \
                       (int)h, (int)w_, th, tw)
    if (c == 32) RHO_HEAD(2);
    else if (c == 64) RHO_HEAD(4);
    else if (c == 96) RHO_HEAD(6);
    else RHO_HEAD(8);
#undef RHO_HEAD
    RHO_LAUNCH_CHECK();
    return 0;
}
