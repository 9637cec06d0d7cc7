// GroupNorm(32 groups, eps 1e-5) statistics over channels-last activations, with the
// torch.cat of the UNet's skip connections read in place as two source tensors.
// (rho_diffusion/layers.py:71-74,122-129; unet_v2.py:729)
//
// Pass 1 (HBM-bound, one read of the activation): each thread owns one channel octet
// (8 consecutive channels = one 16-byte bf16 load) and strides over positions; per-block partial
// sums are combined through LDS in a fixed order => bitwise reproducible statistics.
// Pass 2 (tiny): fp64 combine, mean / rstd, and the per-(sample, channel) affine that the conv
// loader applies (GroupNorm gamma/beta folded with the FiLM scale/shift).
#include "common.h"

extern "C" int rho_gn_nblk(int64_t s) {
    // position blocks per sample of the partial-sum passes.  256 positions per block: with wide channel counts a block
    // covers only 256 / (C/8) positions per iteration, and at the deep levels (S = 4096, C = 512..1024) 2048-position
    // blocks meant 64 workgroups on 256 CUs running 1024 serial iterations each (0.65 TB/s).
    // (round 4: up to 256 blocks, was 64 - a batch of 2 at 128^3 gave the backward reduction 128 workgroups on 256 CUs: 1.3 TB/s)
    int64_t nb = (s + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 256) nb = 256;
    return (int)nb;
}

template <typename T>
__device__ __forceinline__ void load_octet(const T* p, float (&v)[8]);

template <>
__device__ __forceinline__ void load_octet<bf16_raw>(const bf16_raw* p, float (&v)[8]) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xFFFF0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xFFFF0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xFFFF0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xFFFF0000u);
}
template <>
__device__ __forceinline__ void load_octet<float>(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <typename T>
__global__ __launch_bounds__(256) void k_gn_partial(const T* __restrict__ x1, int c1, const T* __restrict__ x2, int c2,
                                                    int64_t s, int nblk, float* __restrict__ partials) {
    __shared__ float red[256 * 17];  // +1 pad: column walk in the final reduce is conflict-free
    const int C = c1 + c2;
    const int OCT = C >> 3;                    // octets per position (4 .. 128), divides 256 or not:
    const int ppi = 256 / OCT;                 // positions per iteration (>= 2 for C <= 1024)
    const int tid = threadIdx.x;
    const int oc = tid % OCT, pl = tid / OCT;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int64_t per = (s + nblk - 1) / nblk;
    const int64_t p0 = (int64_t)blk * per;
    const int64_t p1 = (p0 + per < s) ? p0 + per : s;

    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = sq[j] = 0.0f;

    if (pl < ppi) {
        const int ch = oc * 8;
        const T* src;
        int64_t stride;
        if (ch < c1) {
            src = x1 + (int64_t)n * s * c1 + ch;
            stride = c1;
        } else {
            src = x2 + (int64_t)n * s * c2 + (ch - c1);
            stride = c2;
        }
        // 4 independent 16-byte loads in flight per thread (a single dependent load per iteration leaves the
        // kernel latency-bound at ~2.6 TB/s)
        int64_t p = p0 + pl;
        for (; p + 3 * (int64_t)ppi < p1; p += 4 * (int64_t)ppi) {
            float v0[8], v1[8], v2[8], v3[8];
            load_octet<T>(src + p * stride, v0);
            load_octet<T>(src + (p + ppi) * stride, v1);
            load_octet<T>(src + (p + 2 * (int64_t)ppi) * stride, v2);
            load_octet<T>(src + (p + 3 * (int64_t)ppi) * stride, v3);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sum[j] += (v0[j] + v1[j]) + (v2[j] + v3[j]);
                sq[j] = fmaf(v0[j], v0[j], fmaf(v1[j], v1[j], fmaf(v2[j], v2[j], fmaf(v3[j], v3[j], sq[j]))));
            }
        }
        for (; p < p1; p += ppi) {
            float v[8];
            load_octet<T>(src + p * stride, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sum[j] += v[j];
                sq[j] = fmaf(v[j], v[j], sq[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid * 17 + j] = sum[j];
        red[tid * 17 + 8 + j] = sq[j];
    }
    __syncthreads();
    // thread (oc, j) for j in 0..15 sums over the ppi position lanes, fixed order
    for (int item = tid; item < OCT * 16; item += 256) {
        const int o = item >> 4, j = item & 15;
        float acc = 0.0f;
        for (int q = 0; q < ppi; ++q) acc += red[(q * OCT + o) * 17 + j];
        partials[(((int64_t)n * nblk + blk) * OCT + o) * 16 + j] = acc;
    }
}

extern "C" int rho_gn_partial(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s,
                              float* partials, void* stream) {
    const int64_t C = c1 + (x2 ? c2 : 0);
    if (!x1 || !partials || n <= 0 || s <= 0 || c1 <= 0) return RHO_E_ARG;
    if (!x2) c2 = 0;
    if (C % 32 != 0 || c1 % 8 != 0 || c2 % 8 != 0 || C > 2048) return RHO_E_ALIGN;
    if (C > 2048 || (256 / (C / 8)) < 1) return RHO_E_SHAPE;
    const int nblk = rho_gn_nblk(s);
    dim3 grid(nblk, (unsigned)n), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_gn_partial<bf16_raw>, grid, block, 0, as_stream(stream), (const bf16_raw*)x1, (int)c1,
                           (const bf16_raw*)x2, (int)c2, s, nblk, partials);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_gn_partial<float>, grid, block, 0, as_stream(stream), (const float*)x1, (int)c1, (const float*)x2,
                           (int)c2, s, nblk, partials);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(256) void k_gn_finalize(const float* __restrict__ partials, int c, int64_t s, int nblk,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     int64_t film_stride, float* __restrict__ stats, float* __restrict__ a,
                                                     float* __restrict__ b) {
    __shared__ double chs[2048], chq[2048];
    __shared__ float gmean[32], grstd[32];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int OCT = c >> 3;
    for (int ch = tid; ch < c; ch += 256) {
        const int o = ch >> 3, j = ch & 7;
        double su = 0.0, sq = 0.0;
        for (int k = 0; k < nblk; ++k) {
            const float* p = partials + (((int64_t)n * nblk + k) * OCT + o) * 16;
            su += (double)p[j];
            sq += (double)p[8 + j];
        }
        chs[ch] = su;
        chq[ch] = sq;
    }
    __syncthreads();
    const int cpg = c / 32;
    if (tid < 32) {
        double su = 0.0, sq = 0.0;
        for (int k = 0; k < cpg; ++k) {
            su += chs[tid * cpg + k];
            sq += chq[tid * cpg + k];
        }
        const double cnt = (double)cpg * (double)s;
        const double mean = su / cnt;
        double var = sq / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        gmean[tid] = (float)mean;
        grstd[tid] = rstd;
        if (stats) {
            stats[((int64_t)n * 32 + tid) * 2 + 0] = (float)mean;
            stats[((int64_t)n * 32 + tid) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    for (int ch = tid; ch < c; ch += 256) {
        const int g = ch / cpg;
        const float ga = gamma[ch] * grstd[g];
        float av = ga;
        float bv = beta[ch] - gmean[g] * ga;
        if (scale) {
            const float sc = 1.0f + scale[(int64_t)n * film_stride + ch];
            av *= sc;
            bv = bv * sc + shift[(int64_t)n * film_stride + ch];
        }
        a[(int64_t)n * c + ch] = av;
        b[(int64_t)n * c + ch] = bv;
    }
}

extern "C" int rho_gn_finalize(const float* partials, int64_t n, int64_t c, int64_t s, int64_t nblk, const float* gamma,
                               const float* beta, const float* scale, const float* shift, int64_t film_stride, float* stats,
                               float* a, float* b, void* stream) {
    if (!partials || !gamma || !beta || !a || !b || n <= 0 || c <= 0 || c % 32 != 0 || c > 2048 || (scale && !shift))
        return RHO_E_ARG;
    hipLaunchKernelGGL(k_gn_finalize, dim3((unsigned)n), dim3(256), 0, as_stream(stream), partials, (int)c, s, (int)nblk, gamma,
                       beta, scale, shift, film_stride, stats, a, b);
    RHO_LAUNCH_CHECK();
    return 0;
}

// Two-source finalize (see rho_gn_finalize2 in the header): per-channel sums over nblk partials in a FIXED order
// (blocks dealt round-robin to KP lanes per channel, lanes then combined in lane order, all in fp64), so the
// statistics stay bit-reproducible although a conv's tile partials can number in the thousands.
__device__ __forceinline__ void gn_partial_at(const float* __restrict__ p, int fmt, int nblk, int c, int n, int k, int ch,
                                              double& su, double& sq) {
    if (fmt == 0) {
        const float* q = p + (((int64_t)n * nblk + k) * (c >> 3) + (ch >> 3)) * 16;
        su += (double)q[ch & 7];
        sq += (double)q[8 + (ch & 7)];
    } else {
        const float* q = p + ((int64_t)n * nblk + k) * 2 * c;
        su += (double)q[ch];
        sq += (double)q[c + ch];
    }
}

__global__ __launch_bounds__(256) void k_gn_finalize2(const float* __restrict__ p1, int fmt1, int nblk1, int c1,
                                                      const float* __restrict__ p2, int fmt2, int nblk2, int c2, int64_t s,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      int64_t film_stride, float* __restrict__ stats, float* __restrict__ a,
                                                      float* __restrict__ b, int gpb) {
    // grid (n, 32 / gpb): a workgroup owns gpb (4, or 1 for small batches) of the 32 groups = gpb * cpg consecutive
    // channels (<= 256), so every CU has work and a thread sums nblk / KP partials instead of all of them
    // (74 us -> a few us per GroupNorm at 1024 tiles)
    __shared__ double ls[256], lq[256];
    __shared__ double chs[256], chq[256];
    __shared__ float gmean[4], grstd[4];
    const int n = blockIdx.x, gq = blockIdx.y, tid = threadIdx.x;
    const int c = c1 + c2;
    const int cpg = c / 32;
    const int CB = gpb * cpg;                    // channels of this workgroup
    const int ch0 = gq * CB;
    int KP = 256 / CB;                           // lanes per channel
    if (KP < 1) KP = 1;
    const int cl = tid % CB, kp = tid / CB;
    double su = 0.0, sq = 0.0;
    if (kp < KP) {
        const int ch = ch0 + cl;
        if (ch < c1) {
            for (int k = kp; k < nblk1; k += KP) gn_partial_at(p1, fmt1, nblk1, c1, n, k, ch, su, sq);
        } else {
            for (int k = kp; k < nblk2; k += KP) gn_partial_at(p2, fmt2, nblk2, c2, n, k, ch - c1, su, sq);
        }
    }
    ls[tid] = su;
    lq[tid] = sq;
    __syncthreads();
    if (tid < CB) {
        double tu = 0.0, tq = 0.0;
        for (int q = 0; q < KP; ++q) { tu += ls[q * CB + tid]; tq += lq[q * CB + tid]; }
        chs[tid] = tu;
        chq[tid] = tq;
    }
    __syncthreads();
    if (tid < gpb) {
        double gu = 0.0, gs = 0.0;
        for (int k = 0; k < cpg; ++k) {
            gu += chs[tid * cpg + k];
            gs += chq[tid * cpg + k];
        }
        const double cnt = (double)cpg * (double)s;
        const double mean = gu / cnt;
        double var = gs / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        gmean[tid] = (float)mean;
        grstd[tid] = rstd;
        if (stats) {
            stats[((int64_t)n * 32 + gq * gpb + tid) * 2 + 0] = (float)mean;
            stats[((int64_t)n * 32 + gq * gpb + tid) * 2 + 1] = rstd;
        }
    }
    __syncthreads();
    if (tid < CB) {
        const int ch = ch0 + tid;
        const int g = tid / cpg;
        const float ga = gamma[ch] * grstd[g];
        float av = ga;
        float bv = beta[ch] - gmean[g] * ga;
        if (scale) {
            const float sc = 1.0f + scale[(int64_t)n * film_stride + ch];
            av *= sc;
            bv = bv * sc + shift[(int64_t)n * film_stride + ch];
        }
        a[(int64_t)n * c + ch] = av;
        b[(int64_t)n * c + ch] = bv;
    }
}

extern "C" int rho_gn_finalize2(const float* p1, int fmt1, int64_t nblk1, int64_t c1, const float* p2, int fmt2, int64_t nblk2,
                                int64_t c2, int64_t n, int64_t s, const float* gamma, const float* beta, const float* scale,
                                const float* shift, int64_t film_stride, float* stats, float* a, float* b, void* stream) {
    if (!p1 || !gamma || !beta || !a || !b || n <= 0 || c1 <= 0 || nblk1 <= 0 || (scale && !shift)) return RHO_E_ARG;
    if (!p2) { c2 = 0; nblk2 = 0; }
    const int64_t c = c1 + c2;
    if (c % 32 != 0 || c > 2048 || c1 % 8 != 0 || c2 % 8 != 0 || (p2 && nblk2 <= 0)) return RHO_E_ARG;
    if ((fmt1 != 0 && fmt1 != 1) || (p2 && fmt2 != 0 && fmt2 != 1)) return RHO_E_ARG;
    if (n > 65535) return RHO_E_SHAPE;
    const int gpb = (n * 8 >= 256) ? 4 : 1;      // small batches: one group per workgroup (32 per sample)
    hipLaunchKernelGGL(k_gn_finalize2, dim3((unsigned)n, (unsigned)(32 / gpb)), dim3(256), 0, as_stream(stream), p1, fmt1, (int)nblk1,
                       (int)c1, p2, fmt2, (int)nblk2, (int)c2, s, gamma, beta, scale, shift, film_stride, stats, a, b, gpb);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================
// Backward of  act( GroupNorm(x) * (1 + scale) + shift )  (act = SiLU or identity), recomputing the
// forward from x, the saved statistics and the folded affine (a, b):
//     u = a*x + b,  gq = g * act'(u),  xhat = (x - mean) * rstd,  gamma' = gamma * (1 + scale)
//     dx = rstd * (gamma' * gq - mean_grp(gamma' * gq) - xhat * mean_grp(gamma' * gq * xhat))
// Per-(n, c) sums R1 = sum_pos gq and R2 = sum_pos gq*xhat (pass 1, same partial layout as the
// forward statistics) give every parameter / FiLM gradient and the group means, so pass 2 is a pure
// elementwise   dx = A[n,c]*gq + P[n,grp] + Q[n,grp]*x .

// DROP: the forward applied a dropout mask after the activation (rho_gn_apply_drop): g is multiplied by the regenerated mask
template <typename T, bool DROP = false>
__global__ __launch_bounds__(256) void k_gn_bwd_reduce(const T* __restrict__ g, const T* __restrict__ x1, int c1,
                                                       const T* __restrict__ x2, int c2, int64_t s, int nblk,
                                                       const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ stats, int pre_silu,
                                                       float* __restrict__ partials, DropK dk = DropK{}) {
    __shared__ float red[256 * 17];
    const int C = c1 + c2;
    const int OCT = C >> 3;
    const int ppi = 256 / OCT;
    const int tid = threadIdx.x;
    const int oc = tid % OCT, pl = tid / OCT;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int64_t per = (s + nblk - 1) / nblk;
    const int64_t p0 = (int64_t)blk * per;
    const int64_t p1 = (p0 + per < s) ? p0 + per : s;
    float r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r1[j] = r2[j] = 0.0f;
    if (pl < ppi) {
        const int ch = oc * 8;
        const T* src;
        int64_t stride;
        if (ch < c1) {
            src = x1 + (int64_t)n * s * c1 + ch;
            stride = c1;
        } else {
            src = x2 + (int64_t)n * s * c2 + (ch - c1);
            stride = c2;
        }
        const T* gp = g + (int64_t)n * s * C + ch;
        const int cpg = C / 32;
        float av[8], bv[8], mu[8], rs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            av[j] = a[(int64_t)n * C + ch + j];
            bv[j] = b[(int64_t)n * C + ch + j];
            const int grp = (ch + j) / cpg;
            mu[j] = stats[((int64_t)n * 32 + grp) * 2 + 0];
            rs[j] = stats[((int64_t)n * 32 + grp) * 2 + 1];
        }
        uint64_t doff = 0;
        if constexpr (DROP) doff = *dk.off_dev;
        int64_t p = p0 + pl;
        for (; p + ppi < p1; p += 2 * (int64_t)ppi) {      // two positions (4 loads) in flight
            float xv[8], gv[8], xw[8], gw[8];
            load_octet<T>(src + p * stride, xv);
            load_octet<T>(gp + p * C, gv);
            load_octet<T>(src + (p + ppi) * stride, xw);
            load_octet<T>(gp + (p + ppi) * C, gw);
            if constexpr (DROP) {
                float mv[8], mw[8];
                drop_mult8(dk, doff, ((int64_t)n * s + p) * C + ch, mv);
                drop_mult8(dk, doff, ((int64_t)n * s + p + ppi) * C + ch, mw);
#pragma unroll
                for (int j = 0; j < 8; ++j) { gv[j] *= mv[j]; gw[j] *= mw[j]; }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gq = gv[j], gr = gw[j];
                if (pre_silu == 1) {
                    gq *= dsilu_f(fmaf(av[j], xv[j], bv[j]));
                    gr *= dsilu_f(fmaf(av[j], xw[j], bv[j]));
                } else if (pre_silu) {
                    gq *= dact_other_f(fmaf(av[j], xv[j], bv[j]), pre_silu);
                    gr *= dact_other_f(fmaf(av[j], xw[j], bv[j]), pre_silu);
                }
                r1[j] += gq + gr;
                r2[j] = fmaf(gq, (xv[j] - mu[j]) * rs[j], fmaf(gr, (xw[j] - mu[j]) * rs[j], r2[j]));
            }
        }
        for (; p < p1; p += ppi) {
            float xv[8], gv[8];
            load_octet<T>(src + p * stride, xv);
            load_octet<T>(gp + p * C, gv);
            if constexpr (DROP) {
                float mv[8];
                drop_mult8(dk, doff, ((int64_t)n * s + p) * C + ch, mv);
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[j] *= mv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gq = gv[j];
                if (pre_silu) gq *= (pre_silu == 1 ? dsilu_f(fmaf(av[j], xv[j], bv[j])) : dact_other_f(fmaf(av[j], xv[j], bv[j]), pre_silu));
                r1[j] += gq;
                r2[j] = fmaf(gq, (xv[j] - mu[j]) * rs[j], r2[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid * 17 + j] = r1[j];
        red[tid * 17 + 8 + j] = r2[j];
    }
    __syncthreads();
    for (int item = tid; item < OCT * 16; item += 256) {
        const int o = item >> 4, j = item & 15;
        float acc = 0.0f;
        for (int q = 0; q < ppi; ++q) acc += red[(q * OCT + o) * 17 + j];
        partials[(((int64_t)n * nblk + blk) * OCT + o) * 16 + j] = acc;
    }
}

static inline bool drop_args(float p, uint64_t seed, const uint64_t* off_dev, DropK& dk) {
    if (!(p > 0.0f) || !(p < 1.0f) || !off_dev) return false;
    dk.thr = (uint32_t)((double)p * 4294967296.0);
    dk.inv_keep = 1.0f / (1.0f - p);
    dk.seed = seed;
    dk.off_dev = off_dev;
    return true;
}

// (the dropout form: drop_p in (0, 1), the mask of rho_gn_apply_drop with the same seed / offset; see rho_hip.h)
extern "C" int rho_gn_bwd_reduce_drop(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                                      int64_t s, const float* a, const float* b, const float* stats, int pre_silu, float* partials,
                                      float drop_p, uint64_t drop_seed, const uint64_t* drop_offset_dev, void* stream) {
    if (!g || !x1 || !a || !b || !stats || !partials || n <= 0 || s <= 0) return RHO_E_ARG;
    DropK dk{};
    if (!drop_args(drop_p, drop_seed, drop_offset_dev, dk)) return RHO_E_ARG;
    if (!x2) c2 = 0;
    const int64_t C = c1 + c2;
    if (C % 32 != 0 || c1 % 8 != 0 || c2 % 8 != 0 || C > 2048) return RHO_E_ALIGN;
    const int nblk = rho_gn_nblk(s);
    dim3 grid(nblk, (unsigned)n), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL((k_gn_bwd_reduce<bf16_raw, true>), grid, block, 0, as_stream(stream), (const bf16_raw*)g, (const bf16_raw*)x1,
                           (int)c1, (const bf16_raw*)x2, (int)c2, s, nblk, a, b, stats, pre_silu, partials, dk);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL((k_gn_bwd_reduce<float, true>), grid, block, 0, as_stream(stream), (const float*)g, (const float*)x1, (int)c1,
                           (const float*)x2, (int)c2, s, nblk, a, b, stats, pre_silu, partials, dk);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

extern "C" int rho_gn_bwd_reduce(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                                 int64_t s, const float* a, const float* b, const float* stats, int pre_silu, float* partials,
                                 void* stream) {
    if (!g || !x1 || !a || !b || !stats || !partials || n <= 0 || s <= 0) return RHO_E_ARG;
    if (!x2) c2 = 0;
    const int64_t C = c1 + c2;
    if (C % 32 != 0 || c1 % 8 != 0 || c2 % 8 != 0 || C > 2048) return RHO_E_ALIGN;
    const int nblk = rho_gn_nblk(s);
    dim3 grid(nblk, (unsigned)n), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL((k_gn_bwd_reduce<bf16_raw, false>), grid, block, 0, as_stream(stream), (const bf16_raw*)g, (const bf16_raw*)x1,
                           (int)c1, (const bf16_raw*)x2, (int)c2, s, nblk, a, b, stats, pre_silu, partials, DropK{});
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL((k_gn_bwd_reduce<float, false>), grid, block, 0, as_stream(stream), (const float*)g, (const float*)x1, (int)c1,
                           (const float*)x2, (int)c2, s, nblk, a, b, stats, pre_silu, partials, DropK{});
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// grid (n, 8): a workgroup owns 4 groups (4 * cpg <= 256 channels) of one sample: totals over the position blocks (KP
// lanes per channel, fixed order), group sums, per-sample parameter-gradient rows and the apply coefficients
__global__ __launch_bounds__(256) void k_gn_bwd_finalize(const float* __restrict__ partials, int c, int64_t s, int nblk, int fmt,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ scale, int64_t film_stride,
                                                         const float* __restrict__ stats, float* __restrict__ dgamma_n,
                                                         float* __restrict__ dbeta_n, float* __restrict__ dscale,
                                                         float* __restrict__ dshift, int64_t dfilm_stride,
                                                         float* __restrict__ cA, float* __restrict__ cP, float* __restrict__ cQ) {
    __shared__ double l1[256], l2[256];
    __shared__ float r1s[256], r2s[256];
    const int n = blockIdx.x, gq = blockIdx.y, tid = threadIdx.x;
    const int OCT = c >> 3;
    const int cpg = c / 32;
    const int CB = 4 * cpg;
    const int ch0 = gq * CB;
    int KP = 256 / CB;
    if (KP < 1) KP = 1;
    const int cl = tid % CB, kp = tid / CB;
    double a1 = 0.0, a2 = 0.0;
    if (kp < KP) {
        const int ch = ch0 + cl;
        const int o = ch >> 3, j = ch & 7;
        if (fmt == 0) {
            for (int k = kp; k < nblk; k += KP) {
                const float* p = partials + (((int64_t)n * nblk + k) * OCT + o) * 16;
                a1 += (double)p[j];
                a2 += (double)p[8 + j];
            }
        } else {
            // per-tile sums written by a dgrad launch (rho_conv_desc.gnb_*): [n][tile][2][c], second row = sum of dz * x (raw x)
            for (int k = kp; k < nblk; k += KP) {
                const float* p = partials + ((int64_t)n * nblk + k) * 2 * c + ch;
                a1 += (double)p[0];
                a2 += (double)p[c];
            }
        }
    }
    l1[tid] = a1;
    l2[tid] = a2;
    __syncthreads();
    if (tid < CB) {
        double t1 = 0.0, t2 = 0.0;
        for (int q = 0; q < KP; ++q) { t1 += l1[q * CB + tid]; t2 += l2[q * CB + tid]; }
        if (fmt != 0) {                                   // sum dz * xhat = rstd * (sum dz * x - mean * sum dz)
            const int grp = (ch0 + tid) / cpg;
            const double mean = (double)stats[((int64_t)n * 32 + grp) * 2 + 0], rstd = (double)stats[((int64_t)n * 32 + grp) * 2 + 1];
            t2 = rstd * (t2 - mean * t1);
        }
        r1s[tid] = (float)t1;
        r2s[tid] = (float)t2;
    }
    __syncthreads();
    if (tid < 4) {
        const int g = gq * 4 + tid;
        double s1 = 0.0, s2 = 0.0;
        for (int k = 0; k < cpg; ++k) {
            const int ch = g * cpg + k;
            const float gp = gamma[ch] * (scale ? 1.0f + scale[(int64_t)n * film_stride + ch] : 1.0f);
            s1 += (double)gp * r1s[tid * cpg + k];
            s2 += (double)gp * r2s[tid * cpg + k];
        }
        const float mean = stats[((int64_t)n * 32 + g) * 2 + 0];
        const float rstd = stats[((int64_t)n * 32 + g) * 2 + 1];
        const double M = (double)cpg * (double)s;
        cP[(int64_t)n * 32 + g] = (float)(-(double)rstd * s1 / M + (double)rstd * rstd * s2 * mean / M);
        cQ[(int64_t)n * 32 + g] = (float)(-(double)rstd * rstd * s2 / M);
    }
    if (tid < CB) {
        const int ch = ch0 + tid;
        const int grp = ch / cpg;
        const float rstd = stats[((int64_t)n * 32 + grp) * 2 + 1];
        const float sc = scale ? 1.0f + scale[(int64_t)n * film_stride + ch] : 1.0f;
        cA[(int64_t)n * c + ch] = rstd * gamma[ch] * sc;
        dgamma_n[(int64_t)n * c + ch] = r2s[tid] * sc;
        dbeta_n[(int64_t)n * c + ch] = r1s[tid] * sc;
        if (dscale) {
            dscale[(int64_t)n * dfilm_stride + ch] = gamma[ch] * r2s[tid] + beta[ch] * r1s[tid];
            dshift[(int64_t)n * dfilm_stride + ch] = r1s[tid];
        }
    }
}

// out[c] (+)= sum_n in[n][c]   (deterministic order); blockIdx.y selects the (in, out) pair
__global__ void k_sum_over_n2(const float* __restrict__ in0, float* __restrict__ out0, const float* __restrict__ in1,
                              float* __restrict__ out1, int n, int c, int accumulate) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const float* in = blockIdx.y ? in1 : in0;
    float* out = blockIdx.y ? out1 : out0;
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) acc += in[(int64_t)k * c + ch];
    out[ch] = accumulate ? out[ch] + acc : acc;
}

__global__ void k_sum_over_n(const float* __restrict__ in, float* __restrict__ out, int n, int c, int accumulate) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) acc += in[(int64_t)k * c + ch];
    out[ch] = accumulate ? out[ch] + acc : acc;
}

extern "C" int rho_gn_bwd_finalize(const float* partials, int64_t n, int64_t c, int64_t s, int64_t nblk, int fmt, const float* gamma,
                                   const float* beta, const float* scale, int64_t film_stride, const float* stats,
                                   float* work_nc2, float* dgamma, float* dbeta, int accumulate, float* dscale, float* dshift,
                                   int64_t dfilm_stride, float* cA, float* cP, float* cQ, void* stream) {
    if (!partials || !gamma || !beta || !stats || !work_nc2 || !dgamma || !dbeta || !cA || !cP || !cQ || n <= 0 || c <= 0 ||
        c % 32 != 0 || c > 2048 || (dscale && !dshift) || (fmt != 0 && fmt != 1) || nblk <= 0)
        return RHO_E_ARG;
    float* dg_n = work_nc2;
    float* db_n = work_nc2 + n * c;
    if (n > 65535) return RHO_E_SHAPE;
    hipLaunchKernelGGL(k_gn_bwd_finalize, dim3((unsigned)n, 8), dim3(256), 0, as_stream(stream), partials, (int)c, s, (int)nblk, fmt,
                       gamma, beta, scale, film_stride, stats, dg_n, db_n, dscale, dshift, dfilm_stride, cA, cP, cQ);
    RHO_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_sum_over_n2, dim3((unsigned)((c + 255) / 256), 2), dim3(256), 0, as_stream(stream), dg_n, dgamma, db_n, dbeta,
                       (int)n, (int)c, accumulate);
    RHO_LAUNCH_CHECK();
    return 0;
}

// Position blocks of the elementwise passes: `per` positions per workgroup, finer when that leaves the chip short of
// workgroups (wide, small layers: C = 1024 at 4096 positions would be 128 workgroups), never below 4 positions per thread row.
static int apply_nblk(int64_t s, int64_t per, int64_t C, int64_t n) {
    int64_t nblk = (s + per - 1) / per;
    const int64_t want = (2048 + n - 1) / n;
    if (nblk < want) {
        const int64_t ppi = 256 / (C >> 3) > 0 ? 256 / (C >> 3) : 1;
        const int64_t cap = (s + 4 * ppi - 1) / (4 * ppi);
        nblk = want < cap ? want : cap;
    }
    if (nblk < 1) nblk = 1;
    if (nblk > 1024) nblk = 1024;        // (round 4: was 256 - 512 workgroups for a batch of 2 at 128^3 left the passes at 3.6 - 4.3 TB/s)
    return (int)nblk;
}

template <typename T>
__device__ __forceinline__ void store_octet(T* p, const float (&v)[8]);
template <>
__device__ __forceinline__ void store_octet<bf16_raw>(bf16_raw* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                               pack_bf16x2(v[6], v[7]));
}
template <>
__device__ __forceinline__ void store_octet<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// value as the storage type would hold it (bf16: one rounding; f32: itself)
template <typename T>
__device__ __forceinline__ float cvt_round(float v);
template <>
__device__ __forceinline__ float cvt_round<float>(float v) { return v; }
template <>
__device__ __forceinline__ float cvt_round<bf16_raw>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T, bool DROP = false>
__global__ __launch_bounds__(256) void k_gn_bwd_apply(const T* __restrict__ g, const T* __restrict__ x1, int c1,
                                                      const T* __restrict__ x2, int c2, int64_t s, int nblk,
                                                      const float* __restrict__ a, const float* __restrict__ b, int pre_silu,
                                                      const float* __restrict__ cA, const float* __restrict__ cP,
                                                      const float* __restrict__ cQ, T* __restrict__ dx1, T* __restrict__ dx2,
                                                      int acc1, int acc2, const T* __restrict__ add1, DropK dk = DropK{}) {
    // same thread -> (channel octet, position lane) map as the reducers: the 5 per-channel coefficients live in
    // registers for the whole position walk (the elementwise form re-loaded them and divided indices per element)
    const int C = c1 + c2;
    const int OCT = C >> 3;
    const int ppi = 256 / OCT;
    const int tid = threadIdx.x;
    const int oc = tid % OCT, pl = tid / OCT;
    if (pl >= ppi) return;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int64_t per = (s + nblk - 1) / nblk;
    const int64_t p0 = (int64_t)blk * per;
    const int64_t p1 = (p0 + per < s) ? p0 + per : s;
    const int ch = oc * 8;
    const int cpg = C / 32;
    const bool first = ch < c1;
    const T* xs = first ? x1 + (int64_t)n * s * c1 + ch : x2 + (int64_t)n * s * c2 + (ch - c1);
    T* dst = first ? dx1 + (int64_t)n * s * c1 + ch : dx2 + (int64_t)n * s * c2 + (ch - c1);
    const int64_t stride = first ? c1 : c2;
    const int accf = first ? acc1 : acc2;
    // a further addend of dx1 (the gradient arriving through a residual connection: what a separate rho_add_inplace pass added)
    const T* e1 = (first && add1 != nullptr) ? add1 + (int64_t)n * s * c1 + ch : nullptr;
    const T* gp = g + (int64_t)n * s * C + ch;
    float av[8], bv[8], ca[8], cp[8], cq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int64_t nc = (int64_t)n * C + ch + j;
        const int grp = (ch + j) / cpg;
        av[j] = a[nc]; bv[j] = b[nc]; ca[j] = cA[nc];
        cp[j] = cP[(int64_t)n * 32 + grp]; cq[j] = cQ[(int64_t)n * 32 + grp];
    }
    uint64_t doff = 0;
    if constexpr (DROP) doff = *dk.off_dev;
    for (int64_t p = p0 + pl; p < p1; p += ppi) {
        float xv[8], gv[8], ov[8];
        load_octet<T>(xs + p * stride, xv);
        load_octet<T>(gp + p * C, gv);
        if constexpr (DROP) {
            float mv[8];
            drop_mult8(dk, doff, ((int64_t)n * s + p) * C + ch, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] *= mv[j];
        }
        float ev[8];
        if (accf) load_octet<T>(dst + p * stride, ov);
        if (e1) load_octet<T>(e1 + p * stride, ev);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float gq = gv[j];
            if (pre_silu) gq *= (pre_silu == 1 ? dsilu_f(fmaf(av[j], xv[j], bv[j])) : dact_other_f(fmaf(av[j], xv[j], bv[j]), pre_silu));
            float r = fmaf(ca[j], gq, fmaf(cq[j], xv[j], cp[j]));
            // (the order of the separate passes: the residual's gradient joins the running sum first, rounded to the storage type as
            //  rho_add_inplace stores it, then this term)
            if (accf && e1) r += cvt_round<T>(ov[j] + ev[j]);
            else if (accf) r += ov[j];
            else if (e1) r += ev[j];
            ov[j] = r;
        }
        store_octet<T>(dst + p * stride, ov);
    }
}

static int gn_bwd_apply_impl(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                             int64_t s, const float* a, const float* b, int pre_silu, const float* cA, const float* cP,
                             const float* cQ, void* dx1, void* dx2, int acc1, int acc2, const void* add1, const DropK* dkp, void* stream);

extern "C" int rho_gn_bwd_apply_drop(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                                     int64_t s, const float* a, const float* b, int pre_silu, const float* cA, const float* cP,
                                     const float* cQ, void* dx1, void* dx2, int acc1, int acc2, const void* add1, float drop_p,
                                     uint64_t drop_seed, const uint64_t* drop_offset_dev, void* stream) {
    DropK dk{};
    if (!drop_args(drop_p, drop_seed, drop_offset_dev, dk)) return RHO_E_ARG;
    return gn_bwd_apply_impl(g, x1, c1, x2, c2, dtype, n, s, a, b, pre_silu, cA, cP, cQ, dx1, dx2, acc1, acc2, add1, &dk, stream);
}

extern "C" int rho_gn_bwd_apply(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                                int64_t s, const float* a, const float* b, int pre_silu, const float* cA, const float* cP,
                                const float* cQ, void* dx1, void* dx2, int acc1, int acc2, const void* add1, void* stream) {
    return gn_bwd_apply_impl(g, x1, c1, x2, c2, dtype, n, s, a, b, pre_silu, cA, cP, cQ, dx1, dx2, acc1, acc2, add1, nullptr, stream);
}

static int gn_bwd_apply_impl(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                             int64_t s, const float* a, const float* b, int pre_silu, const float* cA, const float* cP,
                             const float* cQ, void* dx1, void* dx2, int acc1, int acc2, const void* add1, const DropK* dkp, void* stream) {
    if (!g || !x1 || !a || !b || !cA || !cP || !cQ || !dx1 || n <= 0 || s <= 0) return RHO_E_ARG;
    if (!x2) c2 = 0;
    if (c2 > 0 && !dx2) return RHO_E_ARG;
    const int64_t C = c1 + c2;
    if (C % 32 != 0 || c1 % 8 != 0 || c2 % 8 != 0) return RHO_E_ALIGN;
    if (C > 2048) return RHO_E_SHAPE;
    const int nblk = apply_nblk(s, 512, C, n);  // finer than the reducers: nothing to combine afterwards
    dim3 grid((unsigned)nblk, (unsigned)n), block(256);
    if (dtype != RHO_BF16 && dtype != RHO_F32) return RHO_E_ARG;
#define RHO_APPLY(T_, DROP_, DK_)                                                                                                      \
    hipLaunchKernelGGL((k_gn_bwd_apply<T_, DROP_>), grid, block, 0, as_stream(stream), (const T_*)g, (const T_*)x1, (int)c1, (const T_*)x2, \
                       (int)c2, s, nblk, a, b, pre_silu, cA, cP, cQ, (T_*)dx1, (T_*)dx2, acc1, acc2, (const T_*)add1, DK_)
    if (dkp) {
        if (dtype == RHO_BF16) RHO_APPLY(bf16_raw, true, *dkp);
        else RHO_APPLY(float, true, *dkp);
    } else {
        if (dtype == RHO_BF16) RHO_APPLY(bf16_raw, false, DropK{});
        else RHO_APPLY(float, false, DropK{});
    }
#undef RHO_APPLY
    RHO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Materialised normalised activation  y[n,pos,c] = act(a[n,c] * x[n,pos,c] + b[n,c])  over the (virtual) concat
// of one or two sources.  The forward convolutions never need it (their loaders apply the folded affine on the
// fly); the weight gradient does: there every (cout-tile, cin-chunk) workgroup would otherwise redo the
// exp/rcp of SiLU for the same input chunk with a single wave per SIMD to hide it.
template <typename T, bool DROP = false>
__global__ __launch_bounds__(256) void k_gn_apply(const T* __restrict__ x1, int c1, const T* __restrict__ x2, int c2, int64_t s,
                                                  int nblk, const float* __restrict__ a, const float* __restrict__ b,
                                                  int pre_silu, T* __restrict__ y, DropK dk = DropK{}) {
    const int C = c1 + c2;
    const int OCT = C >> 3;
    const int ppi = 256 / OCT;
    const int tid = threadIdx.x;
    const int oc = tid % OCT, pl = tid / OCT;
    if (pl >= ppi) return;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int64_t per = (s + nblk - 1) / nblk;
    const int64_t p0 = (int64_t)blk * per;
    const int64_t p1 = (p0 + per < s) ? p0 + per : s;
    const int ch = oc * 8;
    const bool first = ch < c1;
    const T* xs = first ? x1 + (int64_t)n * s * c1 + ch : x2 + (int64_t)n * s * c2 + (ch - c1);
    const int64_t stride = first ? c1 : c2;
    T* yp = y + (int64_t)n * s * C + ch;
    float av[8], bv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        av[j] = a[(int64_t)n * C + ch + j];
        bv[j] = b[(int64_t)n * C + ch + j];
    }
    uint64_t doff = 0;
    if constexpr (DROP) doff = *dk.off_dev;
    int64_t p = p0 + pl;
    for (; p + 3 * ppi < p1; p += 4 * ppi) {       // four independent 16/32-byte loads in flight per thread
        float xv[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load_octet<T>(xs + (p + u * ppi) * stride, xv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = fmaf(av[j], xv[u][j], bv[j]);
                xv[u][j] = pre_silu == 1 ? silu_f(f) : (pre_silu ? act_other_f(f, pre_silu) : f);
            }
            if constexpr (DROP) {
                float mv[8];
                drop_mult8(dk, doff, ((int64_t)n * s + p + u * ppi) * C + ch, mv);
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[u][j] *= mv[j];
            }
            store_octet<T>(yp + (p + u * ppi) * C, xv[u]);
        }
    }
    for (; p < p1; p += ppi) {
        float xv[8];
        load_octet<T>(xs + p * stride, xv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = fmaf(av[j], xv[j], bv[j]);
            xv[j] = pre_silu == 1 ? silu_f(f) : (pre_silu ? act_other_f(f, pre_silu) : f);
        }
        if constexpr (DROP) {
            float mv[8];
            drop_mult8(dk, doff, ((int64_t)n * s + p) * C + ch, mv);
#pragma unroll
            for (int j = 0; j < 8; ++j) xv[j] *= mv[j];
        }
        store_octet<T>(yp + p * C, xv);
    }
}

static int gn_apply_impl(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s, const float* a,
                         const float* b, int pre_silu, void* y, const DropK* dkp, void* stream);

extern "C" int rho_gn_apply_drop(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s,
                                 const float* a, const float* b, int pre_silu, void* y, float drop_p, uint64_t drop_seed,
                                 const uint64_t* drop_offset_dev, void* stream) {
    DropK dk{};
    if (!drop_args(drop_p, drop_seed, drop_offset_dev, dk)) return RHO_E_ARG;
    return gn_apply_impl(x1, c1, x2, c2, dtype, n, s, a, b, pre_silu, y, &dk, stream);
}

// the mask alone, as bytes (1 = kept): what rho_gn_apply_drop applies to element e of its output (test aid; n elements)
__global__ __launch_bounds__(256) void k_dropout_mask(uint8_t* __restrict__ out, int64_t n, DropK dk) {
    const uint64_t off = *dk.off_dev;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * blockDim.x * 8) {
        float m[8];
        drop_mult8(dk, off, i, m);
        for (int j = 0; j < 8 && i + j < n; ++j) out[i + j] = m[j] != 0.0f ? 1 : 0;
    }
}
extern "C" int rho_dropout_mask(uint8_t* out, int64_t n, float drop_p, uint64_t drop_seed, const uint64_t* drop_offset_dev, void* stream) {
    DropK dk{};
    if (!out || n <= 0 || !drop_args(drop_p, drop_seed, drop_offset_dev, dk)) return RHO_E_ARG;
    int64_t g = (n / 8 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(k_dropout_mask, dim3((unsigned)g), dim3(256), 0, as_stream(stream), out, n, dk);
    RHO_LAUNCH_CHECK();
    return 0;
}

extern "C" int rho_gn_apply(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s,
                            const float* a, const float* b, int pre_silu, void* y, void* stream) {
    return gn_apply_impl(x1, c1, x2, c2, dtype, n, s, a, b, pre_silu, y, nullptr, stream);
}

static int gn_apply_impl(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s, const float* a,
                         const float* b, int pre_silu, void* y, const DropK* dkp, void* stream) {
    if (!x1 || !a || !b || !y || n <= 0 || s <= 0) return RHO_E_ARG;
    if (!x2) c2 = 0;
    const int64_t C = c1 + c2;
    if (C <= 0 || C % 8 != 0 || c1 % 8 != 0 || c2 % 8 != 0) return RHO_E_ALIGN;
    if (C > 2048 || n > 65535) return RHO_E_SHAPE;
    const int nblk = apply_nblk(s, 1024, C, n);
    dim3 grid((unsigned)nblk, (unsigned)n), block(256);
    if (dtype != RHO_BF16 && dtype != RHO_F32) return RHO_E_ARG;
#define RHO_GNA(T_, DROP_, DK_)                                                                                                        \
    hipLaunchKernelGGL((k_gn_apply<T_, DROP_>), grid, block, 0, as_stream(stream), (const T_*)x1, (int)c1, (const T_*)x2, (int)c2, s, nblk, \
                       a, b, pre_silu, (T_*)y, DK_)
    if (dkp) {
        if (dtype == RHO_BF16) RHO_GNA(bf16_raw, true, *dkp);
        else RHO_GNA(float, true, *dkp);
    } else {
        if (dtype == RHO_BF16) RHO_GNA(bf16_raw, false, DropK{});
        else RHO_GNA(float, false, DropK{});
    }
#undef RHO_GNA
    RHO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Channel sums of a channels-last tensor (bias gradients, additive-embedding gradients):
// finalisation of rho_gn_partial's first 8 lanes.   out_nc[n][c] = sum_pos x ;  out_c[c] (+)= sum_n out_nc
__global__ __launch_bounds__(256) void k_chan_sum_finalize(const float* __restrict__ partials, int c, int nblk,
                                                           float* __restrict__ out_nc, int64_t nc_stride, int acc_nc) {
    const int n = blockIdx.x;
    const int OCT = c >> 3;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        const int o = ch >> 3, j = ch & 7;
        double acc = 0.0;
        for (int k = 0; k < nblk; ++k) acc += (double)partials[(((int64_t)n * nblk + k) * OCT + o) * 16 + j];
        float* dst = out_nc + (int64_t)n * nc_stride + ch;
        *dst = acc_nc ? *dst + (float)acc : (float)acc;
    }
}

extern "C" int rho_chan_sum(const void* x, int dtype, int64_t n, int64_t s, int64_t c, float* partials, float* out_nc,
                            int64_t nc_stride, int acc_nc, float* out_c, int acc_c, void* stream) {
    if (!x || !partials || !out_nc || n <= 0 || s <= 0 || c <= 0 || c % 32 != 0) return RHO_E_ARG;
    int rc = rho_gn_partial(x, c, nullptr, 0, dtype, n, s, partials, stream);
    if (rc != 0) return rc;
    const int nblk = rho_gn_nblk(s);
    hipLaunchKernelGGL(k_chan_sum_finalize, dim3((unsigned)n), dim3(256), 0, as_stream(stream), partials, (int)c, nblk, out_nc,
                       nc_stride > 0 ? nc_stride : c, acc_nc);
    if (out_c) {
        if (nc_stride > 0 && nc_stride != c) return RHO_E_ARG;
        hipLaunchKernelGGL(k_sum_over_n, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, as_stream(stream), out_nc, out_c, (int)n,
                           (int)c, acc_c);
    }
    RHO_LAUNCH_CHECK();
    return 0;
}
