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
    int64_t nb = (s + 2047) / 2048;
    if (nb < 1) nb = 1;
    if (nb > 64) nb = 64;
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
        for (int64_t p = p0 + pl; p < p1; p += ppi) {
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
