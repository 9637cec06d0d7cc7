// Tail of the legacy UNet block (rho_diffusion/models/unet.py:117-135, "UNet v1": SURVEY 8f row 4):
//     h = act(conv2(act(conv1 x))) + residual_conv(x) + time_pe[n, c];   out = act(GroupNorm(groups, C)(h))
// The convolutions run on k_conv / k_wgrad (conv.hip, wgrad.hip); what the UNetv2 kernels do not cover is here: GroupNorm with an
// arbitrary group count (v1 builds nn.GroupNorm(8, C); groupnorm.hip is specialised for the 32 groups of GroupNorm32), ReLU / GELU
// beside SiLU, and the activation applied BEFORE a residual add.  HBM-bound elementwise / reduction kernels on channels-last
// [N, S, C] tensors; v1 is a legacy model no shipped script instantiates, so these are written for clarity, not for the last GB/s.
#include <math.h>

#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const T* p, int64_t i);
template <> __device__ __forceinline__ float ldf<float>(const float* p, int64_t i) { return p[i]; }
template <> __device__ __forceinline__ float ldf<bf16_raw>(const bf16_raw* p, int64_t i) { return bf16_to_f32(p[i]); }
template <typename T> __device__ __forceinline__ void stf(T* p, int64_t i, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, int64_t i, float v) { p[i] = v; }
template <> __device__ __forceinline__ void stf<bf16_raw>(bf16_raw* p, int64_t i, float v) { p[i] = f32_to_bf16(v); }

// act codes of the C ABI: 0 identity, 1 SiLU, 2 ReLU, 3 GELU (erf form = nn.GELU() default)
__device__ __forceinline__ float act_f(float u, int act) {
    switch (act) {
        case 1: return u / (1.0f + expf(-u));
        case 2: return u > 0.0f ? u : 0.0f;
        case 3: return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f));
        default: return u;
    }
}
__device__ __forceinline__ float dact_f(float u, int act) {
    switch (act) {
        case 1: { const float s = 1.0f / (1.0f + expf(-u)); return s * (1.0f + u * (1.0f - s)); }
        case 2: return u > 0.0f ? 1.0f : 0.0f;
        case 3: return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * 0.3989422804014327f * expf(-0.5f * u * u);
        default: return 1.0f;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_act_add(const T* __restrict__ x, const T* __restrict__ r, const float* __restrict__ nc,
                                                 T* __restrict__ out, int64_t s, int64_t c, int64_t total, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = act_f(ldf<T>(x, i), act);
        if (r) v += ldf<T>(r, i);
        if (nc) v += nc[(i / (s * c)) * c + i % c];
        stf<T>(out, i, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_act_bwd(const T* __restrict__ x, const T* __restrict__ dout, T* __restrict__ dx,
                                                 int64_t total, int act) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        stf<T>(dx, i, ldf<T>(dout, i) * dact_f(ldf<T>(x, i), act));
}

// block-wide sum of (a, b) in double, fixed order (thread 0 adds the 256 partials): reproducible statistics
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sh) {
    const int tid = threadIdx.x;
    sh[tid] = a;
    sh[256 + tid] = b;
    __syncthreads();
    if (tid == 0) {
        double sa = 0.0, sb = 0.0;
        for (int k = 0; k < 256; ++k) { sa += sh[k]; sb += sh[256 + k]; }
        sh[0] = sa;
        sh[256] = sb;
    }
    __syncthreads();
    a = sh[0];
    b = sh[256];
    __syncthreads();
}

// One workgroup per (group, sample): statistics over its S x cpg elements (biased variance, as nn.GroupNorm), then the apply pass.
template <typename T>
__global__ __launch_bounds__(256) void k_gn_groups_fwd(const T* __restrict__ x, T* __restrict__ y, float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, int64_t s,
                                                       int c, int groups, float eps, int act) {
    __shared__ double sh[512];
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = c / groups;
    const int64_t m = s * cpg;
    const T* xs = x + (int64_t)n * s * c + (int64_t)g * cpg;
    double su = 0.0, sq = 0.0;
    {
        float fs = 0.0f, fq = 0.0f;
        int cnt = 0;
        for (int64_t e = tid; e < m; e += 256) {
            const float v = ldf<T>(xs, (e / cpg) * c + e % cpg);
            fs += v;
            fq = fmaf(v, v, fq);
            if (++cnt == 64) { su += fs; sq += fq; fs = fq = 0.0f; cnt = 0; }      // fp32 runs of 64, carried in fp64
        }
        su += fs;
        sq += fq;
    }
    block_sum2(su, sq, sh);
    const double mean = su / (double)m;
    double var = sq / (double)m - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps)), mu = (float)mean;
    if (tid == 0) {
        stats[((int64_t)n * groups + g) * 2 + 0] = mu;
        stats[((int64_t)n * groups + g) * 2 + 1] = rstd;
    }
    T* ys = y + (int64_t)n * s * c + (int64_t)g * cpg;
    for (int64_t e = tid; e < m; e += 256) {
        const int j = (int)(e % cpg);
        const int64_t off = (e / cpg) * c + j;
        const float xh = (ldf<T>(xs, off) - mu) * rstd;
        stf<T>(ys, off, act_f(fmaf(xh, gamma[g * cpg + j], beta[g * cpg + j]), act));
    }
}

// Backward of y = act(gamma * xhat + beta): per (group, sample) workgroup.  Pass 1: gq = dy * act'(u); per-channel sums of gq and
// gq * xhat (LDS float atomics, then one global atomic per channel: the sum over samples) and the two group sums; pass 2:
// dx = rstd * (gamma gq - mean_grp(gamma gq) - xhat * mean_grp(gamma gq xhat)).
template <typename T>
__global__ __launch_bounds__(256) void k_gn_groups_bwd(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       int64_t s, int c, int groups, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem_v1[];
    double* sh = reinterpret_cast<double*>(smem_v1);                 // [512]
    float* chg = reinterpret_cast<float*>(smem_v1 + 512 * 8);        // [cpg] sum gq * xhat
    float* chb = chg + (c / groups);                                 // [cpg] sum gq
    const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
    const int cpg = c / groups;
    const int64_t m = s * cpg;
    const float mu = stats[((int64_t)n * groups + g) * 2 + 0], rstd = stats[((int64_t)n * groups + g) * 2 + 1];
    const T* xs = x + (int64_t)n * s * c + (int64_t)g * cpg;
    const T* ds = dy + (int64_t)n * s * c + (int64_t)g * cpg;
    for (int j = tid; j < cpg; j += 256) chg[j] = chb[j] = 0.0f;
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int64_t e = tid; e < m; e += 256) {
        const int j = (int)(e % cpg);
        const int64_t off = (e / cpg) * c + j;
        const float ga = gamma[g * cpg + j];
        const float xh = (ldf<T>(xs, off) - mu) * rstd;
        const float gq = ldf<T>(ds, off) * dact_f(fmaf(xh, ga, beta[g * cpg + j]), act);
        atomicAdd(&chb[j], gq);
        atomicAdd(&chg[j], gq * xh);
        s1 += (double)(gq * ga);
        s2 += (double)(gq * ga * xh);
    }
    block_sum2(s1, s2, sh);            // (its barriers also publish the LDS channel sums)
    for (int j = tid; j < cpg; j += 256) {
        atomicAdd(dgamma + g * cpg + j, chg[j]);
        atomicAdd(dbeta + g * cpg + j, chb[j]);
    }
    const float m1 = (float)(s1 / (double)m), m2 = (float)(s2 / (double)m);
    T* dxs = dx + (int64_t)n * s * c + (int64_t)g * cpg;
    for (int64_t e = tid; e < m; e += 256) {
        const int j = (int)(e % cpg);
        const int64_t off = (e / cpg) * c + j;
        const float ga = gamma[g * cpg + j];
        const float xh = (ldf<T>(xs, off) - mu) * rstd;
        const float gq = ldf<T>(ds, off) * dact_f(fmaf(xh, ga, beta[g * cpg + j]), act);
        stf<T>(dxs, off, rstd * (ga * gq - m1 - xh * m2));
    }
}

inline unsigned grid1d(int64_t total) {
    int64_t g = (total + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

}  // namespace

extern "C" int rho_act_add(const void* x, const void* r, const float* nc, void* out, int dtype, int64_t n, int64_t s, int64_t c, int act,
                           void* stream) {
    if (!x || !out || n <= 0 || s <= 0 || c <= 0 || act < 0 || act > 3) return RHO_E_ARG;
    const int64_t total = n * s * c;
    if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_act_add<float>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)r, nc,
                           (float*)out, s, c, total, act);
    else if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_act_add<bf16_raw>, dim3(grid1d(total)), dim3(256), 0, as_stream(stream), (const bf16_raw*)x,
                           (const bf16_raw*)r, nc, (bf16_raw*)out, s, c, total, act);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

extern "C" int rho_act_bwd(const void* x, const void* dout, void* dx, int dtype, int64_t numel, int act, void* stream) {
    if (!x || !dout || !dx || numel <= 0 || act < 0 || act > 3) return RHO_E_ARG;
    if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_act_bwd<float>, dim3(grid1d(numel)), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)dout,
                           (float*)dx, numel, act);
    else if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_act_bwd<bf16_raw>, dim3(grid1d(numel)), dim3(256), 0, as_stream(stream), (const bf16_raw*)x,
                           (const bf16_raw*)dout, (bf16_raw*)dx, numel, act);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

extern "C" int rho_groupnorm_act(const void* x, void* y, float* stats, const float* gamma, const float* beta, int dtype, int64_t n,
                                 int64_t s, int64_t c, int64_t groups, float eps, int act, void* stream) {
    if (!x || !y || !stats || !gamma || !beta || n <= 0 || s <= 0 || c <= 0 || groups <= 0 || c % groups != 0 || act < 0 || act > 3)
        return RHO_E_ARG;
    if (n > 65535 || groups > 65535) return RHO_E_SHAPE;
    dim3 grid((unsigned)groups, (unsigned)n);
    if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_gn_groups_fwd<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, stats, gamma, beta, s,
                           (int)c, (int)groups, eps, act);
    else if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_gn_groups_fwd<bf16_raw>, grid, dim3(256), 0, as_stream(stream), (const bf16_raw*)x, (bf16_raw*)y, stats,
                           gamma, beta, s, (int)c, (int)groups, eps, act);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

extern "C" int rho_groupnorm_act_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta, void* dx,
                                     float* dgamma, float* dbeta, int dtype, int64_t n, int64_t s, int64_t c, int64_t groups, int act,
                                     void* stream) {
    if (!x || !dy || !stats || !gamma || !beta || !dx || !dgamma || !dbeta || n <= 0 || s <= 0 || c <= 0 || groups <= 0 ||
        c % groups != 0 || act < 0 || act > 3)
        return RHO_E_ARG;
    if (n > 65535 || groups > 65535) return RHO_E_SHAPE;
    const size_t lds = 512 * 8 + 2 * (size_t)(c / groups) * sizeof(float);
    if (lds > 64 * 1024) return RHO_E_SHAPE;
    dim3 grid((unsigned)groups, (unsigned)n);
    if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_gn_groups_bwd<float>, grid, dim3(256), lds, as_stream(stream), (const float*)x, (const float*)dy, stats, gamma,
                           beta, (float*)dx, dgamma, dbeta, s, (int)c, (int)groups, act);
    else if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_gn_groups_bwd<bf16_raw>, grid, dim3(256), lds, as_stream(stream), (const bf16_raw*)x, (const bf16_raw*)dy,
                           stats, gamma, beta, (bf16_raw*)dx, dgamma, dbeta, s, (int)c, (int)groups, act);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}
