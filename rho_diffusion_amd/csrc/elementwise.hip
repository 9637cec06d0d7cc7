// HBM-bound kernels of the DDPM loops and the small embedding GEMVs.
// Each kernel is a grid-stride loop over 16-byte-per-lane accesses (coalesced 1 KiB per wave
// instruction), grid capped at 256 CUs x 8 blocks.
#include <cstdlib>

#include "common.h"

static inline int grid_for(int64_t work_items, int block) {
    int64_t g = (work_items + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" int rho_abi_version(void) { return RHO_ABI_VERSION; }

// Process-wide reproducibility switch (RHO_DETERMINISTIC=1, or rho_set_deterministic): kernels that combine partial sums with
// fp32 atomics (linear backward, label-embedding backward) take an ordered path instead; the weight gradient's ordered flush
// needs a workspace and is chosen by the caller (rho_conv_nd_wgrad_ws), who reads this flag.
static int g_deterministic = -1;
extern "C" int rho_get_deterministic(void) {
    if (g_deterministic < 0) {
        const char* e = getenv("RHO_DETERMINISTIC");
        const char* e2 = getenv("RHO_WGRAD_DETERMINISTIC");
        g_deterministic = ((e && atoi(e) != 0) || (e2 && atoi(e2) != 0)) ? 1 : 0;
    }
    return g_deterministic;
}
extern "C" int rho_set_deterministic(int on) {
    const int old = rho_get_deterministic();
    g_deterministic = on ? 1 : 0;
    return old;
}
#ifndef RHO_BUILD_ID
#define RHO_BUILD_ID "unstamped"
#endif
extern "C" const char* rho_build_info(void) {
    return "librho_hip gfx950 (CDNA4) hipcc -O3; MFMA 16x16x32 + 32x32x16 bf16 / 32x32x2 f32; build " RHO_BUILD_ID;
}

// ----------------------------------------------------------------------------- q_sample
// ddpm.py:122-129: x_t = sqrt(abar_t)*x0 + sqrt(1-abar_t)*eps, abar gathered per batch element.
__global__ __launch_bounds__(256) void k_q_sample(const float4* __restrict__ x0, const float4* __restrict__ eps,
                                                  float4* __restrict__ xt, const float* __restrict__ abar,
                                                  const int64_t* __restrict__ t, int64_t per4, int64_t total4,
                                                  int64_t table_len, int32_t* nan_flag) {
    bool bad = false, oob = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / per4;
        int64_t tb = t[b];
        if (tb < 0 || tb >= table_len) { oob = true; tb = tb < 0 ? 0 : table_len - 1; }   // the reference raises IndexError
        const float ab = abar[tb];
        const float sa = sqrtf(ab), sb = sqrtf(1.0f - ab);
        const float4 a = x0[i], e = eps[i];
        float4 r;
        r.x = sa * a.x + sb * e.x;
        r.y = sa * a.y + sb * e.y;
        r.z = sa * a.z + sb * e.z;
        r.w = sa * a.w + sb * e.w;
        bad |= (r.x != r.x) | (r.y != r.y) | (r.z != r.z) | (r.w != r.w);
        xt[i] = r;
    }
    const int any_bad = __any(bad), any_oob = __any(oob);
    if (nan_flag != nullptr && (any_bad || any_oob)) {
        if ((threadIdx.x & 63) == 0) atomicOr(nan_flag, (any_bad ? 1 : 0) | (any_oob ? 4 : 0));
    }
}

__global__ void k_q_sample_tail(const float* x0, const float* eps, float* xt, const float* abar, const int64_t* t,
                                int64_t per_sample, int64_t batch, int64_t table_len, int32_t* nan_flag) {
    // scalar path for per_sample % 4 != 0 (only tiny test shapes)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch * per_sample; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t tb = t[i / per_sample];
        if (tb < 0 || tb >= table_len) {
            if (nan_flag) atomicOr(nan_flag, 4);
            tb = tb < 0 ? 0 : table_len - 1;
        }
        const float ab = abar[tb];
        const float r = sqrtf(ab) * x0[i] + sqrtf(1.0f - ab) * eps[i];
        if (r != r && nan_flag) atomicOr(nan_flag, 1);
        xt[i] = r;
    }
}

extern "C" int rho_q_sample(const float* x0, const float* eps, float* x_t, const float* alpha_bar, const int64_t* t,
                            int64_t batch, int64_t per_sample, int64_t table_len, int32_t* nan_flag, void* stream) {
    if (!x0 || !eps || !x_t || !alpha_bar || !t || batch <= 0 || per_sample <= 0 || table_len <= 0) return RHO_E_ARG;
    if (per_sample % 4 == 0 && (((uintptr_t)x0 | (uintptr_t)eps | (uintptr_t)x_t) & 15) == 0) {
        const int64_t total4 = batch * per_sample / 4;
        hipLaunchKernelGGL(k_q_sample, dim3(grid_for(total4, 256)), dim3(256), 0, as_stream(stream), (const float4*)x0,
                           (const float4*)eps, (float4*)x_t, alpha_bar, t, per_sample / 4, total4, table_len, nan_flag);
    } else {
        hipLaunchKernelGGL(k_q_sample_tail, dim3(grid_for(batch * per_sample, 256)), dim3(256), 0, as_stream(stream), x0, eps,
                           x_t, alpha_bar, t, per_sample, batch, table_len, nan_flag);
    }
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- p_sample_step
// ddpm.py:210-218.  coef row = {1/sqrt(alpha), beta/sqrt(1-abar), 0.8*sqrt(beta)}; t read on device so
// that one captured graph serves every step; t == 0 => no update (q3), t <= 1 => z ignored.
__global__ __launch_bounds__(256) void k_p_sample(float* __restrict__ x, const float* __restrict__ eh,
                                                  const float* __restrict__ z, const float* __restrict__ coef,
                                                  const int32_t* __restrict__ t_dev, int64_t n) {
    const int t = *t_dev;
    if (t <= 0) return;
    const float c0 = coef[3 * t + 0], c1 = coef[3 * t + 1], c2 = (t > 1 && z != nullptr) ? coef[3 * t + 2] : 0.0f;
    const int64_t n4 = n >> 2;
    const bool vec = ((((uintptr_t)x | (uintptr_t)eh | (uintptr_t)z) & 15) == 0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        float4* x4 = (float4*)x;
        const float4* e4 = (const float4*)eh;
        const float4* z4 = (const float4*)z;
        for (int64_t i = tid; i < n4; i += stride) {
            float4 a = x4[i];
            const float4 e = e4[i];
            float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c2 != 0.0f) zz = z4[i];
            a.x = fminf(fmaxf(c0 * (a.x - c1 * e.x) + c2 * zz.x, -1.0f), 1.0f);
            a.y = fminf(fmaxf(c0 * (a.y - c1 * e.y) + c2 * zz.y, -1.0f), 1.0f);
            a.z = fminf(fmaxf(c0 * (a.z - c1 * e.z) + c2 * zz.z, -1.0f), 1.0f);
            a.w = fminf(fmaxf(c0 * (a.w - c1 * e.w) + c2 * zz.w, -1.0f), 1.0f);
            x4[i] = a;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {
            const float zz = (c2 != 0.0f) ? z[i] : 0.0f;
            x[i] = fminf(fmaxf(c0 * (x[i] - c1 * eh[i]) + c2 * zz, -1.0f), 1.0f);
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            const float zz = (c2 != 0.0f) ? z[i] : 0.0f;
            x[i] = fminf(fmaxf(c0 * (x[i] - c1 * eh[i]) + c2 * zz, -1.0f), 1.0f);
        }
    }
}

extern "C" int rho_p_sample_step(float* x, const float* eps_hat, const float* z, const float* coef_table,
                                 const int32_t* t_dev, int64_t n, void* stream) {
    if (!x || !eps_hat || !coef_table || !t_dev || n <= 0) return RHO_E_ARG;
    hipLaunchKernelGGL(k_p_sample, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), x, eps_hat, z, coef_table,
                       t_dev, n);
    RHO_LAUNCH_CHECK();
    return 0;
}

// t <- t - 1 and philox offset += delta: keeps the sampling loop's step state on the device
__global__ void k_step_advance(int32_t* t_dev, uint64_t* offset_dev, uint64_t delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (t_dev) *t_dev -= 1;
        if (offset_dev) *offset_dev += delta;
    }
}
extern "C" int rho_step_advance(int32_t* t_dev, uint64_t* offset_dev, uint64_t delta, void* stream) {
    hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, as_stream(stream), t_dev, offset_dev, delta);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- Philox4x32-10 normals
__device__ __forceinline__ void box_muller(uint32_t u0, uint32_t u1, float& a, float& b) {
    const float f0 = ((float)(u0 >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float f1 = ((float)(u1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * __logf(f0));
    float s, c;
    __sincosf(6.283185307179586f * f1, &s, &c);
    a = r * c;
    b = r * s;
}

__global__ __launch_bounds__(256) void k_philox_normal(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset,
                                                       const uint64_t* __restrict__ offset_dev) {
    const uint64_t base = offset_dev ? *offset_dev : offset;
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r[4];
        philox4x32_10(base + (uint64_t)i, seed, r);
        float v[4];
        box_muller(r[0], r[1], v[0], v[1]);
        box_muller(r[2], r[3], v[2], v[3]);
        const int64_t e = i << 2;
        if (e + 3 < n && (((uintptr_t)out & 15) == 0)) {
            *reinterpret_cast<float4*>(out + e) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            for (int k = 0; k < 4; ++k)
                if (e + k < n) out[e + k] = v[k];
        }
    }
}

extern "C" int rho_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream) {
    if (!out || n <= 0) return RHO_E_ARG;
    hipLaunchKernelGGL(k_philox_normal, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), out, n, seed, offset,
                       offset_dev);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- MSE (+grad)
__global__ __launch_bounds__(256) void k_mse(const float* __restrict__ a, const float* __restrict__ b, float* loss,
                                             float* __restrict__ grad, int64_t n, float inv_n) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        acc += d * d;
        if (grad) grad[i] = 2.0f * d * inv_n;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv_n);
}

extern "C" int rho_mse(const float* a, const float* b, float* loss, float* grad_a, int64_t n, void* stream) {
    if (!a || !b || !loss || n <= 0) return RHO_E_ARG;
    hipError_t e = hipMemsetAsync(loss, 0, sizeof(float), as_stream(stream));
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_mse, dim3(grid_for(n, 256 * 8)), dim3(256), 0, as_stream(stream), a, b, loss, grad_a, n, 1.0f / (float)n);
    RHO_LAUNCH_CHECK();
    return 0;
}

// Ordered form: block b stores its partial to partials[b]; one wave adds them in index order -> the loss is bit-reproducible
// (the atomic form above adds the blocks in arrival order).  grad is the same either way.
__global__ __launch_bounds__(256) void k_mse_part(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ partials,
                                                  float* __restrict__ grad, int64_t n, float inv_n, int vec) {
    __shared__ float red[4];
    float acc = 0.0f;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if (vec) {                                   // 16-byte pieces (all three pointers 16-byte aligned): the HBM-rate path
        const int64_t n4 = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        float4* g4 = reinterpret_cast<float4*>(grad);
        const float k2 = 2.0f * inv_n;
        for (int64_t i = tid; i < n4; i += nthr) {
            const float4 x = a4[i], y = b4[i];
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            acc += d0 * d0; acc += d1 * d1; acc += d2 * d2; acc += d3 * d3;
            if (grad) g4[i] = make_float4(k2 * d0, k2 * d1, k2 * d2, k2 * d3);
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < n; i += nthr) {
        const float d = a[i] - b[i];
        acc += d * d;
        if (grad) grad[i] = 2.0f * d * inv_n;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}
__global__ __launch_bounds__(64) void k_mse_final(const float* __restrict__ partials, int np, float* __restrict__ loss) {
    float acc = 0.0f;
    for (int i = threadIdx.x; i < np; i += 64) acc += partials[i];
    acc = wave_sum(acc);
    if (threadIdx.x == 0) *loss = acc;
}

extern "C" int rho_mse_ws(const float* a, const float* b, float* loss, float* grad_a, int64_t n, float* partials, int64_t n_partials,
                          void* stream) {
    if (!a || !b || !loss || !partials || n <= 0 || n_partials < 1) return RHO_E_ARG;
    int64_t g = grid_for(n, 256 * 8);
    if (g > n_partials) g = n_partials;
    const int vec = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)grad_a) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(k_mse_part, dim3((unsigned)g), dim3(256), 0, as_stream(stream), a, b, partials, grad_a, n, 1.0f / (float)n, vec);
    hipLaunchKernelGGL(k_mse_final, dim3(1), dim3(64), 0, as_stream(stream), partials, (int)g, loss);
    RHO_LAUNCH_CHECK();
    return 0;
}

// mean over all non-batch axes (layers.py:105-110, mean_flat): one workgroup per sample, fp32 in a fixed order
__global__ __launch_bounds__(256) void k_mean_flat(const float* __restrict__ x, float* __restrict__ out, int64_t per_sample) {
    __shared__ float red[4];
    const float* xs = x + (int64_t)blockIdx.x * per_sample;
    float acc = 0.0f;
    for (int64_t i = threadIdx.x; i < per_sample; i += 256) acc += xs[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)per_sample;
}
extern "C" int rho_mean_flat(const float* x, float* out, int64_t batch, int64_t per_sample, void* stream) {
    if (!x || !out || batch <= 0 || per_sample <= 0 || batch > 0x7FFFFFFF) return RHO_E_ARG;
    hipLaunchKernelGGL(k_mean_flat, dim3((unsigned)batch), dim3(256), 0, as_stream(stream), x, out, per_sample);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- AdamW
__global__ __launch_bounds__(256) void k_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                               float wd, float bc1, float rsqrt_bc2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float pi = p[i] * (1.0f - lr * wd);
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        const float denom = sqrtf(vi) * rsqrt_bc2 + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
    }
}

extern "C" int rho_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int32_t step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return RHO_E_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adamw, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, (float)bc1, (float)(1.0 / sqrt(bc2)));
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- embeddings
// One wave per output feature: the weight row stays in registers while the wave walks the batch.
// (B <= a few hundred, in_dim <= 1024: launch-latency bound)
template <int MAXK>  // in_dim <= 64*MAXK
__global__ __launch_bounds__(256) void k_linear(const float* __restrict__ x, const float* __restrict__ w,
                                                const float* __restrict__ bias, const float* __restrict__ add,
                                                float* __restrict__ out, int batch, int in_dim, int out_dim, int act_in,
                                                int act_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= out_dim) return;
    float wr[MAXK];
#pragma unroll
    for (int j = 0; j < MAXK; ++j) {
        const int k = lane + 64 * j;
        wr[j] = (k < in_dim) ? w[(int64_t)o * in_dim + k] : 0.0f;
    }
    const float bo = bias ? bias[o] : 0.0f;
    for (int b = 0; b < batch; ++b) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < MAXK; ++j) {
            const int k = lane + 64 * j;
            if (k < in_dim) {
                float xv = x[(int64_t)b * in_dim + k];
                if (act_in) xv = act_in == 1 ? xv / (1.0f + expf(-xv)) : act_other_f(xv, act_in);
                acc = fmaf(xv, wr[j], acc);
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            float r = acc + bo;
            if (add) r += add[(int64_t)b * out_dim + o];
            if (act_out) r = act_out == 1 ? r / (1.0f + expf(-r)) : act_other_f(r, act_out);
            out[(int64_t)b * out_dim + o] = r;
        }
    }
}

extern "C" int rho_linear(const float* x, const float* w, const float* bias, const float* add, float* out, int64_t batch,
                          int64_t in_dim, int64_t out_dim, int act_in, int act_out, void* stream) {
    if (!x || !w || !out || batch <= 0 || in_dim <= 0 || out_dim <= 0 || in_dim > 2048) return RHO_E_ARG;
    dim3 grid((unsigned)((out_dim + 3) / 4)), block(256);
    if (in_dim <= 256)
        hipLaunchKernelGGL(k_linear<4>, grid, block, 0, as_stream(stream), x, w, bias, add, out, (int)batch, (int)in_dim,
                           (int)out_dim, act_in, act_out);
    else if (in_dim <= 1024)
        hipLaunchKernelGGL(k_linear<16>, grid, block, 0, as_stream(stream), x, w, bias, add, out, (int)batch, (int)in_dim,
                           (int)out_dim, act_in, act_out);
    else
        hipLaunchKernelGGL(k_linear<32>, grid, block, 0, as_stream(stream), x, w, bias, add, out, (int)batch, (int)in_dim,
                           (int)out_dim, act_in, act_out);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- layout
template <typename T>
__device__ __forceinline__ T cvt_out(float v);
template <>
__device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_raw cvt_out<bf16_raw>(float v) { return f32_to_bf16(v); }

template <typename T>
__global__ __launch_bounds__(256) void k_pack_input(const float* __restrict__ x, T* __restrict__ y, int64_t n, int64_t c,
                                                    int64_t s, int64_t cpad) {
    // thread per (sample, position, 16-byte piece): consecutive lanes write consecutive 16-byte pieces of the
    // channels-last rows (one scalar 2-byte store per channel ran at 0.49 TB/s); the reads of channel ch are
    // coalesced across the lanes that share a piece index
    constexpr int PE = 16 / (int)sizeof(T);
    const int64_t ppr = cpad / PE;                      // pieces per row (cpad is a multiple of 32)
    const int64_t total = n * s * ppr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t piece = i % ppr, pos = i / ppr;
        const int64_t b = pos / s, p = pos % s;
        float v[PE];
#pragma unroll
        for (int j = 0; j < PE; ++j) {
            const int64_t ch = piece * PE + j;
            v[j] = ch < c ? x[(b * c + ch) * s + p] : 0.0f;
        }
        uint4 o;
        if constexpr (sizeof(T) == 2) {
            o = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
            o = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
        }
        reinterpret_cast<uint4*>(y)[i] = o;
    }
}

extern "C" int rho_pack_input(const float* x, void* y, int dtype, int64_t n, int64_t c, int64_t s, int64_t cpad, void* stream) {
    if (!x || !y || n <= 0 || c <= 0 || s <= 0 || cpad < c) return RHO_E_ARG;
    if (cpad % 8 != 0) return RHO_E_ALIGN;
    dim3 grid(grid_for(n * s * (cpad / (dtype == RHO_BF16 ? 8 : 4)), 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_pack_input<bf16_raw>, grid, block, 0, as_stream(stream), x, (bf16_raw*)y, n, c, s, cpad);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_pack_input<float>, grid, block, 0, as_stream(stream), x, (float*)y, n, c, s, cpad);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// The two convolutions at the ends of the UNet have one channel on one side (1 -> mc stem, mc -> 1 head): as 3x3x3
// implicit GEMMs they pad that side to 32 and waste 31/32 of the matrix work (1.2 + 1.6 ms of a 137 ms step).  Both are
// 1x1x1 GEMMs in disguise:
//   stem:  y[pos][co] = sum_k W[co][k] * X27[pos][k],  X27[pos][ci * taps + tap] = x[ci][pos + off(tap)]   (k_im2col_taps)
//   head:  out[pos]   = bias + sum_tap T[pos + off(tap)][tap],  T[q][tap] = sum_c W[tap][c] * act[q][c]     (k_tap_gather_sum)
// so the engine's inference plans run them through the 1x1x1 path with these two HBM-rate helpers around it.
// (kernel extents are template parameters: with run-time extents the tap -> (dz, dy, dx) divisions of the 32 columns cost 3000
//  VALU instructions per position, 0.5 ms for 8.4 M positions instead of the 0.15 ms the bytes take)
// Round 4: the 1-channel 3x3x3 case (also the GEMM-shaped backward of the stem / head in training plans) with FOUR threads per
// position, one 16-byte piece (8 taps) each: consecutive threads write consecutive 16 bytes - 1 KB per wave store instead of four
// 64-byte-strided stores per thread (1.5 -> ~3 TB/s on the 0.54 GB operand).
__global__ __launch_bounds__(256) void k_im2col_333_c1(const float* __restrict__ x, bf16_raw* __restrict__ out, int D, int H, int W,
                                                       int64_t total) {
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t i = gi >> 2;                        // output position
    const int piece = (int)(gi & 3);
    if (i >= total) return;
    const int64_t S = (int64_t)D * H * W;
    const int64_t n = i / S;
    const int64_t ps = i - n * S;
    const int w_ = (int)(ps % W), h_ = (int)((ps / W) % H), d_ = (int)(ps / ((int64_t)W * H));
    const float* xs = x + n * S + ps;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int kk = piece * 8 + j;
        float val = 0.0f;
        if (kk < 27) {
            const int dz = kk / 9 - 1, dy = (kk / 3) % 3 - 1, dx = kk % 3 - 1;
            const int z = d_ + dz, y = h_ + dy, xx = w_ + dx;
            if (z >= 0 && z < D && y >= 0 && y < H && xx >= 0 && xx < W) val = xs[((int64_t)dz * H + dy) * W + dx];
        }
        v[j] = val;
    }
    *reinterpret_cast<uint4*>(out + i * 32 + piece * 8) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                                     pack_bf16x2(v[6], v[7]));
}

template <int KD, int KH, int KW>
__global__ __launch_bounds__(256) void k_im2col_taps(const float* __restrict__ x, bf16_raw* __restrict__ out, int cin, int D, int H,
                                                     int W, int cpad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one output position
    if (i >= total) return;
    constexpr int TAPS = KD * KH * KW;
    const int64_t S = (int64_t)D * H * W;
    const int64_t n = i / S;
    const int64_t ps = i - n * S;
    const int w_ = (int)(ps % W), h_ = (int)((ps / W) % H), d_ = (int)(ps / ((int64_t)W * H));
    bf16_raw* o = out + i * cpad;
    const float* xs = x + n * cin * S + ps;
    for (int k0 = 0; k0 < cpad; k0 += 8) {
        float v[8];
        if (cin == 1 && k0 + 8 <= 32) {                 // the common case, fully unrolled: compile-time tap offsets
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.0f;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                if (kk >= k0 && kk < k0 + 8 && kk < TAPS) {
                    const int dz = kk / (KH * KW) - KD / 2, dy = (kk / KW) % KH - KH / 2, dx = kk % KW - KW / 2;
                    const int z = d_ + dz, y = h_ + dy, xx = w_ + dx;
                    if (z >= 0 && z < D && y >= 0 && y < H && xx >= 0 && xx < W) v[kk & 7] = xs[((int64_t)dz * H + dy) * W + dx];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                float val = 0.0f;
                if (k < cin * TAPS) {
                    const int ci = k / TAPS, tap = k - ci * TAPS;
                    const int dz = tap / (KH * KW) - KD / 2, dy = (tap / KW) % KH - KH / 2, dx = tap % KW - KW / 2;
                    const int z = d_ + dz, y = h_ + dy, xx = w_ + dx;
                    if (z >= 0 && z < D && y >= 0 && y < H && xx >= 0 && xx < W) val = xs[ci * S + ((int64_t)dz * H + dy) * W + dx];
                }
                v[j] = val;
            }
        }
        *reinterpret_cast<uint4*>(o + k0) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                       pack_bf16x2(v[6], v[7]));
    }
}

extern "C" int rho_im2col_taps(const float* x, void* out, int dtype, int64_t n, int64_t cin, int64_t d, int64_t h, int64_t w,
                               int kd, int kh, int kw, int64_t cpad, void* stream) {
    if (!x || !out || n <= 0 || cin <= 0 || d <= 0 || h <= 0 || w <= 0) return RHO_E_ARG;
    if (dtype != RHO_BF16) return RHO_E_ARG;
    if (cpad % 8 != 0 || cpad < cin * kd * kh * kw) return RHO_E_ALIGN;
    const int64_t total = n * d * h * w;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (kd == 3 && kh == 3 && kw == 3 && cin == 1 && cpad == 32 && total < (1LL << 40))
        hipLaunchKernelGGL(k_im2col_333_c1, dim3((unsigned)((total * 4 + 255) / 256)), block, 0, as_stream(stream), x, (bf16_raw*)out, (int)d,
                           (int)h, (int)w, total);
    else if (kd == 3 && kh == 3 && kw == 3)
        hipLaunchKernelGGL((k_im2col_taps<3, 3, 3>), grid, block, 0, as_stream(stream), x, (bf16_raw*)out, (int)cin, (int)d, (int)h, (int)w,
                           (int)cpad, total);
    else if (kd == 1 && kh == 3 && kw == 3)
        hipLaunchKernelGGL((k_im2col_taps<1, 3, 3>), grid, block, 0, as_stream(stream), x, (bf16_raw*)out, (int)cin, (int)d, (int)h, (int)w,
                           (int)cpad, total);
    else if (kd == 1 && kh == 1 && kw == 3)
        hipLaunchKernelGGL((k_im2col_taps<1, 1, 3>), grid, block, 0, as_stream(stream), x, (bf16_raw*)out, (int)cin, (int)d, (int)h, (int)w,
                           (int)cpad, total);
    else
        return RHO_E_SHAPE;
    RHO_LAUNCH_CHECK();
    return 0;
}

template <int KD, int KH, int KW>
__global__ __launch_bounds__(256) void k_tap_gather_sum(const bf16_raw* __restrict__ t, const float* __restrict__ bias,
                                                        float* __restrict__ out, int D, int H, int W, int cpad, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one output position
    if (i >= total) return;
    const int64_t S = (int64_t)D * H * W;
    const int64_t n = i / S;
    const int64_t ps = i - n * S;
    const int w_ = (int)(ps % W), h_ = (int)((ps / W) % H), d_ = (int)(ps / ((int64_t)W * H));
    const bf16_raw* tp = t + i * cpad;
    float v[KD * KH * KW];
#pragma unroll
    for (int tap = 0; tap < KD * KH * KW; ++tap) {          // all loads first (independent), then the ordered fp32 sum
        const int dz = tap / (KH * KW) - KD / 2, dy = (tap / KW) % KH - KH / 2, dx = tap % KW - KW / 2;
        const int z = d_ + dz, y = h_ + dy, xx = w_ + dx;
        const bool in = z >= 0 && z < D && y >= 0 && y < H && xx >= 0 && xx < W;
        v[tap] = in ? bf16_to_f32(tp[(((int64_t)dz * H + dy) * W + dx) * cpad + tap]) : 0.0f;
    }
    float acc = bias ? bias[0] : 0.0f;
#pragma unroll
    for (int tap = 0; tap < KD * KH * KW; ++tap) acc += v[tap];
    out[i] = acc;
}

extern "C" int rho_tap_gather_sum(const void* t, int dtype, int64_t n, int64_t d, int64_t h, int64_t w, int kd, int kh, int kw,
                                  int64_t cpad, const float* bias, float* out, void* stream) {
    if (!t || !out || n <= 0 || d <= 0 || h <= 0 || w <= 0) return RHO_E_ARG;
    if (dtype != RHO_BF16) return RHO_E_ARG;
    if (cpad < kd * kh * kw) return RHO_E_ALIGN;
    const int64_t total = n * d * h * w;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (kd == 3 && kh == 3 && kw == 3)
        hipLaunchKernelGGL((k_tap_gather_sum<3, 3, 3>), grid, block, 0, as_stream(stream), (const bf16_raw*)t, bias, out, (int)d, (int)h,
                           (int)w, (int)cpad, total);
    else if (kd == 1 && kh == 3 && kw == 3)
        hipLaunchKernelGGL((k_tap_gather_sum<1, 3, 3>), grid, block, 0, as_stream(stream), (const bf16_raw*)t, bias, out, (int)d, (int)h,
                           (int)w, (int)cpad, total);
    else if (kd == 1 && kh == 1 && kw == 3)
        hipLaunchKernelGGL((k_tap_gather_sum<1, 1, 3>), grid, block, 0, as_stream(stream), (const bf16_raw*)t, bias, out, (int)d, (int)h,
                           (int)w, (int)cpad, total);
    else
        return RHO_E_SHAPE;
    RHO_LAUNCH_CHECK();
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void k_prep_w(const float* __restrict__ w, T* __restrict__ out, int64_t cout, int64_t cin,
                                                int64_t taps, int64_t coutp, int64_t cinp, const int32_t* __restrict__ row_src) {
    const int64_t total = taps * coutp * cinp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ci = i % cinp;
        const int64_t co = (i / cinp) % coutp;
        const int64_t tap = i / (cinp * coutp);
        int64_t src = row_src ? (int64_t)row_src[co] : (co < cout ? co : -1);
        float v = 0.0f;
        if (src >= 0 && src < cout && ci < cin) v = w[(src * cin + ci) * taps + tap];
        out[i] = cvt_out<T>(v);
    }
}

extern "C" int rho_prep_conv_weight(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int64_t taps,
                                    int64_t coutp, int64_t cinp, const int32_t* row_src, void* stream) {
    if (!w || !out || cout <= 0 || cin <= 0 || taps <= 0 || cinp < cin || (!row_src && coutp < cout)) return RHO_E_ARG;
    dim3 grid(grid_for(taps * coutp * cinp, 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_prep_w<bf16_raw>, grid, block, 0, as_stream(stream), w, (bf16_raw*)out, cout, cin, taps, coutp, cinp, row_src);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_prep_w<float>, grid, block, 0, as_stream(stream), w, (float*)out, cout, cin, taps, coutp, cinp, row_src);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// One sub-pixel phase of a conv behind a nearest x2 upsample: on a phased axis (3 taps -> 2) tap r of parity a sums the source
// taps that land on the same source row: a = 0: {0}, {1, 2};  a = 1: {0, 1}, {2}.
__device__ __forceinline__ float phase_weight(const float* __restrict__ w, int64_t co, int64_t ci, int64_t cin, int tap2, int kd, int kh,
                                              int kw, int ph_h, int ph_w) {
    const int kh2 = ph_h ? 2 : kh, kw2 = ph_w ? 2 : kw;
    const int c2 = tap2 % kw2, r2 = (tap2 / kw2) % kh2, dz = tap2 / (kw2 * kh2);
    int rlo = r2, rhi = r2, clo = c2, chi = c2;          // source tap range per axis
    if (ph_h) { if (ph_h == 1) { rlo = r2 ? 1 : 0; rhi = r2 ? 2 : 0; } else { rlo = r2 ? 2 : 0; rhi = r2 ? 2 : 1; } }
    if (ph_w) { if (ph_w == 1) { clo = c2 ? 1 : 0; chi = c2 ? 2 : 0; } else { clo = c2 ? 2 : 0; chi = c2 ? 2 : 1; } }
    const float* base = w + (co * cin + ci) * ((int64_t)kd * kh * kw) + (int64_t)dz * kh * kw;
    float v = 0.0f;
    for (int r = rlo; r <= rhi; ++r)
        for (int c = clo; c <= chi; ++c) v += base[r * kw + c];
    return v;
}

// dgrad = 0: forward layout [tap2][coutp][cinp];  dgrad = 1: data-gradient layout [tap'][rowsp = ceil32(cin)][colsp] with flipped
// taps and transposed channels (as k_prep_w_dgrad), so the forward kernel run on the phase's dY computes its share of dX
template <typename T>
__global__ __launch_bounds__(256) void k_prep_w_phase(const float* __restrict__ w, T* __restrict__ out, int64_t cout, int64_t cin, int kd,
                                                      int kh, int kw, int ph_h, int ph_w, int64_t d1, int64_t d2, int dgrad) {
    const int64_t taps2 = (int64_t)kd * (ph_h ? 2 : kh) * (ph_w ? 2 : kw);
    const int64_t total = taps2 * d1 * d2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i2 = i % d2, i1 = (i / d2) % d1;
        const int tap = (int)(i / (d2 * d1));
        float v = 0.0f;
        if (!dgrad) {                 // d1 = coutp, d2 = cinp
            if (i1 < cout && i2 < cin) v = phase_weight(w, i1, i2, cin, tap, kd, kh, kw, ph_h, ph_w);
        } else {                      // d1 = rows (input channels), d2 = columns (output channels)
            if (i2 < cout && i1 < cin) v = phase_weight(w, i2, i1, cin, (int)(taps2 - 1 - tap), kd, kh, kw, ph_h, ph_w);
        }
        out[i] = cvt_out<T>(v);
    }
}

extern "C" int rho_prep_conv_weight_phase(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int kd, int kh, int kw, int ph_h,
                                          int ph_w, int64_t coutp, int64_t cinp, int dgrad, void* stream) {
    if (!w || !out || cout <= 0 || cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || cinp < cin || coutp < cout) return RHO_E_ARG;
    if (ph_h < 0 || ph_h > 2 || ph_w < 0 || ph_w > 2 || (!ph_h && !ph_w) || (ph_h && kh != 3) || (ph_w && kw != 3)) return RHO_E_ARG;
    const int64_t d1 = dgrad ? cinp : coutp, d2 = dgrad ? coutp : cinp;
    const int64_t total = (int64_t)kd * (ph_h ? 2 : kh) * (ph_w ? 2 : kw) * d1 * d2;
    dim3 grid(grid_for(total, 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_prep_w_phase<bf16_raw>, grid, block, 0, as_stream(stream), w, (bf16_raw*)out, cout, cin, kd, kh, kw, ph_h, ph_w, d1, d2, dgrad);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_prep_w_phase<float>, grid, block, 0, as_stream(stream), w, (float*)out, cout, cin, kd, kh, kw, ph_h, ph_w, d1, d2, dgrad);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// Tap selection for the parity splits of a stride-2 conv (rho_prep_conv_weight_sel): new tap r of an axis takes source tap
// sel >> (4 * r) & 15 of the 3-tap parameter (an axis that keeps its taps passes kh2 = kh and the identity selection).
template <typename T>
__global__ __launch_bounds__(256) void k_prep_w_sel(const float* __restrict__ w, T* __restrict__ out, int64_t cout, int64_t cin, int kd,
                                                    int kh, int kw, int kh2, int kw2, int sel_h, int sel_w, int flip_d, int64_t d1,
                                                    int64_t d2, int dgrad) {
    const int64_t taps2 = (int64_t)kd * kh2 * kw2, taps = (int64_t)kd * kh * kw;
    const int64_t total = taps2 * d1 * d2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i2 = i % d2, i1 = (i / d2) % d1;
        const int tap = (int)(i / (d2 * d1));
        const int c2 = tap % kw2, r2 = (tap / kw2) % kh2, dz = tap / (kw2 * kh2);
        const int ks = ((flip_d ? kd - 1 - dz : dz) * kh + ((sel_h >> (4 * r2)) & 15)) * kw + ((sel_w >> (4 * c2)) & 15);
        const int64_t co = dgrad ? i2 : i1, ci = dgrad ? i1 : i2;
        float v = 0.0f;
        if (co < cout && ci < cin) v = w[(co * cin + ci) * taps + ks];
        out[i] = cvt_out<T>(v);
    }
}

extern "C" int rho_prep_conv_weight_sel(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int kd, int kh, int kw, int kh2,
                                        int kw2, int sel_h, int sel_w, int flip_d, int64_t coutp, int64_t cinp, int dgrad, void* stream) {
    if (!w || !out || cout <= 0 || cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || kh2 <= 0 || kw2 <= 0 || kh2 > 4 || kw2 > 4 || cinp < cin ||
        coutp < cout)
        return RHO_E_ARG;
    for (int r = 0; r < kh2; ++r) if (((sel_h >> (4 * r)) & 15) >= kh) return RHO_E_ARG;
    for (int c = 0; c < kw2; ++c) if (((sel_w >> (4 * c)) & 15) >= kw) return RHO_E_ARG;
    const int64_t d1 = dgrad ? cinp : coutp, d2 = dgrad ? coutp : cinp;
    const int64_t total = (int64_t)kd * kh2 * kw2 * d1 * d2;
    dim3 grid(grid_for(total, 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_prep_w_sel<bf16_raw>, grid, block, 0, as_stream(stream), w, (bf16_raw*)out, cout, cin, kd, kh, kw, kh2, kw2, sel_h, sel_w,
                           flip_d, d1, d2, dgrad);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_prep_w_sel<float>, grid, block, 0, as_stream(stream), w, (float*)out, cout, cin, kd, kh, kw, kh2, kw2, sel_h, sel_w,
                           flip_d, d1, d2, dgrad);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// ================================================================================================ backward helpers

// dgrad weights: out[tap'][ci][co'] = w[src(co')][ci][taps-1-tap']  (flipped taps, transposed channels), so that the
// FORWARD kernel run on dY computes dX (autograd of conv_nd).  rows padded to rowsp (multiple of 32), cols to colsp.
template <typename T>
__global__ __launch_bounds__(256) void k_prep_w_dgrad(const float* __restrict__ w, T* __restrict__ out, int64_t cout, int64_t cin,
                                                      int64_t taps, int64_t rowsp, int64_t colsp, const int32_t* __restrict__ col_src) {
    const int64_t total = taps * rowsp * colsp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = i % colsp;            // original output channel (possibly permuted)
        const int64_t row = (i / colsp) % rowsp;  // original input channel
        const int64_t tap = i / (colsp * rowsp);
        int64_t src = col_src ? (col < cout ? (int64_t)col_src[col] : -1) : (col < cout ? col : -1);
        float v = 0.0f;
        if (src >= 0 && src < cout && row < cin) v = w[(src * cin + row) * taps + (taps - 1 - tap)];
        out[i] = cvt_out<T>(v);
    }
}

extern "C" int rho_prep_conv_weight_dgrad(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int64_t taps,
                                          int64_t rowsp, int64_t colsp, const int32_t* col_src, void* stream) {
    if (!w || !out || cout <= 0 || cin <= 0 || taps <= 0 || rowsp < cin || colsp < cout) return RHO_E_ARG;
    dim3 grid(grid_for(taps * rowsp * colsp, 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_prep_w_dgrad<bf16_raw>, grid, block, 0, as_stream(stream), w, (bf16_raw*)out, cout, cin, taps, rowsp, colsp, col_src);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_prep_w_dgrad<float>, grid, block, 0, as_stream(stream), w, (float*)out, cout, cin, taps, rowsp, colsp, col_src);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ batched weight preparation
// rho_prep_batch: every prepared layout of every convolution of a model (forward / data-gradient / sub-pixel phase / parity
// selection), the padded fp32 biases and the batched FiLM matrix in ONE launch after an optimizer step, driven by a device table of
// rho_prep_op.  The per-tensor entry points above cost ~170 launches of a few microseconds per training step at BASELINE
// configs[2] (2.7 ms) and as many again on the launch-bound 2-D configurations.  A block finds its op by bisection on blk0.
__device__ __forceinline__ float prep_elem(const rho_prep_op& o, int64_t i) {
    const int64_t taps = (int64_t)o.kd * o.kh * o.kw;
    switch (o.kind) {
    case RHO_PREP_FWD: {                       // [taps][coutp = d1][cinp = d2]
        const int64_t ci = i % o.d2, co = (i / o.d2) % o.d1, tap = i / (o.d2 * o.d1);
        const int64_t src = o.perm ? (int64_t)o.perm[co] : (co < o.cout ? co : -1);
        return (src >= 0 && src < o.cout && ci < o.cin) ? o.w[(src * o.cin + ci) * taps + tap] : 0.0f;
    }
    case RHO_PREP_DGRAD: {                     // [taps'][rowsp = d1][colsp = d2], flipped taps, transposed channels
        const int64_t col = i % o.d2, row = (i / o.d2) % o.d1, tap = i / (o.d2 * o.d1);
        const int64_t src = o.perm ? (col < o.cout ? (int64_t)o.perm[col] : -1) : (col < o.cout ? col : -1);
        return (src >= 0 && src < o.cout && row < o.cin) ? o.w[(src * o.cin + row) * taps + (taps - 1 - tap)] : 0.0f;
    }
    case RHO_PREP_PHASE: {
        const int64_t taps2 = (int64_t)o.kd * (o.ph_h ? 2 : o.kh) * (o.ph_w ? 2 : o.kw);
        const int64_t i2 = i % o.d2, i1 = (i / o.d2) % o.d1;
        const int tap = (int)(i / (o.d2 * o.d1));
        if (!o.dgrad) return (i1 < o.cout && i2 < o.cin) ? phase_weight(o.w, i1, i2, o.cin, tap, o.kd, o.kh, o.kw, o.ph_h, o.ph_w) : 0.0f;
        return (i2 < o.cout && i1 < o.cin) ? phase_weight(o.w, i2, i1, o.cin, (int)(taps2 - 1 - tap), o.kd, o.kh, o.kw, o.ph_h, o.ph_w) : 0.0f;
    }
    case RHO_PREP_SEL: {
        const int64_t i2 = i % o.d2, i1 = (i / o.d2) % o.d1;
        const int tap = (int)(i / (o.d2 * o.d1));
        const int c2 = tap % o.kw2, r2 = (tap / o.kw2) % o.kh2, dz = tap / (o.kw2 * o.kh2);
        const int ks = ((o.flip_d ? o.kd - 1 - dz : dz) * o.kh + ((o.sel_h >> (4 * r2)) & 15)) * o.kw + ((o.sel_w >> (4 * c2)) & 15);
        const int64_t co = o.dgrad ? i2 : i1, ci = o.dgrad ? i1 : i2;
        return (co < o.cout && ci < o.cin) ? o.w[(co * o.cin + ci) * taps + ks] : 0.0f;
    }
    default: {                                 // RHO_PREP_VEC: out[i] = w[perm ? perm[i] : i] for i < cout (perm < 0: zero), zero up to d1
        const int64_t src = (o.perm && i < o.cout) ? (int64_t)o.perm[i] : i;
        return (i < o.cout && src >= 0 && src < o.cin) ? o.w[src] : 0.0f;
    }
    }
}

__global__ __launch_bounds__(256) void k_prep_batch(const rho_prep_op* __restrict__ ops, int nops) {
    int lo = 0, hi = nops - 1;
    const int b = (int)blockIdx.x;
    while (lo < hi) {                          // last op whose first block is <= b
        const int mid = (lo + hi + 1) >> 1;
        if (ops[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    const rho_prep_op o = ops[lo];
    const int taps_ = o.kd * o.kh * o.kw;
    if ((o.kind == RHO_PREP_FWD || o.kind == RHO_PREP_DGRAD) && taps_ > 1 && taps_ <= 32) {
        // [row][channel][tap] -> [tap][..][..] through LDS (the elementwise walk reads the parameter with a stride of `taps` floats
        // between consecutive threads: 0.5 TB/s).  A unit = 64 consecutive entries of the output's contiguous axis for one entry of
        // its middle axis: FWD (co, 64 ci): one contiguous read of 64 * taps floats;  DGRAD (ci, 64 co): 64 runs of `taps` floats.
        // Either way every tap is written as one run of 64 contiguous outputs.
        __shared__ float tile[64 * 33];
        const bool fwd = o.kind == RHO_PREP_FWD;
        const int64_t mid = o.d1, inner = o.d2;                   // out [taps][d1][d2]
        const int64_t groups = (inner + 63) / 64;
        const int64_t units = mid * groups;
        for (int64_t u = b - o.blk0; u < units; u += o.nblk) {
            const int64_t m_ = u / groups;
            const int64_t i0 = (u % groups) * 64;
            for (int e = threadIdx.x; e < 64 * taps_; e += 256) {
                float v = 0.0f;
                if (fwd) {
                    // m_ = output row co (maybe permuted / padded), inner = ci: source run w[src][i0 .. i0 + 63][taps]
                    const int64_t src = o.perm ? (int64_t)o.perm[m_] : (m_ < o.cout ? m_ : -1);
                    const int64_t ci = i0 + e / taps_;
                    if (src >= 0 && src < o.cout && ci < o.cin) v = o.w[(src * o.cin + i0) * taps_ + e];
                    tile[(e / taps_) * 33 + e % taps_] = v;
                } else {
                    // m_ = row = input channel ci, inner = col = output channel (maybe permuted): run w[src(col)][m_][taps]
                    const int c = e / taps_, t = e % taps_;
                    const int64_t col = i0 + c;
                    const int64_t src = col < o.cout ? (o.perm ? (int64_t)o.perm[col] : col) : -1;
                    if (src >= 0 && src < o.cout && m_ < o.cin) v = o.w[(src * o.cin + m_) * taps_ + t];
                    tile[c * 33 + t] = v;
                }
            }
            __syncthreads();
            for (int e = threadIdx.x; e < 64 * taps_; e += 256) {
                const int t = e >> 6, c = e & 63;
                if (i0 + c < inner) {
                    const float v = tile[c * 33 + (fwd ? t : taps_ - 1 - t)];
                    const int64_t oi = ((int64_t)t * mid + m_) * inner + i0 + c;
                    if (o.dtype == RHO_BF16) reinterpret_cast<bf16_raw*>(o.out)[oi] = f32_to_bf16(v);
                    else reinterpret_cast<float*>(o.out)[oi] = v;
                }
            }
            __syncthreads();
        }
        return;
    }
    const int64_t stride = (int64_t)o.nblk * 256;
    for (int64_t i = (int64_t)(b - o.blk0) * 256 + threadIdx.x; i < o.total; i += stride) {
        const float v = prep_elem(o, i);
        if (o.dtype == RHO_BF16) reinterpret_cast<bf16_raw*>(o.out)[i] = f32_to_bf16(v);
        else reinterpret_cast<float*>(o.out)[i] = v;
    }
}

extern "C" int rho_prep_batch(const rho_prep_op* ops_dev, int64_t n_ops, int64_t n_blocks, void* stream) {
    if (!ops_dev || n_ops <= 0 || n_blocks <= 0 || n_blocks > 0x7FFFFFFF || n_ops > 0x7FFFFFFF) return RHO_E_ARG;
    hipLaunchKernelGGL(k_prep_batch, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream), ops_dev, (int)n_ops);
    RHO_LAUNCH_CHECK();
    return 0;
}

// nearest x2 on H and/or W of a channels-last tensor, 16-byte pieces (materialised only for the wgrad of Upsample.conv)
__global__ __launch_bounds__(256) void k_upsample2x(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t nd, int h, int w,
                                                    int cpieces, int uh, int uw) {
    const int ho = uh ? 2 * h : h, wo = uw ? 2 * w : w;
    const int64_t total = nd * ho * wo * cpieces;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cp = (int)(i % cpieces);
        int64_t r = i / cpieces;
        const int ow = (int)(r % wo); r /= wo;
        const int oh = (int)(r % ho); r /= ho;
        const int ih = uh ? oh >> 1 : oh, iw = uw ? ow >> 1 : ow;
        y[i] = x[((r * h + ih) * w + iw) * cpieces + cp];
    }
}

extern "C" int rho_upsample2x(const void* x, void* y, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int up_h,
                              int up_w, void* stream) {
    const int esz = dtype == RHO_BF16 ? 2 : 4;
    if (!x || !y || n_times_d <= 0 || h <= 0 || w <= 0 || c <= 0 || (c * esz) % 16) return RHO_E_ARG;
    const int cp = (int)(c * esz / 16);
    const int64_t total = n_times_d * (up_h ? 2 * h : h) * (up_w ? 2 * w : w) * cp;
    hipLaunchKernelGGL(k_upsample2x, dim3(grid_for(total, 256)), dim3(256), 0, as_stream(stream), (const uint4*)x, (uint4*)y, n_times_d,
                       (int)h, (int)w, cp, up_h, up_w);
    RHO_LAUNCH_CHECK();
    return 0;
}

// dx[.., h, w, c] (+)= sum over the 2x2 (or 1x2) children of dy: backward of the nearest upsample
template <typename T>
__global__ __launch_bounds__(256) void k_pool2x_sum(const T* __restrict__ dy, T* __restrict__ dx, int64_t nd, int h, int w, int c,
                                                    int uh, int uw, int accumulate) {
    // one 16-byte channel piece per thread (the scalar form ran at 1.8 TB/s)
    constexpr int PE = 16 / (int)sizeof(T);
    const int ho = uh ? 2 * h : h, wo = uw ? 2 * w : w;
    const int cp = c / PE;
    const int64_t total = nd * h * w * cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pc = (int)(i % cp);
        int64_t r = i / cp;
        const int iw = (int)(r % w); r /= w;
        const int ih = (int)(r % h); r /= h;
        float acc[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) acc[e] = 0.0f;
        for (int a = 0; a <= uh; ++a)
            for (int b = 0; b <= uw; ++b) {
                const int64_t o = ((r * ho + (uh ? 2 * ih + a : ih)) * wo + (uw ? 2 * iw + b : iw)) * c + pc * PE;
                const uint4 u = *reinterpret_cast<const uint4*>(dy + o);
                if constexpr (sizeof(T) == 2) {
                    acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xFFFF0000u);
                    acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xFFFF0000u);
                    acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xFFFF0000u);
                    acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xFFFF0000u);
                } else {
                    acc[0] += __uint_as_float(u.x); acc[1] += __uint_as_float(u.y); acc[2] += __uint_as_float(u.z); acc[3] += __uint_as_float(u.w);
                }
            }
        T* dst = dx + (i / cp) * c + pc * PE;
        if (accumulate) {
            const uint4 u = *reinterpret_cast<const uint4*>(dst);
            if constexpr (sizeof(T) == 2) {
                acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xFFFF0000u);
                acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xFFFF0000u);
                acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xFFFF0000u);
                acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xFFFF0000u);
            } else {
                acc[0] += __uint_as_float(u.x); acc[1] += __uint_as_float(u.y); acc[2] += __uint_as_float(u.z); acc[3] += __uint_as_float(u.w);
            }
        }
        if constexpr (sizeof(T) == 2)
            *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                                                        pack_bf16x2(acc[6], acc[7]));
        else
            *reinterpret_cast<uint4*>(dst) = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                                                        __float_as_uint(acc[3]));
    }
}

extern "C" int rho_pool2x_sum(const void* dy, void* dx, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int up_h,
                              int up_w, int accumulate, void* stream) {
    if (!dy || !dx || n_times_d <= 0 || h <= 0 || w <= 0 || c <= 0) return RHO_E_ARG;
    if (c % 8 != 0) return RHO_E_ALIGN;
    dim3 grid(grid_for(n_times_d * h * w * (c / (dtype == RHO_BF16 ? 8 : 4)), 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_pool2x_sum<bf16_raw>, grid, block, 0, as_stream(stream), (const bf16_raw*)dy, (bf16_raw*)dx, n_times_d, (int)h,
                           (int)w, (int)c, up_h, up_w, accumulate);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_pool2x_sum<float>, grid, block, 0, as_stream(stream), (const float*)dy, (float*)dx, n_times_d, (int)h, (int)w,
                           (int)c, up_h, up_w, accumulate);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- average pooling (K12)
// avg_pool_nd(dims, kernel_size = stride, stride = stride), stride 2 (3-D: (1, 2, 2)), layers.py:91-102 / unet_v2.py:165: the
// Downsample of conv_resample = False and of ResBlock(down = True).  Channels-last, one 16-byte channel piece per thread;
// output extent floor(h / 2) (PyTorch's ceil_mode = False: an odd last row / column is dropped).
template <typename T>
__global__ __launch_bounds__(256) void k_avgpool2x(const T* __restrict__ x, T* __restrict__ y, int64_t nd, int h, int w, int c, int fh,
                                                   int fw) {
    constexpr int PE = 16 / (int)sizeof(T);
    const int ho = fh ? h >> 1 : h, wo = fw ? w >> 1 : w;
    const int cp = c / PE;
    const float scale = 1.0f / (float)((fh ? 2 : 1) * (fw ? 2 : 1));
    const int64_t total = nd * ho * wo * cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pc = (int)(i % cp);
        int64_t r = i / cp;
        const int ow = (int)(r % wo); r /= wo;
        const int oh = (int)(r % ho); r /= ho;
        float acc[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) acc[e] = 0.0f;
        for (int a = 0; a <= fh; ++a)
            for (int b = 0; b <= fw; ++b) {
                const int64_t o = ((r * h + (fh ? 2 * oh + a : oh)) * w + (fw ? 2 * ow + b : ow)) * c + pc * PE;
                const uint4 u = *reinterpret_cast<const uint4*>(x + o);
                if constexpr (sizeof(T) == 2) {
                    acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xFFFF0000u);
                    acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xFFFF0000u);
                    acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xFFFF0000u);
                    acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xFFFF0000u);
                } else {
                    acc[0] += __uint_as_float(u.x); acc[1] += __uint_as_float(u.y); acc[2] += __uint_as_float(u.z); acc[3] += __uint_as_float(u.w);
                }
            }
        T* dst = y + (i / cp) * c + pc * PE;
        if constexpr (sizeof(T) == 2)
            *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(acc[0] * scale, acc[1] * scale), pack_bf16x2(acc[2] * scale, acc[3] * scale),
                                                        pack_bf16x2(acc[4] * scale, acc[5] * scale), pack_bf16x2(acc[6] * scale, acc[7] * scale));
        else
            *reinterpret_cast<uint4*>(dst) = make_uint4(__float_as_uint(acc[0] * scale), __float_as_uint(acc[1] * scale),
                                                        __float_as_uint(acc[2] * scale), __float_as_uint(acc[3] * scale));
    }
}

extern "C" int rho_avgpool2x(const void* x, void* y, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int pool_h, int pool_w,
                             void* stream) {
    if (!x || !y || n_times_d <= 0 || h <= 0 || w <= 0 || c <= 0 || (pool_h && h < 2) || (pool_w && w < 2)) return RHO_E_ARG;
    if (dtype != RHO_BF16 && dtype != RHO_F32) return RHO_E_ARG;
    const int pe = dtype == RHO_BF16 ? 8 : 4;
    if (c % pe != 0) return RHO_E_ALIGN;
    const int64_t total = n_times_d * (pool_h ? h / 2 : h) * (pool_w ? w / 2 : w) * (c / pe);
    dim3 grid(grid_for(total, 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_avgpool2x<bf16_raw>, grid, block, 0, as_stream(stream), (const bf16_raw*)x, (bf16_raw*)y, n_times_d, (int)h, (int)w,
                           (int)c, pool_h, pool_w);
    else
        hipLaunchKernelGGL(k_avgpool2x<float>, grid, block, 0, as_stream(stream), (const float*)x, (float*)y, n_times_d, (int)h, (int)w, (int)c,
                           pool_h, pool_w);
    RHO_LAUNCH_CHECK();
    return 0;
}

// backward: dx[.., ih, iw, :] (+)= dy[.., ih / 2, iw / 2, :] / (window size); rows / columns the pooling dropped receive 0
template <typename T>
__global__ __launch_bounds__(256) void k_avgpool2x_bwd(const T* __restrict__ dy, T* __restrict__ dx, int64_t nd, int h, int w, int c, int fh,
                                                       int fw, int accumulate) {
    constexpr int PE = 16 / (int)sizeof(T);
    const int ho = fh ? h >> 1 : h, wo = fw ? w >> 1 : w;
    const int cp = c / PE;
    const float scale = 1.0f / (float)((fh ? 2 : 1) * (fw ? 2 : 1));
    const int64_t total = nd * h * w * cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int pc = (int)(i % cp);
        int64_t r = i / cp;
        const int iw = (int)(r % w); r /= w;
        const int ih = (int)(r % h); r /= h;
        const int oh = fh ? ih >> 1 : ih, ow = fw ? iw >> 1 : iw;
        float acc[PE];
#pragma unroll
        for (int e = 0; e < PE; ++e) acc[e] = 0.0f;
        if (oh < ho && ow < wo) {
            const uint4 u = *reinterpret_cast<const uint4*>(dy + ((r * ho + oh) * wo + ow) * c + pc * PE);
            if constexpr (sizeof(T) == 2) {
                acc[0] = __uint_as_float(u.x << 16) * scale; acc[1] = __uint_as_float(u.x & 0xFFFF0000u) * scale;
                acc[2] = __uint_as_float(u.y << 16) * scale; acc[3] = __uint_as_float(u.y & 0xFFFF0000u) * scale;
                acc[4] = __uint_as_float(u.z << 16) * scale; acc[5] = __uint_as_float(u.z & 0xFFFF0000u) * scale;
                acc[6] = __uint_as_float(u.w << 16) * scale; acc[7] = __uint_as_float(u.w & 0xFFFF0000u) * scale;
            } else {
                acc[0] = __uint_as_float(u.x) * scale; acc[1] = __uint_as_float(u.y) * scale;
                acc[2] = __uint_as_float(u.z) * scale; acc[3] = __uint_as_float(u.w) * scale;
            }
        }
        T* dst = dx + (i / cp) * c + pc * PE;
        if (accumulate) {
            const uint4 u = *reinterpret_cast<const uint4*>(dst);
            if constexpr (sizeof(T) == 2) {
                acc[0] += __uint_as_float(u.x << 16); acc[1] += __uint_as_float(u.x & 0xFFFF0000u);
                acc[2] += __uint_as_float(u.y << 16); acc[3] += __uint_as_float(u.y & 0xFFFF0000u);
                acc[4] += __uint_as_float(u.z << 16); acc[5] += __uint_as_float(u.z & 0xFFFF0000u);
                acc[6] += __uint_as_float(u.w << 16); acc[7] += __uint_as_float(u.w & 0xFFFF0000u);
            } else {
                acc[0] += __uint_as_float(u.x); acc[1] += __uint_as_float(u.y); acc[2] += __uint_as_float(u.z); acc[3] += __uint_as_float(u.w);
            }
        }
        if constexpr (sizeof(T) == 2)
            *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]),
                                                        pack_bf16x2(acc[6], acc[7]));
        else
            *reinterpret_cast<uint4*>(dst) = make_uint4(__float_as_uint(acc[0]), __float_as_uint(acc[1]), __float_as_uint(acc[2]),
                                                        __float_as_uint(acc[3]));
    }
}

extern "C" int rho_avgpool2x_bwd(const void* dy, void* dx, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int pool_h,
                                 int pool_w, int accumulate, void* stream) {
    if (!dy || !dx || n_times_d <= 0 || h <= 0 || w <= 0 || c <= 0 || (pool_h && h < 2) || (pool_w && w < 2)) return RHO_E_ARG;
    if (dtype != RHO_BF16 && dtype != RHO_F32) return RHO_E_ARG;
    const int pe = dtype == RHO_BF16 ? 8 : 4;
    if (c % pe != 0) return RHO_E_ALIGN;
    dim3 grid(grid_for(n_times_d * h * w * (c / pe), 256)), block(256);
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_avgpool2x_bwd<bf16_raw>, grid, block, 0, as_stream(stream), (const bf16_raw*)dy, (bf16_raw*)dx, n_times_d, (int)h,
                           (int)w, (int)c, pool_h, pool_w, accumulate);
    else
        hipLaunchKernelGGL(k_avgpool2x_bwd<float>, grid, block, 0, as_stream(stream), (const float*)dy, (float*)dx, n_times_d, (int)h, (int)w,
                           (int)c, pool_h, pool_w, accumulate);
    RHO_LAUNCH_CHECK();
    return 0;
}

// Backward of rho_linear (out = W act(x) + b [+ add]):  dW[o,k] (+)= sum_b dout[b,o] act(x[b,k]),  db[o] (+)= sum_b dout[b,o]
__global__ __launch_bounds__(256) void k_linear_bwd_w(const float* __restrict__ dout, const float* __restrict__ x,
                                                      float* __restrict__ dw, float* __restrict__ db, int batch, int in_dim,
                                                      int out_dim, int act_in, int accumulate, int64_t dstride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)out_dim * in_dim) return;
    const int o = (int)(i / in_dim), k = (int)(i % in_dim);
    float acc = 0.0f, accb = 0.0f;
    for (int b = 0; b < batch; ++b) {
        float xv = x[(int64_t)b * in_dim + k];
        if (act_in) xv = act_in == 1 ? xv / (1.0f + expf(-xv)) : act_other_f(xv, act_in);
        const float g = dout[(int64_t)b * dstride + o];
        acc = fmaf(g, xv, acc);
        accb += g;
    }
    dw[i] = accumulate ? dw[i] + acc : acc;
    if (k == 0 && db) db[o] = accumulate ? db[o] + accb : accb;
}

// dx[b,k] += act'(x[b,k]) * sum_o dout[b,o] W[o,k], the o range dealt to gridDim.y slices of 64 that add with fp32 atomics
// (a thread per (b,k) walking all of out_dim was 32 workgroups x 1024 dependent iterations: 0.2-0.4 ms per FiLM linear,
// 4.7 ms per training step for 4 MFLOP of work).  dx must hold the value to accumulate onto (zeroed by the caller side).
__global__ __launch_bounds__(256) void k_linear_bwd_x(const float* __restrict__ dout, const float* __restrict__ w,
                                                      const float* __restrict__ x, float* __restrict__ dx, int batch, int in_dim,
                                                      int out_dim, int act_in, int64_t dstride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)batch * in_dim) return;
    const int b = (int)(i / in_dim), k = (int)(i % in_dim);
    const int o0 = blockIdx.y * 64, o1 = min(o0 + 64, out_dim);
    float acc = 0.0f;
#pragma unroll 8
    for (int o = o0; o < o1; ++o) acc = fmaf(dout[(int64_t)b * dstride + o], w[(int64_t)o * in_dim + k], acc);
    if (act_in == 1) {
        const float u = x[i];
        const float s = 1.0f / (1.0f + expf(-u));
        acc *= s * (1.0f + u * (1.0f - s));
    } else if (act_in) {
        acc *= dact_other_f(x[i], act_in);
    }
    atomicAdd(dx + i, acc);
}

// ordered form (rho_get_deterministic): one thread walks all of out_dim in index order - no atomics, 0.2-0.4 ms per FiLM linear
__global__ __launch_bounds__(256) void k_linear_bwd_x_det(const float* __restrict__ dout, const float* __restrict__ w,
                                                          const float* __restrict__ x, float* __restrict__ dx, int batch, int in_dim,
                                                          int out_dim, int act_in, int64_t dstride, int acc_dx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)batch * in_dim) return;
    const int b = (int)(i / in_dim), k = (int)(i % in_dim);
    float acc = 0.0f;
#pragma unroll 8
    for (int o = 0; o < out_dim; ++o) acc = fmaf(dout[(int64_t)b * dstride + o], w[(int64_t)o * in_dim + k], acc);
    if (act_in == 1) {
        const float u = x[i];
        const float s = 1.0f / (1.0f + expf(-u));
        acc *= s * (1.0f + u * (1.0f - s));
    } else if (act_in) {
        acc *= dact_other_f(x[i], act_in);
    }
    dx[i] = acc_dx ? dx[i] + acc : acc;
}

extern "C" int rho_linear_bwd(const float* dout, int64_t dout_stride, const float* x, const float* w, float* dw, float* db,
                              float* dx, int64_t batch, int64_t in_dim, int64_t out_dim, int act_in, int acc_params, int acc_dx,
                              void* stream) {
    if (!dout || !x || !w || batch <= 0 || in_dim <= 0 || out_dim <= 0) return RHO_E_ARG;
    const int64_t dstride = dout_stride > 0 ? dout_stride : out_dim;
    if (dw) {
        hipLaunchKernelGGL(k_linear_bwd_w, dim3((unsigned)((out_dim * in_dim + 255) / 256)), dim3(256), 0, as_stream(stream), dout, x, dw,
                           db, (int)batch, (int)in_dim, (int)out_dim, act_in, acc_params, dstride);
    }
    if (dx && rho_get_deterministic()) {
        hipLaunchKernelGGL(k_linear_bwd_x_det, dim3((unsigned)((batch * in_dim + 255) / 256)), dim3(256), 0, as_stream(stream), dout, w, x,
                           dx, (int)batch, (int)in_dim, (int)out_dim, act_in, dstride, acc_dx);
    } else if (dx) {
        if (!acc_dx) {
            hipError_t e = hipMemsetAsync(dx, 0, (size_t)batch * in_dim * sizeof(float), as_stream(stream));
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(k_linear_bwd_x, dim3((unsigned)((batch * in_dim + 255) / 256), (unsigned)((out_dim + 63) / 64)), dim3(256), 0,
                           as_stream(stream), dout, w, x, dx, (int)batch, (int)in_dim, (int)out_dim, act_in, dstride);
    }
    RHO_LAUNCH_CHECK();
    return 0;
}

// ExponentialMovingAverage.update (rho_diffusion/ema.py:41-60): shadow -= (1 - frac) * (shadow - param), float32, in the
// reference's operation order (sub, mul, sub; no contraction) so a parameter-by-parameter comparison is bit-exact.
__global__ __launch_bounds__(256) void k_ema_update(float* __restrict__ shadow, const float* __restrict__ param, int64_t n, float omf) {
#pragma clang fp contract(off)
    const int64_t n4 = n >> 2;
    const bool vec = ((((uintptr_t)shadow | (uintptr_t)param) & 15) == 0);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        float4* s4 = reinterpret_cast<float4*>(shadow);
        const float4* p4 = reinterpret_cast<const float4*>(param);
        for (int64_t i = tid; i < n4; i += stride) {
            float4 a = s4[i];
            const float4 b = p4[i];
            float d;
            d = a.x - b.x; d = omf * d; a.x = a.x - d;
            d = a.y - b.y; d = omf * d; a.y = a.y - d;
            d = a.z - b.z; d = omf * d; a.z = a.z - d;
            d = a.w - b.w; d = omf * d; a.w = a.w - d;
            s4[i] = a;
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += stride) {
            float d = shadow[i] - param[i];
            d = omf * d;
            shadow[i] = shadow[i] - d;
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            float d = shadow[i] - param[i];
            d = omf * d;
            shadow[i] = shadow[i] - d;
        }
    }
}

extern "C" int rho_ema_update(float* shadow, const float* param, int64_t n, float one_minus_frac, void* stream) {
    if (!shadow || !param || n <= 0) return RHO_E_ARG;
    hipLaunchKernelGGL(k_ema_update, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), shadow, param, n, one_minus_frac);
    RHO_LAUNCH_CHECK();
    return 0;
}

// dst += src (channels-last activations / gradients), 16-byte pieces
template <typename T>
__global__ __launch_bounds__(256) void k_add_inplace(T* __restrict__ dst, const T* __restrict__ src, int64_t n) {
    constexpr int PE = 16 / (int)sizeof(T);
    const int64_t np = n / PE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (int64_t)gridDim.x * blockDim.x) {
        uint4 a = reinterpret_cast<const uint4*>(dst)[i];
        const uint4 b = reinterpret_cast<const uint4*>(src)[i];
        if constexpr (sizeof(T) == 2) {
            a.x = pack_bf16x2(__uint_as_float(a.x << 16) + __uint_as_float(b.x << 16), __uint_as_float(a.x & 0xFFFF0000u) + __uint_as_float(b.x & 0xFFFF0000u));
            a.y = pack_bf16x2(__uint_as_float(a.y << 16) + __uint_as_float(b.y << 16), __uint_as_float(a.y & 0xFFFF0000u) + __uint_as_float(b.y & 0xFFFF0000u));
            a.z = pack_bf16x2(__uint_as_float(a.z << 16) + __uint_as_float(b.z << 16), __uint_as_float(a.z & 0xFFFF0000u) + __uint_as_float(b.z & 0xFFFF0000u));
            a.w = pack_bf16x2(__uint_as_float(a.w << 16) + __uint_as_float(b.w << 16), __uint_as_float(a.w & 0xFFFF0000u) + __uint_as_float(b.w & 0xFFFF0000u));
        } else {
            a.x = __float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x));
            a.y = __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y));
            a.z = __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z));
            a.w = __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w));
        }
        reinterpret_cast<uint4*>(dst)[i] = a;
    }
}

extern "C" int rho_add_inplace(void* dst, const void* src, int dtype, int64_t n, void* stream) {
    const int pe = dtype == RHO_BF16 ? 8 : 4;
    if (!dst || !src || n <= 0 || n % pe) return RHO_E_ARG;
    if (dtype == RHO_BF16)
        hipLaunchKernelGGL(k_add_inplace<bf16_raw>, dim3(grid_for(n / pe, 256)), dim3(256), 0, as_stream(stream), (bf16_raw*)dst, (const bf16_raw*)src, n);
    else if (dtype == RHO_F32)
        hipLaunchKernelGGL(k_add_inplace<float>, dim3(grid_for(n / pe, 256)), dim3(256), 0, as_stream(stream), (float*)dst, (const float*)src, n);
    else
        return RHO_E_ARG;
    RHO_LAUNCH_CHECK();
    return 0;
}

// x *= *scale_dev  (upstream scalar of loss.backward(), read on the device: no host sync)
__global__ __launch_bounds__(256) void k_scale_dev(float* __restrict__ x, const float* __restrict__ scale_dev, int64_t n) {
    const float sc = *scale_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) x[i] *= sc;
}
extern "C" int rho_scale_by_device_scalar(float* x, const float* scale_dev, int64_t n, void* stream) {
    if (!x || !scale_dev || n <= 0) return RHO_E_ARG;
    hipLaunchKernelGGL(k_scale_dev, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), x, scale_dev, n);
    RHO_LAUNCH_CHECK();
    return 0;
}
