// Per-sample quantile of |x| (dynamic thresholding of GaussianDiffusionPipeline.p_mean_variance,
// rho_diffusion/diffusion/gaussian_diffusion.py:400-415: torch.quantile(x.abs().flatten(1), 0.9, dim=-1)) and the
// fused DDIM update that consumes it (:654-702).
//
// torch.quantile sorts each row; here the two order statistics that the linear interpolation needs are found
// EXACTLY by a most-significant-digit radix select over the IEEE bit patterns (non-negative floats order like their
// bits): 4 passes of 8 bits, each a histogram of the digit among the elements that still match the selected prefix,
// then a 256-bin scan picks the digit.  Both ranks (floor / ceil of q*(N-1)) ride through the same passes.  The model
// output was just written by the UNet's last kernel, so the passes read L2 / Infinity Cache, not HBM; HBM-bound at
// worst (4 reads of B*N floats).  Integer work: results are bit-exact by construction.
#include "common.h"

// The elementwise kernels below restate float32 tensor expressions of the reference op by op (each rounding once);
// hipcc's default -ffp-contract=fast would fuse a*b+c into one fma (1 ulp off on ~1 % of the elements).
#pragma clang fp contract(off)

namespace {
constexpr int SEL_BINS = 256;
// workspace (uint32): hist[4][B][2][256], then state[B][4] = {prefix_lo, prefix_hi, rank_lo, rank_hi}
__host__ __device__ inline size_t sel_hist_words(int64_t B) { return (size_t)4 * B * 2 * SEL_BINS; }
}  // namespace

__global__ __launch_bounds__(256) void k_sel_init(uint32_t* __restrict__ ws, int64_t B, uint32_t rank_lo, uint32_t rank_hi) {
    const size_t nh = sel_hist_words(B);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nh + (size_t)B * 4; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t v = 0;
        if (i >= nh) {
            const int f = (int)((i - nh) & 3);
            v = f == 2 ? rank_lo : (f == 3 ? rank_hi : 0u);
        }
        ws[i] = v;
    }
}

template <int PASS>
__global__ __launch_bounds__(256) void k_sel_hist(const float* __restrict__ x, int64_t N, int64_t B, uint32_t* __restrict__ ws) {
    constexpr int SHIFT = 24 - 8 * PASS;
    __shared__ uint32_t h[2][SEL_BINS];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    h[0][tid] = 0u;
    h[1][tid] = 0u;
    const uint32_t* st = ws + sel_hist_words(B) + (size_t)b * 4;
    const uint32_t p0 = st[0], p1 = st[1];
    const bool same = (PASS == 0) || ((p0 >> (SHIFT + 8)) == (p1 >> (SHIFT + 8)));   // both ranks still in one bucket
    __syncthreads();
    const float* xb = x + (size_t)b * N;
    auto tally = [&](float f) {
        const uint32_t key = __float_as_uint(f) & 0x7FFFFFFFu;
        const uint32_t bin = (key >> SHIFT) & 0xFFu;
        if (PASS == 0) {
            atomicAdd(&h[0][bin], 1u);
        } else {
            const uint32_t hi = key >> (SHIFT + 8);
            if (hi == (p0 >> (SHIFT + 8))) atomicAdd(&h[0][bin], 1u);
            if (!same && hi == (p1 >> (SHIFT + 8))) atomicAdd(&h[1][bin], 1u);
        }
    };
    const int64_t per = (N + gridDim.x - 1) / gridDim.x;
    int64_t i0 = (int64_t)blockIdx.x * per, i1 = i0 + per < N ? i0 + per : N;
    if ((N & 3) == 0 && (per & 3) == 0) {                   // 16-byte pieces (rows are 16-byte aligned when N % 4 == 0)
        for (int64_t i = i0 + 4 * tid; i < i1; i += 4 * 256) {
            const float4 v = *reinterpret_cast<const float4*>(xb + i);
            tally(v.x); tally(v.y); tally(v.z); tally(v.w);
        }
    } else {
        for (int64_t i = i0 + tid; i < i1; i += 256) tally(xb[i]);
    }
    __syncthreads();
    uint32_t* g = ws + (((size_t)PASS * B + b) * 2) * SEL_BINS;
    const uint32_t c0 = h[0][tid], c1 = same ? c0 : h[1][tid];
    if (c0) atomicAdd(g + tid, c0);
    if (c1) atomicAdd(g + SEL_BINS + tid, c1);
}

// one workgroup per sample: pick the digit of both ranks; after the last pass the prefixes ARE the order statistics
template <int PASS>
__global__ __launch_bounds__(256) void k_sel_pick(uint32_t* __restrict__ ws, int64_t B, float weight, float* __restrict__ out) {
    constexpr int SHIFT = 24 - 8 * PASS;
    __shared__ uint32_t cum[2][SEL_BINS];
    const int b = blockIdx.x, tid = threadIdx.x;
    uint32_t* st = ws + sel_hist_words(B) + (size_t)b * 4;
    const uint32_t* g = ws + (((size_t)PASS * B + b) * 2) * SEL_BINS;
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) cum[sel][tid] = g[sel * SEL_BINS + tid];
    __syncthreads();
    // inclusive scan (Hillis-Steele, 256 entries x 2)
    for (int off = 1; off < SEL_BINS; off <<= 1) {
        uint32_t a0 = 0, a1 = 0;
        if (tid >= off) { a0 = cum[0][tid - off]; a1 = cum[1][tid - off]; }
        __syncthreads();
        cum[0][tid] += a0;
        cum[1][tid] += a1;
        __syncthreads();
    }
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) {
        const uint32_t rank = st[2 + sel];
        const uint32_t incl = cum[sel][tid], excl = tid ? cum[sel][tid - 1] : 0u;
        if (excl <= rank && rank < incl) {            // exactly one bin (the counts of the prefix bucket sum to > rank)
            st[sel] |= (uint32_t)tid << SHIFT;
            st[2 + sel] = rank - excl;
        }
    }
    if (PASS == 3) {
        __syncthreads();
        if (tid == 0) {
            const float lo = __uint_as_float(st[0]), hi = __uint_as_float(st[1]);
            // at::lerp: start + w*(end-start) for w < 0.5, else end - (end-start)*(1-w); as one fma like ATen's vector path
            const float diff = hi - lo;
            out[b] = (fabsf(weight) < 0.5f) ? fmaf(weight, diff, lo) : fmaf(weight - 1.0f, diff, hi);
        }
    }
}

extern "C" int64_t rho_abs_quantile_workspace_bytes(int64_t batch) {
    return batch > 0 ? (int64_t)((sel_hist_words(batch) + (size_t)batch * 4) * sizeof(uint32_t)) : 0;
}

extern "C" int rho_abs_quantile(const float* x, int64_t batch, int64_t n, double q, void* workspace, float* out, void* stream) {
    if (!x || !workspace || !out || batch <= 0 || n <= 0 || !(q >= 0.0 && q <= 1.0)) return RHO_E_ARG;
    if (batch > 65535 || n >= (1LL << 32)) return RHO_E_SHAPE;
    // torch.quantile: ranks = q * (n - 1) evaluated in the tensor's dtype (float32), floor / ceil, weight = frac
    const float rank_f = (float)q * (float)(n - 1);
    const float below = floorf(rank_f);
    const uint32_t rank_lo = (uint32_t)below;
    uint32_t rank_hi = (uint32_t)ceilf(rank_f);
    if (rank_hi > (uint32_t)(n - 1)) rank_hi = (uint32_t)(n - 1);
    const float weight = rank_f - below;
    uint32_t* ws = (uint32_t*)workspace;
    hipStream_t st = as_stream(stream);
    const size_t words = sel_hist_words(batch) + (size_t)batch * 4;
    int gi = (int)((words + 255) / 256);
    if (gi > 1024) gi = 1024;
    hipLaunchKernelGGL(k_sel_init, dim3((unsigned)gi), dim3(256), 0, st, ws, batch, rank_lo, rank_hi);
    int nblk = (int)((n + 4095) / 4096);
    if (nblk > 128) nblk = 128;
    if (nblk < 1) nblk = 1;
    dim3 gh((unsigned)nblk, (unsigned)batch), gp((unsigned)batch);
#define RHO_SEL_PASS(P)                                                                      \
    hipLaunchKernelGGL(k_sel_hist<P>, gh, dim3(256), 0, st, x, n, batch, ws);                \
    hipLaunchKernelGGL(k_sel_pick<P>, gp, dim3(256), 0, st, ws, batch, weight, out);
    RHO_SEL_PASS(0)
    RHO_SEL_PASS(1)
    RHO_SEL_PASS(2)
    RHO_SEL_PASS(3)
#undef RHO_SEL_PASS
    RHO_LAUNCH_CHECK();
    return 0;
}

// DDIM update (gaussian_diffusion.py:654-702 with p_mean_variance :400-415 and _predict_eps_from_xstart :462-466),
// x0-prediction model.  Per element, in the reference's own operation order and without FMA contraction:
//   s = max(quantile[b], 1);  x0 = clamp(m, -s, s) / s;  eps = (c_recip * x - x0) / c_recipm1
//   x_prev = x0 * sqrt_abar_prev + coef_eps * eps  (+ sigma * noise when sigma != 0 and t != 0)
__global__ __launch_bounds__(256) void k_ddim_step(const float* __restrict__ x, const float* __restrict__ m, const float* __restrict__ quant,
                                                   const float* __restrict__ noise, float* __restrict__ x_prev, float* __restrict__ pred_x0,
                                                   int64_t per_sample, int64_t total, float c_recip, float c_recipm1,
                                                   float sqrt_abar_prev, float coef_eps, float sigma_masked) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float s = fmaxf(quant[i / per_sample], 1.0f);
        // plain operators: under the file's contract(off) they stay separate, correctly rounded operations (the
        // __fmul_rn-style header inlines carry their own contraction flag and were still fused)
        const float x0 = fminf(fmaxf(m[i], -s), s) / s;
        const float ax = c_recip * x[i];
        const float eps = (ax - x0) / c_recipm1;
        const float p0 = x0 * sqrt_abar_prev, p1 = coef_eps * eps;
        float v = p0 + p1;
        if (noise != nullptr) {
            const float p2 = sigma_masked * noise[i];
            v = v + p2;
        }
        x_prev[i] = v;
        if (pred_x0 != nullptr) pred_x0[i] = x0;
    }
}

extern "C" int rho_ddim_step(const float* x_t, const float* model_out, const float* quantile, const float* noise, float* x_prev,
                             float* pred_xstart, int64_t batch, int64_t per_sample, float c_recip, float c_recipm1,
                             float sqrt_abar_prev, float coef_eps, float sigma_masked, void* stream) {
    if (!x_t || !model_out || !quantile || !x_prev || batch <= 0 || per_sample <= 0) return RHO_E_ARG;
    if (sigma_masked != 0.0f && !noise) return RHO_E_ARG;
    const int64_t total = batch * per_sample;
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_ddim_step, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x_t, model_out, quantile,
                       (sigma_masked != 0.0f) ? noise : nullptr, x_prev, pred_xstart, per_sample, total, c_recip, c_recipm1,
                       sqrt_abar_prev, coef_eps, sigma_masked);
    RHO_LAUNCH_CHECK();
    return 0;
}

// q_sample with explicit coefficient tables: x_t = a[t_b] * x0 + b[t_b] * eps  (gaussian_diffusion.py:294-312: the tables
// are float64 sqrt(abar) / sqrt(1-abar) cast to float32 by _extract_into_tensor, so the caller passes those casts;
// mul, mul, add without contraction like the reference's tensor expression)
__global__ __launch_bounds__(256) void k_q_sample_coef(const float* __restrict__ x0, const float* __restrict__ eps, float* __restrict__ xt,
                                                       const float* __restrict__ ca, const float* __restrict__ cb,
                                                       const int64_t* __restrict__ t, int64_t per_sample, int64_t total,
                                                       int64_t table_len, int32_t* err_flag) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t tb = t[i / per_sample];
        if (tb < 0 || tb >= table_len) {               // the reference's table gather raises IndexError
            if (err_flag != nullptr) atomicOr(err_flag, 4);
            tb = tb < 0 ? 0 : table_len - 1;
        }
        const float p0 = ca[tb] * x0[i], p1 = cb[tb] * eps[i];
        xt[i] = p0 + p1;
    }
}

extern "C" int rho_q_sample_coef(const float* x0, const float* eps, float* x_t, const float* coef_a, const float* coef_b,
                                 const int64_t* t, int64_t batch, int64_t per_sample, int64_t table_len, int32_t* err_flag,
                                 void* stream) {
    if (!x0 || !eps || !x_t || !coef_a || !coef_b || !t || batch <= 0 || per_sample <= 0 || table_len <= 0) return RHO_E_ARG;
    const int64_t total = batch * per_sample;
    int64_t g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_q_sample_coef, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x0, eps, x_t, coef_a, coef_b, t, per_sample,
                       total, table_len, err_flag);
    RHO_LAUNCH_CHECK();
    return 0;
}

// One reverse step of the diffusers-style DDPM scheduler (call site rho_diffusion/diffusion/diffusers.py:214,
// published DDPMScheduler.step), float32, operation order of the published tensor expressions, no contraction:
//   x0 = eps_mode ? (x - sqrt_beta_prod * m) / sqrt_alpha_prod : m;   x0 = clamp(x0, -clip, clip) if clip > 0
//   x_prev = c0 * x0 + c1 * x  (+ sigma * noise when sigma != 0)
// A terminal-SNR-zero schedule has sqrt_alpha_prod = 0 at t = T-1: the division yields +-inf, which the clamp maps to
// +-clip exactly as the tensor expression does.
__global__ __launch_bounds__(256) void k_ddpm_sched_step(const float* __restrict__ x, const float* __restrict__ m,
                                                         const float* __restrict__ noise, float* __restrict__ x_prev,
                                                         float* __restrict__ pred_x0, int64_t total, int eps_mode, float sqrt_beta_prod,
                                                         float sqrt_alpha_prod, float clip, float c0, float c1, float sigma) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float x0 = m[i];
        const float xi = x[i];
        if (eps_mode) {
            const float bm = sqrt_beta_prod * x0;
            const float d = xi - bm;
            x0 = d / sqrt_alpha_prod;
        }
        if (clip > 0.0f) x0 = fminf(fmaxf(x0, -clip), clip);
        const float p0 = c0 * x0, p1 = c1 * xi;
        float v = p0 + p1;
        if (noise != nullptr) {
            const float p2 = sigma * noise[i];
            v = v + p2;
        }
        x_prev[i] = v;
        if (pred_x0 != nullptr) pred_x0[i] = x0;
    }
}

extern "C" int rho_ddpm_sched_step(const float* x_t, const float* model_out, const float* noise, float* x_prev, float* pred_xstart,
                                   int64_t n, int eps_mode, float sqrt_beta_prod, float sqrt_alpha_prod, float clip, float c0,
                                   float c1, float sigma, void* stream) {
    if (!x_t || !model_out || !x_prev || n <= 0) return RHO_E_ARG;
    if (sigma != 0.0f && !noise) return RHO_E_ARG;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_ddpm_sched_step, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x_t, model_out,
                       (sigma != 0.0f) ? noise : nullptr, x_prev, pred_xstart, n, eps_mode, sqrt_beta_prod, sqrt_alpha_prod, clip, c0, c1,
                       sigma);
    RHO_LAUNCH_CHECK();
    return 0;
}
