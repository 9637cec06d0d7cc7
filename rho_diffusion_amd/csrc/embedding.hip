// Embedding path of UNet.forward (rho_diffusion/models/unet_v2.py:699-719) and the timestep draw of training_step:
//   k_timestep_embed   sinusoid of ANY integer t (models/common.py:27-43) + the two time_embed linears (unet_v2.py:521-525)
//                      + the label-embedding add (:702-719), one workgroup per sample
//   k_multi_embed      MultiEmbeddings.forward (models/conditioning.py:115-139): category by exact float equality, rows summed
//   k_multi_embed_bwd  its backward (scatter-add of the embedding gradient into the tables)
//   k_randint          random_timesteps (diffusion/abstract_diffusion.py:163-169) on the device, Philox4x32-10
// All latency-bound: tiny operands, no host synchronisation, graph-capturable.
#include "common.h"

// ----------------------------------------------------------------------------- timestep embedding
// pe[b, 2i] = sin(t_b / omega_i), pe[b, 2i+1] = cos(t_b / omega_i) with omega_i = wavelength^(2i/dim) given as a
// float32 table (dim/2 entries, built once exactly as the reference builds it): the float32 division and the int -> float
// conversion are the reference's; only sinf / cosf differ from the host libm by rounding (<= 2 ulp).
// Then h = W0 pe + b0 ; emb = W2 silu(h) + b2 (+ cond).  Dot products: lane l holds k = l, l + 64, ... (fma chain in
// ascending k), combined by the xor butterfly - the same order as k_linear, so both paths give identical bits.
__global__ __launch_bounds__(256) void k_timestep_embed(const float* __restrict__ omega, const int64_t* __restrict__ t,
                                                        const int32_t* __restrict__ t_scalar, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, const float* __restrict__ w2,
                                                        const float* __restrict__ b2, const float* __restrict__ cond,
                                                        float* __restrict__ pe_out, float* __restrict__ h_out,
                                                        float* __restrict__ emb_out, int dim, int edim, int act) {
    extern __shared__ float sm[];
    float* const pe = sm;               // [dim]
    float* const hs = sm + dim;         // [edim]  silu(h)
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float tf = (float)(t_scalar != nullptr ? (int64_t)(*t_scalar) : t[b]);
    for (int i = tid; i < dim / 2; i += 256) {
        const float arg = tf / omega[i];
        const float s = sinf(arg), c = cosf(arg);
        pe[2 * i] = s;
        pe[2 * i + 1] = c;
        if (pe_out != nullptr) {
            pe_out[(int64_t)b * dim + 2 * i] = s;
            pe_out[(int64_t)b * dim + 2 * i + 1] = c;
        }
    }
    if (w0 == nullptr) return;          // sinusoid only (SinusoidalPositionEmbedding layer)
    __syncthreads();
    for (int o = wave; o < edim; o += 4) {
        float acc = 0.0f;
        for (int k = lane; k < dim; k += 64) acc = fmaf(pe[k], w0[(int64_t)o * dim + k], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            const float h = acc + b0[o];
            if (h_out != nullptr) h_out[(int64_t)b * edim + o] = h;
            hs[o] = act == 1 ? h / (1.0f + expf(-h)) : act_other_f(h, act);
        }
    }
    __syncthreads();
    for (int o = wave; o < edim; o += 4) {
        float acc = 0.0f;
        for (int k = lane; k < edim; k += 64) acc = fmaf(hs[k], w2[(int64_t)o * edim + k], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
            float r = acc + b2[o];
            if (cond != nullptr) r += cond[(int64_t)b * edim + o];
            emb_out[(int64_t)b * edim + o] = r;
        }
    }
}

extern "C" int rho_timestep_embed(const float* omega, const int64_t* t, const int32_t* t_scalar_dev, const float* w0,
                                  const float* b0, const float* w2, const float* b2, const float* cond, float* pe_out,
                                  float* h_out, float* emb_out, int64_t batch, int64_t dim, int64_t edim, int act, void* stream) {
    if (!omega || (!t && !t_scalar_dev) || batch <= 0 || dim <= 0 || (dim & 1)) return RHO_E_ARG;
    if (w0 != nullptr && (!b0 || !w2 || !b2 || !emb_out || edim <= 0)) return RHO_E_ARG;
    if (w0 == nullptr && !pe_out) return RHO_E_ARG;
    if (dim > 8192 || edim > 16384) return RHO_E_SHAPE;
    const size_t lds = (size_t)(dim + (w0 ? edim : 0)) * sizeof(float);
    hipLaunchKernelGGL(k_timestep_embed, dim3((unsigned)batch), dim3(256), lds, as_stream(stream), omega, t, t_scalar_dev, w0, b0, w2,
                       b2, cond, pe_out, h_out, emb_out, (int)dim, (int)edim, act);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- MultiEmbeddings
// y [B, nkeys] float32 labels (nkeys == 1: y [B]); key i has n_i admissible values space[off_i .. off_i + n_i) and an
// embedding table tables[i] = float32 [n_i, dim].  out[b, :] = sum_i tables[i][j_i(b), :] where space_i[j_i] == y[b, i]
// (exact float equality, first match; conditioning.py:132).  A label that matches nothing makes the reference fail with a
// shape error; here it sets *err_flag |= 2 and contributes nothing.
struct MultiEmbedK {
    const float* y;
    const float* space;
    const int32_t* key_off;      // [nkeys + 1]
    const float* const* tables;  // [nkeys] device pointers
    int32_t* idx_out;            // optional [B, nkeys] resolved categories (kept for the backward)
    int nkeys, dim, ystride;
};

__global__ __launch_bounds__(256) void k_multi_embed(const MultiEmbedK p, float* __restrict__ out, int32_t* err_flag) {
    __shared__ int cat[16];
    const int b = blockIdx.x;
    if (threadIdx.x < p.nkeys) {
        const int i = threadIdx.x;
        const float v = p.y[(int64_t)b * p.ystride + (p.ystride == 1 ? 0 : i)];
        int j = -1;
        for (int q = p.key_off[i]; q < p.key_off[i + 1]; ++q)
            if (p.space[q] == v) { j = q - p.key_off[i]; break; }
        cat[i] = j;
        if (p.idx_out != nullptr) p.idx_out[(int64_t)b * p.nkeys + i] = j;
        if (j < 0 && err_flag != nullptr) atomicOr(err_flag, 2);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < p.dim; d += 256) {
        float acc = 0.0f;
        bool first = true;
        for (int i = 0; i < p.nkeys; ++i) {
            if (cat[i] < 0) continue;
            const float e = p.tables[i][(int64_t)cat[i] * p.dim + d];
            acc = first ? e : acc + e;          // emb = e_0 ; emb += e_i  (conditioning.py:134-137)
            first = false;
        }
        out[(int64_t)b * p.dim + d] = acc;
    }
}

extern "C" int rho_multi_embed(const float* y, int64_t y_stride, const float* space, const int32_t* key_off,
                               const float* const* tables, int64_t nkeys, int64_t batch, int64_t dim, float* out,
                               int32_t* idx_out, int32_t* err_flag, void* stream) {
    if (!y || !space || !key_off || !tables || !out || nkeys <= 0 || nkeys > 16 || batch <= 0 || dim <= 0 || y_stride <= 0)
        return RHO_E_ARG;
    MultiEmbedK p{y, space, key_off, tables, idx_out, (int)nkeys, (int)dim, (int)y_stride};
    hipLaunchKernelGGL(k_multi_embed, dim3((unsigned)batch), dim3(256), 0, as_stream(stream), p, out, err_flag);
    RHO_LAUNCH_CHECK();
    return 0;
}

// dtables[i][idx[b, i], :] += demb[b, :]   (nn.Embedding backward; rows repeat across the batch => float atomics;
// one block per (b, key), the tables are tiny)
__global__ __launch_bounds__(256) void k_multi_embed_bwd(const float* __restrict__ demb, const int32_t* __restrict__ idx,
                                                         float* const* dtables, int nkeys, int dim) {
    const int b = blockIdx.x, i = blockIdx.y;
    const int j = idx[(int64_t)b * nkeys + i];
    if (j < 0) return;
    float* const row = dtables[i] + (int64_t)j * dim;
    for (int d = threadIdx.x; d < dim; d += 256) atomicAdd(row + d, demb[(int64_t)b * dim + d]);
}

// ordered form (rho_get_deterministic): one block per key walks the batch in index order
__global__ __launch_bounds__(256) void k_multi_embed_bwd_det(const float* __restrict__ demb, const int32_t* __restrict__ idx,
                                                             float* const* dtables, int nkeys, int dim, int batch) {
    const int i = blockIdx.x;
    for (int b = 0; b < batch; ++b) {
        const int j = idx[(int64_t)b * nkeys + i];
        if (j < 0) continue;
        float* const row = dtables[i] + (int64_t)j * dim;
        for (int d = threadIdx.x; d < dim; d += 256) row[d] += demb[(int64_t)b * dim + d];     // (a thread owns its d: ordered in b)
    }
}

extern "C" int rho_get_deterministic(void);

extern "C" int rho_multi_embed_bwd(const float* demb, const int32_t* idx, float* const* dtables, int64_t nkeys, int64_t batch,
                                   int64_t dim, void* stream) {
    if (!demb || !idx || !dtables || nkeys <= 0 || nkeys > 16 || batch <= 0 || dim <= 0) return RHO_E_ARG;
    if (rho_get_deterministic())
        hipLaunchKernelGGL(k_multi_embed_bwd_det, dim3((unsigned)nkeys), dim3(256), 0, as_stream(stream), demb, idx, dtables, (int)nkeys,
                           (int)dim, (int)batch);
    else
        hipLaunchKernelGGL(k_multi_embed_bwd, dim3((unsigned)batch, (unsigned)nkeys), dim3(256), 0, as_stream(stream), demb, idx, dtables,
                           (int)nkeys, (int)dim);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ----------------------------------------------------------------------------- random timesteps
// Philox4x32-10 (same generator as rho_philox_normal, elementwise.hip): counter = offset + (i >> 2), word i & 3;
// t = floor(u32 * high / 2^32): uniform on [0, high) up to a bias of high / 2^32 (< 2.4e-7 for high <= 1000).
__device__ __forceinline__ void philox_round_e(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void k_randint(int64_t* __restrict__ out, int64_t n, int64_t high, uint64_t seed, uint64_t offset,
                          const uint64_t* __restrict__ offset_dev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ctr = (offset_dev ? *offset_dev : offset) + (uint64_t)(i >> 2);
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round_e(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[i] = (int64_t)(((uint64_t)c[i & 3] * (uint64_t)high) >> 32);
}

extern "C" int rho_randint(int64_t* out, int64_t n, int64_t high, uint64_t seed, uint64_t offset, const uint64_t* offset_dev,
                           void* stream) {
    if (!out || n <= 0 || high <= 0 || high > 0x7FFFFFFFLL) return RHO_E_ARG;
    hipLaunchKernelGGL(k_randint, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), out, n, high, seed, offset,
                       offset_dev);
    RHO_LAUNCH_CHECK();
    return 0;
}
