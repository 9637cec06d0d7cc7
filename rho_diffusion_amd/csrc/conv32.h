// 3x3x3 stride-1 bf16 convolution of a 32-channel tensor to 32 channels (the model_channels = 32 level of the 128^3 configuration:
// ResBlock in- / out-convs at full resolution, unet_v2.py:215,241; their data gradients), included by conv.hip.
//
// k_conv gives such a layer ONE input-channel chunk per 256-position tile: 27 taps x 4 MFMAs = 1.4 us of matrix work per wave
// against ~14 us of per-tile set-up, first-halo latency, weight-ring fill and epilogue (phase stamps) - 450 TF/s with two
// workgroups per CU.  Here the whole weight tensor (27 x 32 x 32 bf16 = 55 KB) lives in the REGISTERS of every wave (216 VGPRs as
// MFMA A operands; one wave per SIMD has 512), a workgroup is persistent over a range of tiles of one sample, the halo tile of
// tile t + 1 (loads issued at the head of tile t, prologue + LDS write dealt behind its taps) lands in the second LDS buffer while
// tile t runs, and nothing but one barrier separates two tiles: no weight ring, no chunk loop, no per-tile decode beyond an
// incremental walk, GroupNorm statistics accumulated in registers over the workgroup's tiles and written once.
#pragma once

#include <type_traits>
#include <utility>

template <int... Is, class F>
__device__ __forceinline__ void static_for_c32_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for_c32(F&& f) {
    static_for_c32_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

#define C32_NP 600                       // halo rows of the 4 x 8 x 8 tile (6 x 10 x 10)
#define C32_ROWS 640                     // rows of an LDS halo buffer (the 40 past the tile take the writes of the unused staging slots)
#define C32_HBUF (C32_ROWS * PITCH)      // bytes per halo buffer
#define C32_WLDS (27 * 32 * 64)          // the whole weight tensor in LDS: [tap][cout][cin] bf16, 64-byte rows, 16-byte slots swizzled

struct C32K {
    const char* x;          // [N][D][H][W][32] bf16
    const char* w;          // prepared weights [27][32][32] bf16 (rows = output channels)
    const float* bias;      // [32]
    const float* pre_a;     // [N][32] folded GroupNorm affine of the prologue, or NULL
    const float* pre_b;
    const char* res;        // residual, the output's shape, or NULL
    const float* res_add;   // per-sample additive row (the FiLM-less embedding), or NULL
    char* y;                // [N][D][H][W][32] bf16
    float* stats;           // [N][wps][2][32] channel sums / sums of squares of the stored output, or NULL
    int D, H, W;
    int tiles_h, tiles_w, tps;      // tiles per sample = (D / 4) * tiles_h * tiles_w
    int wps;                        // workgroups per sample (gridDim.x)
    int pre_silu, res_add_stride;
};

// PRE: the folded GroupNorm affine (+ SiLU) is applied while the halo is staged
template <bool PRE>
__global__ __launch_bounds__(256) void k_conv32(const C32K p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wl = smem + 2 * C32_HBUF;
    float* const coef = reinterpret_cast<float*>(smem + 2 * C32_HBUF + C32_WLDS);  // [2][32] prologue coefficients of this sample
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, piece = tid & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.y, wg = blockIdx.x;
    const int t0 = (int)((long long)p.tps * wg / p.wps), t1 = (int)((long long)p.tps * (wg + 1) / p.wps);   // (host: wps <= tps)

    // ---- weights -> LDS once per workgroup.  Row r = tap * 32 + cout (64 bytes = 32 input channels), its 16-byte slot q stored at
    //      q ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 lane group touches (4 per residue mod 4) then hit 16 different slots
    for (int it = tid; it < 27 * 32 * 4; it += 256) {
        const int r = it >> 2, q = it & 3;
        *reinterpret_cast<uint4*>(wl + r * 64 + ((q ^ ((r >> 2) & 3)) << 4)) = *reinterpret_cast<const uint4*>(p.w + (size_t)it * 16);
    }
    if (PRE && tid < 64) coef[tid] = tid < 32 ? p.pre_a[(size_t)n * 32 + tid] : p.pre_b[(size_t)n * 32 + tid - 32];
    // A operand of (tap, k-step s): row lane & 31 of the tap, slot 2 s + half
    int aoff[2];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const int m = lane & 31;
        aoff[s_] = m * 64 + (((2 * s_ + half) ^ ((m >> 2) & 3)) << 4);
    }

    // ---- this wave's 64 output positions: two 32-column tiles, the conflict-free column -> position map of k_conv (pair_lg = 2)
    int boff[2], opos[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pp = tile_position(wave * 2 + j, lane & 31, 8, 2);
        const int pw = pp & 7, ph = (pp >> 3) & 7, pd = pp >> 6;
        boff[j] = ((pd * 10 + ph) * 10 + pw) * PITCH + 16 * half;
        opos[j] = (pd * p.H + ph) * p.W + pw;                 // offset from the tile's first output position
    }
    // bias (+ the sample's additive row) of this lane's 16 output channels: 8 g + 4 half + e
    float badd[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 8 * g + 4 * half + e;
            badd[4 * g + e] = p.bias[c] + (p.res_add != nullptr ? p.res_add[(size_t)n * p.res_add_stride + c] : 0.0f);
        }
    float ssum[16], ssq[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) ssum[k] = ssq[k] = 0.0f;

    // ---- halo slots of this thread: row (tid >> 2) + 64 i, 16-byte piece tid & 3; byte offset from the tile's first halo position
    //      and validity masks for the first / last tile of an axis (all tiles are full: padding only there)
    unsigned xoff[10];
    unsigned mlo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mhi[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mvalid = 0u;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const int hp = (tid >> 2) + 64 * i;
        const int id = hp / 100, r = hp - id * 100, ih = r / 10, iw = r - ih * 10;
        const unsigned bit = 1u << i;
        if (hp < C32_NP) mvalid |= bit;
        if (id == 0) mlo[0] &= ~bit;
        if (ih == 0) mlo[1] &= ~bit;
        if (iw == 0) mlo[2] &= ~bit;
        if (id == 5) mhi[0] &= ~bit;
        if (ih == 9) mhi[1] &= ~bit;
        if (iw == 9) mhi[2] &= ~bit;
        xoff[i] = (unsigned)((id * p.H + ih) * p.W + iw) * 64u + (unsigned)piece * 16u;
    }
    const int tiles_hw = p.tiles_h * p.tiles_w;
    const char* const xn = p.x + (size_t)n * p.D * p.H * p.W * 64;
    uint4 hv[2][10];                 // two tiles of halo loads in flight (a tile is ~1.5 us of taps, an HBM load up to 2)
    unsigned hmask[2] = {0u, 0u};
    auto tile_origin = [&](int t, int& od0, int& oh0, int& ow0) {
        const int td = t / tiles_hw, r = t - td * tiles_hw, th = r / p.tiles_w, tw = r - th * p.tiles_w;
        od0 = td * 4; oh0 = th * 8; ow0 = tw * 8;
    };
    // global loads of a tile's halo into hv[SET], branch-free, in pieces that ride behind single MFMAs: the tile's mask and base ...
    const char* lxt = xn;
    auto halo_base = [&](int t, auto SET) {
        constexpr int set = decltype(SET)::value;
        int od0, oh0, ow0;
        tile_origin(t, od0, oh0, ow0);
        unsigned m = mvalid;
        m &= (od0 == 0) ? mlo[0] : 0xFFFFFFFFu;
        m &= (oh0 == 0) ? mlo[1] : 0xFFFFFFFFu;
        m &= (ow0 == 0) ? mlo[2] : 0xFFFFFFFFu;
        m &= (od0 + 4 == p.D) ? mhi[0] : 0xFFFFFFFFu;
        m &= (oh0 + 8 == p.H) ? mhi[1] : 0xFFFFFFFFu;
        m &= (ow0 + 8 == p.W) ? mhi[2] : 0xFFFFFFFFu;
        hmask[set] = m;
        lxt = xn + ((ptrdiff_t)((od0 - 1) * p.H + (oh0 - 1)) * p.W + (ow0 - 1)) * 64;
    };
    // ... and one slot (padding / unused slots: any valid address, zeroed at the LDS write)
    auto halo_slot = [&](auto I, auto SET) {
        constexpr int i = decltype(I)::value, set = decltype(SET)::value;
        const char* a = lxt + xoff[i];
        a = ((hmask[set] >> i) & 1u) ? a : xn;
        hv[set][i] = *reinterpret_cast<const uint4*>(a);
    };
    auto halo_load = [&](int t, auto SET) {
        halo_base(t, SET);
        static_for_c32<10>([&](auto I) { halo_slot(I, SET); });
    };
    // slot I of hv[SET]: prologue (the packed form of apply_pre: eight independent SiLU chains for the scheduler to interleave - the
    // staging is transcendental-bound, one wave per SIMD has nothing else to cover a dependent exp -> add -> rcp), zero fill of the
    // padding, LDS write (unused slots land in the 40 spare rows)
    if constexpr (PRE) __syncthreads();                                    // (coef)
    auto stage_slot = [&](auto I, auto SET, int buf) {
        constexpr int i = decltype(I)::value, set = decltype(SET)::value;
        uint4 u = hv[set][i];
        if constexpr (PRE) u = apply_pre<bf16_raw>(u, coef + piece * 8, coef + 32 + piece * 8, p.pre_silu);
        const bool ok = (hmask[set] >> i) & 1u;
        u.x = ok ? u.x : 0u; u.y = ok ? u.y : 0u; u.z = ok ? u.z : 0u; u.w = ok ? u.w : 0u;
        *reinterpret_cast<uint4*>(smem + buf * C32_HBUF + ((tid >> 2) + 64 * i) * PITCH + piece * 16) = u;
    };

    f32x16_t acc[2];
    uint4 fa[3][2], fb[3][2][2];                                           // operands of three taps in flight (read two taps ahead)
    auto rd_one = [&](auto T, auto R, const char* hb) {                    // read R (0..5) of tap T: A k-steps 0 / 1, then B (j, s)
        constexpr int t = decltype(T)::value, r = decltype(R)::value, set = t % 3;
        if constexpr (r < 2) {
            fa[set][r] = *reinterpret_cast<const uint4*>(wl + t * 2048 + aoff[r]);
        } else {
            constexpr int j = (r - 2) >> 1, s_ = (r - 2) & 1;
            constexpr int toff = (((t / 9) * 10 + (t / 3) % 3) * 10 + t % 3) * PITCH;
            fb[set][j][s_] = *reinterpret_cast<const uint4*>(hb + boff[j] + toff + 32 * s_);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;

    // ---- first tile: staged with nothing to hide behind; the second tile's loads go out behind it
    __syncthreads();                                                       // (weights, coef)
    halo_load(t0, S0{});
    static_for_c32<10>([&](auto I) { stage_slot(I, S0{}, 0); });
    if (t0 + 1 < t1) halo_load(t0 + 1, S1{});
    __syncthreads();

    // one tile (t - t0 of parity PAR): taps out of LDS buffer PAR; tile t + 2's halo loads issued into hv[PAR] (tile t's are in LDS
    // since the previous iteration); tile t + 1's halo (hv[PAR ^ 1], loaded one tile ago) through the prologue into the other buffer,
    // one slot behind every eighth MFMA of taps 4 .. 23; the residual of this tile's outputs fetched at its head.  Every
    // MFMA is followed in program order by its share of the reads of tap + 2 and of the staging, pinned by a scheduling barrier
    // (one wave per SIMD issues in order, see wgrad.hip).  Past the last tile the staging writes stale registers into the free buffer.
    auto tile = [&](int t, auto PAR) {
        constexpr int par = decltype(PAR)::value;
        const char* const hb = smem + par * C32_HBUF;
        int od0, oh0, ow0;
        tile_origin(t, od0, oh0, ow0);
        const size_t pbase = ((size_t)(n * p.D + od0) * p.H + oh0) * p.W + ow0;
        uint2 rv[2][4];
        if (p.res != nullptr) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    rv[j][g] = *reinterpret_cast<const uint2*>(p.res + ((pbase + opos[j]) * 32 + 4 * half + 8 * g) * 2);
        }
        if (t + 2 < t1) halo_load(t + 2, PAR);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        static_for_c32<6>([&](auto R) { rd_one(std::integral_constant<int, 0>{}, R, hb); });
        static_for_c32<6>([&](auto R) { rd_one(std::integral_constant<int, 1>{}, R, hb); });
        static_for_c32<27>([&](auto T) {
            constexpr int tp = decltype(T)::value;
            static_for_c32<4>([&](auto M) {
                constexpr int m = decltype(M)::value, j = m >> 1, s_ = m & 1;
                mma_step<bf16_raw>(fa[tp % 3][s_], fb[tp % 3][j][s_], acc[j]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (tp + 2 < 27) {                                // reads of tap + 2: 2, 2, 1, 1 behind the four MFMAs
                    constexpr int r0 = m == 0 ? 0 : m == 1 ? 2 : m == 2 ? 4 : 5;
                    constexpr int r1 = m == 0 ? 2 : m == 1 ? 4 : m == 2 ? 5 : 6;
                    static_for_c32<r1 - r0>([&](auto R) { rd_one(std::integral_constant<int, tp + 2>{}, std::integral_constant<int, r0 + decltype(R)::value>{}, hb); });
                }
                constexpr int g = tp * 4 + m;
                if constexpr (g >= 16 && g < 96 && ((g - 16) & 7) == 0)     // one slot behind every eighth MFMA of taps 4 .. 23
                    stage_slot(std::integral_constant<int, ((g - 16) / 8)>{}, std::integral_constant<int, (par ^ 1)>{}, par ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // the barrier BEFORE the stores: tile t + 1's halo is in LDS and nobody reads this tile's buffer any more; behind the stores
        // it would wait for them (vmcnt counts stores), a write latency per tile
        __syncthreads();
        // ---- epilogue: lane holds, for position column lane & 31 of tile j, channels 8 g + 4 half + {0..3} in acc[j][4 g + e]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t eo = (pbase + opos[j]) * 32 + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[j][4 * g + 0] + badd[4 * g + 0], v1 = acc[j][4 * g + 1] + badd[4 * g + 1];
                float v2 = acc[j][4 * g + 2] + badd[4 * g + 2], v3 = acc[j][4 * g + 3] + badd[4 * g + 3];
                if (p.res != nullptr) {
                    v0 += __uint_as_float(rv[j][g].x << 16); v1 += __uint_as_float(rv[j][g].x & 0xFFFF0000u);
                    v2 += __uint_as_float(rv[j][g].y << 16); v3 += __uint_as_float(rv[j][g].y & 0xFFFF0000u);
                }
                const uint2 o = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                *reinterpret_cast<uint2*>(p.y + (eo + 8 * g) * 2) = o;
                if (p.stats != nullptr) {                    // statistics of the values as stored
                    const float s0 = __uint_as_float(o.x << 16), s1 = __uint_as_float(o.x & 0xFFFF0000u);
                    const float s2 = __uint_as_float(o.y << 16), s3 = __uint_as_float(o.y & 0xFFFF0000u);
                    ssum[4 * g + 0] += s0; ssq[4 * g + 0] = fmaf(s0, s0, ssq[4 * g + 0]);
                    ssum[4 * g + 1] += s1; ssq[4 * g + 1] = fmaf(s1, s1, ssq[4 * g + 1]);
                    ssum[4 * g + 2] += s2; ssq[4 * g + 2] = fmaf(s2, s2, ssq[4 * g + 2]);
                    ssum[4 * g + 3] += s3; ssq[4 * g + 3] = fmaf(s3, s3, ssq[4 * g + 3]);
                }
            }
        }
    };
    for (int t = t0; t < t1; t += 2) {
        tile(t, S0{});
        if (t + 1 < t1) tile(t + 1, S1{});
    }

    // ---- statistics of this workgroup's tiles: lanes of one half hold the same 16 channels for different positions
    if (p.stats != nullptr) {
        float* const red = reinterpret_cast<float*>(smem);                 // [256 threads][32]
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            red[tid * 32 + k] = ssum[k];
            red[tid * 32 + 16 + k] = ssq[k];
        }
        __syncthreads();
        if (tid < 64) {
            const int stat = tid >> 5, c = tid & 31;
            const int g = c >> 3, hf = (c >> 2) & 1, e = c & 3;
            const int k = stat * 16 + 4 * g + e;
            float a = 0.0f;
            for (int w = 0; w < 4; ++w)
                for (int l = 0; l < 32; ++l) a += red[(w * 64 + hf * 32 + l) * 32 + k];
            p.stats[(((size_t)n * p.wps + wg) * 2 + stat) * 32 + c] = a;
        }
    }
}

// ------------------------------------------------------------------------------------------ host
// the launches k_conv32 takes: bf16, 3x3x3, stride 1, one 32-channel source, 32 output channels channels-last, whole 4 x 8 x 8 tiles,
// none of the other fused forms (RHO_CONV32=0: never)
static bool conv32_applies(const rho_conv_desc& d) {
    static const bool on = !(getenv("RHO_CONV32") && atoi(getenv("RHO_CONV32")) == 0);
    if (!on) return false;
    if (d.dtype != RHO_BF16 || d.kd != 3 || d.kh != 3 || d.kw != 3 || d.sh != 1 || d.sw != 1 || d.up_h || d.up_w) return false;
    if (d.x2 || d.c1 != 32 || d.cout != 32 || d.coutp != 32 || d.split != 32 || d.y2 || !d.y) return false;
    if (d.zs_h || d.zs_w || d.ph_h || d.ph_w || d.phd_h || d.phd_w || d.y2_cl || d.res2 || d.gnb_x1 || d.gna_g || d.sk_w) return false;
    if (d.d % 4 || d.h % 8 || d.w_ % 8) return false;
    if ((long long)d.d * d.h * d.w_ * 64 >= (1LL << 32) || (long long)d.n * d.d * d.h * d.w_ >= (1LL << 31)) return false;
    if (d.pre_a && !d.pre_b) return false;
    return true;
}
// workgroups per sample: one persistent workgroup per CU over the whole launch where the batch allows it
static int conv32_wps(const rho_conv_desc& d) {
    const long long tps = (long long)(d.d / 4) * (d.h / 8) * (d.w_ / 8);
    long long w = d.n >= 256 ? 1 : (256 + d.n - 1) / d.n;
    if (w > tps) w = tps;
    return (int)w;
}
static int launch_conv32(const rho_conv_desc& d, hipStream_t st) {
    C32K k{};
    k.x = (const char*)d.x1; k.w = (const char*)d.w; k.bias = d.bias; k.pre_a = d.pre_a; k.pre_b = d.pre_b; k.res = (const char*)d.res;
    k.res_add = d.res_add; k.y = (char*)d.y; k.stats = d.stats;
    k.D = d.d; k.H = d.h; k.W = d.w_;
    k.tiles_h = d.h / 8; k.tiles_w = d.w_ / 8; k.tps = (d.d / 4) * k.tiles_h * k.tiles_w;
    k.wps = conv32_wps(d);
    k.pre_silu = d.pre_silu; k.res_add_stride = d.res_add_stride > 0 ? d.res_add_stride : 32;
    if (d.n > 65535) return RHO_E_SHAPE;
    const size_t lds = (size_t)2 * C32_HBUF + C32_WLDS + 256;
    auto go = [&](auto fn) -> int {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(fn, dim3((unsigned)k.wps, (unsigned)d.n), dim3(256), lds, st, k);
        e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    };
    return d.pre_a ? go(k_conv32<true>) : go(k_conv32<false>);
}
