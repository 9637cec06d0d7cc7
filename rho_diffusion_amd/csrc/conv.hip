// n-D convolution (kernel 1 or 3 per axis, zero padding k/2) for channels-last activations as an
// implicit GEMM on the CDNA4 matrix cores, designed around the 160 KiB LDS of a gfx950 CU:
//
//   * A workgroup (4 waves) owns an output tile of 256 positions (TD x TH x TW, chosen per layer)
//     and BM output channels.  For each 64-byte input-channel chunk (32 bf16 / 16 f32 channels) it
//     stages the tile's input region INCLUDING THE HALO into LDS once -- applying the folded
//     GroupNorm+FiLM affine and SiLU on the way (unet_v2.py:212-216,285-289), zero-filling the
//     padding AFTER the activation -- and then runs all kd*kh*kw taps out of that resident tile
//     at shifted LDS addresses.  Global->LDS traffic is ~2.3x the input instead of 27x, the
//     GN/SiLU work is paid 2.3x instead of 27x, and torch.cat / nearest-upsample / stride never
//     exist as tensors: they are address arithmetic in the loader (unet_v2.py:729,122-131,153).
//   * Orientation is D[cout][position] = W[cout][k] * X[k][position] with 32x32 MFMA tiles
//     (v_mfma_f32_32x32x16_bf16, or the exact-f32 v_mfma_f32_32x32x2_f32): each lane ends up
//     holding 4 consecutive output channels of ONE position, so the epilogue (bias, residual,
//     additive embedding) writes 8/16-byte channels-last pieces with no shuffles.
//   * LDS rows are 64 B of payload on an 80 B pitch: for ds_read_b128 the 16-byte slot index is
//     5*row mod 16, a bijection, so 32 consecutive rows are conflict-free for every lane group.
//   * Weights ([tap][cout][cin], prepared once) stream through a 2-deep LDS ring, one tap ahead.
//
// Replaces conv_nd at every call site of rho_diffusion/models/unet_v2.py (see include/rho_hip.h).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv_common.h"

// Taps per workgroup barrier of the 8-wave M16 variant (with the tap order pinned, see RHO_FENCE: 3 is 3 % faster than 1; the
// LDS weight ring is 9 slots deep then).  The host sizes the ring with the same constant.
#define RHO_GB_WIDE 3
// ... and on the 32-cout variant (4 MFMAs per tap and wave = 128 MFMA cycles between barriers; round 3, c5's mc = 32 levels: conv3
// 17.5 -> 17.1 ms per sampling step in a same-box A/B, gpurun_out/ab2_*; 0 = a barrier per tap)
#ifndef RHO_GB_BM32
#define RHO_GB_BM32 1
#endif

struct ConvK {
    const char* x1;
    const char* x2;
    const float* pre_a;
    const float* pre_b;
    const char* w;
    const float* bias;
    const char* res;
    const float* res_add;
    char* y;
    char* y2;
    int c1, c2, cin;
    int cout, coutp, split;
    int D, H, W;     // input extents (depth carries the batch when merged)
    int Do, Ho, Wo;  // output extents
    long long S_in, S_out;  // positions per sample
    int sh, sw, up_h, up_w, pre_silu, y2_f32;
    int TD, TH, TW, ID, IH, IW, NP;
    int tiles_h, tiles_w;
    int res_add_stride;
    int y2_cl;            // y2 is channels-last [.., cout - split] (dgrad of a concatenated input) instead of channel-major
    const char* res2;     // residual for the y2 region (channels-last only): in-place gradient accumulation
    int pair_lg;          // -1: none; else log2 of the row-index bit that pairs two 8-wide halo rows 8 (mod 16) positions apart
    int zs_h, zs_w;       // zero-stuffed input (dgrad of a stride-2 conv): virtual extent H/W, source extent Hs/Ws
    int Hs, Ws;
    float* stats;         // fused GroupNorm statistics of the output: [N][tps][2][split] (see rho_conv_desc.stats)
    int tps;              // tiles per sample
    int lgTW, lgTH;       // tile extents are powers of two
    float inv_ihw, inv_iw;  // 1 / (IH*IW), 1 / IW: exact small-integer division through one float multiply
    int cofast;           // cout tiles of one position tile on consecutive launch slots of one XCD
    int coef_off;         // > 0: byte offset in LDS of the staged prologue coefficients [2][cin] (3-D tiles: one sample per tile)
    // Sub-pixel phase of a conv behind a nearest x2 upsample (rho_conv_desc.ph_h / ph_w): taps start pad_h / pad_w rows before the
    // output position; the launch writes output row oh * oy_mul + oy_add of a tensor with Ho_out x Wo_out rows per depth slice
    int pad_h, pad_w, oy_mul, oy_add, ox_mul, ox_add, Ho_out, Wo_out;
    int stats_off;        // first statistics tile of this launch within the tps tiles of a sample
    int iy_mul, iy_add, ix_mul, ix_add;   // input row / column of virtual position g: g * mul + add (phd_*: one parity of a 2x tensor)
    // GroupNorm backward reduction fused into a dgrad launch (rho_conv_desc.gnb_*): stats <- per-tile sums of dz and dz * x
    const char* gnb_x1;
    const char* gnb_x2;
    const float* gnb_a;
    const float* gnb_b;
    int gnb_c1, gnb_silu;
    // Split-K (small grids, rho_conv_desc.ws): blockIdx.z = k-split s of ksplit; the split contracts the input-channel chunks
    // [nck * s / ksplit, nck * (s + 1) / ksplit) and stores its fp32 partial tile to slab[s] ([positions][coutp]); k_splitk_reduce
    // adds the slabs, the bias, the residuals and rounds once
    int ksplit;
    float* slab;
    long long slab_stride;      // floats per split
    // Folded 1x1x1 skip convolution (rho_conv_desc.sk_*; FSK instantiations): contracted into the accumulators before the tap loop
    const char* sk_x1;
    const char* sk_x2;
    const char* sk_w;
    const float* sk_bias;
    int sk_c1, sk_c2;
    // GroupNorm backward APPLY fused into the epilogue (rho_conv_desc.gna_*): out += cA * (g * act'(a x0 + b)) + cQ * x0 + cP
    const char* gna_g;
    const float* gna_cA;
    const float* gna_cP;
    const float* gna_cQ;
};

// floor(a / d) for 0 <= a < 2^20 with inv = 1.0f / d: (a + 0.5) * inv is at least 0.5 / d away from an integer, the
// float error is ~1e-7 * a / d.  (The integer division sequence costs ~20 VALU instructions; the per-tile setup of the
// narrow layers ran 1100 VALU instructions of this kind against 430 MFMAs.)
__device__ __forceinline__ int fdiv_small(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }


// Column c (0..31) of MFMA tile m (0..7 = wave*2 + j) -> index of the output position inside the 256-position tile.
// ds_read_b128 serves a wave in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with an odd slot pitch a
// group is conflict-free iff its 16 halo positions are distinct mod 16.  So the columns are ordered group-major and
// each group gets either 16 consecutive positions of one row (TW >= 16) or, for 8-wide tiles, two rows whose halo
// offset is 8 (mod 16) -- the row bit `pair_lg` chosen on the host.  (PMC before this map: SQ_LDS_BANK_CONFLICT was
// 40-50 % of SQ_LDS_IDX_ACTIVE in this kernel.)
// position of rank k (0..15) of 16-position unit u (0..15): 16 positions whose halo rows are distinct mod 16
__device__ __forceinline__ int unit_position(int u, int k, int TW, int pair_lg) {
    if (TW == 8 && pair_lg >= 0) {
        const int lo = u & ((1 << pair_lg) - 1);
        const int row = ((u >> pair_lg) << (pair_lg + 1)) | lo | ((k >> 3) << pair_lg);
        return row * 8 + (k & 7);
    }
    return u * 16 + k;
}
__device__ __forceinline__ int tile_position(int m, int c, int TW, int pair_lg) {
    const int q = c >> 2;
    const int g = (0x96 >> q) & 1;                 // lane group of this column
    const int k = ((q >> 1) << 2) | (c & 3);       // rank inside the group, 0..15
    return unit_position(2 * m + g, k, TW, pair_lg);
}

// The 16x16x32 MFMA layout (M16 variants): a lane holds row / column i = lane & 15 and the 16-byte K piece lane >> 4, so
// a ds_read_b128 lane group {0-3, 12-15 | 20-27} mixes 8 rows at piece p with 8 rows at piece p + 1.  With the odd slot
// pitch that is conflict-free iff both 8-row halves have distinct halo rows (mod 16) of ONE parity: column tile m16 takes
// the ranks of parity (m16 & 1) of the two units of pair m16 >> 1 - half A (i in 0-3, 12-15) from the even unit, half B
// (i in 4-11) from the odd one.  The same 32 positions as the 32-wide tile m16 >> 1 of the other layout.
__device__ __forceinline__ int m16_hb(int i) { return (0x0FF0 >> i) & 1; }
__device__ __forceinline__ int m16_k8(int i) { return m16_hb(i) ? i - 4 : (i < 4 ? i : i - 8); }
__device__ __forceinline__ int tile_position16(int m16, int i, int TW, int pair_lg) {
    return unit_position(2 * (m16 >> 1) + m16_hb(i), 2 * m16_k8(i) + (m16 & 1), TW, pair_lg);
}

// GNA: the 1x1x1 instantiations whose epilogue carries the GroupNorm backward apply (rho_conv_desc.gna_*; no residual operand there:
// its 32 registers are what the extra operand takes)
template <typename T, int KD, int KH, int KW, int BM, int MAXP, int NW, bool M16 = false, bool FSK = false, bool GNA = false>
// (second launch bound = waves per SIMD the register allocation must allow: 2 for the 8-wave variants and for the 4-wave 1x1x1
//  variant, which without it took 241 VGPRs + 64 AGPRs = one workgroup per CU and left its HBM-bound layers at 3.3 TB/s)
__global__ __launch_bounds__(NW * 64, (NW == 8 || (KD * KH * KW == 1 && MAXP <= 10)) ? 2 : 1) void k_conv(const ConvK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CK = ET<T>::CK;
    constexpr int PE = ET<T>::PE;
    constexpr int NT = KD * KH * KW;
    constexpr int NTHR = NW * 64;
    constexpr int RPP = NTHR / 4;          // rows (halo positions / weight rows) covered per staging pass
    constexpr int WCO = NW / 4;            // waves along cout
    constexpr int MT = BM / 32 / WCO;      // 32-row cout tiles per wave
    constexpr int WROWS = (BM + RPP - 1) / RPP;  // weight-tile rows per thread

    char* const halo = smem;
    char* const wbuf = smem + (size_t)p.NP * PITCH;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wpos = wave & 3;             // position quarter of this wave
    const int wco = wave >> 2;             // cout half (NW == 8)
    const int half = lane >> 5;
    const int piece = tid & 3;

    // ---- tile coordinates
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Give every XCD a contiguous range of
    // tiles instead of every 8th one, so that the halo faces shared by neighbouring tiles are re-read from the same L2.
    int bt = blockIdx.x;
    int ct = blockIdx.y;
    if (p.cofast) {
        // several cout tiles and gridDim.x a multiple of 8: the cout tiles of one position tile take consecutive launch
        // slots of the same XCD, so the input tile they all read comes from HBM once and from that L2 afterwards
        const int L = blockIdx.x + gridDim.x * blockIdx.y;
        const int slot = L >> 3;
        ct = slot % (int)gridDim.y;
        bt = (L & 7) * ((int)gridDim.x >> 3) + slot / (int)gridDim.y;
    } else {
        const int nt = gridDim.x, per = nt >> 3;
        if (per > 0) {
            const int full = per << 3;                      // tiles covered by the 8 equal ranges; the tail keeps its index
            if (bt < full) bt = (bt & 7) * per + (bt >> 3);
        }
    }
    const int tw_i = bt % p.tiles_w;
    const int th_i = (bt / p.tiles_w) % p.tiles_h;
    const int td_i = bt / (p.tiles_w * p.tiles_h);
    const int co0 = ct * BM;
    const bool splitk = (KD != 3) && p.ksplit > 1;            // (2-D / 1-D launches have gridDim.z = 1 otherwise: z carries the k-split)
    const int n = splitk ? 0 : (int)blockIdx.z;
    const int od0 = td_i * p.TD, oh0 = th_i * p.TH, ow0 = tw_i * p.TW;
    const int gd_base = od0 - (KD / 2);
    const int gh_base = p.up_h ? (oh0 / 2 - 1) : (oh0 * p.sh - p.pad_h);
    const int gw_base = p.up_w ? (ow0 / 2 - 1) : (ow0 * p.sw - p.pad_w);

    // ---- halo slots owned by this thread: global linear input position (or -1 = zero padding,
    //      -2 = beyond the tile) and sample index for the prologue coefficients
    constexpr int NSMP = (KD == 3) ? 1 : MAXP;  // 3-D tiles never straddle samples: the sample is blockIdx.z
    int spos[MAXP];
    int ssmp[NSMP];
    {
        const int ihw = p.IH * p.IW;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int hp = (tid >> 2) + RPP * i;
            int pos = -2, smp = 0;
            if (hp < p.NP) {
                const int id = fdiv_small(hp, p.inv_ihw);
                const int r = hp - id * ihw;
                const int ih = fdiv_small(r, p.inv_iw);
                const int iw = r - ih * p.IW;
                const int gd = gd_base + id;
                int gh = gh_base + ih, gw = gw_base + iw;
                bool ok = ((unsigned)gd < (unsigned)p.D) & ((unsigned)gh < (unsigned)p.H) & ((unsigned)gw < (unsigned)p.W);
                if (p.zs_h) { ok = ok && ((gh & 1) == 0) && ((gh >> 1) < p.Hs); gh >>= 1; }
                if (p.zs_w) { ok = ok && ((gw & 1) == 0) && ((gw >> 1) < p.Ws); gw >>= 1; }
                if (ok) {
                    pos = ((n * p.D + gd) * p.Hs + (gh * p.iy_mul + p.iy_add)) * p.Ws + (gw * p.ix_mul + p.ix_add);
                    if constexpr (KD != 3) smp = (int)((long long)pos / p.S_in);
                } else {
                    pos = -1;
                }
            }
            spos[i] = pos;
            if constexpr (KD != 3) ssmp[i] = smp;
        }
    }

    // ---- per-lane LDS byte offsets of the B (activation) fragments, per column tile and tap axis
    // (M16: j = pair of 16-wide column tiles; the odd tile of a pair is one halo row = +PITCH further, stride 1 only)
    int offd[2], offh[2][KH], offw[2][KW];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pp = M16 ? tile_position16(2 * (wpos * 2 + j), lane & 15, p.TW, p.pair_lg)
                           : tile_position(wpos * 2 + j, lane & 31, p.TW, p.pair_lg);
        const int pw = pp & (p.TW - 1);
        const int ph = (pp >> p.lgTW) & (p.TH - 1);
        const int pd = pp >> (p.lgTW + p.lgTH);
        offd[j] = pd * p.IH * p.IW * PITCH + 16 * (M16 ? (lane >> 4) : half);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int ih = p.up_h ? (((ph + kh - 1) >> 1) + 1) : (ph * p.sh + kh);
            offh[j][kh] = ih * p.IW * PITCH;
        }
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const int iw = p.up_w ? (((pw + kw - 1) >> 1) + 1) : (pw * p.sw + kw);
            offw[j][kw] = iw * PITCH;
        }
    }
    // M16: MFMA row i of cout tile t (16 couts) lives in LDS row 32 * (t >> 1) + 16 * hb(i) + 2 * k8(i) + (t & 1)
    const int a_off = M16 ? (wco * (BM / WCO) + 16 * m16_hb(lane & 15) + 2 * m16_k8(lane & 15)) * PITCH + 16 * (lane >> 4)
                          : (wco * (BM / WCO) + (lane & 31)) * PITCH + 16 * half;

    f32x16_t acc[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.0f;

    int ck_lo = 0, nck = p.cin / CK;
    if (splitk) {
        const int nall = nck, kz = (int)blockIdx.z;
        ck_lo = (int)((long long)nall * kz / p.ksplit);
        nck = (int)((long long)nall * (kz + 1) / p.ksplit) - ck_lo;       // >= 1: the host keeps ksplit <= chunks
    }
    const size_t wrow_bytes = (size_t)p.cin * sizeof(T);

    // weight tile: BM rows x 4 pieces of 16 B; thread -> row (tid>>2) + 64*k, piece tid&3.
    // (plain scalars, no arrays by reference: those end up in scratch)
    const bool w_active = (BM >= RPP) || (tid < BM * 4);
    // address = uniform 64-bit base (tap, chunk, cout tile: scalar ALU) + one 32-bit per-thread offset, the `saddr` form of
    // global_load: per-tap 64-bit pointers in VGPRs (27 taps x 2 registers, hoisted out of the chunk loop) spilled to scratch
    const char* const w_src0 = p.w + (size_t)co0 * wrow_bytes + (size_t)ck_lo * 64;      // (a chunk is 64 bytes of every weight row)
    // (re-declared opaque at every chunk: otherwise base + tap * stride + offset is hoisted as 27 per-lane 64-bit pointers anyway)
    unsigned w_voff = (unsigned)(tid >> 2) * (unsigned)wrow_bytes + (unsigned)piece * 16u;         // < 128 rows x 16 KiB
    const size_t w_tap_stride = (size_t)p.coutp * wrow_bytes;
    int w_row = tid >> 2;                  // LDS row of this thread's weight row(s); rows RPP apart keep the permutation
    if constexpr (M16) w_row = (w_row & ~31) + 16 * m16_hb(w_row & 15) + 2 * m16_k8(w_row & 15) + ((w_row >> 4) & 1);
    const int w_dst0 = w_row * PITCH + piece * 16;
    // Weights: G taps form one barrier step (G = 3 for the narrow-cout variants, where 8 MFMAs per barrier would be
    // barrier-bound), fetched PD steps ahead of their use (L2 latency ~1.5k cycles vs 0.25-0.5k cycles of MFMA per
    // tap) into a register ring of PD sets; set (step % PD) holds that step's G tiles; the LDS ring stays 2 deep.
    constexpr int G = 1;   // measured: G = 3 on the narrow-cout variants changes nothing (barriers are not their limiter)
    constexpr int NS = NT / G;
    constexpr int PD = (NS % 3 == 0) ? 3 : 1;
    constexpr int SLOT = G * BM * PITCH;
    uint4 wq0[PD][G], wq1[PD][G];
#pragma unroll
    for (int i = 0; i < PD; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) wq0[i][g] = wq1[i][g] = make_uint4(0u, 0u, 0u, 0u);
#define RHO_LOAD_W(set_, ck_, st_)                                                                       \
    do {                                                                                                \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                                              \
            const char* ws_ = w_src0 + (size_t)((st_) * G + g_) * w_tap_stride + (size_t)(ck_) * 64;    \
            if (w_active) wq0[set_][g_] = *reinterpret_cast<const uint4*>(ws_ + w_voff);                \
            if constexpr (WROWS == 2) wq1[set_][g_] = *reinterpret_cast<const uint4*>(ws_ + RPP * wrow_bytes + w_voff); \
        }                                                                                               \
    } while (0)
// PIPE paths inside the chunk loop: the fetch pointer `wl` walks tap by tap (a loop-carried scalar: nothing to hoist, two scalar
// adds per tap; 27 hoisted 64-bit bases overflowed the SGPR file and came back as v_readlane per use).
#define RHO_LOAD_WP(set_, st_)                                                                          \
    do {                                                                                                \
        if (w_active) wq0[set_][0] = *reinterpret_cast<const uint4*>(wl + w_voff);                      \
        if constexpr (WROWS == 2) wq1[set_][0] = *reinterpret_cast<const uint4*>(wl + RPP * wrow_bytes + w_voff); \
        if (((st_) + LD) % NS == NS - 1) wl = w_src0 + (size_t)min(ck + ((st_) + LD) / NS + 1, nck - 1) * 64; \
        else wl += w_tap_stride;                                                                        \
    } while (0)
#define RHO_STORE_W(buf_, set_)                                                                         \
    do {                                                                                                \
        _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                                              \
            char* wd_ = wbuf + (size_t)(buf_) * SLOT + g_ * BM * PITCH + w_dst0;                        \
            if (w_active) *reinterpret_cast<uint4*>(wd_) = wq0[set_][g_];                               \
            if constexpr (WROWS == 2) *reinterpret_cast<uint4*>(wd_ + RPP * PITCH) = wq1[set_][g_];     \
        }                                                                                               \
    } while (0)

    // Pipelined variant (3 | NS, i.e. every kernel with more than one tap): an R-slot LDS weight ring, slot = step % R.
    // Step q's tile is fetched at step q-LD into register set q % 3, handed to LDS at the end of step q-D (two steps in
    // flight) and read from step q-1 on, at least one barrier AFTER it was written - so the fragment reads of step q+1
    // are issued while step q's MFMAs run, instead of right after a barrier with every wave of the workgroup waiting.
    // GB = steps per barrier: a write at step q needs a barrier in (q, q+D-1] before its first read and one in
    // (q+D-R, q] after the slot's last read, i.e. D = GB+1 and R >= 2*GB+1 (R = 9 for GB = 3 keeps the slot static).
    constexpr bool PIPE = (NS % 3 == 0) && (G == 1);
    // measured on the 8-wave 128-cout variant: GB = 3, R = 9 is 3 % SLOWER than a barrier per tap (131 vs 127 ms per
    // c3 step) - the per-tap barrier is not what limits this kernel - so GB stays 1.
    // (the slot of a step is st % RS: the ring depth must divide the taps of a chunk - 9 | 27 and 9 | 9, not the 3 taps of 1-D)
    constexpr int GB = (NS % 9 == 0 && ((BM == 128 && NW == 8 && M16) || (RHO_GB_BM32 && BM == 32 && !M16))) ? RHO_GB_WIDE : 1;
    constexpr int RS = (GB == 3) ? 9 : 3;          // LDS ring slots
    constexpr int DS = GB + 1;                     // store distance
    constexpr int LD = DS + 2;                     // load distance
    // Folded GroupNorm affine of the prologue: read from global memory at every chunk's landing it is one exposed L2 round
    // trip per chunk; a 3-D tile lies in one sample, so that sample's coefficients are staged in LDS once per tile.
    const float* coef = nullptr;
    if constexpr (NT != 1 && KD == 3) {
        if (p.coef_off > 0 && p.pre_a != nullptr) {
            float* cw_ = reinterpret_cast<float*>(smem + p.coef_off);
            for (int i = tid; i < p.cin; i += NTHR) {
                cw_[i] = p.pre_a[(size_t)n * p.cin + i];
                cw_[p.cin + i] = p.pre_b[(size_t)n * p.cin + i];
            }
            coef = cw_;
            __syncthreads();
        }
    }
    if constexpr (FSK) {
        // ---- folded 1x1x1 skip convolution: acc += sk_w[co][:] . cat(sk_x1, sk_x2)[position][:] over the tile's 256 output positions,
        // one "tap" of the M16 schedule per 32-channel chunk, BEFORE the tap loop touches LDS.  The activations are staged in the
        // tile's own order (row r = position r of the tile: the odd 16-column tile of a pair is one row further, as in the halo)
        // into two buffers over the bytes the halo tile and the weight ring use afterwards; chunk c + 1 is fetched while chunk c
        // runs, one barrier per chunk.
        static_assert(M16 && MT == 2 && KD == 3, "folded skip: the 64-couts-per-wave 16x16x32 layout of the 3-D kernels");
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        constexpr int XR = 256 / RPP;                       // activation rows per thread
        const int skc = p.sk_c1 + p.sk_c2, nsk = skc / CK;
        char* const sa = smem;                               // [2][256 rows]
        char* const sw = smem + 2 * 256 * PITCH;             // [2][BM rows]
        int gp[XR];
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int r = (tid >> 2) + RPP * i;
            const int od = od0 + (r >> (p.lgTW + p.lgTH)), oh = oh0 + ((r >> p.lgTW) & (p.TH - 1)), ow = ow0 + (r & (p.TW - 1));
            const bool ok = od < p.Do && oh < p.Ho && ow < p.Wo;
            gp[i] = ok ? ((n * p.Do + od) * p.Ho + oh) * p.Wo + ow : -1;
        }
        int cb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            cb[j] = tile_position16(2 * (wpos * 2 + j), lane & 15, p.TW, p.pair_lg) * PITCH + 16 * (lane >> 4);
        const unsigned skw_voff = (unsigned)(tid >> 2) * (unsigned)skc * (unsigned)sizeof(T) + (unsigned)piece * 16u;
        const char* const skw0 = p.sk_w + (size_t)co0 * skc * sizeof(T);
        uint4 xr[XR], wr0 = make_uint4(0u, 0u, 0u, 0u), wr1 = make_uint4(0u, 0u, 0u, 0u);
        auto sk_load = [&](int c) {
            const int ch = c * CK;
            const char* src;
            int cs, csrc;
            if (ch < p.sk_c1) { src = p.sk_x1; cs = p.sk_c1; csrc = ch; } else { src = p.sk_x2; cs = p.sk_c2; csrc = ch - p.sk_c1; }
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int ps = gp[i] > 0 ? gp[i] : 0;
                xr[i] = *reinterpret_cast<const uint4*>(src + ((size_t)ps * cs + csrc) * sizeof(T) + piece * 16);
            }
            const char* ws_ = skw0 + (size_t)c * 64;
            if (w_active) wr0 = *reinterpret_cast<const uint4*>(ws_ + skw_voff);
            if constexpr (WROWS == 2) wr1 = *reinterpret_cast<const uint4*>(ws_ + (size_t)RPP * skc * sizeof(T) + skw_voff);
        };
        auto sk_store = [&](int slot) {
#pragma unroll
            for (int i = 0; i < XR; ++i) {
                const int r = (tid >> 2) + RPP * i;
                *reinterpret_cast<uint4*>(sa + (slot * 256 + r) * PITCH + piece * 16) = gp[i] >= 0 ? xr[i] : make_uint4(0u, 0u, 0u, 0u);
            }
            char* const wd = sw + slot * BM * PITCH + w_dst0;
            if (w_active) *reinterpret_cast<uint4*>(wd) = wr0;
            if constexpr (WROWS == 2) *reinterpret_cast<uint4*>(wd + RPP * PITCH) = wr1;
        };
        auto sk_mm = [&](const u32x4_t (&fa)[2], const u32x4_t (&fb)[2], int tp, int jp) {
            f32x16_t& c = acc[tp][jp];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int q = 4 * (2 * tt + jj);
                    f32x4_t v = {c[q], c[q + 1], c[q + 2], c[q + 3]};
                    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[tt]), __builtin_bit_cast(bf16x8_t, fb[jj]), v, 0, 0, 0);
                    c[q] = v[0]; c[q + 1] = v[1]; c[q + 2] = v[2]; c[q + 3] = v[3];
                }
        };
        sk_load(0);
        for (int c = 0; c < nsk; ++c) {
            const int slot = c & 1;
            sk_store(slot);                                  // (slot last read in iteration c - 2: every wave has passed the barrier of c - 1)
            if (c + 1 < nsk) sk_load(c + 1);
            __syncthreads();
            u32x4_t fa01[2], fa23[2], fb0[2], fb1[2];
            const char* const wA = sw + slot * BM * PITCH + a_off;
            const char* const xB = sa + slot * 256 * PITCH;
            fa01[0] = *reinterpret_cast<const u32x4_t*>(wA);
            fa01[1] = *reinterpret_cast<const u32x4_t*>(wA + PITCH);
            fb0[0] = *reinterpret_cast<const u32x4_t*>(xB + cb[0]);
            fb0[1] = *reinterpret_cast<const u32x4_t*>(xB + cb[0] + PITCH);
            fb1[0] = *reinterpret_cast<const u32x4_t*>(xB + cb[1]);
            fb1[1] = *reinterpret_cast<const u32x4_t*>(xB + cb[1] + PITCH);
            fa23[0] = *reinterpret_cast<const u32x4_t*>(wA + 32 * PITCH);
            fa23[1] = *reinterpret_cast<const u32x4_t*>(wA + 33 * PITCH);
            sk_mm(fa01, fb0, 0, 0);
            sk_mm(fa01, fb1, 0, 1);
            sk_mm(fa23, fb0, 1, 0);
            sk_mm(fa23, fb1, 1, 1);
        }
        __syncthreads();                                     // the tap loop's prologue overwrites these LDS bytes
    }
    const char* wl = w_src0;
    if constexpr (NT != 1) {
    if constexpr (PIPE) {
        RHO_LOAD_W(0, 0, 0);
        RHO_LOAD_W(1, 0, 1);
        RHO_LOAD_W(2, 0, 2);
        RHO_STORE_W(0, 0);
        RHO_STORE_W(1, 1);
        if constexpr (DS > 2) RHO_STORE_W(2, 2);
#pragma unroll
        for (int q = 3; q < LD; ++q) RHO_LOAD_W(q % 3, min(q / NS, nck - 1), q % NS);
#pragma unroll
        for (int q = 3; q < DS; ++q) RHO_STORE_W(q % RS, q % 3);
        wl = w_src0 + (size_t)(LD % NS) * w_tap_stride + (size_t)min(LD / NS, nck - 1) * 64;   // step LD: fetched at tap 0 of chunk 0
    } else {
        // steps 0 .. PD-1 of chunk 0 in flight (PD <= NS), step 0 landed in LDS slot 0
#pragma unroll
        for (int i = 0; i < PD; ++i) RHO_LOAD_W(i, 0, i);
        RHO_STORE_W(0, 0);
    }
    int cur = 0;

    // Halo staging.  HPF (one workgroup per CU, nothing else to hide behind): the NEXT chunk's global loads are
    // issued into registers while the last taps of the current chunk run on the matrix cores, and only the
    // prologue + LDS write happens between chunks (issue-early / write-late).  Otherwise (>= 2 workgroups per CU
    // overlap each other) load and write in small batches to keep the register footprint down.
    constexpr bool HPF = (BM >= 128) && (MAXP * RPP <= 640);
    constexpr int TPF = (NT > 6) ? NT - 6 : 0;        // tap at which the next chunk's loads are issued
    uint4 hv[HPF ? MAXP : 1];
    auto halo_src = [&](int ck_, const char*& src, int& cs, int& csrc) {
        const int c = (ck_lo + ck_) * CK;
        if (c < p.c1) { src = p.x1; cs = p.c1; csrc = c; } else { src = p.x2; cs = p.c2; csrc = c - p.c1; }
    };
#define RHO_HALO_LOAD(ck_)                                                                                         \
    do {                                                                                                          \
        const char* src_; int cs_, csrc_;                                                                         \
        halo_src(ck_, src_, cs_, csrc_);                                                                          \
        /* branch-free (padding lanes read position 0 and are zeroed at write time): loads under an exec mask */  \
        /* are invisible to the compiler's counted vmcnt waits, which would then drain them immediately        */  \
        _Pragma("unroll") for (int i = 0; i < MAXP; ++i) {                                                        \
            const int ps_ = spos[i] > 0 ? spos[i] : 0;                                                            \
            hv[i] = *reinterpret_cast<const uint4*>(src_ + ((size_t)ps_ * cs_ + csrc_) * sizeof(T) + piece * 16); \
        }                                                                                                         \
    } while (0)
    if constexpr (HPF) RHO_HALO_LOAD(0);

    for (int ck = 0; ck < nck; ++ck) {
        if constexpr (M16) asm volatile("" : "+v"(w_voff));
        // ---- stage the halo tile of this channel chunk (previous chunk's reads are fenced by the
        //      barrier that closed its last tap)
        {
            const int c = (ck_lo + ck) * CK;
            if constexpr (HPF) {
#pragma unroll
                for (int i = 0; i < MAXP; ++i) {
                    if (spos[i] != -2) {
                        uint4 u = spos[i] >= 0 ? hv[i] : make_uint4(0u, 0u, 0u, 0u);
                        if (p.pre_a != nullptr && spos[i] >= 0) {
                            if (KD == 3 && coef != nullptr) {
                                u = apply_pre<T>(u, coef + c + piece * PE, coef + p.cin + c + piece * PE, p.pre_silu);
                            } else {
                                const int smp = (KD == 3) ? n : ssmp[(KD == 3) ? 0 : i];
                                const size_t co = (size_t)smp * p.cin + c + piece * PE;
                                u = apply_pre<T>(u, p.pre_a + co, p.pre_b + co, p.pre_silu);
                            }
                        }
                        const int hp = (tid >> 2) + RPP * i;
                        *reinterpret_cast<uint4*>(halo + hp * PITCH + piece * 16) = u;
                    }
                }
            } else {
                const char* src;
                int cs, csrc;
                halo_src(ck, src, cs, csrc);
#ifdef RHO_PROBE_GB10   /* probe build: the whole small halo (10 pieces) in one batch - one exposed load latency per chunk instead of two */
                constexpr int GB = (MAXP == 10) ? 10 : ((MAXP % 7 == 0) ? 7 : 5);
#else
                constexpr int GB = (MAXP % 7 == 0) ? 7 : 5;  // loads in flight per batch (divides MAXP)
#endif
#pragma unroll
                for (int i0 = 0; i0 < MAXP; i0 += GB) {
                    uint4 v[GB];
#pragma unroll
                    for (int q = 0; q < GB; ++q) {
                        const int i = i0 + q;
                        if (i < MAXP) {
                            v[q] = make_uint4(0u, 0u, 0u, 0u);
                            if (spos[i] >= 0)
                                v[q] = *reinterpret_cast<const uint4*>(src + ((size_t)spos[i] * cs + csrc) * sizeof(T) + piece * 16);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < GB; ++q) {
                        const int i = i0 + q;
                        if (i < MAXP) {
                            if (spos[i] != -2) {
                                uint4 u = v[q];
                                if (p.pre_a != nullptr && spos[i] >= 0) {
                                    if (KD == 3 && coef != nullptr) {
                                        u = apply_pre<T>(u, coef + c + piece * PE, coef + p.cin + c + piece * PE, p.pre_silu);
                                    } else {
                                        const int smp = (KD == 3) ? n : ssmp[(KD == 3) ? 0 : i];
                                        const size_t co = (size_t)smp * p.cin + c + piece * PE;
                                        u = apply_pre<T>(u, p.pre_a + co, p.pre_b + co, p.pre_silu);
                                    }
                                }
                                const int hp = (tid >> 2) + RPP * i;
                                *reinterpret_cast<uint4*>(halo + hp * PITCH + piece * 16) = u;
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();

        if constexpr (PIPE && M16) {
            // 16 MFMAs (16x16x32) per tap in four phases of 2 x 2 tiles: (a01,F) (a01,S) (a23,S) (a23,F), where F / S are the
            // two column-tile pairs in an order that alternates with the tap.  Five 2-fragment register sets (a01, a23, bS and
            // bF[2]); every set is read from LDS two phases (>= 128 MFMA cycles) before its first use:
            //   phase 0 reads a23 (phase 2) | 1 reads the next tap's F | 2 reads the next tap's a01 | 3 reads the next tap's S.
            static_assert(MT == 2, "M16 tap schedule: 64 couts per wave");
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
            u32x4_t a01[2], a23[2], bS[2][2], bF[2][2];
            auto rdA = [&](int slot, int tp, u32x4_t (&fa)[2]) {
                const char* w = wbuf + (size_t)slot * SLOT + a_off + tp * 32 * PITCH;
                fa[0] = *reinterpret_cast<const u32x4_t*>(w);
                fa[1] = *reinterpret_cast<const u32x4_t*>(w + PITCH);
            };
            // stride 1, no upsampling (M16): a tap shifts every lane by the same (kd * IH * IW + kh * IW + kw) rows - a scalar
            // added to one per-lane base per column pair, instead of 27 x 2 per-lane offsets kept in VGPRs across the chunk loop
            const int bbase[2] = {offd[0] + offh[0][0] + offw[0][0], offd[1] + offh[1][0] + offw[1][0]};
            // (opaque per chunk: computed by the scalar ALU at each use - hoisted out of the chunk loop the 27 offsets and the 27
            //  weight bases overflow the SGPR file and come back as v_readlane per use)
            int ihs = p.IH, iws = p.IW;
            asm volatile("" : "+s"(ihs), "+s"(iws));
            auto rdB = [&](int tap, int jp, u32x4_t (&fb)[2]) {
                const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
                const char* b = halo + (bbase[jp] + ((kd * ihs + kh) * iws + kw) * PITCH);
                fb[0] = *reinterpret_cast<const u32x4_t*>(b);
                fb[1] = *reinterpret_cast<const u32x4_t*>(b + PITCH);
            };
            auto mm = [&](const u32x4_t (&fa)[2], const u32x4_t (&fb)[2], int tp, int jp) {   // tp, jp: compile-time after unrolling
                f32x16_t& c = acc[tp][jp];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int q = 4 * (2 * tt + jj);
                        f32x4_t v = {c[q], c[q + 1], c[q + 2], c[q + 3]};
                        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[tt]), __builtin_bit_cast(bf16x8_t, fb[jj]),
                                                                    v, 0, 0, 0);
                        c[q] = v[0]; c[q + 1] = v[1]; c[q + 2] = v[2]; c[q + 3] = v[3];
                    }
            };
            rdA(0, 0, a01);
            rdB(0, 0, bF[0]);
            rdB(0, 1, bS[0]);
            // MFMA intrinsics carry no chain: instruction selection gathers a tap's 16 MFMAs and the scheduler then bursts the
            // fragment reads right before the barrier's lgkmcnt(0).  An empty volatile asm that passes the fragments a group is
            // about to use pins the order: the group's MFMAs come after it (data), the reads issued before it stay before it
            // (memory clobber) - so every read is issued two groups (>= 128 MFMA cycles) ahead of its first use.
#define RHO_PHASE() __builtin_amdgcn_sched_barrier(0)
#define RHO_FENCE(A_, B_)                                                                                             \
    do {                                                                                                              \
        asm volatile("" : "+v"(A_[0]), "+v"(A_[1]), "+v"(B_[0]), "+v"(B_[1]) : : "memory");                           \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    } while (0)
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                RHO_LOAD_WP((st + LD) % 3, st);
                if constexpr (HPF) {
                    if (st == TPF) {
                        __builtin_amdgcn_sched_barrier(0);     // keep the halo loads younger than the weight fetch above
                        RHO_HALO_LOAD(min(ck + 1, nck - 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                const int slot = st % RS, nslot = (st + 1) % RS;
                const bool more = st + 1 < NS;                 // next chunk's first tap: after its halo is staged
                const int jf = st & 1, js = jf ^ 1;            // column-tile pair of F / S in this tap (swapped in the next)
                const int fc = st & 1, fn = fc ^ 1;            // fragment buffers of this / the next tap
                // group 0: this tap's second cout pair
                rdA(slot, 1, a23);
                RHO_FENCE(a01, bF[fc]);
                mm(a01, bF[fc], 0, jf);
                // group 1
                if (more) rdB(st + 1, js, bF[fn]);
                RHO_FENCE(a01, bS[fc]);
                mm(a01, bS[fc], 0, js);
                // group 2: the last fragment reads of the tap, two groups (>= 128 MFMA cycles) before a barrier's lgkmcnt(0)
                if (more) rdA(nslot, 0, a01);
                if (more) rdB(st + 1, jf, bS[fn]);
                RHO_FENCE(a23, bS[fc]);
                mm(a23, bS[fc], 1, js);
                // group 3: the LDS weight store (fetched two taps ago; late in the tap its vmcnt wait is free)
                RHO_STORE_W((st + DS) % RS, (st + DS) % 3);
                RHO_FENCE(a23, bF[fc]);
                mm(a23, bF[fc], 1, jf);
                if (st % GB == GB - 1) __syncthreads();
                RHO_PHASE();
            }
        } else if constexpr (PIPE) {
            // fragments: X = k-half 0 of the current step (already in flight), Y = k-half 1
            uint4 xa[MT], xb[2], ya[MT], yb[2];
            auto frag_reads = [&](int tap, int slot, int s_, uint4 (&fa)[MT], uint4 (&fb)[2]) {
                const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
                const char* wcur = wbuf + (size_t)slot * SLOT + a_off + 32 * s_;
                const int dtap = kd * p.IH * p.IW * PITCH + 32 * s_;
                fb[0] = *reinterpret_cast<const uint4*>(halo + (dtap + offd[0] + offh[0][kh] + offw[0][kw]));
                fb[1] = *reinterpret_cast<const uint4*>(halo + (dtap + offd[1] + offh[1][kh] + offw[1][kw]));
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) fa[mi] = *reinterpret_cast<const uint4*>(wcur + mi * 32 * PITCH);
            };
            frag_reads(0, 0, 0, xa, xb);
#pragma unroll
            for (int st = 0; st < NS; ++st) {
                {   // step st + LD -> register set (st + LD) % 3 (its previous tile went to LDS at the end of step st - 1);
                    // unconditional and clamped, so the loads in flight are a static count (counted vmcnt, no drain)
                    const int nst = (st + LD) % NS;
                    const int nckk = min(ck + (st + LD) / NS, nck - 1);
                    RHO_LOAD_W((st + LD) % 3, nckk, nst);
                }
                if constexpr (HPF) {
                    if (st == TPF) {
                        __builtin_amdgcn_sched_barrier(0);     // keep the halo loads younger than the weight fetch above
                        RHO_HALO_LOAD(min(ck + 1, nck - 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                frag_reads(st, st % RS, 1, ya, yb);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    mma_step<T>(xa[mi], xb[0], acc[mi][0]);
                    mma_step<T>(xa[mi], xb[1], acc[mi][1]);
                }
                if (st + 1 < NS) frag_reads(st + 1, (st + 1) % RS, 0, xa, xb);  // next chunk's first step: after its halo is staged
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    mma_step<T>(ya[mi], yb[0], acc[mi][0]);
                    mma_step<T>(ya[mi], yb[1], acc[mi][1]);
                }
                RHO_STORE_W((st + DS) % RS, (st + DS) % 3);    // step st + DS (slot last read in step st + DS - RS)
                if (st % GB == GB - 1) __syncthreads();
            }
        } else {
            // ---- all taps out of the resident tile, G taps per barrier step
    #pragma unroll
            for (int st = 0; st < NS; ++st) {
                const bool has_next = !(ck == nck - 1 && st == NS - 1);
                // step q = ck*NS + st: its register set (st % PD) went to LDS one step ago -> refill it with step q + PD
                // now, so the fetch has PD steps of MFMA work to hide under; at the end of this step, step q + 1 (fetched
                // PD - 1 steps ago) is handed to the other LDS slot.  Unconditional (past the end it re-reads the last
                // chunk's tiles) so that the number of loads in flight is static and the compiler waits with a counted
                // vmcnt instead of draining to 0.
                {
                    const int nst = (st + PD < NS) ? st + PD : st + PD - NS;
                    const int nckk = (st + PD < NS) ? ck : min(ck + 1, nck - 1);
                    RHO_LOAD_W(st % PD, nckk, nst);
                }
                // next chunk's halo: issued AFTER this step's weight fetch (vmcnt retires in order, so the counted waits
                // for the next PD-1 steps' weights do not drain these loads) and unconditionally (static count)
                if constexpr (HPF) {
                    if (st * G == TPF) {
                        __builtin_amdgcn_sched_barrier(0);     // keep the halo loads younger than the weight fetch above
                        RHO_HALO_LOAD(min(ck + 1, nck - 1));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
    #pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int tap = st * G + g;
                    const int kd = tap / (KH * KW), kh = (tap / KW) % KH, kw = tap % KW;
                    const char* wcur = wbuf + (size_t)cur * SLOT + g * BM * PITCH + a_off;
                    const int dtap = kd * p.IH * p.IW * PITCH;
                    const char* b0p = halo + (dtap + offd[0] + offh[0][kh] + offw[0][kw]);
                    const char* b1p = halo + (dtap + offd[1] + offh[1][kh] + offw[1][kw]);
    #pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const uint4 b0 = *reinterpret_cast<const uint4*>(b0p + 32 * s);
                        const uint4 b1 = *reinterpret_cast<const uint4*>(b1p + 32 * s);
    #pragma unroll
                        for (int mi = 0; mi < MT; ++mi) {
                            const uint4 a = *reinterpret_cast<const uint4*>(wcur + mi * 32 * PITCH + 32 * s);
                            mma_step<T>(a, b0, acc[mi][0]);
                            mma_step<T>(a, b1, acc[mi][1]);
                        }
                    }
                }
                if (has_next) RHO_STORE_W(cur ^ 1, (st + 1) % PD);
                __syncthreads();
                cur ^= 1;
            }
            }
    }

    } else {
        // ---- 1x1x1: a GEMM over channel chunks.  There is no halo and one tap per chunk, so the generic path (stage a
        // chunk, barrier, 8 MFMAs, barrier) exposed one global-load latency per chunk (0.88 ms for the 512 -> 1536 qkv
        // projection, 0.2 TFLOP).  Here chunk c + DX is fetched into a register ring while chunk c runs, activations and
        // weights are double-buffered in LDS, and one barrier per chunk orders both the hand-over and the slot reuse.
        constexpr int XS = (256 + RPP - 1) / RPP;          // activation rows per thread (the tile is 256 linear positions)
        constexpr int DX = (NW == 8) ? 4 : 2;              // chunks in flight (4-wave variants: two workgroups per CU share the registers)
        constexpr int HSLOT = 256 * PITCH;
        char* const wb2 = smem + 2 * HSLOT;                // [2][BM rows]
        uint4 xq[DX][XS], wx0[DX], wx1[DX];
#pragma unroll
        for (int i = 0; i < DX; ++i) wx0[i] = wx1[i] = make_uint4(0u, 0u, 0u, 0u);
        auto issue = [&](auto SET, int ckk) {
            constexpr int S_ = decltype(SET)::value;
            const int cc = min(ckk, nck - 1);              // clamped: static number of loads in flight (counted vmcnt)
            const int c = (ck_lo + cc) * CK;
            const char* src;
            int cs, csrc;
            if (c < p.c1) { src = p.x1; cs = p.c1; csrc = c; } else { src = p.x2; cs = p.c2; csrc = c - p.c1; }
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                const int ps = spos[i] > 0 ? spos[i] : 0;
                xq[S_][i] = *reinterpret_cast<const uint4*>(src + ((size_t)ps * cs + csrc) * sizeof(T) + piece * 16);
            }
            const char* ws = w_src0 + (size_t)cc * 64;
            if (w_active) wx0[S_] = *reinterpret_cast<const uint4*>(ws + w_voff);
            if constexpr (WROWS == 2) wx1[S_] = *reinterpret_cast<const uint4*>(ws + RPP * wrow_bytes + w_voff);
        };
        // The folded GroupNorm affine of the prologue: fetched from global memory inside the chunk loop it would be the
        // youngest load in flight at every hand-over, and waiting for it (vmcnt is in-order) drains the whole register
        // ring: one exposed memory latency per chunk (qkv projection: 0.8 ms at 260 TFLOP/s).  A tile that lies in one
        // sample (block-uniform test) stages that sample's coefficients in LDS once; tiles straddling samples keep the
        // global fetch.
        float* const coef = reinterpret_cast<float*>(smem + 2 * HSLOT + 2 * BM * PITCH);   // [2][nck * CK]
        const bool has_pre = p.pre_a != nullptr;
        bool coef_lds = false;
        if (has_pre) {
            const int pos0 = ((n * p.D + gd_base) * p.Hs + gh_base) * p.Ws + gw_base;     // the tile's first position
            const int smp_ref = (int)((long long)pos0 / p.S_in);
            bool mine = !(p.up_h | p.up_w | p.zs_h | p.zs_w) && p.sh == 1 && p.sw == 1;
#pragma unroll
            for (int i = 0; i < XS; ++i)
                if (spos[i] >= 0 && ssmp[i] != smp_ref) mine = false;
            coef_lds = __syncthreads_and(mine ? 1 : 0) != 0;
            if (coef_lds) {
                const int cpad = nck * CK, cbase = ck_lo * CK;          // (this launch's chunks: all of them unless k-split)
                for (int i = tid; i < cpad; i += NTHR) {
                    const bool in = cbase + i < p.cin;
                    coef[i] = in ? p.pre_a[(size_t)smp_ref * p.cin + cbase + i] : 0.0f;
                    coef[cpad + i] = in ? p.pre_b[(size_t)smp_ref * p.cin + cbase + i] : 0.0f;
                }
                __syncthreads();
            }
        }
        auto land = [&](auto SET, int ck, auto LDSCOEF) {   // registers -> LDS slot ck & 1 (prologue applied)
            constexpr int S_ = decltype(SET)::value;
            const int c = ck * CK;
            char* const hs = smem + (ck & 1) * HSLOT;
#pragma unroll
            for (int i = 0; i < XS; ++i) {
                if (spos[i] != -2) {
                    uint4 u = spos[i] >= 0 ? xq[S_][i] : make_uint4(0u, 0u, 0u, 0u);
                    if (has_pre && spos[i] >= 0) {
                        if constexpr (decltype(LDSCOEF)::value) {
                            const float* const ca = coef + c + piece * PE;
                            u = apply_pre<T>(u, ca, ca + nck * CK, p.pre_silu);
                        } else {
                            const size_t co = (size_t)ssmp[i] * p.cin + ck_lo * CK + c + piece * PE;
                            u = apply_pre<T>(u, p.pre_a + co, p.pre_b + co, p.pre_silu);
                        }
                    }
                    const int hp = (tid >> 2) + RPP * i;
                    *reinterpret_cast<uint4*>(hs + hp * PITCH + piece * 16) = u;
                }
            }
            char* const wd = wb2 + (ck & 1) * BM * PITCH + w_dst0;
            if (w_active) *reinterpret_cast<uint4*>(wd) = wx0[S_];
            if constexpr (WROWS == 2) *reinterpret_cast<uint4*>(wd + RPP * PITCH) = wx1[S_];
        };
        auto compute = [&](int ck) {
            const char* const hs = smem + (ck & 1) * HSLOT;
            const char* const wcur = wb2 + (ck & 1) * BM * PITCH + a_off;
            const char* b0p = hs + (offd[0] + offh[0][0] + offw[0][0]);
            const char* b1p = hs + (offd[1] + offh[1][0] + offw[1][0]);
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
                const uint4 b0 = *reinterpret_cast<const uint4*>(b0p + 32 * s_);
                const uint4 b1 = *reinterpret_cast<const uint4*>(b1p + 32 * s_);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const uint4 a = *reinterpret_cast<const uint4*>(wcur + mi * 32 * PITCH + 32 * s_);
                    mma_step<T>(a, b0, acc[mi][0]);
                    mma_step<T>(a, b1, acc[mi][1]);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        issue(I0{}, 0); issue(I1{}, 1);
        if constexpr (DX == 4) { issue(I2{}, 2); issue(I3{}, 3); }
        // Groups of DX chunks run without a branch (same number of loads in flight on every path: the hand-over waits are
        // counted, `vmcnt(9)`-style, instead of draining); the last nck % DX chunks have nothing left to fetch.
#define RHO_ONE_STEP(K_, ISET, LC)                                                                  \
        land(ISET{}, ck0 + K_, LC{});                                                               \
        __syncthreads();                                                                            \
        issue(ISET{}, ck0 + K_ + DX);                                                               \
        compute(ck0 + K_);
#define RHO_ONE_TAIL(K_, ISET, LC)                                                                  \
        if (ck0 + K_ < nck) {                                                                       \
            land(ISET{}, ck0 + K_, LC{});                                                           \
            __syncthreads();                                                                        \
            compute(ck0 + K_);                                                                      \
        }
        if (!has_pre || coef_lds) {
            int ck0 = 0;
            for (; ck0 + DX <= nck; ck0 += DX) {
                RHO_ONE_STEP(0, I0, std::true_type)
                RHO_ONE_STEP(1, I1, std::true_type)
                if constexpr (DX == 4) {
                    RHO_ONE_STEP(2, I2, std::true_type)
                    RHO_ONE_STEP(3, I3, std::true_type)
                }
            }
            RHO_ONE_TAIL(0, I0, std::true_type)
            if constexpr (DX == 4) {
                RHO_ONE_TAIL(1, I1, std::true_type)
                RHO_ONE_TAIL(2, I2, std::true_type)
            }
        } else {
            int ck0 = 0;
            for (; ck0 + DX <= nck; ck0 += DX) {
                RHO_ONE_STEP(0, I0, std::false_type)
                RHO_ONE_STEP(1, I1, std::false_type)
                if constexpr (DX == 4) {
                    RHO_ONE_STEP(2, I2, std::false_type)
                    RHO_ONE_STEP(3, I3, std::false_type)
                }
            }
            RHO_ONE_TAIL(0, I0, std::false_type)
            if constexpr (DX == 4) {
                RHO_ONE_TAIL(1, I1, std::false_type)
                RHO_ONE_TAIL(2, I2, std::false_type)
            }
        }
#undef RHO_ONE_STEP
#undef RHO_ONE_TAIL
    }
#undef RHO_PHASE
#undef RHO_FENCE
#undef RHO_LOAD_W
#undef RHO_LOAD_WP
#undef RHO_STORE_W
#undef RHO_HALO_LOAD
    // The epilogue reads its parameters (output / residual / bias / statistics pointers, output geometry) from the kernel-argument
    // segment again instead of from `p`: held in SGPRs across the tap loop they pushed the kernel past the scalar register file
    // (120 spills to VGPR lanes: ~300 v_readlane per tile in the set-up and 4 per epilogue item, all VALU issue slots).  The asm
    // keeps the reloads below the loop.
    const __attribute__((address_space(4))) ConvK* qp = (const __attribute__((address_space(4))) ConvK*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(qp));
    const __attribute__((address_space(4))) ConvK& q = *qp;
    // ---- epilogue: lane holds, for position column (lane&31) of tile j, channels
    //      co0 + 32*mi + 8*rg + 4*half + {0,1,2,3}  in acc[mi][j][4*rg + {0..3}]
    if (splitk) {
        // k-split launch: the raw fp32 partial sums of this split go to its slab ([launch positions][coutp]; the host only splits
        // launches whose outputs are all channels-last); bias, residuals and the rounding happen once, in k_splitk_reduce
        float* const slab = q.slab + (size_t)blockIdx.z * q.slab_stride;
#pragma unroll
        for (int jx = 0; jx < (M16 ? 4 : 2); ++jx) {
            const int pp = M16 ? tile_position16(wpos * 4 + jx, lane & 15, q.TW, q.pair_lg) : tile_position(wpos * 2 + jx, lane & 31, q.TW, q.pair_lg);
            const int pw = pp & (q.TW - 1);
            const int ph = (pp >> q.lgTW) & (q.TH - 1);
            const int pd = pp >> (q.lgTW + q.lgTH);
            const int od = od0 + pd, oh = oh0 + ph, ow = ow0 + pw;
            if (od >= q.Do || oh >= q.Ho || ow >= q.Wo) continue;
            const size_t L = ((size_t)od * q.Ho + oh) * q.Wo + ow;
            const int j = M16 ? (jx >> 1) : jx;
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
                for (int rx = 0; rx < (M16 ? 2 : 4); ++rx) {
                    const int rg = M16 ? 2 * rx + (jx & 1) : rx;
                    const int co = M16 ? co0 + wco * (BM / WCO) + 16 * (2 * mi + rx) + 4 * (lane >> 4)
                                       : co0 + wco * (BM / WCO) + mi * 32 + rg * 8 + half * 4;
                    *reinterpret_cast<float4*>(slab + L * q.coutp + co) =
                        make_float4(acc[mi][j][rg * 4 + 0], acc[mi][j][rg * 4 + 1], acc[mi][j][rg * 4 + 2], acc[mi][j][rg * 4 + 3]);
                }
            }
        }
        return;
    }
    // (round 4: the second channels-last region of a launch - the gradient of the second source of a concatenated input - takes the
    //  same transposed path when its width is whole 16-byte pieces: the fused GroupNorm-backward apply needs both regions here)
    const bool reg2 = q.y2_cl && co0 >= q.split && ((q.cout - q.split) % PE == 0);
    const bool cl_region = (co0 < q.split) || reg2;
    if (cl_region) {
        const int rw = reg2 ? q.cout - q.split : q.split;          // row width of this region's tensor
        const int rc0 = reg2 ? co0 - q.split : co0;                // first channel of this tile inside it
        char* const ry = reg2 ? q.y2 : q.y;
        const char* const rrp = reg2 ? q.res2 : q.res;
        const float* const radd = reg2 ? nullptr : q.res_add;
        // Channels-last output (every block of this launch region: BM divides split).  The accumulator layout gives a
        // lane 4 channels of one position, i.e. 8-byte stores (and residual loads) scattered over 32 rows per
        // instruction.  Transpose through LDS instead (the halo / weight space is free now): fp32 rows of BM channels,
        // 128 positions per pass, then every lane moves one 16-byte piece and the 16-byte pieces of a row are
        // consecutive lanes.  (Probe: without any epilogue the 64-cout layers run 23 % faster; the scattered form
        // also wrote 1.36x the algorithmic bytes.)
        constexpr int ROWB = BM * 4 + 16;                 // odd number of 16-byte slots: conflict-free column writes
        constexpr int PPR = BM / PE;                      // 16-byte output pieces per row
        static_assert(NTHR % PPR == 0, "a thread keeps one channel piece across its rows (statistics accumulators)");
        char* const stg = smem;
        float ssum[PE], ssq[PE];                          // GroupNorm partial sums of this thread's channel piece
#pragma unroll
        for (int e = 0; e < PE; ++e) ssum[e] = ssq[e] = 0.0f;
        {
            __syncthreads();                              // tap loop done with the LDS bytes reused here
            {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            // M16: group rg of acc[mi][j] = cout tile 2 * mi + (rg >> 1), column tile 2 * j + (rg & 1)
                            const int lr = M16 ? wpos * 64 + (2 * j + (rg & 1)) * 16 + (lane & 15) : wpos * 64 + j * 32 + (lane & 31);
                            const int cl = M16 ? wco * (BM / WCO) + 16 * (2 * mi + (rg >> 1)) + 4 * (lane >> 4)
                                               : wco * (BM / WCO) + mi * 32 + rg * 8 + half * 4;
                            *reinterpret_cast<float4*>(stg + lr * ROWB + cl * 4) =
                                make_float4(acc[mi][j][rg * 4 + 0], acc[mi][j][rg * 4 + 1], acc[mi][j][rg * 4 + 2], acc[mi][j][rg * 4 + 3]);
                        }
                }
            }
            // Item k of this thread: row tid / PPR + k * (NTHR / PPR), 16-byte channel piece tid % PPR.  Output offsets and
            // the residual loads are issued BEFORE the barrier that publishes the staged rows: their latency runs under the
            // barrier and the LDS reads instead of after them (one exposed global latency per tile otherwise).
            constexpr int NIT = 256 * PPR / NTHR;
            static_assert(256 * PPR % NTHR == 0, "whole items per thread");
            int eoff[NIT];                                   // output position (the host guarantees < 2^31 positions), -1 = outside
            uint4 rres[GNA ? 1 : NIT];
            uint4 gx[NIT];                                  // gnb: this thread's piece of the forward input at its output positions
            float bia[PE];                                  // bias (+ the per-sample additive term, 3-D: one sample per tile)
            float gna[PE], gnb[PE];                         // gnb: folded affine of the forward prologue, this thread's channels
            const bool gnb_on = q.gnb_x1 != nullptr;
            // gna: the GroupNorm backward apply of this thread's channels rides in the store loop (rho_conv_desc.gna_*)
            const bool gna_on = GNA && q.gna_g != nullptr;
            uint4 gg[GNA ? NIT : 1];
            float gca[GNA ? PE : 1], gcp[GNA ? PE : 1], gcq[GNA ? PE : 1];
            {
                const int piece = tid % PPR;
                const int gco = co0 + piece * PE;           // the piece lies in one concat source (widths are multiples of 32)
                const bool g_first = gco < q.gnb_c1;
                const char* const gsrc = g_first ? q.gnb_x1 : q.gnb_x2;
                // (statistics: one tensor of `split` channels; apply: the concat of both regions, cout channels)
                const int gC = gna_on ? q.cout : q.split;
                const int gcs = g_first ? q.gnb_c1 : gC - q.gnb_c1, gch = g_first ? gco : gco - q.gnb_c1;
                if (gnb_on) {
                    const int nsg = n + bt / q.tps;         // the tile's sample (statistics are only fused for one-sample tiles)
#pragma unroll
                    for (int e = 0; e < PE; ++e) {
                        gna[e] = q.gnb_a[(size_t)nsg * gC + gco + e];
                        gnb[e] = q.gnb_b[(size_t)nsg * gC + gco + e];
                    }
                    if constexpr (GNA) {
                        const int cpg = gC / 32;
#pragma unroll
                        for (int e = 0; e < PE; ++e) {
                            const int ce = (gco + e < gC) ? gco + e : gC - 1;        // (padding pieces of the last tile: never stored)
                            gca[e] = q.gna_cA[(size_t)nsg * gC + ce];
                            gcp[e] = q.gna_cP[(size_t)nsg * 32 + ce / cpg];
                            gcq[e] = q.gna_cQ[(size_t)nsg * 32 + ce / cpg];
                        }
                    }
                }
#pragma unroll
                for (int q4 = 0; q4 < PE / 4; ++q4) {
                    const float4 b4 = *reinterpret_cast<const float4*>(q.bias + co0 + piece * PE + q4 * 4);
                    bia[q4 * 4 + 0] = b4.x; bia[q4 * 4 + 1] = b4.y; bia[q4 * 4 + 2] = b4.z; bia[q4 * 4 + 3] = b4.w;
                    if constexpr (FSK) {
                        if (q.sk_bias != nullptr) {          // the folded skip convolution's own bias
                            const float4 s4 = *reinterpret_cast<const float4*>(q.sk_bias + co0 + piece * PE + q4 * 4);
                            bia[q4 * 4 + 0] += s4.x; bia[q4 * 4 + 1] += s4.y; bia[q4 * 4 + 2] += s4.z; bia[q4 * 4 + 3] += s4.w;
                        }
                    }
                    if constexpr (KD == 3) {
                        if (radd != nullptr) {
                            const float4 e = *reinterpret_cast<const float4*>(radd + (long long)n * q.res_add_stride + co0 + piece * PE + q4 * 4);
                            bia[q4 * 4 + 0] += e.x; bia[q4 * 4 + 1] += e.y; bia[q4 * 4 + 2] += e.z; bia[q4 * 4 + 3] += e.w;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    const int lr = tid / PPR + k * (NTHR / PPR);
                    const int pp = M16 ? tile_position16((lr >> 6) * 4 + ((lr >> 4) & 3), lr & 15, q.TW, q.pair_lg)
                                       : tile_position((lr >> 6) * 2 + ((lr >> 5) & 1), lr & 31, q.TW, q.pair_lg);
                    const int pw = pp & (q.TW - 1);
                    const int ph = (pp >> q.lgTW) & (q.TH - 1);
                    const int pd = pp >> (q.lgTW + q.lgTH);
                    const int od = od0 + pd, oh = oh0 + ph, ow = ow0 + pw;
                    // (the last tile of the second region may reach past its tensor)
                    const bool ok = od < q.Do && oh < q.Ho && ow < q.Wo && (rc0 + piece * PE + PE <= rw);
                    const int L = ((n * q.Do + od) * q.Ho_out + (oh * q.oy_mul + q.oy_add)) * q.Wo_out + (ow * q.ox_mul + q.ox_add);
                    eoff[k] = ok ? L : -1;
                    if constexpr (!GNA) {
                        rres[k] = make_uint4(0u, 0u, 0u, 0u);
                        if (ok && rrp != nullptr)   // (32 x 32 -> 64-bit multiply: one v_mad_u64_u32 instead of a 64 x 64 sequence)
                            rres[k] = *reinterpret_cast<const uint4*>(rrp + ((size_t)(unsigned)L * (unsigned)rw + (unsigned)(rc0 + piece * PE)) * sizeof(T));
                    }
                    gx[k] = make_uint4(0u, 0u, 0u, 0u);
                    if (ok && gnb_on) gx[k] = *reinterpret_cast<const uint4*>(gsrc + ((size_t)(unsigned)L * (unsigned)gcs + (unsigned)gch) * sizeof(T));
                    if constexpr (GNA) {
                        gg[k] = make_uint4(0u, 0u, 0u, 0u);
                        if (ok) gg[k] = *reinterpret_cast<const uint4*>(q.gna_g + ((size_t)(unsigned)L * (unsigned)gC + (unsigned)gco) * sizeof(T));
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < NIT; ++k) {
                if (eoff[k] < 0) continue;
                const int lr = tid / PPR + k * (NTHR / PPR), piece = tid % PPR;
                const int L = eoff[k];
                const int co = co0 + piece * PE;
                float v[PE];
#pragma unroll
                for (int q4 = 0; q4 < PE / 4; ++q4) {
                    const float4 a4 = *reinterpret_cast<const float4*>(stg + lr * ROWB + (piece * PE + q4 * 4) * 4);
                    v[q4 * 4 + 0] = a4.x + bia[q4 * 4 + 0]; v[q4 * 4 + 1] = a4.y + bia[q4 * 4 + 1];
                    v[q4 * 4 + 2] = a4.z + bia[q4 * 4 + 2]; v[q4 * 4 + 3] = a4.w + bia[q4 * 4 + 3];
                }
                if (KD != 3 && radd != nullptr) {
                    const long long ns = (long long)L / q.S_out;
#pragma unroll
                    for (int q4 = 0; q4 < PE / 4; ++q4) {
                        const float4 e = *reinterpret_cast<const float4*>(radd + ns * q.res_add_stride + co + q4 * 4);
                        v[q4 * 4 + 0] += e.x; v[q4 * 4 + 1] += e.y; v[q4 * 4 + 2] += e.z; v[q4 * 4 + 3] += e.w;
                    }
                }
                const size_t eo = (size_t)(unsigned)L * (unsigned)rw + (unsigned)(rc0 + piece * PE);
                if constexpr (GNA) {
                    // + cA * (g * act'(a x0 + b)) + cQ * x0 + cP: rho_gn_bwd_apply's term, from the operands that pass read
                    float xv[PE], gv[PE];
                    if constexpr (sizeof(T) == 2) {
                        const uint4 a_ = gx[k], b_ = gg[k];
                        xv[0] = __uint_as_float(a_.x << 16); xv[1] = __uint_as_float(a_.x & 0xFFFF0000u);
                        xv[2] = __uint_as_float(a_.y << 16); xv[3] = __uint_as_float(a_.y & 0xFFFF0000u);
                        xv[4] = __uint_as_float(a_.z << 16); xv[5] = __uint_as_float(a_.z & 0xFFFF0000u);
                        xv[6] = __uint_as_float(a_.w << 16); xv[7] = __uint_as_float(a_.w & 0xFFFF0000u);
                        gv[0] = __uint_as_float(b_.x << 16); gv[1] = __uint_as_float(b_.x & 0xFFFF0000u);
                        gv[2] = __uint_as_float(b_.y << 16); gv[3] = __uint_as_float(b_.y & 0xFFFF0000u);
                        gv[4] = __uint_as_float(b_.z << 16); gv[5] = __uint_as_float(b_.z & 0xFFFF0000u);
                        gv[6] = __uint_as_float(b_.w << 16); gv[7] = __uint_as_float(b_.w & 0xFFFF0000u);
                    } else {
                        const uint4 a_ = gx[k], b_ = gg[k];
                        xv[0] = __uint_as_float(a_.x); xv[1] = __uint_as_float(a_.y); xv[2] = __uint_as_float(a_.z); xv[3] = __uint_as_float(a_.w);
                        gv[0] = __uint_as_float(b_.x); gv[1] = __uint_as_float(b_.y); gv[2] = __uint_as_float(b_.z); gv[3] = __uint_as_float(b_.w);
                    }
#pragma unroll
                    for (int e = 0; e < PE; ++e) {
                        float gq = gv[e];
                        if (q.gnb_silu) gq *= dsilu_f(fmaf(gna[e], xv[e], gnb[e]));
                        v[e] += fmaf(gca[e], gq, fmaf(gcq[e], xv[e], gcp[e]));
                    }
                }
                if constexpr (sizeof(T) == 2) {
                    if (!GNA && rrp != nullptr) {
                        const uint4 r = rres[GNA ? 0 : k];
                        v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xFFFF0000u);
                        v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xFFFF0000u);
                        v[4] += __uint_as_float(r.z << 16); v[5] += __uint_as_float(r.z & 0xFFFF0000u);
                        v[6] += __uint_as_float(r.w << 16); v[7] += __uint_as_float(r.w & 0xFFFF0000u);
                    }
                    const uint4 o = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
                    {   // streaming store: the 1 GB outputs do not fit the caches, keep L2 for the halo re-reads (whole step -0.5 %)
                        typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
                        const u32x4_nt ov = {o.x, o.y, o.z, o.w};
                        __builtin_nontemporal_store(ov, reinterpret_cast<u32x4_nt*>(ry + eo * 2));
                    }
                    if (q.stats != nullptr) {             // statistics of the values as stored (what a reader would see)
                        v[0] = __uint_as_float(o.x << 16); v[1] = __uint_as_float(o.x & 0xFFFF0000u);
                        v[2] = __uint_as_float(o.y << 16); v[3] = __uint_as_float(o.y & 0xFFFF0000u);
                        v[4] = __uint_as_float(o.z << 16); v[5] = __uint_as_float(o.z & 0xFFFF0000u);
                        v[6] = __uint_as_float(o.w << 16); v[7] = __uint_as_float(o.w & 0xFFFF0000u);
                    }
                } else {
                    if (!GNA && rrp != nullptr) {
                        const uint4 r = rres[GNA ? 0 : k];
                        v[0] += __uint_as_float(r.x); v[1] += __uint_as_float(r.y); v[2] += __uint_as_float(r.z); v[3] += __uint_as_float(r.w);
                    }
                    *reinterpret_cast<float4*>(ry + eo * 4) = make_float4(v[0], v[1], v[2], v[3]);
                }
                if (q.stats != nullptr) {
                    if (gnb_on && !gna_on) {
                        // dgrad of a conv behind GroupNorm (+SiLU): v = d act(a x + b) as stored; the norm's backward needs the
                        // per-channel sums of dz = v * act'(a x + b) and of dz * x (rho_gn_bwd_finalize, fmt 1)
                        float xv[PE];
                        if constexpr (sizeof(T) == 2) {
                            const uint4 q = gx[k];
                            xv[0] = __uint_as_float(q.x << 16); xv[1] = __uint_as_float(q.x & 0xFFFF0000u);
                            xv[2] = __uint_as_float(q.y << 16); xv[3] = __uint_as_float(q.y & 0xFFFF0000u);
                            xv[4] = __uint_as_float(q.z << 16); xv[5] = __uint_as_float(q.z & 0xFFFF0000u);
                            xv[6] = __uint_as_float(q.w << 16); xv[7] = __uint_as_float(q.w & 0xFFFF0000u);
                        } else {
                            const uint4 q = gx[k];
                            xv[0] = __uint_as_float(q.x); xv[1] = __uint_as_float(q.y); xv[2] = __uint_as_float(q.z); xv[3] = __uint_as_float(q.w);
                        }
#pragma unroll
                        for (int e = 0; e < PE; ++e) {
                            float dz = v[e];
                            if (q.gnb_silu) dz *= dsilu_f(fmaf(gna[e], xv[e], gnb[e]));
                            ssum[e] += dz;
                            ssq[e] = fmaf(dz, xv[e], ssq[e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < PE; ++e) {
                            ssum[e] += v[e];
                            ssq[e] = fmaf(v[e], v[e], ssq[e]);
                        }
                    }
                }
            }
        }
        if (q.stats != nullptr) {
            // threads tid = piece (mod PPR) hold partials of the same channels: combine through LDS in thread order
            // (fixed order => reproducible), one (channel, statistic) per finishing thread
            __syncthreads();
            float* red = reinterpret_cast<float*>(stg);                     // [NTHR][2 * PE]
#pragma unroll
            for (int e = 0; e < PE; ++e) {
                red[tid * (2 * PE) + e] = ssum[e];
                red[tid * (2 * PE) + PE + e] = ssq[e];
            }
            __syncthreads();
            const int ns = n + bt / q.tps, ts = bt % q.tps + q.stats_off;
            for (int item = tid; item < PPR * 2 * PE; item += NTHR) {
                const int piece = item / (2 * PE), e2 = item % (2 * PE);
                float accv = 0.0f;
                for (int q = 0; q < NTHR / PPR; ++q) accv += red[(q * PPR + piece) * (2 * PE) + e2];
                const int stat = e2 / PE, ch = co0 + piece * PE + (e2 % PE);
                q.stats[(((size_t)ns * q.tps + ts) * 2 + stat) * q.split + ch] = accv;
            }
        }
        return;
    }
#pragma unroll
    for (int jx = 0; jx < (M16 ? 4 : 2); ++jx) {
        const int pp = M16 ? tile_position16(wpos * 4 + jx, lane & 15, q.TW, q.pair_lg) : tile_position(wpos * 2 + jx, lane & 31, q.TW, q.pair_lg);
        const int pw = pp & (q.TW - 1);
        const int ph = (pp >> q.lgTW) & (q.TH - 1);
        const int pd = pp >> (q.lgTW + q.lgTH);
        const int od = od0 + pd, oh = oh0 + ph, ow = ow0 + pw;
        if (od >= q.Do || oh >= q.Ho || ow >= q.Wo) continue;
        const long long L = (((long long)n * q.Do + od) * q.Ho + oh) * q.Wo + ow;
        const long long ns = L / q.S_out;
        const long long ps = L - ns * q.S_out;
        const int j = M16 ? (jx >> 1) : jx;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int rx = 0; rx < (M16 ? 2 : 4); ++rx) {
                const int rg = M16 ? 2 * rx + (jx & 1) : rx;
                const int co = M16 ? co0 + wco * (BM / WCO) + 16 * (2 * mi + rx) + 4 * (lane >> 4)
                                   : co0 + wco * (BM / WCO) + mi * 32 + rg * 8 + half * 4;
                const float4 bv = *reinterpret_cast<const float4*>(q.bias + co);
                float v0 = acc[mi][j][rg * 4 + 0] + bv.x;
                float v1 = acc[mi][j][rg * 4 + 1] + bv.y;
                float v2 = acc[mi][j][rg * 4 + 2] + bv.z;
                float v3 = acc[mi][j][rg * 4 + 3] + bv.w;
                if (cl_region) {
                    if (q.res_add != nullptr) {
                        const float4 e = *reinterpret_cast<const float4*>(q.res_add + ns * q.res_add_stride + co);
                        v0 += e.x; v1 += e.y; v2 += e.z; v3 += e.w;
                    }
                    const size_t eo = (size_t)L * q.split + co;
                    if constexpr (sizeof(T) == 2) {
                        if (q.res != nullptr) {
                            const uint2 r = *reinterpret_cast<const uint2*>(q.res + eo * 2);
                            v0 += __uint_as_float(r.x << 16); v1 += __uint_as_float(r.x & 0xFFFF0000u);
                            v2 += __uint_as_float(r.y << 16); v3 += __uint_as_float(r.y & 0xFFFF0000u);
                        }
                        *reinterpret_cast<uint2*>(q.y + eo * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                    } else {
                        if (q.res != nullptr) {
                            const float4 r = *reinterpret_cast<const float4*>(q.res + eo * 4);
                            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                        }
                        *reinterpret_cast<float4*>(q.y + eo * 4) = make_float4(v0, v1, v2, v3);
                    }
                } else if (q.y2_cl) {
                    const int w2 = q.cout - q.split;
                    const size_t eo = (size_t)L * w2 + (co - q.split);
                    if (co < q.cout) {                 // widths are multiples of 4: a 4-channel piece is all in or all out
                        if constexpr (sizeof(T) == 2) {
                            if (q.res2 != nullptr) {
                                const uint2 r = *reinterpret_cast<const uint2*>(q.res2 + eo * 2);
                                v0 += __uint_as_float(r.x << 16); v1 += __uint_as_float(r.x & 0xFFFF0000u);
                                v2 += __uint_as_float(r.y << 16); v3 += __uint_as_float(r.y & 0xFFFF0000u);
                            }
                            *reinterpret_cast<uint2*>(q.y2 + eo * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
                        } else {
                            if (q.res2 != nullptr) {
                                const float4 r = *reinterpret_cast<const float4*>(q.res2 + eo * 4);
                                v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                            }
                            *reinterpret_cast<float4*>(q.y2 + eo * 4) = make_float4(v0, v1, v2, v3);
                        }
                    }
                } else {
                    const float vv[4] = {v0, v1, v2, v3};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (co + e < q.cout) {
                            const size_t eo = ((size_t)ns * (q.cout - q.split) + (co + e - q.split)) * q.S_out + ps;
                            if (q.y2_f32 || sizeof(T) == 4)
                                reinterpret_cast<float*>(q.y2)[eo] = vv[e];
                            else
                                reinterpret_cast<bf16_raw*>(q.y2)[eo] = f32_to_bf16(vv[e]);
                        }
                    }
                }
            }
        }
    }
}

struct SplitMap {          // launch grid position (od, oh, ow) -> output row ((od * Ho_out) + oh * oy_mul + oy_add) * Wo_out + ow * ox_mul + ox_add
    int Ho, Wo, Ho_out, Wo_out, oy_mul, oy_add, ox_mul, ox_add;      // (identity unless the launch is one sub-pixel phase of its output)
};
struct SplitOut {          // the two channels-last output regions of a launch: co < split -> y (+ res, res_add), else y2 (+ res2)
    int cout, coutp, split;
    const float* bias;
    const float* res_add;
    long long res_add_stride, S_out;
    const char* res;
    char* y;
    const char* res2;
    char* y2;
};
// Second half of a k-split conv: out[L][co] = round( sum_s slab[s][L][co] + bias[co] + res_add[sample(L)][co] + res[L][co] ), the
// slabs added in split order (fixed => reproducible), the same operand order as the fused epilogue after the contraction.
template <typename T>
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ slab, long long slab_stride, int ksplit, long long positions,
                                                       SplitOut o, SplitMap m) {
    const int c4 = o.coutp / 4;
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= positions * c4) return;
    const long long Ls = item / c4;                      // position in the launch's own grid = slab row
    int co = (int)(item - Ls * c4) * 4;
    if (co >= o.cout) return;                            // padding columns of the last cout tile (widths are multiples of 4)
    const size_t so = (size_t)Ls * o.coutp + co;
    float4 a = *reinterpret_cast<const float4*>(slab + so);
    for (int s = 1; s < ksplit; ++s) {
        const float4 b = *reinterpret_cast<const float4*>(slab + (size_t)s * slab_stride + so);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    long long L = Ls;
    if (m.oy_mul != 1 || m.ox_mul != 1) {
        const int ow = (int)(Ls % m.Wo);
        const long long r = Ls / m.Wo;
        const int oh = (int)(r % m.Ho);
        L = ((r / m.Ho) * m.Ho_out + (oh * m.oy_mul + m.oy_add)) * m.Wo_out + (ow * m.ox_mul + m.ox_add);
    }
    const float4 bv = *reinterpret_cast<const float4*>(o.bias + co);
    float v0 = a.x + bv.x, v1 = a.y + bv.y, v2 = a.z + bv.z, v3 = a.w + bv.w;
    const bool first = co < o.split;
    if (first && o.res_add != nullptr) {
        const float4 e = *reinterpret_cast<const float4*>(o.res_add + (L / o.S_out) * o.res_add_stride + co);
        v0 += e.x; v1 += e.y; v2 += e.z; v3 += e.w;
    }
    const int width = first ? o.split : o.cout - o.split;
    const char* const res = first ? o.res : o.res2;
    char* const y = first ? o.y : o.y2;
    if (!first) co -= o.split;
    const size_t eo = (size_t)L * width + co;
    if constexpr (sizeof(T) == 2) {
        if (res != nullptr) {
            const uint2 r = *reinterpret_cast<const uint2*>(res + eo * 2);
            v0 += __uint_as_float(r.x << 16); v1 += __uint_as_float(r.x & 0xFFFF0000u);
            v2 += __uint_as_float(r.y << 16); v3 += __uint_as_float(r.y & 0xFFFF0000u);
        }
        *reinterpret_cast<uint2*>(y + eo * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
    } else {
        if (res != nullptr) {
            const float4 r = *reinterpret_cast<const float4*>(res + eo * 4);
            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
        }
        *reinterpret_cast<float4*>(y + eo * 4) = make_float4(v0, v1, v2, v3);
    }
}

#include "conv32.h"

// ------------------------------------------------------------------------------------------ host
namespace {

using namespace rho_conv;

// Variant query (rho_conv_variant): the same dispatch as a launch, but the chosen instantiation is written as text
// instead of being launched - so the name can never disagree with what rho_conv_nd_fwd runs.
struct VariantOut {
    char* buf;
    int cap;
};
thread_local VariantOut* g_variant = nullptr;

template <typename T, int KD, int KH, int KW, int BM, int MAXP, int NW, bool M16 = false, bool FSK = false, bool GNA = false>
int launch_one(const ConvK& k, dim3 grid, size_t lds, hipStream_t st) {
    if (g_variant != nullptr) {
        snprintf(g_variant->buf, (size_t)g_variant->cap, "k_conv<%s,%d,%d,%d,BM=%d,MAXP=%d,NW=%d,M16=%d>%s%s", sizeof(T) == 2 ? "bf16" : "f32",
                 KD, KH, KW, BM, MAXP, NW, (int)M16, FSK ? "+skip" : "", GNA ? "+gn_apply" : "");
        return 0;
    }
    auto fn = k_conv<T, KD, KH, KW, BM, MAXP, NW, M16, FSK, GNA>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, grid, dim3(NW * 64), lds, st, k);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

// np = halo positions of the chosen tile.  BM = 128 runs 8 waves (slots per thread: 5 for np <= 640, else 14).
template <typename T, int KD, int KH, int KW>
int launch_bm(const ConvK& k, int BM, int np, dim3 grid, size_t lds, bool m16, hipStream_t st) {
    // sub-pixel phase kernels (2-tap axes; 3x1x1: the even-even parity of a stride-2 split): stride 1 - the small-halo variants only
    constexpr bool PHASE = (KH == 2 || KW == 2 || (KD == 3 && KH == 1 && KW == 1));
    if constexpr (PHASE) { if (np > 640) return RHO_E_SHAPE; }
    if constexpr (sizeof(T) == 2 && KD * KH * KW > 1 && (KD * KH * KW) % 3 == 0) {
        // bf16, stride 1, no upsampling, regular halo: the 16x16x32 MFMA layout (holds a higher clock under load)
        if (m16 && np <= 640) {
            if constexpr (KD == 3 && KH == 3 && KW == 3) {
                if (k.sk_w != nullptr) {                   // (conv_impl only sets it for this geometry)
                    if (BM == 128) return launch_one<T, KD, KH, KW, 128, 5, 8, true, true>(k, grid, lds, st);
                    if (BM == 64) return launch_one<T, KD, KH, KW, 64, 10, 4, true, true>(k, grid, lds, st);
                }
            }
            if (BM == 128) return launch_one<T, KD, KH, KW, 128, 5, 8, true>(k, grid, lds, st);
            if (BM == 64) return launch_one<T, KD, KH, KW, 64, 10, 4, true>(k, grid, lds, st);
        }
    }
    if (k.sk_w != nullptr) return RHO_E_ARG;               // a folded skip needs one of the two variants above
    if (k.gna_g != nullptr) {                              // GroupNorm backward apply in the epilogue: the 1x1x1 instantiations
        if constexpr (KD * KH * KW == 1) {
            if (np > 640) return RHO_E_SHAPE;
            if (BM == 128) return launch_one<T, 1, 1, 1, 128, 5, 8, false, false, true>(k, grid, lds, st);
            if (BM == 64) return launch_one<T, 1, 1, 1, 64, 10, 4, false, false, true>(k, grid, lds, st);
            return launch_one<T, 1, 1, 1, 32, 10, 4, false, false, true>(k, grid, lds, st);
        }
        return RHO_E_ARG;
    }
    if (BM == 128) {
        if (np <= 640) return launch_one<T, KD, KH, KW, 128, 5, 8>(k, grid, lds, st);
        if constexpr (!PHASE) return launch_one<T, KD, KH, KW, 128, 14, 8>(k, grid, lds, st);
    }
#define RHO_CASE(bm)                                                                   \
    if (BM == bm) {                                                                    \
        if (np <= 640) return launch_one<T, KD, KH, KW, bm, 10, 4>(k, grid, lds, st);  \
        if constexpr (!PHASE) return launch_one<T, KD, KH, KW, bm, 28, 4>(k, grid, lds, st); \
    }
    RHO_CASE(32)
    RHO_CASE(64)
#undef RHO_CASE
    return RHO_E_ARG;
}

template <typename T>
int launch_taps(const rho_conv_desc& d, const ConvK& k, int BM, int np, dim3 grid, size_t lds, bool m16, hipStream_t st) {
    if (d.kd == 3 && d.kh == 3 && d.kw == 3) return launch_bm<T, 3, 3, 3>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 1 && d.kh == 3 && d.kw == 3) return launch_bm<T, 1, 3, 3>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 1 && d.kh == 1 && d.kw == 3) return launch_bm<T, 1, 1, 3>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 1 && d.kh == 1 && d.kw == 1) return launch_bm<T, 1, 1, 1>(k, BM, np, grid, lds, m16, st);
    // sub-pixel phases of a conv behind a nearest x2 upsample (3-D / 2-D: both inner axes; 1-D: the last)
    if (d.kd == 3 && d.kh == 2 && d.kw == 2) return launch_bm<T, 3, 2, 2>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 1 && d.kh == 2 && d.kw == 2) return launch_bm<T, 1, 2, 2>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 1 && d.kh == 1 && d.kw == 2) return launch_bm<T, 1, 1, 2>(k, BM, np, grid, lds, m16, st);
    // parity split of a 3-D stride-2 conv (1- and 2-tap inner axes)
    if (d.kd == 3 && d.kh == 1 && d.kw == 1) return launch_bm<T, 3, 1, 1>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 3 && d.kh == 1 && d.kw == 2) return launch_bm<T, 3, 1, 2>(k, BM, np, grid, lds, m16, st);
    if (d.kd == 3 && d.kh == 2 && d.kw == 1) return launch_bm<T, 3, 2, 1>(k, BM, np, grid, lds, m16, st);
    return RHO_E_ARG;
}

}  // namespace

static bool m16_env_on() {
    static const bool on = !(getenv("RHO_CONV_M16") && atoi(getenv("RHO_CONV_M16")) == 0);
    return on;
}

static int conv_impl(const rho_conv_desc* dp, void* stream, int64_t* stats_tiles, int64_t* ws_want = nullptr) {
    if (!dp) return RHO_E_ARG;
    const rho_conv_desc& d = *dp;
    if (!d.x1 || !d.w || !d.bias) return RHO_E_ARG;
    if (d.dtype != RHO_F32 && d.dtype != RHO_BF16) return RHO_E_ARG;
    const int CK = d.dtype == RHO_BF16 ? 32 : 16;
    const int c2 = d.x2 ? d.c2 : 0;
    const int cin = d.c1 + c2;
    if (d.c1 <= 0 || d.c1 % CK || c2 % CK) return RHO_E_ALIGN;
    if (d.cout <= 0 || d.coutp < d.cout || d.coutp % 32 || d.split < 0 || d.split > d.cout || d.split % 32) return RHO_E_ARG;
    if (d.split > 0 && !d.y) return RHO_E_ARG;
    if (d.split < d.cout && !d.y2) return RHO_E_ARG;
    if (d.split < d.cout && d.split > 0 && d.coutp != d.cout && !d.y2_cl) return RHO_E_ARG;  // mixed layouts need exact tiling
    if (d.y2_cl && ((d.cout - d.split) % 4 != 0)) return RHO_E_ALIGN;
    if ((d.zs_h || d.zs_w) && (d.up_h || d.up_w || d.sh != 1 || d.sw != 1 || d.out_h <= 0 || d.out_w <= 0)) return RHO_E_ARG;
    if (d.split == d.cout && d.coutp != d.cout) return RHO_E_ARG;               // channels-last rows are not padded
    if ((d.sh != 1 && d.sh != 2) || (d.sw != 1 && d.sw != 2)) return RHO_E_ARG;
    if ((d.up_h && d.sh != 1) || (d.up_w && d.sw != 1)) return RHO_E_ARG;
    if ((d.up_h && d.kh != 3) || (d.up_w && d.kw != 3)) return RHO_E_ARG;
    if (d.n <= 0 || d.d <= 0 || d.h <= 0 || d.w_ <= 0) return RHO_E_ARG;
    // sub-pixel phases: a 2-tap axis at the source resolution, stride 1, the output rows of one parity
    if (d.ph_h < 0 || d.ph_h > 2 || d.ph_w < 0 || d.ph_w > 2) return RHO_E_ARG;
    if ((d.ph_h && ((d.kh != 2 && d.kh != 1) || d.sh != 1 || d.up_h || d.zs_h)) || (d.ph_w && ((d.kw != 2 && d.kw != 1) || d.sw != 1 || d.up_w || d.zs_w)))
        return RHO_E_ARG;
    if ((!d.ph_h && !d.phd_h && d.kh == 2) || (!d.ph_w && !d.phd_w && d.kw == 2)) return RHO_E_ARG;
    if ((d.ph_h || d.ph_w) && (d.split != d.cout || d.kd == 2)) return RHO_E_ARG;      // channels-last outputs only
    if (d.phd_h < 0 || d.phd_h > 2 || d.phd_w < 0 || d.phd_w > 2) return RHO_E_ARG;
    if ((d.phd_h && ((d.kh != 2 && d.kh != 1) || d.sh != 1 || d.up_h || d.zs_h || d.ph_h)) ||
        (d.phd_w && ((d.kw != 2 && d.kw != 1) || d.sw != 1 || d.up_w || d.zs_w || d.ph_w)))
        return RHO_E_ARG;
    if ((d.ph_h || d.phd_h || d.ph_w || d.phd_w) && d.kd * d.kh * d.kw == 1) return RHO_E_ARG;   // (the 1x1x1 path merges the axes)
    if ((d.phd_h || d.phd_w) && (d.kd == 2 || d.pre_a)) return RHO_E_ARG;

    // the 32 -> 32 channel 3x3x3 layers: the persistent register-resident-weights kernel (conv32.h)
    if (conv32_applies(d)) {
        if (stats_tiles) { *stats_tiles = conv32_wps(d); return 0; }
        if (ws_want) { *ws_want = 0; return 0; }
        if (g_variant != nullptr) { snprintf(g_variant->buf, (size_t)g_variant->cap, "k_conv32<bf16>"); return 0; }
        if (d.stats && d.split != d.cout) return RHO_E_ARG;
        return launch_conv32(d, as_stream(stream));
    }

    // output extents per sample (padding k/2).  Zero-stuffed input (dgrad of a stride-2 conv): the
    // virtual input and the output both have the forward conv's input extent out_h / out_w.
    const int hv = d.zs_h ? d.out_h : d.h, wv = d.zs_w ? d.out_w : d.w_;   // virtual input extents
    // (phases: the launch's own output grid is the source grid; it lands on every second row / column of the real output)
    const int ho = (d.ph_h || d.phd_h) ? d.h : d.up_h ? d.h * 2 : (hv + 2 * (d.kh / 2) - d.kh) / d.sh + 1;
    const int wo = (d.ph_w || d.phd_w) ? d.w_ : d.up_w ? d.w_ * 2 : (wv + 2 * (d.kw / 2) - d.kw) / d.sw + 1;
    const int ho_out = d.ph_h ? 2 * ho : ho, wo_out = d.ph_w ? 2 * wo : wo;
    const int n_phase = (d.ph_h ? 2 : 1) * (d.ph_w ? 2 : 1);
    const int phase_idx = (d.ph_h ? d.ph_h - 1 : 0) * (d.ph_w ? 2 : 1) + (d.ph_w ? d.ph_w - 1 : 0);
    const int dout = d.d;

    // axes that carry no kernel extent are merged with the batch so tiles stay full:
    //   1x1x1: everything is one long W axis;  1xkxk: depth*batch is the tile's depth axis.
    ConvK k{};
    int gridz = d.n;
    // (a 1-tap phased axis sits on the output row: the even parity of a stride-2 conv's forward / data gradient)
    k.pad_h = d.kh == 1 ? 0 : d.ph_h ? 2 - d.ph_h : d.phd_h ? d.phd_h - 1 : d.kh / 2;
    k.pad_w = d.kw == 1 ? 0 : d.ph_w ? 2 - d.ph_w : d.phd_w ? d.phd_w - 1 : d.kw / 2;
    k.iy_mul = d.phd_h ? 2 : 1; k.iy_add = d.phd_h ? d.phd_h - 1 : 0;
    k.ix_mul = d.phd_w ? 2 : 1; k.ix_add = d.phd_w ? d.phd_w - 1 : 0;
    k.oy_mul = d.ph_h ? 2 : 1; k.oy_add = d.ph_h ? d.ph_h - 1 : 0;
    k.ox_mul = d.ph_w ? 2 : 1; k.ox_add = d.ph_w ? d.ph_w - 1 : 0;
    if (d.kd == 1 && d.kh == 1 && d.kw == 1) {
        if (d.sh != 1 || d.sw != 1 || d.up_h || d.up_w) return RHO_E_ARG;
        k.D = 1; k.H = 1; k.W = d.n * d.d * d.h * d.w_;
        k.Do = 1; k.Ho = 1; k.Wo = k.W;
        gridz = 1;
    } else if (d.kd == 1) {
        k.D = d.n * d.d; k.H = hv; k.W = wv;
        k.Do = k.D; k.Ho = ho; k.Wo = wo;
        gridz = 1;
    } else {
        k.D = d.d; k.H = hv; k.W = wv;
        k.Do = dout; k.Ho = ho; k.Wo = wo;
    }
    k.Hs = k.H; k.Ws = k.W;
    if (d.zs_h) k.Hs = d.h;
    if (d.zs_w) k.Ws = d.w_;
    if (d.phd_h) k.Hs = 2 * d.h;
    if (d.phd_w) k.Ws = 2 * d.w_;
    k.zs_h = d.zs_h; k.zs_w = d.zs_w;
    k.y2_cl = d.y2_cl; k.res2 = (const char*)d.res2;
    if ((long long)d.n * d.d * d.h * d.w_ * (d.phd_h ? 2 : 1) * (d.phd_w ? 2 : 1) >= (1LL << 31) || (long long)d.n * dout * ho_out * wo_out >= (1LL << 31)) return RHO_E_SHAPE;
    k.S_in = (long long)d.d * d.h * d.w_ * (d.phd_h ? 2 : 1) * (d.phd_w ? 2 : 1);
    k.S_out = (long long)dout * ho_out * wo_out;
    k.Ho_out = (d.kd == 1 && d.kh == 1 && d.kw == 1) ? 1 : ho_out;
    k.Wo_out = (d.kd == 1 && d.kh == 1 && d.kw == 1) ? k.Wo : wo_out;

    // cout tile
    int BM = 32;
    if (d.coutp % 128 == 0 && (d.split % 128 == 0)) BM = 128;
    else if (d.coutp % 64 == 0 && (d.split % 64 == 0)) BM = 64;

    const size_t lds_cap = 160 * 1024;
    const int taps = d.kd * d.kh * d.kw;
    {
        // A/B knob: short contractions (<= N input-channel chunks) on the 4-wave 64-cout tile, two workgroups per CU, instead of the
        // 8-wave 128-cout tile whose waves all sit in set-up / epilogue at the same time (0 = off)
        static const int bm64_chunks = getenv("RHO_BM64_MAX_CHUNKS") ? atoi(getenv("RHO_BM64_MAX_CHUNKS")) : 0;
        if (BM == 128 && bm64_chunks > 0 && d.kd == 3 && taps > 1 && cin / CK <= bm64_chunks) BM = 64;
    }
    const bool m16 = d.dtype == RHO_BF16 && taps > 1 && taps % 3 == 0 && d.sh == 1 && d.sw == 1 && !d.up_h && !d.up_w;
    const int WSLOTS = (taps % 3 == 0) ? ((((BM == 128 && m16) || (RHO_GB_BM32 && BM == 32)) && RHO_GB_WIDE == 3 && taps % 9 == 0) ? 9 : 3) : 2;   // LDS weight-ring depth (matches the kernel's PIPE / RS)
    int np_cap = (int)((lds_cap - (size_t)WSLOTS * BM * PITCH) / PITCH);
    if (np_cap > 28 * 64) np_cap = 28 * 64;
    // prefer the small-halo (2 blocks / CU) configuration when it exists
    TileChoice t = choose_tile(d, k.D, k.Do, k.Ho, k.Wo, 640);
    if (!t.ok) t = choose_tile(d, k.D, k.Do, k.Ho, k.Wo, np_cap);
    if (!t.ok) return RHO_E_SHAPE;

    k.x1 = (const char*)d.x1; k.x2 = (const char*)d.x2; k.pre_a = d.pre_a; k.pre_b = d.pre_b;
    k.w = (const char*)d.w; k.bias = d.bias; k.res = (const char*)d.res; k.res_add = d.res_add;
    k.y = (char*)d.y; k.y2 = (char*)d.y2;
    k.c1 = d.c1; k.c2 = c2; k.cin = cin;
    k.cout = d.cout; k.coutp = d.coutp; k.split = d.split;
    k.res_add_stride = d.res_add_stride > 0 ? d.res_add_stride : d.split;
    k.sh = d.sh; k.sw = d.sw; k.up_h = d.up_h; k.up_w = d.up_w; k.pre_silu = d.pre_silu; k.y2_f32 = d.y2_f32;
    k.TD = t.TD; k.TH = t.TH; k.TW = t.TW; k.ID = t.ID; k.IH = t.IH; k.IW = t.IW; k.NP = t.NP;
    k.tiles_h = cdiv(k.Ho, t.TH); k.tiles_w = cdiv(k.Wo, t.TW);
    k.lgTW = 0; k.lgTH = 0;
    while ((1 << k.lgTW) < t.TW) ++k.lgTW;
    while ((1 << k.lgTH) < t.TH) ++k.lgTH;
    if ((1 << k.lgTW) != t.TW || (1 << k.lgTH) != t.TH) return RHO_E_SHAPE;      // choose_tile only returns powers of two
    k.inv_ihw = 1.0f / (float)(t.IH * t.IW); k.inv_iw = 1.0f / (float)t.IW;
    // 8-wide tiles: pair rows (pd*TH + ph) differing in one bit whose halo offset is 8 (mod 16) positions
    k.pair_lg = -1;
    if (t.TW == 8 && d.sw == 1 && d.sh == 1 && !d.up_h && !d.up_w) {
        int lgTH = 0;
        while ((1 << lgTH) < t.TH) ++lgTH;
        for (int b = 0; b < 5 && k.pair_lg < 0; ++b) {
            const long long delta = (b < lgTH) ? (long long)(1 << b) * t.IW : (long long)(1 << (b - lgTH)) * t.IH * t.IW;
            if ((1 << b) < t.TD * t.TH && delta % 16 == 8) k.pair_lg = b;
        }
    }
    if (d.pre_a && !d.pre_b) return RHO_E_ARG;

    const long long tiles = (long long)cdiv(k.Do, t.TD) * k.tiles_h * k.tiles_w;
    if (tiles > 0x7FFFFFFFLL || d.coutp / BM > 65535 || gridz > 65535) return RHO_E_SHAPE;
    k.cofast = (d.coutp / BM > 1 && tiles % 8 == 0) ? 1 : 0;
    dim3 grid((unsigned)tiles, (unsigned)(d.coutp / BM), (unsigned)gridz);
    // fused output statistics: only where a tile belongs to one sample and the whole output is channels-last
    int64_t tps_geo = 0;                                 // tiles per sample where every tile lies in one sample, else 0
    if (d.kd == 3) tps_geo = tiles * n_phase;            // phases: the launches of one output share the buffer, each its own tile range
    else if (taps == 1 && k.S_out % 256 == 0 && t.TW == 256) tps_geo = k.S_out / 256;
    const int64_t tps = (d.split == d.cout && d.split > 0) ? tps_geo : 0;
    if (stats_tiles) { *stats_tiles = tps; return 0; }
    // k-split (rho_conv_desc.ws): merged-batch launches (2-D / 1-D kernels with taps) whose grid leaves most CUs idle
    int ksplit = 1;
    const long long positions = (long long)k.Do * k.Ho * k.Wo;
    {
        const long long wgs = tiles * (d.coutp / BM);
        const int chunks = cin / CK;
        static const bool split_env = !(getenv("RHO_CONV_SPLITK") && atoi(getenv("RHO_CONV_SPLITK")) == 0);
        if (split_env && d.kd == 1 && (d.split == d.cout || d.y2_cl) && !d.stats && !d.gna_g && wgs <= 128 && chunks >= 4) {
            int want = (int)(512 / wgs);                               // about two workgroups per CU
            if (want > 16) want = 16;
            const int per = taps == 1 ? 4 : 2;                         // chunks per split at least: the set-up of a tile is paid per split
            if (want > chunks / per) want = chunks / per;
            if (ws_want) { *ws_want = want >= 2 ? (int64_t)want * positions * d.coutp * (int64_t)sizeof(float) : 0; return 0; }
            if (d.ws != nullptr && want >= 2) {
                const long long fit = d.ws_bytes / (positions * d.coutp * (long long)sizeof(float));
                if (fit < want) want = (int)fit;
                if (want >= 2) ksplit = want;
            }
        } else if (ws_want) { *ws_want = 0; return 0; }
    }
    k.ksplit = ksplit; k.slab = (float*)d.ws; k.slab_stride = positions * d.coutp;
    if (ksplit > 1) grid.z = (unsigned)ksplit;
    k.stats = nullptr; k.tps = 1;
    if (d.stats) {
        if (tps <= 0) return RHO_E_ARG;
        k.stats = d.stats; k.tps = (int)tps;
        k.stats_off = phase_idx * (int)tiles;
    }
    k.gnb_x1 = nullptr;
    k.gna_g = nullptr;
    if (d.gna_g) {
        // GroupNorm backward apply in the epilogue: both output regions on the transposed path, one sample per tile
        const int pe = d.dtype == RHO_BF16 ? 8 : 4;
        if (taps != 1 || d.res || d.res2 || d.stats || !d.gnb_x1 || !d.gnb_a || !d.gnb_b || !d.gna_cA || !d.gna_cP || !d.gna_cQ || d.split <= 0 || d.cout % 32 ||
            d.gnb_c1 <= 0 || d.gnb_c1 > d.cout || d.gnb_c1 % 32 || ((d.gnb_c1 < d.cout) != (d.gnb_x2 != nullptr)) || tps_geo <= 0 ||
            (d.split < d.cout && (!d.y2_cl || (d.cout - d.split) % pe != 0)) || d.gnb_silu < 0 || d.gnb_silu > 1 || n_phase != 1)
            return RHO_E_ARG;
        k.gnb_x1 = (const char*)d.gnb_x1; k.gnb_x2 = (const char*)d.gnb_x2; k.gnb_a = d.gnb_a; k.gnb_b = d.gnb_b;
        k.gnb_c1 = d.gnb_c1; k.gnb_silu = d.gnb_silu;
        k.gna_g = (const char*)d.gna_g; k.gna_cA = d.gna_cA; k.gna_cP = d.gna_cP; k.gna_cQ = d.gna_cQ;
        k.tps = (int)tps_geo;
    } else if (d.gnb_x1) {
        if (!d.stats || !d.gnb_a || !d.gnb_b || d.gnb_c1 <= 0 || d.gnb_c1 > d.split || d.gnb_c1 % 32 || (d.split - d.gnb_c1) % 32 ||
            ((d.gnb_c1 < d.split) != (d.gnb_x2 != nullptr)))
            return RHO_E_ARG;
        k.gnb_x1 = (const char*)d.gnb_x1; k.gnb_x2 = (const char*)d.gnb_x2; k.gnb_a = d.gnb_a; k.gnb_b = d.gnb_b;
        k.gnb_c1 = d.gnb_c1; k.gnb_silu = d.gnb_silu;
    }
    size_t lds = (size_t)t.NP * PITCH + (size_t)WSLOTS * BM * PITCH;
    k.sk_w = nullptr;
    size_t lds_sk = 0;
    if (d.sk_w) {
        // folded 1x1x1 skip: the 16x16x32 3x3x3 variants only (bf16, stride 1, whole output channels-last, same-size input)
        if (!d.sk_x1 || d.sk_c1 <= 0 || d.sk_c1 % CK || d.sk_c2 < 0 || d.sk_c2 % CK || ((d.sk_c2 > 0) != (d.sk_x2 != nullptr))) return RHO_E_ARG;
        if (!(m16 && m16_env_on() && d.kd == 3 && d.kh == 3 && d.kw == 3 && t.NP <= 640 && (BM == 64 || BM == 128) && d.split == d.cout &&
              !d.zs_h && !d.zs_w && !d.ph_h && !d.ph_w && !d.phd_h && !d.phd_w))
            return RHO_E_ARG;
        k.sk_x1 = (const char*)d.sk_x1; k.sk_x2 = (const char*)d.sk_x2; k.sk_w = (const char*)d.sk_w; k.sk_bias = d.sk_bias;
        k.sk_c1 = d.sk_c1; k.sk_c2 = d.sk_c2;
        lds_sk = (size_t)2 * 256 * PITCH + (size_t)2 * BM * PITCH;
    }
    k.coef_off = 0;
    if (taps > 1 && d.kd == 3 && d.pre_a && lds + (size_t)2 * cin * sizeof(float) <= (BM == 128 ? lds_cap : lds_cap / 2)) {
        k.coef_off = (int)lds;                                    // after the halo tile and the weight ring
        lds += (size_t)2 * cin * sizeof(float);
    }
    if (taps == 1)        // 1x1x1: double-buffered activations + weights + the prologue coefficients of one sample
        lds = (size_t)2 * 256 * PITCH + (size_t)2 * BM * PITCH + (d.pre_a ? (size_t)2 * k.cin * sizeof(float) : 0);
    if (k.coef_off > 0 && (size_t)k.coef_off < lds_sk) { lds -= (size_t)2 * cin * sizeof(float); k.coef_off = 0; }   // (the skip phase's buffers would overlap them)
    if (lds < lds_sk) lds = lds_sk;
    if (d.stats) { const size_t lr = (size_t)(BM == 128 ? 512 : 256) * 16 * sizeof(float); if (lds < lr) lds = lr; }
    const size_t lds_epi = (size_t)256 * (BM * 4 + 16);       // epilogue transpose staging (fp32 rows, all 256 positions at once)
    if ((d.split > 0 || d.y2_cl) && lds < lds_epi) lds = lds_epi;
    hipStream_t st = as_stream(stream);
    const bool m16_env = m16_env_on();
    const int rc = d.dtype == RHO_BF16 ? launch_taps<bf16_raw>(d, k, BM, t.NP, grid, lds, m16 && m16_env, st)
                                       : launch_taps<float>(d, k, BM, t.NP, grid, lds, false, st);
    if (rc != 0 || ksplit == 1 || g_variant != nullptr) return rc;
    const long long items = positions * (d.coutp / 4);
    const dim3 rgrid((unsigned)((items + 255) / 256));
    const SplitMap sm{k.Ho, k.Wo, k.Ho_out, k.Wo_out, k.oy_mul, k.oy_add, k.ox_mul, k.ox_add};
    const SplitOut so{d.cout, d.coutp, d.split, d.bias, d.res_add, (long long)k.res_add_stride, k.S_out, (const char*)d.res, (char*)d.y,
                      (const char*)d.res2, (char*)d.y2};
    if (d.dtype == RHO_BF16) hipLaunchKernelGGL(k_splitk_reduce<bf16_raw>, rgrid, dim3(256), 0, st, k.slab, k.slab_stride, ksplit, positions, so, sm);
    else hipLaunchKernelGGL(k_splitk_reduce<float>, rgrid, dim3(256), 0, st, k.slab, k.slab_stride, ksplit, positions, so, sm);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int64_t rho_conv_workspace_bytes(const rho_conv_desc* dp) {
    int64_t b = 0;
    return conv_impl(dp, nullptr, nullptr, &b) == 0 ? b : 0;
}

extern "C" int rho_conv_nd_fwd(const rho_conv_desc* dp, void* stream) { return conv_impl(dp, stream, nullptr); }

extern "C" int rho_conv_variant(const rho_conv_desc* dp, char* buf, int cap) {
    if (!buf || cap < 64) return RHO_E_ARG;
    buf[0] = 0;
    VariantOut vo{buf, cap};
    g_variant = &vo;
    const int rc = conv_impl(dp, nullptr, nullptr);
    g_variant = nullptr;
    return rc;
}

extern "C" int64_t rho_conv_stats_tiles(const rho_conv_desc* dp) {
    int64_t t = 0;
    return conv_impl(dp, nullptr, &t) == 0 ? t : 0;
}
