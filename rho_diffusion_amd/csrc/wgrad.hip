// Weight gradient of the n-D convolution:  dW[tap][co][ci] = sum_{n,pos} dY[n,pos,co] * Xact[n,pos (+) tap,ci]
// (autograd of conv_nd, rho_diffusion/layers.py:77-88; Xact = SiLU(GroupNorm*FiLM(x)) is RECOMPUTED
// in the loader from the saved pre-norm activation and the folded affine, never stored).
//
// Same LDS-halo organisation as the forward kernel, turned into a reduction over positions:
//   * a workgroup owns one (64-cout x 32-cin) weight tile for ALL taps and walks a slab of 256-position
//     output tiles; per tile it stages the input halo chunk (with the prologue) and the dY tile once,
//   * the taps are dealt to the four waves (7+7+7+6 of 27), so the whole 27x64x32 fp32 accumulator
//     (221 KB) lives in registers (224 per lane) across the slab and is flushed with fp32 atomics once
//     per workgroup: 2 x 128-byte contiguous segments per wave instruction (MI355X_MICROARCH, float atomics),
//   * the contraction index is the POSITION, but both operands are channels-last, so the MFMA
//     fragments (8 consecutive k per lane) are produced by gfx950's transposing LDS read
//     ds_read_b64_tr_b16 straight from the row-major tiles - no transposed copies.
// The exact-f32 variant (v_mfma_f32_32x32x2_f32) needs one k per lane, i.e. plain ds_read_b32.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "conv_common.h"

// LDS rows are UNPADDED: a transposing read touches, per half-wave, 4 consecutive k-rows x 64 contiguous bytes;
// with 64-byte halo rows those are 256 contiguous bytes = all 64 banks once.  The 128-byte dY rows get the same
// property by swapping their two 64-byte halves on rows with bit 1 set.
#define XP 64          // halo row: 32 bf16 / 16 f32 channels
#define DYP 128        // dY row: 64 bf16 / 32 f32 channels
__device__ __forceinline__ int dy_swz(int row) { return ((row >> 1) & 1) << 6; }

struct WgradK {
    const char* x1;
    const char* x2;
    const float* pre_a;
    const float* pre_b;
    const char* dy;
    float* dw;
    float* dbias;       // optional [coutp]: channel sums of dY (bias gradient), accumulated by the cin-chunk-0 workgroups
    int c1, c2, cin;
    int dyw;            // row width (elements) of dY
    int coutp;          // rows of the dw buffer
    int D, H, W, Do, Ho, Wo;
    long long S_in;
    int sh, sw, pre_silu;
    int TD, TH, TW, ID, IH, IW, NP;
    int lgTW, lgTH;
    int tiles_d, tiles_h, tiles_w, n_batch;
    int tiles_total, tiles_per_block;
    // sub-pixel phase of a conv behind a nearest x2 upsample (rho_conv_desc.ph_h / ph_w): taps start pad_h / pad_w rows before
    // the output position; dY row of output row oh: oh * oy_mul + oy_add in a tensor of Ho_out x Wo_out rows per depth slice
    int pad_h, pad_w, oy_mul, oy_add, ox_mul, ox_add, Ho_out, Wo_out;
    int xcd_map;        // 1: workgroups of one position slab (all cout tiles x cin chunks) on consecutive launch slots of one XCD
    int pairc;          // f32 1x1x1: a workgroup owns TWO 16-channel input chunks (the upper 16 B columns of the MFMA carry the second)
    // Deterministic flush (rho_conv_nd_wgrad_ws): instead of fp32 atomics into dw / dbias (whose arrival order differs run to run)
    // every accumulator owner STORES its partial tile to its own slab [taps][coutp][cin] + [coutp]; k_wgrad_slab_reduce then adds
    // the slabs in slab order.  Owner = the position slab of the workgroup (bx), x 4 + wave where the waves split positions.
    float* slab;
    long long slab_stride;      // floats per slab (taps * coutp * cin + coutp)
    long long slab_bias_off;    // offset of the bias partial inside a slab (taps * coutp * cin)
};

// one element of a workgroup's partial weight gradient: ordered slab store (deterministic mode) or fp32 atomic
// (base = the owner's slab or dw: resolved once per flush from the kernel-argument segment, see RHO_WG_FLUSH_ARGS)
__device__ __forceinline__ void wg_flush(bool det, float* base, size_t off, float v) {
    if (det) base[off] = v;
    else atomicAdd(base + off, v);
}
// The flush parameters are read from the kernel-argument segment AFTER the tile loop: held in SGPRs across it they pushed the
// main bf16 variant past the scalar register file (8 spills to VGPR lanes).  The empty asm keeps the reload below the loop.
#define RHO_WG_FLUSH_ARGS()                                                                                                          \
    const __attribute__((address_space(4))) WgradK* qp_ = (const __attribute__((address_space(4))) WgradK*)__builtin_amdgcn_kernarg_segment_ptr(); \
    asm volatile("" : "+s"(qp_));                                                                                                    \
    const __attribute__((address_space(4))) WgradK& kq = *qp_

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}).  Register arrays indexed
// through a lambda parameter end up in scratch memory unless the index is a constant expression.
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// source of the LDS-DMA lanes that must deliver zeros (conv padding, rows beyond the tile, channels beyond dy_width)
__device__ uint4 g_wgrad_zero[4];

// 32(row/col) x 16(k) bf16 MFMA operand out of a row-major [k][channel] LDS tile: lane (col = l&31,
// half = l>>5) gets k = 8*half + {0..7}.  Per 16-lane group the transposing read takes row addresses
// from lanes 4q+p (row q, columns 4p..4p+3) and returns column i of the 4 rows to lane i.
// r0 / r1: byte offsets of this lane's k-rows (4t + q, t = 0/1) incl. the column offset.
__device__ __forceinline__ uint4 tr_frag(const char* lds, int r0, int r1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + r0));
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(lds + r1));
    uint4 f;
    f.x = (uint32_t)(uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
    f.y = (uint32_t)(uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
    f.z = (uint32_t)(uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
    f.w = (uint32_t)(uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
    return f;
}

// PRE: the folded GroupNorm affine + SiLU is applied while the halo is staged (p.pre_a != NULL); a compile-time switch so that
// the common training path (input materialised by rho_gn_apply, PRE = false) keeps its staging code free of branches.
// GEO = 1: the tile geometry is a compile-time constant (4 x 8 x 8 output positions, 10 x 10 halo rows, stride 1 - every stride-1
// 3x3x3 layer of the 3-D configurations): the LDS offsets of the k-steps become immediates of the transposing reads.
template <typename T, int KD, int KH, int KW, int MAXP, bool PRE, int GEO = 0>
__global__ __launch_bounds__(256) void k_wgrad(const WgradK p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int CK = ET<T>::CK;
    constexpr int PE = ET<T>::PE;
    constexpr int NT = KD * KH * KW;
    constexpr bool IS_BF16 = sizeof(T) == 2;
    // taps dealt to waves; else positions dealt to waves.  (f32, 9 taps: positions - dealing 9 taps to 4 waves leaves one wave idle,
    // and the f32 accumulators of all 9 taps fit one wave: 5 tap-pair tiles of 16 registers, see F32 below)
    constexpr bool TAPSPLIT = IS_BF16 ? (NT >= 9) : (NT > 9);
    constexpr int TPW = TAPSPLIT ? (NT + 3) / 4 : NT;  // taps per wave
    constexpr int COT = 128 / (int)sizeof(T);        // cout tile: 64 bf16 / 32 f32
    constexpr int MT = COT / 32;
    // exact f32 (v_mfma_f32_32x32x2_f32): a 64-byte chunk is 16 input channels, half the 32 columns of the MFMA's B operand.
    // Round 3: the other 16 columns carry the SAME channels at the wave's NEXT tap (a per-lane tap offset), so one MFMA reduces
    // two taps (r02 fed them zeros: c2's weight gradient ran at 0.18 of the f32 matrix peak, 184 of 314 ms per training step).
    constexpr int NACC = IS_BF16 ? TPW : (TPW + 1) / 2;
    // ... and a 1x1x1 launch has no second tap: there the upper columns carry the NEXT 16-channel chunk (p.pairc; the workgroup stages
    // both chunks, the second at halo rows 256 .. 511: slots 4 .. 7 of every thread)
    constexpr bool PAIRC = !IS_BF16 && NT == 1;
    constexpr int NKS = TAPSPLIT ? 16 : 4;           // 16-position k-steps of this wave per tile
    constexpr bool KEEP_REL = (MAXP <= 10);          // big-halo (strided) variant recomputes instead of holding registers
    // PIPE (bf16, regular halo): the LDS tiles are DOUBLE-BUFFERED (2 x (40 KB halo + 32 KB dY) = 144 of the 160 KB) and the staging
    // of tile t+1 (registers -> LDS) plus the global loads of tile t+2 are dealt one slot per k-step INTO the MFMA phase of tile t,
    // one barrier per tile.  The single-buffered form ran load-wait / LDS-write / barrier / address set-up as a serial phase per
    // tile with nothing to overlap it (one wave per SIMD: 224 accumulator registers) - PMC r01h: the matrix pipe busy 57 % of the
    // wave's cycles, 19 % in staging VALU, 18 % parked in waits.
    constexpr bool PIPE = KEEP_REL;          // (round 3: the exact-f32 variants too - their serial load-wait / LDS-write phase cost 47 % of the tile)
    constexpr int XBUF = MAXP * 64 * XP;
    constexpr int DBUF = 256 * DYP;
    constexpr int BUF = XBUF + DBUF;

    char* const halo = smem;                          // MAXP*64 rows (rows >= NP are written with zeros, never read)
    char* const dyt = smem + (size_t)MAXP * 64 * XP;  // (PIPE: buffer b = [halo | dY] at smem + b * BUF)

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform by construction: keeps the tap offsets in scalar registers
    const int piece = tid & 3;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  The cout-tile x cin-chunk workgroups of one position
    // slab read the same dY tile / X rows: with the plain 3-D grid the second chunk of a slab ran rounds later on whatever XCD and
    // fetched its half of every 128-byte line from HBM again.  Here the launch slots k = L / 8 of XCD e = L % 8 walk slab
    // (k / pairs) * 8 + e through all its pairs back to back, so they run side by side out of one L2.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_map) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int pairs = gridDim.y * gridDim.z;
        const int e = L & 7, k = L >> 3;
        const int pi = k % pairs;
        bx = (k / pairs) * 8 + e;
        bz = pi % (int)gridDim.z;
        by = pi / (int)gridDim.z;
    }
    const int co0 = by * COT;
    const bool pairc = PAIRC && p.pairc != 0;
    const int c = bz * CK * (pairc ? 2 : 1);        // (first) input-channel chunk of this workgroup
    const char* src;
    int cs, csrc;
    if (c < p.c1) { src = p.x1; cs = p.c1; csrc = c; } else { src = p.x2; cs = p.c2; csrc = c - p.c1; }
    src += (size_t)csrc * sizeof(T) + piece * 16;
    const bool has2 = pairc && (c + CK < p.cin);    // second chunk of the pair (may lie in the other concat source)
    const char* src2 = src;
    int cs2 = cs;
    if (has2) {
        const int cb = c + CK;
        if (cb < p.c1) { src2 = p.x1 + (size_t)cb * sizeof(T) + piece * 16; cs2 = p.c1; }
        else { src2 = p.x2 + (size_t)(cb - p.c1) * sizeof(T) + piece * 16; cs2 = p.c2; }
    }

    // halo slot -> (id, ih, iw) and its offset (in positions) from the tile's first halo position: fixed for every tile
    int sdec[MAXP], srel[KEEP_REL ? MAXP : 1];
    {
        const int ihw = p.IH * p.IW;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int hp0 = (tid >> 2) + 64 * i;
            // (pair mode: slots 4 .. 7 hold the second chunk's copy of rows 0 .. 255)
            const bool sec = PAIRC && pairc && i >= 4 && i < 8;
            const int hp = sec ? hp0 - 256 : hp0;
            int v = -1, rel = 0;
            if (hp < p.NP && (!sec || has2) && !(PAIRC && pairc && !sec && i >= 4)) {
                const int id = hp / ihw;
                const int r = hp - id * ihw;
                const int ih = r / p.IW;
                const int iw = r - ih * p.IW;
                v = (id << 20) | (ih << 10) | iw;
                rel = (id * p.H + ih) * p.W + iw;
            }
            sdec[i] = v;
            if constexpr (KEEP_REL) srel[i] = rel;
        }
    }
    // dY rows of this thread: (pd, ph, pw) and the offset from the tile's first output position
    const int dpiece = tid & 7;
    const bool dch_ok = (co0 + dpiece * PE) < p.dyw;
    const char* const dsrc = p.dy + (size_t)(dch_ok ? co0 + dpiece * PE : 0) * sizeof(T);
    int ddec[8], drel[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (tid >> 3) + 32 * i;
        const int pw = row & (p.TW - 1), ph = (row >> p.lgTW) & (p.TH - 1), pd = row >> (p.lgTW + p.lgTH);
        ddec[i] = (pd << 20) | (ph << 10) | pw;
        drel[i] = (pd * p.Ho_out + ph * p.oy_mul) * p.Wo_out + pw * p.ox_mul;
    }

    // taps of this wave
    int tap_of[TPW], tapoff[TPW];
#pragma unroll
    for (int ti = 0; ti < TPW; ++ti) {
        const int tap = TAPSPLIT ? wave + 4 * ti : ti;
        tap_of[ti] = tap;
        // a slot past the last tap (27 = 4*7 - 1) still runs, on the last tap's window, and is not flushed:
        // branch-free inner loop (an exec-masked branch per tap pair serialises LDS latency with the MFMAs)
        const int tc = tap < NT ? tap : NT - 1;
        const int kd = tc / (KH * KW), kh = (tc / KW) % KH, kw = tc % KW;
        tapoff[ti] = ((kd * p.IH + kh) * p.IW + kw) * XP;
    }

    f32x16_t acc[NACC][MT];
#pragma unroll
    for (int ti = 0; ti < NACC; ++ti)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][mi][r] = 0.0f;

    // lane geometry of the transposing reads; the LDS row offsets of this lane's k-rows depend only on the k-step,
    // so they are computed once (not per tile and k-step: the decode costs ~20 VALU incl. integer multiplies)
    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pq = li & 3;
    const int colb = 16 * (grp & 1) + 4 * pq;          // channel of this lane's 8-byte address within a 32-wide tile
    const int ks0 = TAPSPLIT ? 0 : wave * NKS;
    auto xrow_of = [&](int j, int tt) {
        const int pp = 16 * (ks0 + j) + 8 * (grp >> 1) + 4 * tt + q;
        const int pw = pp & (p.TW - 1), ph = (pp >> p.lgTW) & (p.TH - 1), pd = pp >> (p.lgTW + p.lgTH);
        return ((pd * p.IH + ph * p.sh) * p.IW + pw * p.sw) * XP + colb * 2;
    };
    constexpr bool DMA = PIPE && !PRE;                 // double-buffered LDS-DMA path (below)
    int xrow[(KEEP_REL && !DMA) ? NKS : 1][2];
    int arow[MT];                                      // dY fragment base of this lane (k-step 0, t = 0)
    if constexpr (IS_BF16) {
        if constexpr (KEEP_REL && !DMA) {
#pragma unroll
            for (int j = 0; j < NKS; ++j)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) xrow[j][tt] = xrow_of(j, tt);
        }
        const int pp0 = 16 * ks0 + 8 * (grp >> 1) + q;   // + 16*j + 4*tt: bit 1 of the row index is q's
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) arow[mi] = pp0 * DYP + ((mi * 64) ^ dy_swz(q)) + colb * 2;
    }

    const int tile0 = bx * p.tiles_per_block;
    const int tile1 = min(tile0 + p.tiles_per_block, p.tiles_total);

    // Staging is software-pipelined across tiles: the global loads of tile t+1 (branch-free, padding lanes read a
    // clamped address and are zeroed later, so all of them are in flight together and visible to the counted
    // vmcnt) are issued before the MFMA phase of tile t and consumed after it.
    uint4 xv[MAXP], dv[8];
    // bias gradient = channel sums of dY: the dY tile passes through this thread's registers anyway (8 rows x one
    // 16-byte channel piece per tile); only the workgroups of input-channel chunk 0 add them up (uniform branch)
    constexpr int DPE = 16 / (int)sizeof(T);
    const bool do_bias = (p.dbias != nullptr) && (bz == 0);
    float bsum[DPE];
#pragma unroll
    for (int e = 0; e < DPE; ++e) bsum[e] = 0.0f;
    int xpos[MAXP];          // global input position of each halo slot (-1 = zero padding) of the tile in flight
    unsigned dok = 0;        // validity bits of the 8 dY rows of the tile in flight
    int n_cur = 0;

    int base = 0, dbase = 0, gd_base = 0, gh_base = 0, gw_base = 0;   // wave-uniform tile origin (scalar unit)
    bool full = true;
    auto decode = [&](int tl) {
        int t = tl;
        const int tw_i = t % p.tiles_w; t /= p.tiles_w;
        const int th_i = t % p.tiles_h; t /= p.tiles_h;
        const int td_i = t % p.tiles_d;
        const int n = t / p.tiles_d;
        n_cur = n;
        const int od0 = td_i * p.TD, oh0 = th_i * p.TH, ow0 = tw_i * p.TW;
        gd_base = od0 - (KD / 2); gh_base = oh0 * p.sh - p.pad_h; gw_base = ow0 * p.sw - p.pad_w;
        base = ((n * p.D + gd_base) * p.H + gh_base) * p.W + gw_base;
        dbase = ((n * p.Do + od0) * p.Ho_out + oh0 * p.oy_mul + p.oy_add) * p.Wo_out + ow0 * p.ox_mul + p.ox_add;
        full = od0 + p.TD <= p.Do && oh0 + p.TH <= p.Ho && ow0 + p.TW <= p.Wo;
    };
    auto issue_x = [&](auto LO, auto HI) {
#pragma unroll
        for (int i = decltype(LO)::value; i < decltype(HI)::value; ++i) {
            const int id = sdec[i] >> 20, ih = (sdec[i] >> 10) & 1023, iw = sdec[i] & 1023;
            const bool ok = (sdec[i] >= 0) & ((unsigned)(gd_base + id) < (unsigned)p.D) & ((unsigned)(gh_base + ih) < (unsigned)p.H) &
                            ((unsigned)(gw_base + iw) < (unsigned)p.W);
            const int rel = KEEP_REL ? srel[KEEP_REL ? i : 0] : (id * p.H + ih) * p.W + iw;
            const int pos = ok ? base + rel : 0;
            xpos[i] = ok ? pos : -1;
            if (PAIRC && i >= 4 && i < 8)      // (slots of the second chunk; without pair mode they are invalid and read position 0 of src2 = src)
                xv[i] = *reinterpret_cast<const uint4*>(src2 + (size_t)pos * cs2 * sizeof(T));
            else
                xv[i] = *reinterpret_cast<const uint4*>(src + (size_t)pos * cs * sizeof(T));
        }
    };
    auto issue_dy = [&]() {
        if (full) {
            dok = dch_ok ? 0xFFu : 0u;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                dv[i] = *reinterpret_cast<const uint4*>(dsrc + (size_t)(dch_ok ? dbase + drel[i] : 0) * p.dyw * sizeof(T));
        } else {
            const int od_lim = p.Do - (gd_base + KD / 2), oh_lim = p.Ho - (gh_base + p.pad_h) / p.sh, ow_lim = p.Wo - (gw_base + p.pad_w) / p.sw;
            dok = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pd = ddec[i] >> 20, ph = (ddec[i] >> 10) & 1023, pw = ddec[i] & 1023;
                const bool ok = (pd < od_lim) & (ph < oh_lim) & (pw < ow_lim) & dch_ok;
                dok |= (ok ? 1u : 0u) << i;
                dv[i] = *reinterpret_cast<const uint4*>(dsrc + (size_t)(ok ? dbase + drel[i] : 0) * p.dyw * sizeof(T));
            }
        }
    };
    // halo chunk (prologue applied, zero padding after the activation) -> LDS
    auto store_x = [&](auto LO, auto HI) {
#pragma unroll
        for (int i = decltype(LO)::value; i < decltype(HI)::value; ++i) {
            uint4 u = xpos[i] >= 0 ? xv[i] : make_uint4(0u, 0u, 0u, 0u);
            if constexpr (PRE) {
                if (xpos[i] >= 0) {
                    const int smp = (KD == 3) ? n_cur : (int)((unsigned)xpos[i] / (unsigned)p.S_in);
                    const size_t co = (size_t)smp * p.cin + c + ((PAIRC && pairc && i >= 4) ? CK : 0) + piece * PE;
                    u = apply_pre<T>(u, p.pre_a + co, p.pre_b + co, p.pre_silu);
                }
            }
            const int hp = (tid >> 2) + 64 * i;
            *reinterpret_cast<uint4*>(halo + hp * XP + piece * 16) = u;
        }
    };
    auto store_dy = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (tid >> 3) + 32 * i;
            const uint4 u = ((dok >> i) & 1u) ? dv[i] : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(dyt + row * DYP + ((dpiece * 16) ^ dy_swz(row))) = u;
            if (do_bias) {
                if constexpr (IS_BF16) {
                    bsum[0] += __uint_as_float(u.x << 16); bsum[1] += __uint_as_float(u.x & 0xFFFF0000u);
                    bsum[2] += __uint_as_float(u.y << 16); bsum[3] += __uint_as_float(u.y & 0xFFFF0000u);
                    bsum[4] += __uint_as_float(u.z << 16); bsum[5] += __uint_as_float(u.z & 0xFFFF0000u);
                    bsum[6] += __uint_as_float(u.w << 16); bsum[7] += __uint_as_float(u.w & 0xFFFF0000u);
                } else {
                    bsum[0] += __uint_as_float(u.x); bsum[1] += __uint_as_float(u.y);
                    bsum[2] += __uint_as_float(u.z); bsum[3] += __uint_as_float(u.w);
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using IH = std::integral_constant<int, MAXP / 2>;
    using IM = std::integral_constant<int, MAXP>;

    // exact-f32 reduction of one staged tile (halo_ / dyt_: the LDS images of that tile)
    auto f32_tile = [&](const char* halo_, const char* dyt_, auto&& between) {
                // exact f32: k = position pair (2*kk + half); columns 0-15 = the 16 channels at tap 2s, columns 16-31 = the same
        // channels at tap 2s + 1 of this wave (an odd tap count leaves the last slot's upper half on a repeat that is not flushed)
        const int kk0 = TAPSPLIT ? 0 : wave * 32;
        const int kk1 = TAPSPLIT ? 128 : wave * 32 + 32;
        const int col = lane & 31;
        int toff2[NACC];
    #pragma unroll
        for (int s_ = 0; s_ < NACC; ++s_)
            toff2[s_] = ((col >= 16 && 2 * s_ + 1 < TPW) ? tapoff[(2 * s_ + 1 < TPW) ? 2 * s_ + 1 : 0] : tapoff[2 * s_]) + (col & 15) * 4 +
                        ((PAIRC && has2 && col >= 16) ? 256 * XP : 0);
        // operands of k-step kk + 1 are read from LDS before the MFMAs of k-step kk are issued (one wave per SIMD: nothing else
        // covers the ~100 cycles of a dependent ds_read in front of every 5 MFMAs)
        auto rd = [&](int kk, float& av, float (&bv)[NACC]) {
            const int pp = 2 * kk + half;
            const int pw = pp & (p.TW - 1), ph = (pp >> p.lgTW) & (p.TH - 1), pd = pp >> (p.lgTW + p.lgTH);
            av = *reinterpret_cast<const float*>(dyt_ + pp * DYP + ((col * 4) ^ dy_swz(pp)));
            const int xr = ((pd * p.IH + ph * p.sh) * p.IW + pw * p.sw) * XP;
    #pragma unroll
            for (int s_ = 0; s_ < NACC; ++s_) bv[s_] = *reinterpret_cast<const float*>(halo_ + xr + toff2[s_]);
        };
        constexpr int NIT2 = TAPSPLIT ? 64 : 16;        // (kk1 - kk0 is 128 or 32)
        // Round 4: a wave issues IN ORDER and has the SIMD to itself (one wave per SIMD: 144 KB of LDS), so the ~30 address / read /
        // bookkeeping instructions of a k-step, scheduled as one run behind its five back-to-back MFMAs, only started to issue once the
        // LAST of them had entered the pipe - 64 cycles of cover for >= 120 cycles of issue, the matrix pipe idle for the rest (the
        // reduction alone ran at 0.62 of the f32 rate).  Now every MFMA is followed, in program order, by ONE operand read of k-step
        // kk + 2 (three operand sets rotate) and a fifth of the other work, pinned by the data-dependence fence in front of the next
        // MFMA (the MFMA comes after it through its operands, the reads issued before it stay before it through the memory clobber)
        // and a scheduling barrier behind the group: each group issues inside its own MFMA's 64 cycles.
        float av[3], bv[3][NACC];
        rd(kk0, av[0], bv[0]);
        rd(min(kk0 + 1, kk1 - 1), av[1], bv[1]);
        static_for<2 * NIT2>([&](auto KS) {
            constexpr int ks = decltype(KS)::value;
            constexpr int cur = ks % 3, nn = (ks + 2) % 3;
            const int kn = min(kk0 + ks + 2, kk1 - 1);               // (past the end: a harmless re-read of the last k-step)
            const int pp = 2 * kn + half;
            const int pw = pp & (p.TW - 1), ph = (pp >> p.lgTW) & (p.TH - 1), pd = pp >> (p.lgTW + p.lgTH);
            const int xr = ((pd * p.IH + ph * p.sh) * p.IW + pw * p.sw) * XP;
    #pragma unroll
            for (int s_ = 0; s_ < NACC; ++s_) {
                asm volatile("" : "+v"(av[cur]), "+v"(bv[cur][s_]) : : "memory");
                acc[s_][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur], bv[cur][s_], acc[s_][0], 0, 0, 0);
                if (s_ == 0) av[nn] = *reinterpret_cast<const float*>(dyt_ + pp * DYP + ((col * 4) ^ dy_swz(pp)));
                bv[nn][s_] = *reinterpret_cast<const float*>(halo_ + xr + toff2[s_]);
                if constexpr ((ks & 1) == 0) {
                    if (s_ == NACC / 2) between(std::integral_constant<int, ks / 2>{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    if constexpr (DMA) {
        // ------------------------------------------------------------------ double-buffered LDS tiles filled by LDS-DMA
        // Phase of tile t: the 18 global_load_lds_dwordx4 of tile t+1 (into the other buffer; their addresses were computed during phase
        // t-1), the address set-up of tile t+2 and 16 k-steps of MFMAs out of this tile's buffer, then __syncthreads(): a vmcnt(0) in
        // front of it retires the DMA, the barrier publishes the buffer.  No staging registers, no ds_write, no serial load-wait /
        // write / set-up phase per tile.  The LDS image is lane-linear (16 bytes x tid per 4 KB slot), which is exactly the halo layout
        // (64-byte rows, 4 pieces) and the dY layout (128-byte rows, 8 pieces); the dY half-swap of rows with bit 1 set and the zero
        // fill of padding / out-of-range rows are applied on the SOURCE side (swapped channel piece; a zero page in the code object).
        //
        // Round 4: ONE wave per SIMD issues IN ORDER, so whatever sits between two MFMAs in program order delays the second one.  The
        // compiler's own schedule ran the 18 DMAs as one burst at the top of the phase, the slot addresses as exec-masked branches with
        // 64-bit multiplies (~25 instructions per slot) and the transposing reads in runs of 12 - 16 between groups of MFMAs: removing
        // the address work alone returned 15 % of the launch, the DMAs 11 % (timing probes).  Now (a) a slot address is a
        // wave-uniform 64-bit tile base + a per-lane 32-bit offset fixed for the launch, its validity one bit of a per-tile mask (only
        // the first / last tile of a dimension has padding or overhang: per-lane masks for those six cases are built once) - 6 VALU per
        // slot, no branch; (b) the tile walk is incremental (no divisions); (c) every MFMA is followed, in program order, by its share
        // of the k-step's reads, of the DMAs and of the set-up, and a scheduling barrier pins that order.
        const int dsw = dpiece ^ (((tid >> 4) & 1) << 2);                  // channel piece this thread FETCHES for its dY slots
        const bool dsw_ok = (co0 + dsw * PE) < p.dyw;
        const char* const zpage = reinterpret_cast<const char*>(g_wgrad_zero);
        const char* const srcu = (c < p.c1) ? p.x1 + (size_t)c * sizeof(T) : p.x2 + (size_t)(c - p.c1) * sizeof(T);   // wave-uniform
        const char* srcu2 = srcu;
        if (has2) {
            const int cb = c + CK;
            srcu2 = (cb < p.c1) ? p.x1 + (size_t)cb * sizeof(T) : p.x2 + (size_t)(cb - p.c1) * sizeof(T);
        }
        // per-lane byte offset of every slot from the tile's base, and the validity masks (bit i: halo slot i, bit 16 + i: dY slot i)
        unsigned xoff[MAXP], doff[8];
        unsigned mlo[3], mhi[3], mvalid = dsw_ok ? 0x00FF0000u : 0u;
        {
            const int gdL = (p.tiles_d - 1) * p.TD - (KD / 2), ghL = (p.tiles_h - 1) * p.TH * p.sh - p.pad_h, gwL = (p.tiles_w - 1) * p.TW * p.sw - p.pad_w;
            const int odL = p.Do - (p.tiles_d - 1) * p.TD, ohL = p.Ho - (p.tiles_h - 1) * p.TH, owL = p.Wo - (p.tiles_w - 1) * p.TW;
#pragma unroll
            for (int k = 0; k < 3; ++k) { mlo[k] = 0xFFFFFFFFu; mhi[k] = 0xFFFFFFFFu; }
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int id = sdec[i] >> 20, ih = (sdec[i] >> 10) & 1023, iw = sdec[i] & 1023;
                const unsigned bit = 1u << i;
                if (sdec[i] >= 0) mvalid |= bit;
                if (id < (KD / 2)) mlo[0] &= ~bit;
                if (ih < p.pad_h) mlo[1] &= ~bit;
                if (iw < p.pad_w) mlo[2] &= ~bit;
                if (gdL + id >= p.D) mhi[0] &= ~bit;
                if (ghL + ih >= p.H) mhi[1] &= ~bit;
                if (gwL + iw >= p.W) mhi[2] &= ~bit;
                const int cs_i = (PAIRC && i >= 4 && i < 8) ? cs2 : cs;
                xoff[i] = (unsigned)srel[i] * (unsigned)(cs_i * (int)sizeof(T)) + (unsigned)piece * 16u;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pd = ddec[i] >> 20, ph = (ddec[i] >> 10) & 1023, pw = ddec[i] & 1023;
                const unsigned bit = 0x10000u << i;
                if (pd >= odL) mhi[0] &= ~bit;
                if (ph >= ohL) mhi[1] &= ~bit;
                if (pw >= owL) mhi[2] &= ~bit;
                doff[i] = (unsigned)drel[i] * (unsigned)(p.dyw * (int)sizeof(T)) + (unsigned)(dsw_ok ? co0 + dsw * PE : 0) * (unsigned)sizeof(T);
            }
        }
        // the tile whose addresses are being set up (wave-uniform: scalar unit)
        int a_n = 0, a_d = 0, a_h = 0, a_w = 0, a_tl = tile0;
        const char* xt = srcu;
        const char* xt2 = srcu2;
        const char* dt = p.dy;
        unsigned mk = 0;
        auto tile_first = [&](int tl) {
            int t = tl;
            a_w = t % p.tiles_w; t /= p.tiles_w;
            a_h = t % p.tiles_h; t /= p.tiles_h;
            a_d = t % p.tiles_d;
            a_n = t / p.tiles_d;
            a_tl = tl;
        };
        // (branch-free - selects on the scalar unit: a branch would split the k-loop's basic block and with it the schedule - and in
        //  pieces small enough to ride behind one MFMA each; past the end the walk stays on the last tile: a harmless re-read)
        // (the carries as scalar compare + select in asm - the compiler turns a boolean into an integer on the vector unit - which also
        //  pins every piece where it is written)
        int t_cw = 0, t_ch = 0;
        auto s_eq = [](int a, int b) { int r; asm volatile("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; };
        auto s_lt = [](int a, int b) { int r; asm volatile("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc"); return r; };
        auto tile_next_w = [&]() {
            const int adv = s_lt(a_tl + 1, tile1);
            const int w1 = a_w + adv;                      // (not advancing: a_w < tiles_w stays, no carry)
            t_cw = s_eq(w1, p.tiles_w);
            a_tl += adv;
            a_w = w1 - t_cw * p.tiles_w;
            asm volatile("" : "+s"(a_w), "+s"(a_tl));
        };
        auto tile_next_h = [&]() {
            const int h1 = a_h + t_cw;
            t_ch = s_eq(h1, p.tiles_h);
            a_h = h1 - t_ch * p.tiles_h;
            asm volatile("" : "+s"(a_h));
        };
        auto tile_next_d = [&]() {
            const int d1 = a_d + t_ch;
            const int cd = s_eq(d1, p.tiles_d);
            a_n += cd;
            a_d = d1 - cd * p.tiles_d;
            asm volatile("" : "+s"(a_n), "+s"(a_d));
        };
        int t_b = 0, t_db = 0;
        auto tile_base_x0 = [&]() {
            const int gd = a_d * p.TD - (KD / 2), gh = a_h * p.TH * p.sh - p.pad_h, gw = a_w * p.TW * p.sw - p.pad_w;
            t_b = ((a_n * p.D + gd) * p.H + gh) * p.W + gw;
            asm volatile("" : "+s"(t_b));
        };
        auto tile_base_x1 = [&]() {
            xt = srcu + (ptrdiff_t)t_b * (ptrdiff_t)(cs * (int)sizeof(T));
            // (pure arithmetic with its only use in the NEXT iteration: without a pin the compiler sinks it - and the slot addresses
            //  below - out of the k-loop into the latch block, a serial run again)
            asm volatile("" : "+s"(xt));
            if constexpr (PAIRC) {
                xt2 = srcu2 + (ptrdiff_t)t_b * (ptrdiff_t)(cs2 * (int)sizeof(T));
                asm volatile("" : "+s"(xt2));
            }
        };
        auto tile_base_dy0 = [&]() {
            t_db = ((a_n * p.Do + a_d * p.TD) * p.Ho_out + a_h * p.TH * p.oy_mul + p.oy_add) * p.Wo_out + a_w * p.TW * p.ox_mul + p.ox_add;
            asm volatile("" : "+s"(t_db));
        };
        auto tile_base_dy1 = [&]() {
            dt = p.dy + (ptrdiff_t)t_db * (ptrdiff_t)(p.dyw * (int)sizeof(T));
            asm volatile("" : "+s"(dt));
        };
        unsigned t_m = 0;
        auto tile_mask0 = [&]() {
            unsigned m = mvalid;
            m &= (a_d == 0) ? mlo[0] : 0xFFFFFFFFu;
            m &= (a_h == 0) ? mlo[1] : 0xFFFFFFFFu;
            m &= (a_w == 0) ? mlo[2] : 0xFFFFFFFFu;
            t_m = m;
            asm volatile("" : "+v"(t_m));
        };
        auto tile_mask1 = [&]() {
            unsigned m = t_m;
            m &= (a_d == p.tiles_d - 1) ? mhi[0] : 0xFFFFFFFFu;
            m &= (a_h == p.tiles_h - 1) ? mhi[1] : 0xFFFFFFFFu;
            m &= (a_w == p.tiles_w - 1) ? mhi[2] : 0xFFFFFFFFu;
            mk = m;
            asm volatile("" : "+v"(mk));
        };
        const char* ax[MAXP];
        const char* ad[8];
        auto addr_slot = [&](auto SL) {                                    // global source of slot SL of the tile set up last
            constexpr int sl = decltype(SL)::value;
            if constexpr (sl < MAXP) {
                const char* a = ((PAIRC && sl >= 4 && sl < 8) ? xt2 : xt) + xoff[sl];
                ax[sl] = ((mk >> sl) & 1u) ? a : zpage;
                asm volatile("" : "+v"(ax[sl]));
            } else {
                constexpr int i = sl - MAXP;
                const char* a = dt + doff[i];
                ad[i] = ((mk >> (16 + i)) & 1u) ? a : zpage;
                asm volatile("" : "+v"(ad[i]));
            }
        };
        constexpr int NSLOT = MAXP + 8;
        // LDS-DMA as inline asm (cdna_hip_programming.md 5.7): through the builtin the compiler treats the DMA as a store that may
        // alias every later ds_read and drains vmcnt(0) in front of the first fragment read of the phase - the whole HBM -> LDS
        // latency serial again.  As asm it is invisible to the compiler's wait bookkeeping; the one wait it needs (all of them
        // landed before the buffer is published) is the explicit vmcnt(0) in front of the end-of-phase barrier below.
        const unsigned ldsw = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + (unsigned)wave * 1024u);
        // M0 (the DMA's LDS base) is left as set: nothing the compiler emits for this kernel reads M0 (no LDS-direct / GWS / movrel;
        // tests/test_cabi.py checks the disassembly of the built library for exactly that), and the save / restore pair around every
        // DMA was 36 scalar instructions per tile in MFMA gaps that have room for seven.
        auto glds16 = [&](const char* gsrc, unsigned lds_dst) {            // wave-uniform LDS base + 16 bytes x lane
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(gsrc), "s"(lds_dst) : "memory");
        };
        auto dma_slot = [&](auto SL, int wb) {
            constexpr int sl = decltype(SL)::value;
            if constexpr (sl < MAXP) glds16(ax[sl], ldsw + wb + sl * 4096);
            else glds16(ad[sl - MAXP], ldsw + wb + XBUF + (sl - MAXP) * 4096);
        };
        // What rides behind MFMA g of the G a wave issues per tile: the DMAs over the first 9/16 of the phase (so that the last of them
        // has the rest of it to land), then - after the nine scalar / mask pieces of the tile walk - the slot addresses of tile t+2
        // (slot s only after its DMA of tile t+1 has been issued: it overwrites the address register).
        auto filler = [&](auto GI, auto GN, int wb) {
            constexpr int g = decltype(GI)::value, G = decltype(GN)::value;
            constexpr int GD = (G * 9 + 15) / 16;
            constexpr int GA0 = G >= 16 ? 10 : 0;
            static_for<NSLOT>([&](auto SL) {
                if constexpr ((decltype(SL)::value * GD) / NSLOT == g) dma_slot(SL, wb);
            });
            if constexpr (g == 0) tile_next_w();
            if constexpr (g == (GA0 ? 1 : 0)) tile_next_h();
            if constexpr (g == (GA0 ? 2 : 0)) tile_next_d();
            if constexpr (g == (GA0 ? 3 : 0)) tile_base_x0();
            if constexpr (g == (GA0 ? 4 : 0)) tile_base_x1();
            if constexpr (g == (GA0 ? 5 : 0)) tile_base_dy0();
            if constexpr (g == (GA0 ? 6 : 0)) tile_base_dy1();
            if constexpr (g == (GA0 ? 8 : 0)) tile_mask0();
            if constexpr (g == (GA0 ? 9 : 0)) tile_mask1();
            static_for<NSLOT>([&](auto SL) {
                if constexpr (GA0 + (decltype(SL)::value * (G - GA0)) / NSLOT == g) addr_slot(SL);
            });
        };
        using I0_ = std::integral_constant<int, 0>;
        using I1_ = std::integral_constant<int, 1>;
        if (tile0 < tile1) {
            tile_first(tile0);
            tile_base_x0();
            tile_base_x1();
            tile_base_dy0();
            tile_base_dy1();
            tile_mask0();
            tile_mask1();
            static_for<NSLOT>([&](auto SL) { addr_slot(SL); });
            filler(I0_{}, I1_{}, 0);                        // all DMAs of the first tile, then the addresses of the second
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // LDS row offset of position pp of the tile = sum over the set bits k of pp of C_k (the pw / ph / pd fields are disjoint bit
        // ranges of pp): xrow_of(j, tt) = xb[tt] + sum of kc[] over the bits of j.  kc[] is wave-uniform (scalar registers), so the
        // 32 per-k-step row offsets of the single-buffered path shrink to two vector registers.
        auto bitc = [&](int k) {
            return (k < p.lgTW ? (p.sw << k) : k < p.lgTW + p.lgTH ? (p.sh * p.IW) << (k - p.lgTW) : (p.IH * p.IW) << (k - p.lgTW - p.lgTH)) * XP;
        };
        int kc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) kc[k] = bitc(4 + k);
        int xb[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) xb[tt] = xrow_of(0, tt);          // j = 0: the lane's part (+ the wave's k-step base if !TAPSPLIT)
        if constexpr (IS_BF16) {
            // Fragments of k-step j + 1 are read from LDS while the 14 MFMAs of k-step j issue (two register sets): with one wave
            // per SIMD nothing else covers the ~100+ cycles of a transposing read.  Read r of a k-step: the dY halves first (every
            // MFMA of the k-step needs them), then the taps in the order the MFMAs consume them.  The barrier that publishes tile
            // t + 1 sits in the MIDDLE of the last k-step (its operands left LDS a k-step ago, so nobody still reads this buffer;
            // the DMAs were issued in the first 9 k-steps) and the first fragments of tile t + 1 are read behind the remaining
            // MFMAs - the tile seam costs no exposed LDS latency.
            constexpr int NRD = 2 * (MT + TPW), NMM = TPW * MT, G = NKS * NMM;
            constexpr int MB = (NMM - 1) / 2;            // MFMA of the last k-step that the barrier follows
            s16x4_t fa[2][MT][2], fb[2][TPW][2];
            int vt[TPW][2], ab[MT];                      // lane bases of the transposing reads: halo (per tap and half), dY (per cout half)
            auto set_bases = [&](int buf) {
#pragma unroll
                for (int ti = 0; ti < TPW; ++ti)
#pragma unroll
                    for (int tt = 0; tt < (GEO == 1 ? 1 : 2); ++tt) vt[ti][tt] = xb[tt] + (tapoff[ti] + buf);
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) ab[mi] = arow[mi] + buf + XBUF;
            };
            auto rd_one = [&](auto J, auto R) {
                constexpr int j = decltype(J)::value, r = decltype(R)::value, s_ = j & 1, h = r & 1;
                if constexpr (r < 2 * MT) {
                    fa[s_][r >> 1][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4_t*)(smem + (ab[r >> 1] + (j * 16 + 4 * h) * DYP)));
                } else {
                    constexpr int ti = (r - 2 * MT) >> 1;
                    if constexpr (GEO == 1) {
                        // position bits 4, 5 (ph) -> 20, 40 halo rows; bits 6, 7 (pd) -> 100, 200; bit 2 (pw, the second half) -> 4
                        constexpr int cj = (((j & 1) ? 20 : 0) + ((j & 2) ? 40 : 0) + ((j & 4) ? 100 : 0) + ((j & 8) ? 200 : 0) + 4 * h) * XP;
                        fb[s_][ti][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + vt[ti][0] + cj));
                    } else {
                        const int sj = ((j & 1) ? kc[0] : 0) + ((j & 2) ? kc[1] : 0) + ((j & 4) ? kc[2] : 0) + ((j & 8) ? kc[3] : 0);   // scalar
                        fb[s_][ti][h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + (vt[ti][h] + sj)));
                    }
                }
            };
            auto frag = [&](const s16x4_t (&f)[2]) {
                uint4 u;
                u.x = (uint32_t)(uint16_t)f[0][0] | ((uint32_t)(uint16_t)f[0][1] << 16);
                u.y = (uint32_t)(uint16_t)f[0][2] | ((uint32_t)(uint16_t)f[0][3] << 16);
                u.z = (uint32_t)(uint16_t)f[1][0] | ((uint32_t)(uint16_t)f[1][1] << 16);
                u.w = (uint32_t)(uint16_t)f[1][2] | ((uint32_t)(uint16_t)f[1][3] << 16);
                return u;
            };
            // bias gradient = channel sums of dY: this thread's 8 pieces of the tile being reduced (LDS slot tid * 16 of each 4 KB dY
            // slot holds channel piece dsw, written by this wave's own DMA); only the workgroups of input-channel chunk 0 run the
            // copy of the loop that carries it - one piece per two k-steps, read in one and summed behind MFMAs of the next
            uint4 ub[2];
            auto bias_filler = [&](auto GI, int rdb) {
                constexpr int g = decltype(GI)::value;
                static_for<8>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    // piece i owns MFMAs [i G / 8, (i + 1) G / 8): read at a quarter of them, summed in two halves at 0.6 and 0.85
                    constexpr int gr = (i * G + G / 4) / 8, ga = (i * G + (G * 6) / 10) / 8, ga2 = (i * G + (G * 85) / 100) / 8;
                    if constexpr (g == gr) ub[i & 1] = *reinterpret_cast<const uint4*>(smem + rdb + XBUF + tid * 16 + i * 4096);
                    if constexpr (g == ga) {
                        const uint4 u = ub[i & 1];
                        bsum[0] += __uint_as_float(u.x << 16); bsum[1] += __uint_as_float(u.x & 0xFFFF0000u);
                        bsum[2] += __uint_as_float(u.y << 16); bsum[3] += __uint_as_float(u.y & 0xFFFF0000u);
                    }
                    if constexpr (g == ga2) {
                        const uint4 u = ub[i & 1];
                        bsum[4] += __uint_as_float(u.z << 16); bsum[5] += __uint_as_float(u.z & 0xFFFF0000u);
                        bsum[6] += __uint_as_float(u.w << 16); bsum[7] += __uint_as_float(u.w & 0xFFFF0000u);
                    }
                });
            };
            auto tiles = [&](auto BIAS) {
                if (tile0 < tile1) {
                    set_bases(0);
                    static_for<NRD>([&](auto R) { rd_one(I0_{}, R); });
                }
                for (int tl = tile0; tl < tile1; ++tl) {
                    const int rdb = ((tl - tile0) & 1) * BUF, wrb = BUF - rdb;
                    static_for<NKS>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        static_for<NMM>([&](auto M) {
                            constexpr int m = decltype(M)::value, ti = m / MT, mi = m % MT;
                            mma_step<T>(frag(fa[j & 1][mi]), frag(fb[j & 1][ti]), acc[ti][mi]);
                            __builtin_amdgcn_sched_barrier(0);           // (the MFMA first: its fillers issue under it, not in front of it)
                            if constexpr (j + 1 < NKS) {
                                static_for<NRD>([&](auto R) {
                                    if constexpr ((decltype(R)::value * NMM) / NRD == m) rd_one(std::integral_constant<int, j + 1>{}, R);
                                });
                            } else {
                                if constexpr (m == MB) {
                                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile tl + 1 have landed ...
                                    __syncthreads();                                   // ... everyone's have; and nobody reads buffer rdb any more
                                    set_bases(wrb);
                                }
                                if constexpr (m > MB) {                                // k-step 0 of tile tl + 1 (register set 0; this k-step runs on set 1)
                                    static_for<NRD>([&](auto R) {
                                        if constexpr (MB + 1 + (decltype(R)::value * (NMM - MB - 1)) / NRD == m) rd_one(I0_{}, R);
                                    });
                                }
                            }
                            filler(std::integral_constant<int, j * NMM + m>{}, std::integral_constant<int, G>{}, wrb);
                            if constexpr (decltype(BIAS)::value) bias_filler(std::integral_constant<int, j * NMM + m>{}, rdb);
                            __builtin_amdgcn_sched_barrier(0);
                        });
                    });
                }
            };
            static_assert((NKS & 1) == 0, "the last k-step must run on register set 1");
            if (do_bias) tiles(std::true_type{});
            else tiles(std::false_type{});
        } else {
            for (int tl = tile0; tl < tile1; ++tl) {
                const int rdb = ((tl - tile0) & 1) * BUF, wrb = BUF - rdb;
                // exact f32: the same double-buffered DMA staging, the f32 reduction (two taps / chunks per MFMA) on buffer rdb
                constexpr int NIT2_ = TAPSPLIT ? 64 : 16;
                f32_tile(smem + rdb, smem + rdb + XBUF, [&](auto IT) { filler(IT, std::integral_constant<int, NIT2_>{}, wrb); });
                if (do_bias) {
                    // bias gradient = channel sums of dY: this thread's 8 pieces of the tile just reduced (LDS slot tid * 16 of
                    // each 4 KB dY slot holds channel piece dsw); only the workgroups of input-channel chunk 0 (uniform branch)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const uint4 u = *reinterpret_cast<const uint4*>(smem + rdb + XBUF + tid * 16 + i * 4096);
                        bsum[0] += __uint_as_float(u.x); bsum[1] += __uint_as_float(u.y);
                        bsum[2] += __uint_as_float(u.z); bsum[3] += __uint_as_float(u.w);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of tile tl + 1 have landed ...
                __syncthreads();                                   // ... everyone's have; and buffer rdb is free
            }
        }
    } else {
        constexpr bool PF = (MAXP <= 10);      // big-halo (strided) variant: no cross-tile prefetch, it would spill
        if (PF && tile0 < tile1) {
            decode(tile0);
            issue_x(I0{}, IM{});
            issue_dy();
        }
        for (int tl = tile0; tl < tile1; ++tl) {
            if constexpr (PF) {
                __syncthreads();   // previous tile's fragments consumed
                store_x(I0{}, IM{});
                store_dy();
                __syncthreads();
                // next tile's loads fly under this tile's MFMAs (last: harmless re-read)
                decode(min(tl + 1, tile1 - 1));
                issue_x(I0{}, IM{});
                issue_dy();
            } else {
                decode(tl);
                issue_x(I0{}, IH{});
                issue_dy();
                __syncthreads();
                store_x(I0{}, IH{});
                store_dy();
                issue_x(IH{}, IM{});
                store_x(IH{}, IM{});
                __syncthreads();
            }

            // ---- reduce this tile's positions
            if constexpr (IS_BF16) {
                auto kstep = [&](int j, int xr0, int xr1) {
                    uint4 a[MT];
    #pragma unroll
                    for (int mi = 0; mi < MT; ++mi) a[mi] = tr_frag(dyt, arow[mi] + j * 16 * DYP, arow[mi] + (j * 16 + 4) * DYP);
    #pragma unroll
                    for (int ti = 0; ti < TPW; ++ti) {
                        const uint4 b = tr_frag(halo, xr0 + tapoff[ti], xr1 + tapoff[ti]);
    #pragma unroll
                        for (int mi = 0; mi < MT; ++mi) mma_step<T>(a[mi], b, acc[ti][mi]);
                    }
                };
                if constexpr (KEEP_REL) {
    #pragma unroll
                    for (int j = 0; j < NKS; ++j) kstep(j, xrow[j][0], xrow[j][1]);
                } else {
    #pragma unroll 2
                    for (int j = 0; j < NKS; ++j) kstep(j, xrow_of(j, 0), xrow_of(j, 1));   // rolled: keeps the decode out of registers
                }
            } else {
                f32_tile(halo, dyt, [](auto) {});
            }
        }

    }

    // ---- bias gradient: threads tid = dpiece (mod 8) hold sums of the same channels -> LDS -> one atomic per channel
    if (do_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);               // [256][DPE]
        // (LDS-DMA path: this thread summed channel piece dpiece ^ 4 when bit 1 of its dY rows is set - the half-swap lives on the
        //  source side there; slot it under the piece it really holds)
        const int bslot = DMA ? ((tid & ~7) | (dpiece ^ (((tid >> 4) & 1) << 2))) : tid;
#pragma unroll
        for (int e = 0; e < DPE; ++e) red[bslot * DPE + e] = bsum[e];
        __syncthreads();
        if (tid < 8 * DPE) {
            const int piece = tid / DPE, e = tid % DPE;
            float accv = 0.0f;
            for (int q = 0; q < 32; ++q) accv += red[(q * 8 + piece) * DPE + e];
            const int co = co0 + piece * DPE + e;
            RHO_WG_FLUSH_ARGS();
            if (kq.slab != nullptr) {
                // (where the waves own slabs of their own, the workgroup's channel sums go to the first and zeros to the rest)
                if (co < kq.coutp) {
                    float* const bs = kq.slab + (size_t)(TAPSPLIT ? bx : bx * 4) * (size_t)kq.slab_stride + (size_t)kq.slab_bias_off + co;
                    bs[0] = co < kq.dyw ? accv : 0.0f;
                    if constexpr (!TAPSPLIT) { bs[kq.slab_stride] = 0.0f; bs[2 * kq.slab_stride] = 0.0f; bs[3 * kq.slab_stride] = 0.0f; }
                }
            } else if (co < kq.dyw && co < kq.coutp) atomicAdd(kq.dbias + co, accv);
        }
    }

    // ---- flush: lane holds ci = c + (lane&31), rows co0 + 32*mi + (r&3) + 8*(r>>2) + 4*half
    // (deterministic mode: slab of this accumulator's owner - the workgroup, or the wave where waves split the positions of a tile)
    RHO_WG_FLUSH_ARGS();
    const bool fdet = kq.slab != nullptr;
    float* const fbase = fdet ? kq.slab + (size_t)(TAPSPLIT ? bx : bx * 4 + wave) * (size_t)kq.slab_stride : kq.dw;
    if constexpr (IS_BF16) {
        const int ci = c + (lane & 31);
#pragma unroll
        for (int ti = 0; ti < TPW; ++ti) {
            if (tap_of[ti] < NT) {
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (co < kq.coutp) wg_flush(fdet, fbase, ((size_t)tap_of[ti] * kq.coutp + co) * kq.cin + ci, acc[ti][mi][r]);
                    }
            }
        }
    } else {
        // f32: lane column (lane & 31) = channel c + (column & 15) at tap 2s (columns 0-15) / 2s + 1 (columns 16-31)
        const int ci = c + (lane & 15);
        const bool upper = (lane & 31) >= 16;
#pragma unroll
        for (int s_ = 0; s_ < NACC; ++s_) {
            int tap = upper ? ((2 * s_ + 1 < TPW) ? tap_of[(2 * s_ + 1 < TPW) ? 2 * s_ + 1 : 0] : NT) : tap_of[2 * s_];
            int cif = ci;
            if (PAIRC && has2 && upper) { tap = tap_of[0]; cif = ci + CK; }      // 1x1x1 pair mode: the same tap, the next chunk
            if (tap < NT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (co < kq.coutp) wg_flush(fdet, fbase, ((size_t)tap * kq.coutp + co) * kq.cin + cif, acc[s_][0][r]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 1x1x1 (bf16)
// dW[co][ci] = sum_pos dY[pos][co] * X[pos][ci]: a plain GEMM with the positions as K.  The generic kernel above gives a
// 1-tap convolution 8 MFMAs per wave and staged tile (100 - 180 TFLOP/s, 24 ms per training step for 27 launches);
// here a workgroup owns 64 couts x 128 cins (4 chunks, one per wave), stages 128 positions of X (4 planes of 64-byte
// rows) and dY per iteration with the next tile's loads in flight, and each wave runs 16 MFMAs per tile on transposing
// reads of the unpadded planes.  48 KB LDS: three workgroups per CU.
__global__ __launch_bounds__(256) void k_wgrad1(const WgradK p, long long npos, int tiles_total, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = 128;                            // positions per tile
    char* const xpl = smem;                            // [4 chunks][TP rows][64 B]
    char* const dyt = smem + 4 * TP * XP;              // [TP rows][128 B], halves swapped on rows with bit 1 set
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    // (slab -> XCD map as in k_wgrad: the cout-tile x channel-group workgroups of a position slab side by side on one XCD)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_map) {
        const int L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const int pairs = gridDim.y * gridDim.z;
        const int e = L & 7, k = L >> 3;
        const int pi = k % pairs;
        bx = (k / pairs) * 8 + e;
        bz = pi % (int)gridDim.z;
        by = pi / (int)gridDim.z;
    }
    const int co0 = by * 64;
    const int cg0 = bz * 128;                          // first input channel of this workgroup's group
    const int nchunk = min(4, (p.cin - cg0) / 32);     // chunks that exist (cin is a multiple of 32)

    // staging geometry: X pieces idx = tid + 256*i -> (row = idx/16, chunk = (idx%16)/4, piece = idx%4)
    const int xrow0 = tid >> 4, xpc = tid & 15, xch = xpc >> 2, xpiece = xpc & 3;
    const int xc = cg0 + xch * 32;                     // channel of this thread's pieces
    const bool xch_ok = xch < nchunk;
    const char* xsrc;
    int xcs;
    {
        const int cc = xch_ok ? xc : cg0;
        if (cc < p.c1) { xsrc = p.x1 + ((size_t)cc * 2 + xpiece * 16); xcs = p.c1; }
        else { xsrc = p.x2 + ((size_t)(cc - p.c1) * 2 + xpiece * 16); xcs = p.c2; }
    }
    const int drow0 = tid >> 3, dpiece = tid & 7;
    const bool dch_ok = (co0 + dpiece * 8) < p.dyw;
    const char* const dsrc = p.dy + (size_t)(dch_ok ? co0 + dpiece * 8 : 0) * 2;

    f32x16_t acc[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.0f;
    const bool do_bias = (p.dbias != nullptr) && (bz == 0);
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.0f;

    const int grp = lane >> 4, li = lane & 15, q = li >> 2, pq = li & 3;
    const int colb = 16 * (grp & 1) + 4 * pq;
    const int rbase = 8 * (grp >> 1) + q;              // k-row of this lane within a 16-position k-step (t = 0)
    const int xoff = wave * TP * XP + rbase * XP + colb * 2;
    int aoff[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) aoff[mi] = rbase * DYP + ((mi * 64) ^ dy_swz(q)) + colb * 2;

    uint4 xv[8], dv[4];
    auto issue = [&](int tl) {
        const long long pos0 = (long long)tl * TP;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            long long pos = pos0 + xrow0 + 16 * i;
            if (pos >= npos) pos = npos - 1;                           // clamped, zeroed at store time
            xv[i] = *reinterpret_cast<const uint4*>(xsrc + (size_t)pos * xcs * 2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            long long pos = pos0 + drow0 + 32 * i;
            if (pos >= npos) pos = npos - 1;
            dv[i] = *reinterpret_cast<const uint4*>(dsrc + (size_t)pos * p.dyw * 2);
        }
    };
    const int tile0 = bx * tiles_per_block;
    const int tile1 = min(tile0 + tiles_per_block, tiles_total);
    if (tile0 < tile1) issue(tile0);
    for (int tl = tile0; tl < tile1; ++tl) {
        const long long pos0 = (long long)tl * TP;
        __syncthreads();                                               // previous tile's fragments consumed
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = xrow0 + 16 * i;
            const bool ok = xch_ok && (pos0 + row < npos);
            *reinterpret_cast<uint4*>(xpl + xch * TP * XP + row * XP + xpiece * 16) = ok ? xv[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = drow0 + 32 * i;
            const bool ok = dch_ok && (pos0 + row < npos);
            const uint4 u = ok ? dv[i] : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(dyt + row * DYP + ((dpiece * 16) ^ dy_swz(row))) = u;
            if (do_bias) {
                bsum[0] += __uint_as_float(u.x << 16); bsum[1] += __uint_as_float(u.x & 0xFFFF0000u);
                bsum[2] += __uint_as_float(u.y << 16); bsum[3] += __uint_as_float(u.y & 0xFFFF0000u);
                bsum[4] += __uint_as_float(u.z << 16); bsum[5] += __uint_as_float(u.z & 0xFFFF0000u);
                bsum[6] += __uint_as_float(u.w << 16); bsum[7] += __uint_as_float(u.w & 0xFFFF0000u);
            }
        }
        __syncthreads();
        issue(min(tl + 1, tile1 - 1));                                 // next tile's loads fly under this tile's MFMAs
        if (wave < nchunk) {
#pragma unroll
            for (int j = 0; j < TP / 16; ++j) {
                const uint4 b = tr_frag(xpl, xoff + j * 16 * XP, xoff + (j * 16 + 4) * XP);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const uint4 a = tr_frag(dyt, aoff[mi] + j * 16 * DYP, aoff[mi] + (j * 16 + 4) * DYP);
                    mma_step<bf16_raw>(a, b, acc[mi]);
                }
            }
        }
    }
    if (do_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);                   // [256][8]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 64) {
            const int piece = tid >> 3, e = tid & 7;
            float accv = 0.0f;
            for (int qq = 0; qq < 32; ++qq) accv += red[(qq * 8 + piece) * 8 + e];
            const int co = co0 + piece * 8 + e;
            if (p.slab != nullptr) { if (co < p.coutp) p.slab[(size_t)bx * (size_t)p.slab_stride + (size_t)p.slab_bias_off + co] = co < p.dyw ? accv : 0.0f; }
            else if (co < p.dyw && co < p.coutp) atomicAdd(p.dbias + co, accv);
        }
    }
    if (wave < nchunk) {
        const int ci = cg0 + wave * 32 + (lane & 31);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (co < p.coutp) wg_flush(p.slab != nullptr, p.slab != nullptr ? p.slab + (size_t)bx * (size_t)p.slab_stride : p.dw, (size_t)co * p.cin + ci, acc[mi][r]);
            }
    }
}

namespace {
using namespace rho_conv;

struct VariantOut {
    char* buf;
    int cap;
};
thread_local VariantOut* g_wvariant = nullptr;      // see rho_conv_variant (conv.hip)

template <typename T, int KD, int KH, int KW>
int launch_wgrad(const WgradK& k, int maxp, dim3 grid, size_t lds, hipStream_t st) {
    // the compile-time tile geometry of k_wgrad<..., GEO = 1>
    const bool geo488 = k.TD == 4 && k.TH == 8 && k.TW == 8 && k.IH == 10 && k.IW == 10 && k.sh == 1 && k.sw == 1;
    if (g_wvariant != nullptr) {
        snprintf(g_wvariant->buf, (size_t)g_wvariant->cap, "k_wgrad<%s,%d,%d,%d,MAXP=%d,PRE=%d%s>", sizeof(T) == 2 ? "bf16" : "f32", KD, KH,
                 KW, maxp <= 10 ? 10 : 28, k.pre_a != nullptr ? 1 : 0,
                 (geo488 && sizeof(T) == 2 && KD == 3 && KH == 3 && KW == 3 && maxp <= 10 && k.pre_a == nullptr) ? ",GEO=1" : "");
        return 0;
    }
    auto go = [&](auto fn) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(fn, grid, dim3(256), lds, st, k);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    };
    const bool pre = k.pre_a != nullptr;
    if constexpr (KH == 2 || KW == 2) {      // sub-pixel phases: stride 1, small halo, no prologue (checked by the caller)
        if (maxp > 10 || pre) return RHO_E_SHAPE;
        return go(k_wgrad<T, KD, KH, KW, 10, false>);
    }
    if constexpr (sizeof(T) == 2 && KD == 3 && KH == 3 && KW == 3) {
        if (geo488 && !pre && maxp <= 10) return go(k_wgrad<T, KD, KH, KW, 10, false, 1>);
    }
    if (maxp <= 10) return pre ? go(k_wgrad<T, KD, KH, KW, 10, true>) : go(k_wgrad<T, KD, KH, KW, 10, false>);
    return pre ? go(k_wgrad<T, KD, KH, KW, 28, true>) : go(k_wgrad<T, KD, KH, KW, 28, false>);
}

template <typename T>
int launch_wgrad_taps(const rho_conv_desc& d, const WgradK& k, int maxp, dim3 grid, size_t lds, hipStream_t st) {
    if (d.kd == 3 && d.kh == 3 && d.kw == 3) return launch_wgrad<T, 3, 3, 3>(k, maxp, grid, lds, st);
    if (d.kd == 1 && d.kh == 3 && d.kw == 3) return launch_wgrad<T, 1, 3, 3>(k, maxp, grid, lds, st);
    if (d.kd == 1 && d.kh == 1 && d.kw == 3) return launch_wgrad<T, 1, 1, 3>(k, maxp, grid, lds, st);
    if (d.kd == 1 && d.kh == 1 && d.kw == 1) return launch_wgrad<T, 1, 1, 1>(k, maxp, grid, lds, st);
    // sub-pixel phases of a conv behind a nearest x2 upsample
    if (d.kd == 3 && d.kh == 2 && d.kw == 2) return launch_wgrad<T, 3, 2, 2>(k, maxp, grid, lds, st);
    if (d.kd == 1 && d.kh == 2 && d.kw == 2) return launch_wgrad<T, 1, 2, 2>(k, maxp, grid, lds, st);
    if (d.kd == 1 && d.kh == 1 && d.kw == 2) return launch_wgrad<T, 1, 1, 2>(k, maxp, grid, lds, st);
    return RHO_E_ARG;
}

inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
}  // namespace

// desc describes the FORWARD convolution (x1/x2, prologue, geometry); dy is its output gradient
// (channels-last, row width dy_width >= cout, extra channels must be zero); dw is an fp32 buffer
// [taps][coutp][c1+c2] that this call ACCUMULATES into (zero it first).  up_h/up_w are not supported:
// materialise the upsampled input (rho_upsample2x) and pass it as x1.
static int wgrad_impl(const rho_conv_desc* dp, const void* dy, int64_t dy_width, float* dw, float* dbias, void* stream,
                      float* ws = nullptr, int64_t ws_bytes = 0, int64_t* ws_want = nullptr);

extern "C" int rho_conv_nd_wgrad(const rho_conv_desc* dp, const void* dy, int64_t dy_width, float* dw, float* dbias, void* stream) {
    if (!dp || !dy || !dw) return RHO_E_ARG;
    return wgrad_impl(dp, dy, dy_width, dw, dbias, stream);
}

// Deterministic form: the same launch, flushed through ordered slabs in `ws` instead of fp32 atomics (bit-reproducible run to run
// and across data-parallel replicas).  ws must hold rho_conv_wgrad_workspace_bytes(desc, dy_width) bytes; it is scratch.
extern "C" int rho_conv_nd_wgrad_ws(const rho_conv_desc* dp, const void* dy, int64_t dy_width, float* dw, float* dbias, void* ws,
                                    int64_t ws_bytes, void* stream) {
    if (!dp || !dy || !dw || !ws || ws_bytes <= 0) return RHO_E_ARG;
    return wgrad_impl(dp, dy, dy_width, dw, dbias, stream, (float*)ws, ws_bytes);
}

extern "C" int64_t rho_conv_wgrad_workspace_bytes(const rho_conv_desc* dp, int64_t dy_width) {
    if (!dp) return 0;
    int64_t b = 0;
    return wgrad_impl(dp, nullptr, dy_width, nullptr, nullptr, nullptr, nullptr, 0, &b) == 0 ? b : 0;
}

// dw[i] (+ dbias) += slab[0][i] + slab[1][i] + ... in slab order: the fixed summation order of the deterministic flush
__global__ __launch_bounds__(256) void k_wgrad_slab_reduce(const float* __restrict__ slab, int nslab, long long stride, float* __restrict__ dw,
                                                           long long nw, float* __restrict__ dbias, int nb) {
    const long long total = nw + (dbias ? nb : 0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float* dst = i < nw ? dw + i : dbias + (i - nw);
        float v = *dst;
        const float* sp = slab + i;
        for (int s_ = 0; s_ < nslab; ++s_) v += sp[(long long)s_ * stride];
        *dst = v;
    }
}

static int launch_slab_reduce(const float* ws, int nslab, long long stride, float* dw, long long nw, float* dbias, int nb, hipStream_t st) {
    long long g = (nw + nb + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_wgrad_slab_reduce, dim3((unsigned)g), dim3(256), 0, st, ws, nslab, stride, dw, nw, dbias, nb);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

extern "C" int rho_conv_wgrad_variant(const rho_conv_desc* dp, int64_t dy_width, char* buf, int cap) {
    if (!dp || !buf || cap < 64) return RHO_E_ARG;
    buf[0] = 0;
    VariantOut vo{buf, cap};
    g_wvariant = &vo;
    const int rc = wgrad_impl(dp, nullptr, dy_width, nullptr, nullptr, nullptr);
    g_wvariant = nullptr;
    return rc;
}

static int wgrad_impl(const rho_conv_desc* dp, const void* dy, int64_t dy_width, float* dw, float* dbias, void* stream, float* ws,
                      int64_t ws_bytes, int64_t* ws_want) {
    const rho_conv_desc& d = *dp;
    if (!d.x1) return RHO_E_ARG;
    if (d.dtype != RHO_F32 && d.dtype != RHO_BF16) return RHO_E_ARG;
    if (d.up_h || d.up_w || d.zs_h || d.zs_w) return RHO_E_ARG;
    const int CK = d.dtype == RHO_BF16 ? 32 : 16;
    const int PE = d.dtype == RHO_BF16 ? 8 : 4;
    const int COT = d.dtype == RHO_BF16 ? 64 : 32;
    const int c2 = d.x2 ? d.c2 : 0;
    const int cin = d.c1 + c2;
    if (d.c1 <= 0 || d.c1 % CK || c2 % CK) return RHO_E_ALIGN;
    if (dy_width <= 0 || dy_width % PE || d.coutp <= 0 || d.coutp % 32) return RHO_E_ALIGN;
    if ((d.sh != 1 && d.sh != 2) || (d.sw != 1 && d.sw != 2)) return RHO_E_ARG;
    if (d.pre_a && !d.pre_b) return RHO_E_ARG;

    if (d.phd_h || d.phd_w || d.ph_h < 0 || d.ph_h > 2 || d.ph_w < 0 || d.ph_w > 2) return RHO_E_ARG;
    if ((d.ph_h && (d.kh != 2 || d.sh != 1)) || (d.ph_w && (d.kw != 2 || d.sw != 1)) || (!d.ph_h && d.kh == 2) || (!d.ph_w && d.kw == 2) ||
        ((d.ph_h || d.ph_w) && (d.pre_a || d.kd == 2)))
        return RHO_E_ARG;
    // (phases: the launch's output grid is the source grid; dY is the full-resolution gradient, read at one parity)
    const int ho = d.ph_h ? d.h : (d.h + 2 * (d.kh / 2) - d.kh) / d.sh + 1;
    const int wo = d.ph_w ? d.w_ : (d.w_ + 2 * (d.kw / 2) - d.kw) / d.sw + 1;
    WgradK k{};
    k.pad_h = d.ph_h ? 2 - d.ph_h : d.kh / 2; k.pad_w = d.ph_w ? 2 - d.ph_w : d.kw / 2;
    k.oy_mul = d.ph_h ? 2 : 1; k.oy_add = d.ph_h ? d.ph_h - 1 : 0;
    k.ox_mul = d.ph_w ? 2 : 1; k.ox_add = d.ph_w ? d.ph_w - 1 : 0;
    int nb = d.n;
    if (d.kd == 1 && d.kh == 1 && d.kw == 1) {
        if (d.sh != 1 || d.sw != 1) return RHO_E_ARG;
        k.D = 1; k.H = 1; k.W = d.n * d.d * d.h * d.w_;
        k.Do = 1; k.Ho = 1; k.Wo = k.W;
        nb = 1;
    } else if (d.kd == 1) {
        k.D = d.n * d.d; k.H = d.h; k.W = d.w_;
        k.Do = k.D; k.Ho = ho; k.Wo = wo;
        nb = 1;
    } else {
        k.D = d.d; k.H = d.h; k.W = d.w_;
        k.Do = d.d; k.Ho = ho; k.Wo = wo;
    }
    if ((long long)d.n * d.d * d.h * d.w_ * k.oy_mul * k.ox_mul >= (1LL << 31)) return RHO_E_SHAPE;
    k.S_in = (long long)d.d * d.h * d.w_;
    k.Ho_out = k.Ho * k.oy_mul; k.Wo_out = k.Wo * k.ox_mul;

    if (d.dtype == RHO_BF16 && d.kd == 1 && d.kh == 1 && d.kw == 1 && !d.pre_a) {
        // 1x1x1 without prologue: the GEMM-shaped kernel
        WgradK k1{};
        k1.x1 = (const char*)d.x1; k1.x2 = (const char*)d.x2; k1.dy = (const char*)dy; k1.dw = dw; k1.dbias = dbias;
        k1.c1 = d.c1; k1.c2 = c2; k1.cin = cin; k1.dyw = (int)dy_width; k1.coutp = d.coutp;
        const long long npos = (long long)d.n * d.d * d.h * d.w_;
        const int tiles_total = (int)((npos + 127) / 128);
        const int pairs = cdiv(d.coutp, 64) * cdiv(cin, 128);
        int splits = cdiv(1536, pairs);
        if (splits >= 8) splits &= ~7;
        if (splits < 1) splits = 1;
        if (splits > tiles_total) splits = tiles_total;
        const int tpb = cdiv(tiles_total, splits);
        splits = cdiv(tiles_total, tpb);
        static const bool xcd_env1 = !(getenv("RHO_WGRAD_XCD") && atoi(getenv("RHO_WGRAD_XCD")) == 0);
        k1.xcd_map = (xcd_env1 && splits % 8 == 0 && (long long)splits * pairs < (1LL << 31)) ? 1 : 0;
        if (cdiv(d.coutp, 64) > 65535 || cdiv(cin, 128) > 65535) return RHO_E_SHAPE;
        dim3 grid((unsigned)splits, (unsigned)cdiv(d.coutp, 64), (unsigned)cdiv(cin, 128));
        const long long nw1 = (long long)d.coutp * cin;
        k1.slab_bias_off = nw1; k1.slab_stride = nw1 + d.coutp;
        if (ws_want) { *ws_want = (int64_t)splits * k1.slab_stride * (int64_t)sizeof(float); return 0; }
        if (g_wvariant != nullptr) {
            snprintf(g_wvariant->buf, (size_t)g_wvariant->cap, "k_wgrad1<bf16>");
            return 0;
        }
        if (ws) {
            if (ws_bytes < (int64_t)splits * k1.slab_stride * (int64_t)sizeof(float)) return RHO_E_ARG;
            k1.slab = ws;
        }
        hipLaunchKernelGGL(k_wgrad1, grid, dim3(256), (size_t)(4 * 128 * XP + 128 * DYP), as_stream(stream), k1, npos, tiles_total, tpb);
        hipError_t e1 = hipGetLastError();
        if (e1 != hipSuccess) return (int)e1;
        return ws ? launch_slab_reduce(ws, splits, k1.slab_stride, dw, nw1, dbias, d.coutp, as_stream(stream)) : 0;
    }
    const size_t lds_cap = 160 * 1024;
    int np_cap = (int)((lds_cap - 256 * DYP) / XP);
    if (np_cap > 28 * 64) np_cap = 28 * 64;
    TileChoice t = choose_tile(d, k.D, k.Do, k.Ho, k.Wo, 640);
    if (!t.ok) t = choose_tile(d, k.D, k.Do, k.Ho, k.Wo, np_cap);
    if (!t.ok) return RHO_E_SHAPE;

    k.x1 = (const char*)d.x1; k.x2 = (const char*)d.x2; k.pre_a = d.pre_a; k.pre_b = d.pre_b;
    k.dy = (const char*)dy; k.dw = dw; k.dbias = dbias;
    k.c1 = d.c1; k.c2 = c2; k.cin = cin; k.dyw = (int)dy_width; k.coutp = d.coutp;
    k.sh = d.sh; k.sw = d.sw; k.pre_silu = d.pre_silu;
    k.TD = t.TD; k.TH = t.TH; k.TW = t.TW; k.ID = t.ID; k.IH = t.IH; k.IW = t.IW; k.NP = t.NP;
    k.lgTW = ilog2(t.TW); k.lgTH = ilog2(t.TH);
    k.tiles_d = cdiv(k.Do, t.TD); k.tiles_h = cdiv(k.Ho, t.TH); k.tiles_w = cdiv(k.Wo, t.TW);
    k.n_batch = nb;
    const long long tiles = (long long)nb * k.tiles_d * k.tiles_h * k.tiles_w;
    if (tiles > 0x7FFFFFFFLL) return RHO_E_SHAPE;
    k.tiles_total = (int)tiles;
    // enough workgroups to fill 256 CUs, long enough slabs to amortise the 27x64x32 fp32 flush
    // One workgroup per CU (144 KB of LDS) and equal slabs: the launch runs in rounds of n_cu workgroups.  Pick the slab count so
    // that splits x pairs fills 3..6 whole rounds (was ceil(1024 / pairs): 6 pairs -> 1026 workgroups = a fifth round of two,
    // 20 % of the launch, on every layer whose channel count is 3 * 2^k - the concatenated inputs of the output blocks).
    // f32 1x1x1: a workgroup takes two input-channel chunks (k_wgrad's PAIRC), so the chunk axis of the grid halves
    const bool pairc = d.dtype == RHO_F32 && d.kd * d.kh * d.kw == 1 && t.NP == 256 && cdiv(t.NP, 64) <= 10;
    k.pairc = pairc ? 1 : 0;
    const int nchunk = pairc ? cdiv(cin / CK, 2) : cin / CK;
    const int pairs = cdiv(d.coutp, COT) * nchunk;
    static const int n_cu = []() { hipDeviceProp_t pr; int dev = 0; return (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256; }();
    int splits = cdiv(4 * n_cu, pairs);
    double best = -1.0;
    static const int r_lo = getenv("RHO_WGRAD_ROUNDS_MIN") ? atoi(getenv("RHO_WGRAD_ROUNDS_MIN")) : 3;      // (A/B knobs)
    static const int r_hi = getenv("RHO_WGRAD_ROUNDS_MAX") ? atoi(getenv("RHO_WGRAD_ROUNDS_MAX")) : 6;
    for (int r = r_lo; r <= r_hi; ++r) {
        int sp = (r * n_cu) / pairs;
        if (sp >= 8) sp &= ~7;                 // multiples of 8: the slab -> XCD map of the kernel needs whole groups of 8 slabs
        if (sp < 1) continue;
        const long long total = (long long)sp * pairs;
        const double eff = (double)total / (double)(cdiv((int)total, n_cu) * n_cu) - 0.002 * (r > 4 ? r - 4 : 4 - r) * (r_lo == 3 && r_hi == 6 ? 1.0 : 0.0);
        if (eff > best) { best = eff; splits = sp; }
    }
    if (splits < 1) splits = 1;
    if (splits > k.tiles_total) splits = k.tiles_total;
    k.tiles_per_block = cdiv(k.tiles_total, splits);
    splits = cdiv(k.tiles_total, k.tiles_per_block);
    if (cdiv(d.coutp, COT) > 65535 || nchunk > 65535) return RHO_E_SHAPE;
    dim3 grid((unsigned)splits, (unsigned)cdiv(d.coutp, COT), (unsigned)nchunk);
    static const bool xcd_env = !(getenv("RHO_WGRAD_XCD") && atoi(getenv("RHO_WGRAD_XCD")) == 0);
    k.xcd_map = (xcd_env && splits % 8 == 0 && (long long)splits * pairs < (1LL << 31)) ? 1 : 0;
    const int maxp = cdiv(t.NP, 64);
    size_t lds = (size_t)(maxp <= 10 ? 10 : 28) * 64 * XP + 256 * DYP;
    if (maxp <= 10 && !d.pre_a) lds *= 2;     // LDS-DMA path (both dtypes since round 3): double-buffered tiles (2 x 72 KB)
    hipStream_t st = as_stream(stream);
    // deterministic flush: one slab per accumulator owner (the kernel's TAPSPLIT decides whether that is the workgroup or the wave)
    const int nt = d.kd * d.kh * d.kw;
    const bool tapsplit = d.dtype == RHO_BF16 ? (nt >= 9) : (nt > 9);
    const int nslab = splits * (tapsplit ? 1 : 4);
    const long long nw = (long long)nt * d.coutp * cin;
    k.slab_bias_off = nw; k.slab_stride = nw + d.coutp;
    if (ws_want) { *ws_want = (int64_t)nslab * k.slab_stride * (int64_t)sizeof(float); return 0; }
    if (ws && g_wvariant == nullptr) {
        if (ws_bytes < (int64_t)nslab * k.slab_stride * (int64_t)sizeof(float)) return RHO_E_ARG;
        k.slab = ws;
    }
    const int rc = d.dtype == RHO_BF16 ? launch_wgrad_taps<bf16_raw>(d, k, maxp, grid, lds, st) : launch_wgrad_taps<float>(d, k, maxp, grid, lds, st);
    if (rc != 0 || !ws || g_wvariant != nullptr) return rc;
    return launch_slab_reduce(ws, nslab, k.slab_stride, dw, nw, dbias, d.coutp, st);
}

// fp32 [taps][coutp][cin_buf] accumulation buffer -> parameter-gradient layout [cout][cin][taps] (fp32),
// undoing the qkv row permutation if any; accumulate = 1 adds to the existing .grad.
__global__ __launch_bounds__(256) void k_wgrad_finalize(const float* __restrict__ dw, float* __restrict__ grad, int64_t cout,
                                                        int64_t cin, int64_t taps, int64_t coutp, int64_t cinb,
                                                        const int32_t* __restrict__ row_src, int accumulate) {
    const int64_t total = cout * cin * taps;   // over buffer rows r (permuted order), written to original rows
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t tap = i % taps;
        const int64_t ci = (i / taps) % cin;
        const int64_t r = i / (taps * cin);
        const int64_t dst_row = row_src ? (int64_t)row_src[r] : r;
        if (dst_row < 0 || dst_row >= cout) continue;
        const float v = dw[(tap * coutp + r) * cinb + ci];
        float* g = grad + (dst_row * cin + ci) * taps + tap;
        *g = accumulate ? *g + v : v;
    }
}

extern "C" int rho_wgrad_finalize(const float* dw, float* grad, int64_t cout, int64_t cin, int64_t taps, int64_t coutp,
                                  int64_t cin_buf, const int32_t* row_src, int accumulate, void* stream) {
    if (!dw || !grad || cout <= 0 || cin <= 0 || taps <= 0 || coutp < cout || cin_buf < cin) return RHO_E_ARG;
    int64_t g = (cout * cin * taps + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_wgrad_finalize, dim3((unsigned)g), dim3(256), 0, as_stream(stream), dw, grad, cout, cin, taps, coutp, cin_buf,
                       row_src, accumulate);
    RHO_LAUNCH_CHECK();
    return 0;
}

// Weight gradient of one sub-pixel phase (buffer [kd * kh' * kw' taps][coutp][cin_buf], kh' / kw' = 2 on the phased axes) into the
// parameter gradient [cout][cin][kd * kh * kw]: original tap (kz, ky, kx) of parity a received the phase tap its row was summed
// into (a = 0: {0}, {1, 2};  a = 1: {0, 1}, {2}), so its gradient is that phase tap's - summed over the phases by accumulate = 1.
__global__ __launch_bounds__(256) void k_wgrad_finalize_phase(const float* __restrict__ dw, float* __restrict__ grad, int64_t cout,
                                                              int64_t cin, int kd, int kh, int kw, int ph_h, int ph_w, int64_t coutp,
                                                              int64_t cinb, int accumulate) {
    const int64_t taps = (int64_t)kd * kh * kw;
    const int kh2 = ph_h ? 2 : kh, kw2 = ph_w ? 2 : kw;
    const int64_t total = cout * cin * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % taps);
        const int64_t ci = (i / taps) % cin;
        const int64_t r = i / (taps * cin);
        const int kx = tap % kw, ky = (tap / kw) % kh, kz = tap / (kw * kh);
        const int r2 = ph_h == 0 ? ky : ph_h == 1 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
        const int c2 = ph_w == 0 ? kx : ph_w == 1 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
        const float v = dw[((int64_t)((kz * kh2 + r2) * kw2 + c2) * coutp + r) * cinb + ci];
        float* g = grad + i;
        *g = accumulate ? *g + v : v;
    }
}

extern "C" int rho_wgrad_finalize_phase(const float* dw, float* grad, int64_t cout, int64_t cin, int kd, int kh, int kw, int ph_h, int ph_w,
                                        int64_t coutp, int64_t cin_buf, int accumulate, void* stream) {
    if (!dw || !grad || cout <= 0 || cin <= 0 || kd <= 0 || kh <= 0 || kw <= 0 || coutp < cout || cin_buf < cin) return RHO_E_ARG;
    if (ph_h < 0 || ph_h > 2 || ph_w < 0 || ph_w > 2 || (!ph_h && !ph_w) || (ph_h && kh != 3) || (ph_w && kw != 3)) return RHO_E_ARG;
    int64_t g = (cout * cin * kd * kh * kw + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_wgrad_finalize_phase, dim3((unsigned)g), dim3(256), 0, as_stream(stream), dw, grad, cout, cin, kd, kh, kw, ph_h, ph_w,
                       coutp, cin_buf, accumulate);
    RHO_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ batched finalize
// rho_wgrad_finalize_batch: the [taps][coutp][cin_buf] accumulation buffers of MANY convolutions (and their channel-sum vectors =
// bias gradients) into the parameter-gradient layout in one launch, from a device table of rho_wfin_op.  The per-tensor calls
// above are ~170 launches of a few microseconds per training step at BASELINE configs[2] and as many on the launch-bound 2-D
// configurations.  kind 0 = rho_wgrad_finalize (weights; a bias is the cin = taps = 1 case), kind 1 = ALL sub-pixel phases of one
// conv behind a nearest x2 upsample (their buffers `phase_stride` floats apart, in the order (a, b) for a in H phases for b in W
// phases) summed into the 3-tap parameter gradient - one op, so no two ops of a launch touch the same gradient element.
__device__ __forceinline__ void wfin_elem(const rho_wfin_op& o, int64_t i) {
    const int64_t taps = (int64_t)o.kd * o.kh * o.kw;
    const int tap = (int)(i % taps);
    const int64_t ci = (i / taps) % o.cin;
    const int64_t r = i / (taps * o.cin);
    if (o.kind == 2) {          // i walks [ci][tap]: grad[ci][tap] += dw[taps - 1 - tap][ci]
        const int64_t cc = i / taps;
        if (cc < o.cin) o.grad[i] += o.dw[(int64_t)(taps - 1 - tap) * o.cinb + cc];
        return;
    }
    if (o.kind == 0) {
        const int64_t dst_row = o.row_src ? (int64_t)o.row_src[r] : r;
        if (dst_row < 0 || dst_row >= o.cout) return;
        float* g = o.grad + (dst_row * o.cin + ci) * taps + tap;
        *g += o.dw[((int64_t)tap * o.coutp + r) * o.cinb + ci];
        return;
    }
    const int kx = tap % o.kw, ky = (tap / o.kw) % o.kh, kz = tap / (o.kw * o.kh);
    const int kh2 = o.up_h ? 2 : o.kh, kw2 = o.up_w ? 2 : o.kw;
    const int nb = o.up_w ? 2 : 1;
    float v = 0.0f;
    for (int a = (o.up_h ? 1 : 0); a <= (o.up_h ? 2 : 0); ++a)
        for (int b = (o.up_w ? 1 : 0); b <= (o.up_w ? 2 : 0); ++b) {
            const int r2 = a == 0 ? ky : a == 1 ? (ky == 0 ? 0 : 1) : (ky == 2 ? 1 : 0);
            const int c2 = b == 0 ? kx : b == 1 ? (kx == 0 ? 0 : 1) : (kx == 2 ? 1 : 0);
            const int pidx = (o.up_h ? a - 1 : 0) * nb + (o.up_w ? b - 1 : 0);
            v += o.dw[(int64_t)pidx * o.phase_stride + ((int64_t)((kz * kh2 + r2) * kw2 + c2) * o.coutp + r) * o.cinb + ci];
        }
    o.grad[i] += v;
}

__global__ __launch_bounds__(256) void k_wgrad_finalize_batch(const rho_wfin_op* __restrict__ ops, int nops) {
    __shared__ float tile[32 * 65];            // [tap][64 channels] (+1: conflict-free column reads)
    int lo = 0, hi = nops - 1;
    const int b = (int)blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (ops[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    const rho_wfin_op o = ops[lo];
    const int taps = o.kd * o.kh * o.kw;
    if (o.kind == 0 && taps > 1 && taps <= 32) {
        // [tap][row][channel] -> [row][channel][tap] through LDS: the buffer is read in 256-byte channel runs (one per tap), the
        // gradient written as one contiguous run of 64 * taps floats - both sides coalesced (the elementwise walk reads the buffer
        // with a stride of coutp * cin_buf floats between consecutive threads)
        const int64_t cg = (o.cin + 63) / 64;                      // 64-channel groups per row
        const int64_t units = o.cout * cg;
        for (int64_t u = b - o.blk0; u < units; u += o.nblk) {
            const int64_t r = u / cg;
            const int ci0 = (int)(u % cg) * 64;
            const int nc = (int)min((int64_t)64, o.cin - ci0);
            for (int e = threadIdx.x; e < taps * 64; e += 256) {
                const int t = e >> 6, c = e & 63;
                tile[t * 65 + c] = c < nc ? o.dw[((int64_t)t * o.coutp + r) * o.cinb + ci0 + c] : 0.0f;
            }
            __syncthreads();
            const int64_t dst_row = o.row_src ? (int64_t)o.row_src[r] : r;
            if (dst_row >= 0 && dst_row < o.cout) {
                float* g = o.grad + (dst_row * o.cin + ci0) * taps;
                for (int e = threadIdx.x; e < nc * taps; e += 256) g[e] += tile[(e % taps) * 65 + e / taps];
            }
            __syncthreads();
        }
        return;
    }
    const int64_t stride = (int64_t)o.nblk * 256;
    for (int64_t i = (int64_t)(b - o.blk0) * 256 + threadIdx.x; i < o.total; i += stride) wfin_elem(o, i);
}

extern "C" int rho_wgrad_finalize_batch(const rho_wfin_op* ops_dev, int64_t n_ops, int64_t n_blocks, void* stream) {
    if (!ops_dev || n_ops <= 0 || n_blocks <= 0 || n_blocks > 0x7FFFFFFF || n_ops > 0x7FFFFFFF) return RHO_E_ARG;
    hipLaunchKernelGGL(k_wgrad_finalize_batch, dim3((unsigned)n_blocks), dim3(256), 0, as_stream(stream), ops_dev, (int)n_ops);
    RHO_LAUNCH_CHECK();
    return 0;
}
