/*
 * rho_hip.h -- C ABI of librho_hip.so: the MI355X (gfx950) kernels behind the
 * rho_diffusion DDPM / UNetv2 hot path.
 *
 * The reference (intel/rho-diffusion) has no FFI of its own: its hot path is stock
 * PyTorch ops called from Python.  Each entry point below therefore cites the reference
 * *operator* (file:line under /root/reference) whose arithmetic it replaces; the Python
 * classes in rho_diffusion_amd/ that mirror the reference's class surface call these
 * through ctypes.
 *
 * Conventions
 *  - Every function is asynchronous on the caller's hipStream_t (passed as void*),
 *    allocates nothing, never synchronises, and returns 0 on success, a negative
 *    RHO_E_* code for bad arguments, or a positive hipError_t from the launch.
 *  - Pointers are device pointers unless stated otherwise.
 *  - dtype: RHO_F32 = 0 (float), RHO_BF16 = 1 (bfloat16 storage, fp32 accumulate).
 *  - "channels-last" activations are [N, D, H, W, C] contiguous (2-D: D = 1, 1-D: D = H = 1).
 *    Tensors at the pipeline boundary (x_t, eps, eps_hat) keep the reference's
 *    [N, C, *spatial] float32 layout.
 */
#ifndef RHO_HIP_H
#define RHO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RHO_F32 0
#define RHO_BF16 1

#define RHO_E_ARG (-1)      /* inconsistent or unsupported argument */
#define RHO_E_ALIGN (-2)    /* pointer / channel count not aligned as required */
#define RHO_E_SHAPE (-3)    /* tile / shape constraint violated */

/* ABI version: bumped on ANY signature / struct-layout change (2: table_len in rho_q_sample(_coef), fmt in rho_gn_bwd_finalize,
 * rho_conv_desc grew; 3: round-3 additions).  A loader must compare rho_abi_version() with the header it was written against
 * before calling anything else (hip.py does; a build with all symbols but older signatures would be called with shifted arguments). */
#define RHO_ABI_VERSION 8
int rho_abi_version(void);
/* static string: target arch + build flags */
const char* rho_build_info(void);

/* ------------------------------------------------------------------ diffusion loops */

/* q_sample: x_t = sqrt(abar[t_b]) * x0 + sqrt(1 - abar[t_b]) * eps     (per batch element b)
 * replaces DDPM.forward_process, rho_diffusion/diffusion/ddpm.py:104-130 (+ the gather of
 * abstract_diffusion.py:171-220).  x0/eps/x_t: float32 [B, per_sample]; abar: float32 [T];
 * t: int64 [B].  nan_flag (optional, int32[1]): bit 0 is set if any x_t is NaN (the device-side form of the host check at
 * ddpm.py:268-272), bit 2 if some t[b] lies outside [0, table_len) - the reference raises IndexError there; the kernel then
 * reads the clamped row instead of out-of-bounds memory.  table_len = number of entries of alpha_bar. */
int rho_q_sample(const float* x0, const float* eps, float* x_t, const float* alpha_bar,
                 const int64_t* t, int64_t batch, int64_t per_sample, int64_t table_len, int32_t* nan_flag, void* stream);

/* p_sample_step: x <- clamp((x - beta/sqrt(1-abar) * eps_hat)/sqrt(alpha) + 0.8*sqrt(beta)*z, -1, 1)
 * replaces ddpm.py:210-218 (one reverse update, caller skips t == 0; z may be NULL => 0,
 * the t <= 1 case of ddpm.py:196-199).  coef = {1/sqrt(alpha_t), beta_t/sqrt(1-abar_t), 0.8*sqrt(beta_t)}
 * as float32[3] on the DEVICE (row t of a [T,3] table built once per schedule) so a captured
 * HIP graph can be replayed with a different row pointer-free: the row index is read from
 * *t_dev (int32[1], device).  x updated in place; n elements. */
int rho_p_sample_step(float* x, const float* eps_hat, const float* z, const float* coef_table,
                      const int32_t* t_dev, int64_t n, void* stream);

/* Device-resident loop state of reverse_process (ddpm.py:195): *t_dev -= 1 and *offset_dev += delta,
 * so that one captured HIP graph replays every step without host involvement. Either may be NULL. */
int rho_step_advance(int32_t* t_dev, uint64_t* offset_dev, uint64_t delta, void* stream);

/* Counter-based Philox4x32-10 standard normals (Box-Muller), replaces torch.randn_like at
 * ddpm.py:101-102,171,197.  out float32[n]; stream of (seed, offset) is reproducible and
 * independent of launch geometry.  offset is read from *offset_dev (uint64[1], device) when
 * offset_dev != NULL (graph-replayable), else from `offset`. */
int rho_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t offset,
                      const uint64_t* offset_dev, void* stream);

/* mean((a-b)^2) -> loss[0] (float32, zeroed by this call) and optionally grad_a = 2(a-b)/n.
 * replaces nn.MSELoss at ddpm.py:280 (+ its backward). */
int rho_mse(const float* a, const float* b, float* loss, float* grad_a, int64_t n, void* stream);

/* The same with an ORDERED reduction (ABI 7): block partials go to partials[n_partials] (scratch, >= 1; 1024 is plenty) and are added
 * in index order, so the loss value is bit-reproducible run to run (rho_mse adds its blocks with an fp32 atomic, in arrival order).
 * grad_a is identical in both.  The Python binding always uses this form. */
int rho_mse_ws(const float* a, const float* b, float* loss, float* grad_a, int64_t n, float* partials, int64_t n_partials, void* stream);

/* mean over all non-batch axes of a float32 [batch, per_sample] tensor -> out[batch]: mean_flat of layers.py:105-110 (ABI 7). */
int rho_mean_flat(const float* x, float* out, int64_t batch, int64_t per_sample, void* stream);

/* Reproducibility switch (ABI 7).  The reference's CPU path is deterministic; here three backward kernels combine partial sums
 * with fp32 atomics by default (weight gradient, the dx of rho_linear_bwd, rho_multi_embed_bwd), so gradients differ in the last
 * bits run to run.  rho_set_deterministic(1) (or RHO_DETERMINISTIC=1 / RHO_WGRAD_DETERMINISTIC=1 in the environment) makes the
 * latter two take ordered paths and tells callers to use rho_conv_nd_wgrad_ws for the weight gradient; returns the old value. */
int rho_set_deterministic(int on);
int rho_get_deterministic(void);

/* Fused multi-tensor-free AdamW over one flat float32 parameter arena
 * (torch.optim.AdamW built at abstract_diffusion.py:103-119): decoupled weight decay,
 * bias correction from `step` (1-based).  p, g, m, v: float32[n]. */
int rho_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
              float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* ------------------------------------------------------------------ embeddings */

/* Timestep embedding of UNet.forward (unet_v2.py:699-701): the interleaved sinusoid of models/common.py:27-43 for ANY integer
 * t (no table, no range limit) followed by time_embed = Linear(dim -> edim), SiLU, Linear(edim -> edim) (unet_v2.py:521-525) and
 * the label-embedding add (:702-719), one workgroup per sample:
 *   pe[b, 2i] = sin(t_b / omega[i]), pe[b, 2i+1] = cos(t_b / omega[i])      omega: float32 [dim / 2] = wavelength^(2i / dim)
 *   h = w0 pe + b0;   emb = w2 silu(h) + b2 (+ cond[b, :])
 * t is int64[B]; when t_scalar_dev != NULL every b uses *t_scalar_dev (int32[1], device) instead (graph-replayable sampling
 * step).  pe_out [B, dim] / h_out [B, edim] (optional) keep what the backward needs.  w0 == NULL: sinusoid only into pe_out
 * (the registry layer SinusoidalPositionEmbedding, common.py:46-80).  act (ABI 7): the activation between the two linears, an
 * ACTIVATION CODE - 0 identity, 1 SiLU, 2 ReLU, 3 GELU (erf), 4 Tanh, 5 Sigmoid, 6 ELU(1) - the elementwise, parameter-free entries
 * of the reference's activation registry (registry.py:162-170, resolved at unet_v2.py:518-519).  The same codes are what the `pre_silu`
 * argument of rho_gn_apply / rho_gn_bwd_reduce / rho_gn_bwd_apply and the `act_in` / `act_out` arguments of rho_linear / rho_linear_bwd
 * take (0 / 1 keep their old meaning); the conv loaders (rho_conv_desc.pre_silu, gnb_silu) and rho_head_conv3d know 0 / 1 only - a
 * network with another activation materialises its activated conv inputs with rho_gn_apply. */
int rho_timestep_embed(const float* omega, const int64_t* t, const int32_t* t_scalar_dev, const float* w0, const float* b0,
                       const float* w2, const float* b2, const float* cond, float* pe_out, float* h_out, float* emb_out,
                       int64_t batch, int64_t dim, int64_t edim, int act, void* stream);

/* MultiEmbeddings.forward (models/conditioning.py:115-139): out[b, :] = sum over keys i of tables[i][j, :] with j the position
 * of y[b, i] in that key's value list (exact float equality, conditioning.py:132).  y float32 [B, nkeys] with row stride
 * y_stride (y_stride == 1: one label per sample, shared by all keys as in the reference's 1-D branch); space = the value lists
 * concatenated (float32), key i owning space[key_off[i] .. key_off[i+1]); tables = device array of nkeys float32 [n_i, dim]
 * pointers.  idx_out (optional int32 [B, nkeys]) keeps the categories for the backward.  A label found in no list sets
 * *err_flag |= 2 (the reference fails with a shape error there).  No host synchronisation (the reference's torch.where is one). */
int rho_multi_embed(const float* y, int64_t y_stride, const float* space, const int32_t* key_off, const float* const* tables,
                    int64_t nkeys, int64_t batch, int64_t dim, float* out, int32_t* idx_out, int32_t* err_flag, void* stream);

/* Backward of rho_multi_embed: dtables[i][idx[b, i], :] += demb[b, :] (autograd of nn.Embedding, conditioning.py:58-60). */
int rho_multi_embed_bwd(const float* demb, const int32_t* idx, float* const* dtables, int64_t nkeys, int64_t batch, int64_t dim,
                        void* stream);

/* random_timesteps (abstract_diffusion.py:163-169: randint(0, timesteps, (B,)) with replacement) on the device:
 * out int64[n] uniform on [0, high), Philox4x32-10 stream (seed, offset) as rho_philox_normal (offset read from *offset_dev
 * when given).  Removes the CPU draw + H2D copy of every training step. */
int rho_randint(int64_t* out, int64_t n, int64_t high, uint64_t seed, uint64_t offset, const uint64_t* offset_dev, void* stream);

/* out[b, o] = bias[o] + sum_k act(x[b, k]) * w[o, k] (+ add[b, o])   all float32.
 * act_in: 0 = identity, 1 = SiLU.  act_out: 0 = identity, 1 = SiLU.
 * replaces the nn.Linear / SiLU chains of unet_v2.py:521-525 (time_embed), :229-235
 * (emb_layers of every ResBlock, batched by concatenating their weights) and the label
 * embedding add of :702-719. */
int rho_linear(const float* x, const float* w, const float* bias, const float* add, float* out,
               int64_t batch, int64_t in_dim, int64_t out_dim, int act_in, int act_out, void* stream);

/* ------------------------------------------------------------------ layout */

/* [N, C, S] float32 (reference layout) -> channels-last [N, S, Cpad] dtype, channels >= C zeroed. */
int rho_pack_input(const float* x, void* y, int dtype, int64_t n, int64_t c, int64_t s, int64_t cpad, void* stream);

/* Stem convolution with Cin * taps <= Cpad (conv_nd(dims, in_channels = 1, mc, 3), unet_v2.py:535) as a 1x1x1 GEMM: the
 * im2col operand  out[n][pos][ci * taps + tap] = x[n][ci][pos + offset(tap)]  (zero outside the volume and for k >= Cin * taps),
 * channels-last bf16; taps are ordered (kd, kh, kw) as the rows of weight.reshape(Cout, Cin * taps). */
int rho_im2col_taps(const float* x, void* out, int dtype, int64_t n, int64_t cin, int64_t d, int64_t h, int64_t w,
                    int kd, int kh, int kw, int64_t cpad, void* stream);

/* Head convolution with Cout = 1 (zero_module(conv_nd(dims, mc, out_channels = 1, 3)), unet_v2.py:679-683) as a 1x1x1 GEMM
 * T[n][q][tap] = sum_c W[0][c][tap] * act[n][q][c] followed by  out[n][pos] = bias[0] + sum_tap T[n][pos + offset(tap)][tap]
 * (fp32 sum in tap order, out-of-volume taps skipped = zero padding).  t: channels-last bf16 [N, D, H, W, Cpad]. */
int rho_tap_gather_sum(const void* t, int dtype, int64_t n, int64_t d, int64_t h, int64_t w, int kd, int kh, int kw,
                       int64_t cpad, const float* bias, float* out, void* stream);

/* Conv weight [Cout, Cin, kd, kh, kw] float32 (PyTorch layout) -> [taps][CoutP][CinP] dtype,
 * zero padded.  Optional row permutation `row_src` (int32[CoutP], device; -1 = zero row) lets
 * the qkv projection be re-ordered (legacy vs new attention order, unet_v2.py:384,419-431). */
int rho_prep_conv_weight(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int64_t taps,
                         int64_t coutp, int64_t cinp, const int32_t* row_src, void* stream);

/* ------------------------------------------------------------------ GroupNorm (32 groups) */

/* Pass 1: per-(sample, block, channel-octet) partial sums over channels-last input that may be
 * the virtual concat of two tensors (torch.cat at unet_v2.py:729 is never materialised).
 * partials: float32 [N][nblk][C/8][16] (8 sums, 8 sums of squares).  Returns required nblk via
 * rho_gn_nblk.  replaces the statistics half of GroupNorm32.forward, layers.py:71-74. */
int rho_gn_nblk(int64_t s);
int rho_gn_partial(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype,
                   int64_t n, int64_t s, float* partials, void* stream);

/* Pass 2: mean / rstd per (n, group) (eps 1e-5, biased variance) and the folded per-(n, channel)
 * affine used by the conv prologue:   y = a*x + b  with
 *   a = rstd*gamma*(1+scale),  b = (beta - mean*rstd*gamma)*(1+scale) + shift
 * where scale/shift ([N, C] each, row stride film_stride floats) are the FiLM halves of
 * unet_v2.py:285-289 (NULL => no FiLM: plain GroupNorm affine).  stats: float32 [N][32][2]. */
int rho_gn_finalize(const float* partials, int64_t n, int64_t c, int64_t s, int64_t nblk,
                    const float* gamma, const float* beta, const float* scale, const float* shift,
                    int64_t film_stride, float* stats, float* a, float* b, void* stream);

/* ------------------------------------------------------------------ convolution */

typedef struct rho_conv_desc {
    const void* x1;        /* channels-last input, c1 channels */
    const void* x2;        /* optional second input (virtual concat after x1), c2 channels */
    const float* pre_a;    /* optional prologue y = act(a*x+b), [N, c1+c2]; NULL = raw input */
    const float* pre_b;
    const void* w;         /* prepared weights [taps][coutp][c1+c2] */
    const float* bias;     /* [coutp] float32 */
    const void* res;       /* optional residual, channels-last [.., split], added in the epilogue */
    const float* res_add;  /* optional per-(n, co) float32 added in the epilogue (additive timestep embedding,
                              unet_v2.py:291): element (n, co) at res_add[n * res_add_stride + co] */
    void* y;               /* channels-last output for co < split: [N, Do, Ho, Wo, split] */
    void* y2;              /* channel-major output for split <= co < cout: [N, cout-split, Do*Ho*Wo] */
    int32_t dtype;         /* of x1, x2, w, res, y */
    int32_t y2_f32;        /* 1: y2 is float32 regardless of dtype */
    int32_t c1, c2;
    int32_t cout, coutp, split;
    int32_t n, d, h, w_;   /* input extents (pre-upsample) */
    int32_t kd, kh, kw;    /* 1 or 3 each; allowed: (3,3,3) (1,3,3) (1,1,3) (1,1,1) */
    int32_t sh, sw;        /* stride on H / W (1 or 2); depth stride is always 1 (unet_v2.py:153) */
    int32_t up_h, up_w;    /* nearest x2 upsample folded into the loader (unet_v2.py:122-131) */
    int32_t pre_silu;      /* apply SiLU after the affine prologue */
    int32_t res_add_stride;
    /* --- backward-data use of the same kernel (dgrad = conv of dY with flipped, transposed weights) */
    int32_t y2_cl;         /* 1: y2 is channels-last [.., cout-split] (gradient of the 2nd concat source) */
    int32_t zs_h, zs_w;    /* input is dY of a stride-2 conv, read zero-stuffed: virtual extent out_h/out_w */
    int32_t out_h, out_w;  /* forward-conv input extents (only with zs_*) */
    const void* res2;      /* optional residual for the y2 region (channels-last): in-place grad accumulation */
    /* --- GroupNorm statistics of the OUTPUT, fused into the epilogue (saves the separate read of rho_gn_partial) */
    float* stats;          /* optional: per-tile channel sums of the stored (rounded) output, float32
                              [N][tiles][2][split] with tiles = rho_conv_stats_tiles(desc): row 0 = sum, row 1 = sum of
                              squares over the tile's positions; combined in fixed order by rho_gn_finalize2 */
    /* --- sub-pixel phases of a conv behind a nearest x2 upsample (Upsample, unet_v2.py:122-134): on an upsampled axis the three
     * taps of output row 2i + a read only two source rows (a = 0: rows i-1, i with weights w0, w1 + w2; a = 1: rows i, i+1 with
     * w0 + w1, w2), so one launch per parity runs a 2-tap axis (kh / kw = 2, weights from rho_prep_conv_weight_phase) on the
     * SOURCE tensor and writes every second output row / column: 12 instead of 27 taps in 3-D.  ph_h / ph_w: 0 = axis not
     * upsampled, 1 / 2 = parity 0 / 1.  y (and res, stats) describe the full-resolution output [N, D, 2H, 2W, cout].
     * A 1-tap phased axis (kh / kw = 1, taps start at the output row) is the even parity of a stride-2 conv's data gradient, see
     * rho_prep_conv_weight_sel. */
    int32_t ph_h, ph_w;
    /* data gradient of such a phase: the input (dY of the full-resolution output) is read at parity phd - 1 of the phased axis
     * (d.h / d.w_ are the source-resolution extents, the tensor has twice as many rows / columns), 2-tap axis, taps start
     * phd - 1 rows before the output position, dense output; the four launches accumulate into dX through `res`.  With kh / kw
     * = 1 (one tap at the output row) or 2 this is also the forward of a stride-2 conv split by input parity
     * (rho_prep_conv_weight_sel). */
    int32_t phd_h, phd_w;
    /* --- GroupNorm BACKWARD reduction fused into a dgrad launch (the output is d act(a * x + b), x the forward input): with
     * gnb_x1 set, `stats` receives per tile and channel  row 0 = sum of dz,  row 1 = sum of dz * x  (dz = output * act'(a x + b),
     * the output as stored) instead of the output's own moments: what rho_gn_bwd_reduce reads two tensors to compute
     * (rho_gn_bwd_finalize with fmt = 1 takes this layout).  Replaces the reductions of GroupNorm32's autograd backward. */
    const void* gnb_x1;    /* forward input, channels-last, first gnb_c1 channels of the output's `split` */
    const void* gnb_x2;    /* second concat source (split - gnb_c1 channels) or NULL */
    int32_t gnb_c1;
    int32_t gnb_silu;      /* act = SiLU (else identity) */
    const float* gnb_a;    /* [N][split] folded affine of the forward prologue (pre_a / pre_b of the forward conv) */
    const float* gnb_b;
    /* --- k-split of launches too small to fill the chip (2-D / 1-D layers at low resolution: 1024 positions x 512 couts are 16
     * workgroups on 256 CUs).  With a workspace the launch is split over the input-channel chunks into up to 16 partial launches
     * (grid z) whose fp32 tiles land in `ws` and are added in split order, with bias / residuals / rounding, by a second kernel
     * (reproducible; the partial sums are fp32 either way).  ws = NULL: never split.  rho_conv_workspace_bytes(desc) is the size
     * the preferred split wants (0: this launch is not split); a smaller workspace lowers the split count. */
    void* ws;
    int64_t ws_bytes;
    /* --- a 1x1x1 convolution of ANOTHER input folded into this launch: out = conv(x) + sk_w * cat(sk_x1, sk_x2) + sk_bias, i.e. the
     * ResBlock's `skip_connection(x) + out_layers(h)` (unet_v2.py:245-256,293) without the launch, the tensor and the residual
     * read of the skip branch.  sk_w: prepared weights [1][coutp][sk_c1 + sk_c2] (rho_prep_conv_weight of the 1x1x1 parameter),
     * sk_bias [coutp] float32 or NULL; sk_x1 / sk_x2 channels-last at the OUTPUT resolution.  bf16 3x3x3 stride-1 launches with a
     * channels-last output only (RHO_E_ARG otherwise: the caller then launches the skip conv on its own and passes it as `res`). */
    const void* sk_x1;
    const void* sk_x2;
    const void* sk_w;
    const float* sk_bias;
    int32_t sk_c1, sk_c2;
    /* --- GroupNorm BACKWARD apply pass fused into a data-gradient launch: with gna_g set the launch writes
     *     out = conv(x) [+ res / res2]  +  cA * (g * act'(a * x0 + b)) + cQ * x0 + cP
     * per channel of its (one or two) channels-last outputs - the second operand is what rho_gn_bwd_apply computes from g = the
     * gradient of act(GroupNorm(x0)), channels-last [positions][cout] over BOTH output regions, and x0 = the forward input of that
     * norm (gnb_x1 / gnb_x2 / gnb_c1, gnb_a / gnb_b = its folded affine, gnb_silu its activation 0 / 1).  cA [N][cout],
     * cP / cQ [N][32] are rho_gn_bwd_finalize's coefficients.  Used for the ResBlock whose input reaches its output through a
     * 1x1x1 skip convolution (unet_v2.py:245-256): dX = skip^T(dY) + GroupNorm-backward(in-conv path) in ONE launch instead of a
     * data-gradient launch plus an apply pass that re-reads and re-writes dX.  Needs one-sample tiles (rho_conv_stats_tiles > 0),
     * no `stats`, a second region (if any) that is channels-last (y2_cl) and a multiple of 8 (bf16) / 4 (f32) channels wide. */
    const void* gna_g;
    const float* gna_cA;
    const float* gna_cP;
    const float* gna_cQ;
} rho_conv_desc;

/* n-D convolution, zero padding k/2, as an LDS-halo-staged implicit GEMM on MFMA.
 * replaces conv_nd (layers.py:77-88) at every call site of unet_v2.py (ResBlock convs :215,
 * :241; skip 1x1 :256; Downsample :153-162; Upsample+conv :122-134; attention qkv / proj_out
 * :323,:331 with the residual of :342; stem :535; head :679-683) with the GroupNorm affine +
 * FiLM + SiLU (:212-216, :285-289) applied while the halo tile is staged. */
int rho_conv_nd_fwd(const rho_conv_desc* desc, void* stream);

/* Weights of one sub-pixel phase (see rho_conv_desc.ph_h): from the fp32 parameter [cout][cin][kd][kh][kw] (kh / kw = 3 on a
 * phased axis) to the launch layout [kd * kh' * kw' taps][coutp][cinp] with the two taps of a phased axis summed as described
 * there; ph_h / ph_w as in the descriptor.  dgrad = 1: the data-gradient layout instead, [taps][cinp rows = ceil32(cin)][coutp
 * columns] with flipped taps and transposed channels (as rho_prep_conv_weight_dgrad), for a launch with phd_h / phd_w. */
int rho_prep_conv_weight_phase(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int kd, int kh, int kw, int ph_h,
                               int ph_w, int64_t coutp, int64_t cinp, int dgrad, void* stream);

/* Parity split of a STRIDE-2 conv (Downsample, unet_v2.py:153-162), again on stride-1 launches.  Per strided axis
 * y[o] = w0 x[2o-1] + w1 x[2o] + w2 x[2o+1]:
 *   forward  = one launch per INPUT parity, accumulated through `res` (rho_conv_desc.phd_h: even rows: 1 tap {w1}; odd rows:
 *              2 taps {w0, w2} starting one row before the output row) - the loader never stages the 2x halo of a strided tile;
 *   backward = one launch per parity of dX (rho_conv_desc.ph_h: dx[2m] = w1 dy[m]; dx[2m+1] = w2 dy[m] + w0 dy[m+1]) instead of
 *              a 27-tap conv over a zero-stuffed dY in which three of four multiply-adds are zeros.
 * This packs the taps such a launch uses: new tap r of an axis is source tap (sel >> 4r) & 15 (kh2 / kw2 new taps; an axis that
 * keeps its taps passes its size and 0x210); flip_d mirrors the depth taps (data gradient); dgrad selects the data-gradient
 * layout [taps][ceil32(cin)][ceilCK(cout)] instead of [taps][coutp][cinp]. */
int rho_prep_conv_weight_sel(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int kd, int kh, int kw, int kh2, int kw2,
                             int sel_h, int sel_w, int flip_d, int64_t coutp, int64_t cinp, int dgrad, void* stream);

/* Number of output tiles per sample the launch of `desc` uses (the middle extent of desc->stats), or 0 when fused
 * statistics are not available for this geometry (channel-major outputs, tiles that straddle samples: 1-D / 2-D
 * kernels, 1x1x1 with positions-per-sample not a multiple of 256); callers then fall back to rho_gn_partial. */
int64_t rho_conv_stats_tiles(const rho_conv_desc* desc);

/* The 1-channel ends of the 3-D UNet as single launches (bf16 inference plans; contraction over the 27 taps / taps as output rows,
 * intermediates in LDS - see csrc/ends.hip).
 *   rho_stem_conv3d: y[n, d, h, w, co] = bias[co] + sum_tap W[co][tap] x[n, 0, (d, h, w) + off(tap)]   (zero padding 1), the stem
 *     `conv_nd(3, 1, mc, 3, padding=1)` (unet_v2.py:535).  x float32 [N, 1, D, H, W]; w = the prepared 1x1x1 weights of the im2col form
 *     [coutp][32] bf16 (rho_prep_conv_weight of the parameter reshaped [cout, 27, 1, 1, 1]); y bf16 channels-last; cout 32 or 64.
 *     stats: optional fused GroupNorm statistics of the stored output, float32 [N][rho_stem_conv3d_tiles(d, h, w)][2][cout]
 *     (the layout of rho_conv_desc.stats, read by rho_gn_finalize2).
 *   rho_head_conv3d: out[n, 0, d, h, w] = bias + sum_tap sum_c W[tap][c] act(a[n, c] x[(d, h, w) + off(tap)][c] + b[n, c]), the head
 *     `GroupNorm32 -> SiLU -> conv_nd(3, mc, 1, 3, padding=1)` (unet_v2.py:679-683).  x bf16 channels-last [N, D, H, W, C], C in
 *     {32, 64, 96, 128}; pre_a / pre_b [N, C] float32 (NULL: raw input), pre_silu; w = prepared [32][C] bf16 (rows = taps, 27 used);
 *     out float32 [N, 1, D, H, W] (the reference layout). */
int64_t rho_stem_conv3d_tiles(int64_t d, int64_t h, int64_t w);
int rho_stem_conv3d(const float* x, const void* w, const float* bias, void* y, float* stats, int64_t n, int64_t d, int64_t h, int64_t w_,
                    int64_t cout, void* stream);
int rho_head_conv3d(const void* x, const float* pre_a, const float* pre_b, int pre_silu, const void* w, const float* bias, float* out,
                    int64_t n, int64_t d, int64_t h, int64_t w_, int64_t c, void* stream);

/* Batched form of rho_wgrad_finalize / rho_wgrad_finalize_phase (ABI 7): a device table of ops, op j on launch blocks
 * [blk0, blk0 + nblk) (ascending, contiguous from 0).  total = cout * cin * kd * kh * kw (elements of the parameter gradient walked,
 * in buffer-row order).  kind 0: grad[row_src ? row_src[r] : r][ci][tap] += dw[tap][r][ci] (dw rows coutp, row width cinb; a bias
 * gradient is cin = kd = kh = kw = cinb = 1 with coutp = the width of the channel-sum vector).  kind 1: all sub-pixel phases of a conv
 * behind a nearest x2 upsample (up_h / up_w say which axes are phased; the phase buffers lie phase_stride floats apart in the order
 * of rho_prep_conv_weight_phase's phases) summed into the 3-tap gradient.  kind 2: grad[ci][taps - 1 - r] += dw[r][ci] for r < cout
 * rows (cout = the tap count kd * kh * kw, `cin` channels, buffer row width cinb): the weight gradient of a one-output-channel conv
 * computed as a GEMM against the im2col of its output gradient (taps-as-rows, mirrored).  Always accumulates; no two ops of a launch may
 * share a gradient. */
typedef struct rho_wfin_op {
    const float* dw;
    float* grad;
    const int32_t* row_src;
    int64_t cout, cin, coutp, cinb, total, phase_stride;
    int32_t kind, kd, kh, kw, up_h, up_w;
    int32_t blk0, nblk;
} rho_wfin_op;
int rho_wgrad_finalize_batch(const rho_wfin_op* ops_dev, int64_t n_ops, int64_t n_blocks, void* stream);

/* Batched weight preparation (ABI 7): ONE launch writes every prepared layout a model needs after an optimizer step - what the
 * per-tensor rho_prep_conv_weight / _dgrad / _phase / _sel calls above write, the zero-padded (and, for the qkv projection,
 * row-gathered) fp32 bias vectors, and plain fp32 copies (the batched FiLM matrix of ResBlock.emb_layers, unet_v2.py:225-232) -
 * from a table of rho_prep_op in DEVICE memory.  Op j owns launch blocks [blk0, blk0 + nblk) (blk0 ascending, contiguous, op 0 at 0;
 * n_blocks = their sum); field meanings per kind are those of the per-tensor call of the same name:
 *   RHO_PREP_FWD   out[kd*kh*kw][d1 = coutp][d2 = cinp], perm = row_src;   RHO_PREP_DGRAD  out[taps][d1 = rowsp][d2 = colsp], perm = col_src;
 *   RHO_PREP_PHASE ph_h / ph_w / dgrad, d1 x d2 as rho_prep_conv_weight_phase;   RHO_PREP_SEL kh2 / kw2 / sel_h / sel_w / flip_d / dgrad;
 *   RHO_PREP_VEC   out[d1] (a general gather): out[i] = w[perm ? perm[i] : i] for i < cout (source length cin; perm[i] < 0: zero), 0 beyond -
 *                  the padded bias vectors, plain copies, and layouts no other kind describes (the head conv's taps-as-rows form).
 * total = elements of `out`; dtype = RHO_BF16 / RHO_F32 of `out`. */
enum { RHO_PREP_FWD = 0, RHO_PREP_DGRAD = 1, RHO_PREP_PHASE = 2, RHO_PREP_SEL = 3, RHO_PREP_VEC = 4 };
typedef struct rho_prep_op {
    const float* w;
    void* out;
    const int32_t* perm;
    int64_t cout, cin, d1, d2, total;
    int32_t kind, dtype;
    int32_t kd, kh, kw, kh2, kw2, ph_h, ph_w, sel_h, sel_w, flip_d, dgrad;
    int32_t blk0, nblk;
    int32_t pad_;
} rho_prep_op;
int rho_prep_batch(const rho_prep_op* ops_dev, int64_t n_ops, int64_t n_blocks, void* stream);

/* Workspace the k-split of `desc` wants (rho_conv_desc.ws), in bytes; 0 when the launch would not be split: kernels with a depth
 * extent (kd = 3: the batch is grid z there), launches with fused output statistics or channel-major outputs, grids of more than
 * 128 workgroups, fewer than 4 input-channel chunks.  Every kd = 1 launch qualifies otherwise - INCLUDING 1x1x1 launches (also the
 * qkv / proj_out projections of a 3-D model on a small grid).  Size = splits (<= 16) x positions x coutp x 4 bytes; at most
 * 128 workgroups x 256 positions x 128 couts x 16 splits x 4 B = 268 MB.  Depends on geometry only: one allocation of the maximum
 * over a plan's descriptors serves all of them (launches on one stream are ordered); a plan keeps it for its lifetime (counted by
 * _Plan.nbytes(); 64 MB at BASELINE configs[0], 0 at configs[2] - every grid there fills the chip). */
int64_t rho_conv_workspace_bytes(const rho_conv_desc* desc);

/* Test / profiling aid: the name of the kernel instantiation rho_conv_nd_fwd would launch for `desc`
 * ("k_conv<bf16,3,3,3,BM=128,MAXP=5,NW=8,M16=1>"), written NUL-terminated into buf (cap >= 64).  Nothing is launched.
 * Lets the parity tests assert that the variants they check against the oracle cover every variant a benchmark
 * plan (BASELINE configs c3 / c5) launches.  Same return codes as rho_conv_nd_fwd. */
int rho_conv_variant(const rho_conv_desc* desc, char* buf, int cap);

/* ------------------------------------------------------------------ synthetic data */

/* Spherical-harmonic density fields of the reference's SphericalHarmonicDataset, evaluated on the device in float64 and
 * cast to float32 once (data/synthetic.py:45-124 compute_spherical_harmonic on the linspace(-2, 2, G)^3 grid of :172-174,
 * complex min-max normalisation :115-119, abs :124, float32 :303):  out[b] = float32 [G, G, G] for (l, m) = lm[b] (int32 [B, 2]
 * on the device; the field depends on |m| only).  workspace: rho_sph_harm_workspace_bytes(batch, grid) bytes.
 * minmax_in (optional, float64 [B, 4] = min.real, min.imag, max.real, max.imag on the device) replaces the complex (min, max)
 * pair of the normalisation - for m = 1, l >= 2 the reference's pair is decided by rounding noise of scipy (sph_harm.hip);
 * minmax_out (optional, same layout) receives the pair used. */
int64_t rho_sph_harm_workspace_bytes(int64_t batch, int64_t grid);
int rho_sph_harm_fields(const int32_t* lm, int64_t batch, int64_t grid, float* out, void* workspace, const double* minmax_in,
                        double* minmax_out, void* stream);

/* ------------------------------------------------------------------ attention */

/* Flash-style self attention, fp32 online softmax (QKVAttentionLegacy / QKVAttention,
 * unet_v2.py:365-436, with the head order resolved at weight-prep time).
 * qk: channels-last [B, T, 2C] (q of head h at h*ch, k at C + h*ch); vt: channel-major
 * [B, C, T]; out: channels-last [B, T, C].  logits = (q.k) * ch^-0.5 (== scaling q and k by
 * ch^-0.25 each, :385,:420).  ch in {16, 32, 64, 128, 256}.  lse (optional, float32 [B, heads, T]) receives the
 * base-2 log-sum-exp of the scaled logits per query, which rho_attention_bwd recomputes from. */
int rho_attention_fwd(const void* qk, const void* vt, void* out, float* lse, int dtype, int64_t batch, int64_t t,
                      int64_t heads, int64_t ch, void* stream);

/* ExponentialMovingAverage.update, rho_diffusion/ema.py:41-60: shadow -= one_minus_frac * (shadow - param),
 * float32, element-wise over n values (a whole flat parameter arena or one tensor); one_minus_frac = 1 - decay *
 * (1 - exp(-step / 2000)) is computed by the caller as the reference does. */
int rho_ema_update(float* shadow, const float* param, int64_t n, float one_minus_frac, void* stream);

/* ================================================================== backward (training) */

/* Weight gradient of rho_conv_nd_fwd (autograd of conv_nd, layers.py:77-88): desc describes the FORWARD
 * conv (inputs + prologue, which is recomputed in the loader); dy is its output gradient, channels-last with
 * row width dy_width (>= cout, padding channels zero); dw is an fp32 buffer [taps][coutp][c1+c2] that the
 * call accumulates into with fp32 atomics (zero it first).  up_h/up_w unsupported: pass the materialised
 * upsampled input (rho_upsample2x).  dbias (optional, float32 [coutp], zeroed by the caller) additionally receives the
 * bias gradient = per-channel sums of dy over all positions (the dY tiles pass through the kernel anyway; this replaces
 * a separate rho_chan_sum read of dy). */
int rho_conv_nd_wgrad(const rho_conv_desc* desc, const void* dy, int64_t dy_width, float* dw, float* dbias, void* stream);

/* Deterministic form of rho_conv_nd_wgrad (ABI 7): every workgroup (or wave, where the waves of a workgroup split positions) STORES its
 * partial [taps][coutp][cin] (+ [coutp] channel sums) to a slab of its own in `ws`, and a second launch adds the slabs to dw / dbias
 * in slab order - the same values as the atomic flush up to the summation order, bit-identical run to run and across replicas.
 * ws: scratch of at least rho_conv_wgrad_workspace_bytes(desc, dy_width) bytes (170 - 340 MB per layer at BASELINE configs[2]);
 * costs one write + one read of it per launch (+2.1 % per training step there). */
int rho_conv_nd_wgrad_ws(const rho_conv_desc* desc, const void* dy, int64_t dy_width, float* dw, float* dbias, void* ws,
                         int64_t ws_bytes, void* stream);
int64_t rho_conv_wgrad_workspace_bytes(const rho_conv_desc* desc, int64_t dy_width);

/* As rho_conv_variant, for the kernel rho_conv_nd_wgrad would launch ("k_wgrad<bf16,3,3,3,MAXP=10>", "k_wgrad1<bf16>"). */
int rho_conv_wgrad_variant(const rho_conv_desc* desc, int64_t dy_width, char* buf, int cap);

/* dw buffer -> parameter-gradient layout [cout][cin][taps] float32 (undoing the qkv row gather). */
int rho_wgrad_finalize(const float* dw, float* grad, int64_t cout, int64_t cin, int64_t taps, int64_t coutp,
                       int64_t cin_buf, const int32_t* row_src, int accumulate, void* stream);

/* ... of one sub-pixel phase (rho_conv_desc.ph_h / ph_w; dw = [kd * kh' * kw' taps][coutp][cin_buf] from rho_conv_nd_wgrad on the
 * phase's forward descriptor): every original tap of the [cout][cin][kd * kh * kw] parameter gradient takes the phase tap its row /
 * column was summed into; call once per phase with accumulate = 1 (the phases add up). */
int rho_wgrad_finalize_phase(const float* dw, float* grad, int64_t cout, int64_t cin, int kd, int kh, int kw, int ph_h, int ph_w,
                             int64_t coutp, int64_t cin_buf, int accumulate, void* stream);

/* Weights for the data gradient: out[tap'][ci][co'] = w[src(co')][ci][taps-1-tap'] so that rho_conv_nd_fwd
 * applied to dY yields dX.  rows padded to rowsp, cols to colsp. */
int rho_prep_conv_weight_dgrad(const float* w, void* out, int dtype, int64_t cout, int64_t cin, int64_t taps,
                               int64_t rowsp, int64_t colsp, const int32_t* col_src, void* stream);

/* Backward of act(GroupNorm32(x) * (1 + scale) + shift) (layers.py:71-74 + unet_v2.py:285-289), recomputed
 * from x, the saved stats and the folded affine (a, b):
 *   reduce  : per-(n, c) sums of g*act'(u) and g*act'(u)*xhat            (one pass over g and x)
 *   finalize: (fmt 0: the reduce pass's partials, nblk = rho_gn_nblk(s); fmt 1: the per-tile sums of dz and dz * x a dgrad
 *             launch wrote through rho_conv_desc.gnb_*, [N][nblk tiles][2][C] - the reduce pass is not run then)
 *             dgamma / dbeta (summed over samples, optionally accumulated), FiLM gradients dscale / dshift
 *             ([N, C] at row stride dfilm_stride), and the coefficients cA [N,C], cP / cQ [N,32] of pass 3;
 *             work_nc2 is float32 [2][N][C] scratch
 *   apply   : dx = cA*g*act'(u) + cP + cQ*x, written (or accumulated) into the one or two source gradients.  add1 (ABI 7, may be
 *             NULL): a further addend of dx1, same shape - the gradient that reaches x1 through a residual connection
 *             (unet_v2.py:293, :342), folded in here instead of a rho_add_inplace pass of its own. */
int rho_gn_bwd_reduce(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                      int64_t s, const float* a, const float* b, const float* stats, int pre_silu, float* partials,
                      void* stream);
int rho_gn_bwd_finalize(const float* partials, int64_t n, int64_t c, int64_t s, int64_t nblk, int fmt, const float* gamma,
                        const float* beta, const float* scale, int64_t film_stride, const float* stats,
                        float* work_nc2, float* dgamma, float* dbeta, int accumulate, float* dscale, float* dshift,
                        int64_t dfilm_stride, float* cA, float* cP, float* cQ, void* stream);
int rho_gn_bwd_apply(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n,
                     int64_t s, const float* a, const float* b, int pre_silu, const float* cA, const float* cP,
                     const float* cQ, void* dx1, void* dx2, int acc1, int acc2, const void* add1, void* stream);

/* Dropout forms (ABI 7): nn.Dropout(p) of ResBlock.out_layers (unet_v2.py:239: normalization, activation, Dropout, conv) as a
 * counter-based mask over the activated tensor [N, S, C] - element e keeps its value, scaled by 1 / (1 - p), iff word (e & 3) of
 * Philox4x32-10(counter = *drop_offset_dev + (e >> 2), key = drop_seed) >= p * 2^32.  rho_gn_apply_drop writes dropout(act(a x + b));
 * the two backward passes multiply the incoming gradient by the same regenerated mask (nothing is stored).  drop_p in (0, 1);
 * drop_offset_dev: uint64[1] on the device (the caller advances it once per training forward, rho_step_advance).  rho_dropout_mask
 * writes the mask alone as bytes (test aid: lets the CPU oracle apply the very mask the kernels used). */
int rho_gn_apply_drop(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s, const float* a,
                      const float* b, int pre_silu, void* y, float drop_p, uint64_t drop_seed, const uint64_t* drop_offset_dev,
                      void* stream);
int rho_gn_bwd_reduce_drop(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s,
                           const float* a, const float* b, const float* stats, int pre_silu, float* partials, float drop_p,
                           uint64_t drop_seed, const uint64_t* drop_offset_dev, void* stream);
int rho_gn_bwd_apply_drop(const void* g, const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s,
                          const float* a, const float* b, int pre_silu, const float* cA, const float* cP, const float* cQ, void* dx1,
                          void* dx2, int acc1, int acc2, const void* add1, float drop_p, uint64_t drop_seed,
                          const uint64_t* drop_offset_dev, void* stream);
int rho_dropout_mask(uint8_t* out, int64_t n, float drop_p, uint64_t drop_seed, const uint64_t* drop_offset_dev, void* stream);

/* rho_gn_finalize over one or two sources (the virtual concat), each with its own partial-sum format:
 *   fmt 0: rho_gn_partial's layout  [n][nblk][c/8][16]  (8 sums, 8 sums of squares per channel octet)
 *   fmt 1: a convolution's fused epilogue statistics  [n][nblk][2][c]  (rho_conv_desc.stats, nblk = tiles)
 * Source 2 is optional (p2 = NULL).  Channels of source 1 come first, c = c1 + c2, groups = 32. */
int rho_gn_finalize2(const float* p1, int fmt1, int64_t nblk1, int64_t c1, const float* p2, int fmt2, int64_t nblk2,
                     int64_t c2, int64_t n, int64_t s, const float* gamma, const float* beta, const float* scale,
                     const float* shift, int64_t film_stride, float* stats, float* a, float* b, void* stream);

/* Materialised prologue  y[n,pos,:] = act(a[n,:] * concat(x1,x2)[n,pos,:] + b[n,:])  (channels-last, dtype of x),
 * bit-identical to what the convolution loaders compute on the fly from the same a / b (GroupNorm32 + FiLM + SiLU,
 * layers.py:71-74, unet_v2.py:285-289).  Used in front of rho_conv_nd_wgrad so that the weight gradient reads
 * ready activations.  C = c1 + c2 a multiple of 8, c1 and c2 multiples of 8. */
int rho_gn_apply(const void* x1, int64_t c1, const void* x2, int64_t c2, int dtype, int64_t n, int64_t s,
                 const float* a, const float* b, int pre_silu, void* y, void* stream);

/* ---- GaussianDiffusionPipeline sampling path -------------------------------------------------
 * Dynamic thresholding, gaussian_diffusion.py:400-415: out[b] = torch.quantile(|x[b, :]|, q) ("linear"
 * interpolation, evaluated in float32 exactly as ATen does: rank = float(q) * float(n-1), floor / ceil order
 * statistics found EXACTLY by a radix select, then lerp).  x: float32 [batch, n] contiguous; workspace: at least
 * rho_abs_quantile_workspace_bytes(batch) bytes of device memory (contents undefined on entry). */
int64_t rho_abs_quantile_workspace_bytes(int64_t batch);
int rho_abs_quantile(const float* x, int64_t batch, int64_t n, double q, void* workspace, float* out, void* stream);

/* q_sample with explicit float32 coefficient tables: x_t = a[t_b] * x0 + b[t_b] * eps  (GaussianDiffusionPipeline.q_sample,
 * gaussian_diffusion.py:294-312, a = float(sqrt(abar)), b = float(sqrt(1-abar)) as _extract_into_tensor casts them).
 * x0 / eps / x_t: float32 [batch, per_sample]; t: int64 [batch] on the device; a t[b] outside [0, table_len) sets
 * *err_flag |= 4 (optional flag) and reads the clamped row, as rho_q_sample. */
int rho_q_sample_coef(const float* x0, const float* eps, float* x_t, const float* coef_a, const float* coef_b,
                      const int64_t* t, int64_t batch, int64_t per_sample, int64_t table_len, int32_t* err_flag, void* stream);

/* One DDIM update for an x0-predicting model, gaussian_diffusion.py:654-702 (+ :400-415, :462-466), float32:
 *   s = max(quantile[b], 1);  x0 = clamp(model_out, -s, s) / s;  eps = (c_recip * x_t - x0) / c_recipm1
 *   x_prev = x0 * sqrt_abar_prev + coef_eps * eps + sigma_masked * noise
 * c_recip = sqrt(1/abar_t), c_recipm1 = sqrt(1/abar_t - 1), coef_eps = sqrt(1 - abar_prev - sigma^2),
 * sigma_masked = (t != 0) * sigma, all float32 scalars prepared by the caller as the reference computes them.
 * noise may be NULL when sigma_masked == 0 (eta = 0); pred_xstart may be NULL. */
int rho_ddim_step(const float* x_t, const float* model_out, const float* quantile, const float* noise, float* x_prev,
                  float* pred_xstart, int64_t batch, int64_t per_sample, float c_recip, float c_recipm1,
                  float sqrt_abar_prev, float coef_eps, float sigma_masked, void* stream);

/* One reverse step of the diffusers-style DDPM scheduler the reference's DiffusersDDPMPipeline drives
 * (rho_diffusion/diffusion/diffusers.py:214; published DDPMScheduler.step; PARITY UNPINNED - third-party arithmetic):
 *   x0 = eps_mode ? (x_t - sqrt_beta_prod * model_out) / sqrt_alpha_prod : model_out;  clamp to +-clip when clip > 0;
 *   x_prev = c0 * x0 + c1 * x_t + sigma * noise        (noise may be NULL when sigma == 0; pred_xstart may be NULL)
 * float32, n elements, scalars prepared by the caller from the scheduler tables. */
int rho_ddpm_sched_step(const float* x_t, const float* model_out, const float* noise, float* x_prev, float* pred_xstart,
                        int64_t n, int eps_mode, float sqrt_beta_prod, float sqrt_alpha_prod, float clip, float c0, float c1,
                        float sigma, void* stream);

/* Channel sums of a channels-last tensor (conv bias gradients; additive-embedding gradients):
 * out_nc[n*nc_stride + c] (+)= sum_pos x[n,pos,c];  out_c[c] (+)= sum_n out_nc[n][c] (optional).
 * partials: scratch sized like rho_gn_partial's. */
int rho_chan_sum(const void* x, int dtype, int64_t n, int64_t s, int64_t c, float* partials, float* out_nc,
                 int64_t nc_stride, int acc_nc, float* out_c, int acc_c, void* stream);

/* Nearest x2 upsample (materialised only for the wgrad of Upsample.conv) and its backward (2x2 / 1x2 sum). */
int rho_upsample2x(const void* x, void* y, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int up_h,
                   int up_w, void* stream);
int rho_pool2x_sum(const void* dy, void* dx, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int up_h,
                   int up_w, int accumulate, void* stream);

/* avg_pool_nd with kernel = stride = 2 on H and / or W (layers.py:91-102; Downsample of conv_resample = False, unet_v2.py:165, and the
 * h_upd / x_upd of ResBlock(down = True), :221-224), channels-last, output extents floor(h / 2), floor(w / 2). */
int rho_avgpool2x(const void* x, void* y, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int pool_h, int pool_w,
                  void* stream);
/* its backward: dx [.., h, w, c] (+)= dy[.., h / 2, w / 2, c] / window; rows / columns dropped by the floor receive zero. */
int rho_avgpool2x_bwd(const void* dy, void* dx, int dtype, int64_t n_times_d, int64_t h, int64_t w, int64_t c, int pool_h, int pool_w,
                      int accumulate, void* stream);

/* Backward of rho_linear: dw[o,k] (+)= sum_b dout[b,o] act(x[b,k]); db[o] (+)= sum_b dout[b,o];
 * dx[b,k] (+)= act'(x[b,k]) sum_o dout[b,o] w[o,k].  dw/db/dx may be NULL; dout rows are dout_stride floats apart
 * (0 = out_dim) so a slice of the batched FiLM gradient can be passed in place. */
int rho_linear_bwd(const float* dout, int64_t dout_stride, const float* x, const float* w, float* dw, float* db,
                   float* dx, int64_t batch, int64_t in_dim, int64_t out_dim, int act_in, int acc_params, int acc_dx,
                   void* stream);

/* x *= *scale_dev over n float32 elements: the upstream scalar of loss.backward() applied to d(loss)/d(pred)
 * without a host round trip. */
int rho_scale_by_device_scalar(float* x, const float* scale_dev, int64_t n, void* stream);

/* dst += src over n elements of `dtype` (gradient accumulation where a tensor has several consumers). */
int rho_add_inplace(void* dst, const void* src, int dtype, int64_t n, void* stream);

/* Backward of rho_attention_fwd (recompute from lse).  o / dout channels-last [B,T,C]; delta_ws float32
 * [B,heads,T] scratch; outputs dqk [B,T,2C] (dq | dk) and dv [B,T,C], channels-last in `dtype`, with explicit row
 * strides in elements (0 = dense) so both can be column ranges of one [B,T,3C] buffer = the qkv projection's dY. */
int rho_attention_bwd(const void* qk, const void* vt, const void* o, const void* dout, const float* lse,
                      float* delta_ws, void* dqk, int64_t dqk_row_stride, void* dv, int64_t dv_row_stride, int dtype,
                      int64_t batch, int64_t t, int64_t heads, int64_t ch, void* stream);

/* ------------------------------------------------------------------ legacy UNet ("UNet v1", rho_diffusion/models/unet.py:30-269)
 * The tail of AbstractUNetBlock.forward (unet.py:117-135) on channels-last [N, S, C] tensors of `dtype`:
 *     h = act(conv2(act(conv1 x))) + residual_conv(x) + time_pe[n, c];   out = act(GroupNorm(groups, C)(h)).
 * act: 0 identity, 1 SiLU, 2 ReLU, 3 GELU (erf form, the nn.GELU() default the registry resolves "GELU" to). */

/* out = act(x) + r + nc[n, c]   (r and nc optional): unet.py:121-129 after the two convolutions. */
int rho_act_add(const void* x, const void* r, const float* nc, void* out, int dtype, int64_t n, int64_t s, int64_t c, int act,
                void* stream);
/* dx = dout * act'(x): autograd of the activation in front of the add. */
int rho_act_bwd(const void* x, const void* dout, void* dx, int dtype, int64_t numel, int act, void* stream);
/* y = act(GroupNorm(groups, C, eps)(x) * gamma + beta), nn.GroupNorm semantics (biased variance) for ANY group count
 * (unet.py:109-112 builds GroupNorm(8, C); the UNetv2 kernels are specialised for GroupNorm32); stats [N, groups, 2] =
 * (mean, rstd) is written for the backward. */
int rho_groupnorm_act(const void* x, void* y, float* stats, const float* gamma, const float* beta, int dtype, int64_t n, int64_t s,
                      int64_t c, int64_t groups, float eps, int act, void* stream);
/* its backward: dx, and dgamma[c] / dbeta[c] ACCUMULATED (fp32 atomics over samples: zero them first). */
int rho_groupnorm_act_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta, void* dx,
                          float* dgamma, float* dbeta, int dtype, int64_t n, int64_t s, int64_t c, int64_t groups, int act,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RHO_HIP_H */
