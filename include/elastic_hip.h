/*
 * elastic_hip.h -- C ABI of libelastic_hip.so: the MI355X (gfx950) kernels behind the patched global/local
 * denoising loop of ElasticDiffusion.generate_image().
 *
 * The reference (MoayedHajiAli/ElasticDiffusion-official) is pure Python and has NO FFI / plugin boundary; its glue
 * is eager torch ops inside /root/reference/elastic_diffusion.py ("ED:n" below) and
 * /root/reference/elastic_diffusion_w_controlnet.py ("EDC:n").  Each entry point names the reference lines it
 * replaces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - return value: hipError_t as int (0 == hipSuccess); never throws, never allocates, never synchronises;
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors), NCHW contiguous unless stated;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); NULL = default stream;
 *   - "dtype" arguments: ED_F32 / ED_F16 / ED_BF16 select the element type of the buffer crossing the torch model
 *     boundary (UNet / VAE / ControlNet tensors).  All latent-space state is fp32;
 *   - latent-space glue entry points (everything up to ed_tile_accumulate_normalise, ed_assemble_rows,
 *     ed_phase_epilogue): arithmetic is fp32 with contraction disabled, in the operation order of the reference's
 *     torch-CPU path, so results are bit-identical to it for fp32 model tensors;
 *   - entry points inside the UNet (ed_geglu .. ed_flash_attention): 16-bit tensors, fp32 math, every value rounded to
 *     the 16-bit type where the torch kernel it replaces rounds; ed_flash_attention accumulates in fp32 on the MFMA
 *     units and rounds the softmax probabilities to the I/O type before the second contraction;
 *   - re-entrant, stateless; one host thread per process, one process per GPU.
 */
#ifndef ELASTIC_HIP_H
#define ELASTIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ED_F32 = 0, ED_F16 = 1, ED_BF16 = 2 };

/* ABI version, bumped on any signature change. */
int ed_version(void);

/* Name of the last HIP error code (hipGetErrorString), for the Python wrapper's RuntimeError. */
const char* ed_error_string(int err);

/*
 * ed_gather_views -- ED:706-757 crop_with_context (S == 1: one contiguous window per view) + ED:845 cat, and the
 * pad-to-model-size of ED:404-411 when a view is smaller than the UNet's native size.
 *   latent  f32 [B,C,H,W]
 *   out     dtype [(V*B),C,PH,PW]   row = v*B + b
 *   win_y0/win_x0  int32[V]  window origin in the latent (context included); windows are Sh x Sw
 *   (off_y,off_x)  where the window lands inside the PH x PW row; outside it the row is `frame` (f32 [C,PH,PW],
 *                  the noised-background strips of ED:366-391) or 0 when frame == NULL
 *   Source positions outside the latent read 0 (used by ed_tile_gather_pad).  divisor != 1 divides every value
 *   taken from the latent (IEEE division).
 */
int ed_gather_views(const float* latent, void* out, int dtype, int B, int C, int H, int W,
                    const int32_t* win_y0, const int32_t* win_x0, int V, int Sh, int Sw,
                    int PH, int PW, int off_y, int off_x, const float* frame, float divisor, void* stream);

/*
 * ed_scatter_centres -- ED:852-861: write each view's centre into the full-resolution buffer where it is still
 * exactly 0, in view order (first writer wins, decided by VALUE like the reference's `!= 0` mask).
 * Implemented as a gather per destination pixel over the (<= 2 x 2) views whose centre covers it, ascending view
 * index, stop at the first value != 0; no read-modify-write, `local` need not be zeroed.
 *   pred    dtype [(V*B),C,PH,PW]  UNet output rows, row = v*B + b, view v = rb*n_col_blocks + cb
 *   local   f32 [B,C,H,W]
 *   row_blk int32[H*2]  the row-blocks rb whose centre covers latent row Y (ascending, -1 = none)
 *   row_src int32[H*2]  the row inside the PH x PW prediction that holds latent row Y for that block
 *   col_blk / col_src   same for columns
 */
int ed_scatter_centres(const void* pred, int dtype, float* local, int B, int C, int H, int W,
                       int PH, int PW, int n_col_blocks,
                       const int32_t* row_blk, const int32_t* row_src,
                       const int32_t* col_blk, const int32_t* col_src, void* stream);

/*
 * ed_pick_assemble -- ED:560-630 random_nearest_downsample (value path) + ED:436 cat([latent]*2) + ED:404-411
 * background_pad, for K resampling steps at once.  The random choice itself stays on the host (torch CPU
 * generator, parity): idx holds one pick in [0,4) per reduced pixel and step.
 *   latent  f32 [B,C,H,W]
 *   idx     uint8 [K, h*w]
 *   src_row int32[2h] / src_col int32[2w]  latent row/col behind each line of the 2h x 2w pick grid (ED:565-613)
 *   frame   f32 [C,PH,PW] or NULL; the reduced latent sits at (off_y,off_x) inside the PH x PW model input
 *   out     dtype [(K*2*B),C,PH,PW]  row = (k*2 + j)*B + b, j = 0 uncond copy, 1 cond copy
 *   low     f32 [K,B,C,h,w] or NULL  the picked reduced-resolution latents (ED:678, 686)
 */
int ed_pick_assemble(const float* latent, const uint8_t* idx, const int32_t* src_row, const int32_t* src_col,
                     const float* frame, void* out, int dtype, float* low,
                     int K, int B, int C, int H, int W, int h, int w, int PH, int PW, int off_y, int off_x,
                     void* stream);

/*
 * ed_unpad_direction -- ED:429-430 crop padding + ED:439-440 chunk(2), direction = cond - uncond, for K steps.
 *   unet_out dtype [(K*2*B),C,PH,PW]  rows as written by ed_pick_assemble
 *   dirs     f32 [K,B,C,h,w]
 *   uncond_last f32 [B,C,h,w] or NULL: uncond score of step K-1 (ED:687)
 */
int ed_unpad_direction(const void* unet_out, int dtype, float* dirs, float* uncond_last,
                       int K, int B, int C, int h, int w, int PH, int PW, int off_y, int off_x, void* stream);

/*
 * ed_fill_directions -- ED:633-647 fill_in_from_downsampled_direction applied for K steps in order + ED:688.
 * Per full-resolution pixel: the LAST step whose pick mask covers it supplies the value; if none, step K-1
 * (fill_all).  Value = nearest-upsampled direction of that step at the pixel.
 *   dirs    f32 [K,B,C,h,w]
 *   stamp   int8 [h*w, 4]  stamp[n][q] = last step k whose pick at reduced pixel n was q, -1 if never (built on the
 *                          host from the same draws as idx: the K masks of ED:550-556 folded into one table)
 *   inv_row int32[H*2] / inv_col int32[W*2]  lines of the 2h x 2w pick grid that fold onto latent row/col
 *                                            (ED:446-465 restore_mask_shape, ED:622-628), -1 = none
 *   up_row int32[H] / up_col int32[W]        F.interpolate(nearest) source index reduced<-full (ED:636)
 *   down_row int32[h] / down_col int32[w]    F.interpolate(nearest) source index full<-reduced (ED:688)
 *   target  f32 [B,C,H,W];  low_dir f32 [B,C,h,w] or NULL
 */
int ed_fill_directions(const float* dirs, const int8_t* stamp,
                       const int32_t* inv_row, const int32_t* inv_col,
                       const int32_t* up_row, const int32_t* up_col,
                       const int32_t* down_row, const int32_t* down_col,
                       float* target, float* low_dir,
                       int K, int B, int C, int H, int W, int h, int w, void* stream);

/*
 * ed_cfg_ddim_step -- ED:1031/1053 noise = local + g * direction, then diffusers DDIMScheduler.step (eta = 0,
 * epsilon prediction; call sites ED:1033, 1054):
 *   x0   = (x - sqrt_beta_t * noise) / sqrt_alpha_t
 *   prev = sqrt_alpha_prev * x0 + sqrt_one_minus_alpha_prev * noise
 * All buffers f32 [n]; coefficient scalars are computed on the host in fp32 exactly like diffusers does.
 */
int ed_cfg_ddim_step(const float* local, const float* direction, const float* x, float* prev, float* x0,
                     float g, float sqrt_beta_t, float sqrt_alpha_t, float sqrt_alpha_prev,
                     float sqrt_one_minus_alpha_prev, int64_t n, void* stream);

/*
 * ed_undo_step -- ED:692-704 RePaint re-noising: n_sub sequential x <- a_k * x + b_k * noise_k in registers.
 *   noise f32 [n_sub, n] (drawn on the host generator, parity);  coef f32 [n_sub,2] = (sqrt(1-beta), sqrt(beta))
 */
int ed_undo_step(const float* x_in, const float* noise, const float* coef, float* x_out,
                 int n_sub, int64_t n, void* stream);

/*
 * ed_rrg_update -- ED:885-940 reduced_resolution_guidance on cached low-res scores + ED:1078:
 *   eps_low = low_uncond + g * low_dir;  x0_low = (low_latent - sqrt_beta_t * eps_low) / sqrt_alpha_t
 *   up      = nearest-upsample(x0_low);  out = prev - ((norm * (x0 - up)) * weight)
 * which is the closed form of the reference's autograd of weight * mse_loss(up, x0) w.r.t. x0 (norm = 2/numel of
 * ONE sample, ED:927-936).  Full-res buffers f32 [B,C,H,W], low-res f32 [B,C,h,w].
 */
int ed_rrg_update(const float* prev, const float* x0, const float* low_latent, const float* low_uncond,
                  const float* low_dir, const int32_t* up_row, const int32_t* up_col, float* out,
                  float g, float sqrt_beta_t, float sqrt_alpha_t, float norm, float weight,
                  int B, int C, int H, int W, int h, int w, void* stream);

/*
 * ed_gather2d -- generic table-driven 2-D gather used for ED:868-883 nearest_interpolate, the ControlNet
 * condition handling (EDC:457-461 zero pad, EDC:932-949 upsample + per-view crop) and debugging:
 *   out[n,c,i,j] = (rows[n,i] < 0 || cols[n,j] < 0) ? 0 : in[src_n[n], c, rows[n,i], cols[n,j]]
 *   in  in_dtype [Bin,C,H,W];  out out_dtype [N,C,oh,ow];  rows int32[N,oh];  cols int32[N,ow];  src_n int32[N]
 */
int ed_gather2d(const void* in, int in_dtype, void* out, int out_dtype, int C, int H, int W,
                const int32_t* src_n, const int32_t* rows, const int32_t* cols, int N, int oh, int ow, void* stream);

/*
 * ed_tile_gather_pad -- ED:287-300: tiles of the zero-padded latent for the tiled VAE decode, already divided by
 * the VAE scaling factor (ED:269).  Same kernel as ed_gather_views with out-of-range reads = 0.
 *   latent f32 [B,C,H,W] -> tiles dtype [(T*B),C,Ts,Ts], tile origin (tile_y0[t]-pad, tile_x0[t]-pad)
 */
int ed_tile_gather_pad(const float* latent, void* tiles, int dtype, int B, int C, int H, int W,
                       const int32_t* tile_y0, const int32_t* tile_x0, int T, int Ts, float scaling_factor,
                       void* stream);

/*
 * ed_tile_accumulate_normalise -- ED:271 (img/2 + 0.5).clamp(0,1) per tile, ED:303-308 image += centre;
 * count += 1; image / count.  Gather form: per output pixel, sum the covering tiles in ascending tile order.
 *   decoded dtype [(T*B),Cimg,TP,TP] raw VAE output (TP = Ts*scale), row = t*B + b, tile t = rb*n_col_tiles + cb
 *   image   f32 [B,Cimg,HP,WP]
 *   row_tile int32[HP*MAXC] / row_src int32[HP*MAXC]: tiles (row index rb, ascending, -1 = none) covering pixel
 *   row Y and the row inside the decoded tile; same for columns.  MAXC = ED_TILE_MAXC.
 */
#define ED_TILE_MAXC 4
int ed_tile_accumulate_normalise(const void* decoded, int dtype, float* image, int B, int Cimg, int HP, int WP,
                                 int TP, int n_col_tiles,
                                 const int32_t* row_tile, const int32_t* row_src,
                                 const int32_t* col_tile, const int32_t* col_src, void* stream);

/* ---- fused memory-bound kernels inside the UNet (model side of the boundary ED:422-426; csrc/unet_kernels.hip) ---- */

/*
 * ed_geglu -- the GEGLU activation of the transformer feed-forward (diffusers GEGLU, reached through ED:422):
 *   out[m,i] = in[m,i] * gelu(in[m,I+i])   (exact erf GELU), 16-bit in/out, fp32 math, torch's rounding points.
 *   in dtype [M, 2I] row-major, out dtype [M, I]; I % 8 == 0; dtype = ED_F16 | ED_BF16.
 */
int ed_geglu(const void* in, void* out, int dtype, int64_t M, int I, void* stream);

/*
 * ed_groupnorm -- GroupNorm over NCHW 16-bit activations with optional fused SiLU (ResnetBlock2D norm1/norm2,
 * conv_norm_out) and optional [N,HW,C] token-layout output (Transformer2DModel.norm + the permute that follows).
 *   x dtype [N,C,HW] (NCHW contiguous), gamma/beta dtype [C], out dtype [N,C,HW] or [N,HW,C];
 *   conv_bias dtype [C] or NULL, chan_bias dtype [N,C] or NULL: when given, the tensor that is normalised is
 *   round16(round16(x + conv_bias[c]) + chan_bias[n,c]) -- the bias of the convolution that produced x (run bias-free:
 *   MIOpen would add it in a separate broadcast kernel) and the time-embedding add `h + temb[:, :, None, None]` that
 *   precedes norm2 in ResnetBlock2D, folded into both passes with the roundings of the kernels they replace;
 *   workspace: caller-owned fp32 scratch of ed_groupnorm_workspace(N, C, HW, G) bytes, or NULL.  Groups larger than
 *   64 K elements are then normalised by two fully parallel launches (per-chunk Welford partials, merged in order --
 *   deterministic) instead of one workgroup per (sample, group); with NULL (or small groups: workspace size 0) the
 *   single-launch kernel runs.
 *   HW % 8 == 0, C % G == 0 (and (C/G) % 4 == 0 for tokens_out); dtype = ED_F16 | ED_BF16.
 */
int ed_groupnorm(const void* x, const void* gamma, const void* beta, const void* conv_bias, const void* chan_bias,
                 void* out, float* workspace, int dtype, int N, int C, int HW, int G, float eps, int act_silu,
                 int tokens_out, void* stream);
int64_t ed_groupnorm_workspace(int N, int C, int HW, int G);

/*
 * ed_bias_residual_add -- ResnetBlock2D's closing add with the convolution biases folded in:
 *   out[n,c,p] = round16(res[n,c,p] (+ res_bias[c])) + round16(h[n,c,p] + h_bias[c])
 * h = conv2 output (bias-free), res = the block input or the bias-free 1x1 shortcut convolution; either bias may be NULL.
 * 16-bit; channels_last == 0: [N,C,HW] NCHW contiguous, HW % 8 == 0; channels_last != 0: [N,HW,C] memory (torch
 * channels_last), C % 8 == 0; dtype = ED_F16 | ED_BF16.
 */
int ed_bias_residual_add(const void* h, const void* h_bias, const void* res, const void* res_bias, void* out, int dtype,
                         int N, int C, int HW, int channels_last, void* stream);

/*
 * ed_layernorm -- LayerNorm over the last dimension of [M, D] 16-bit activations (BasicTransformerBlock.norm1/2/3):
 * one wavefront per row, two passes over registers (mean, centred variance), one read + one write per element.
 *   D % 8 == 0, D <= 2048; gamma / beta dtype [D]; dtype = ED_F16 | ED_BF16.
 */
int ed_layernorm(const void* x, const void* gamma, const void* beta, void* out, int dtype, int64_t M, int D, float eps,
                 void* stream);
/*
 * ---- the fp32-residual-stream mode of the UNet (round 6; models.UNet2DConditionModel.residual_fp32) -----------------------------
 * The tolerance mode for configurations whose plain-fp16 latent ends outside north_star's 1e-3 (cfg2): the tensor that persists from
 * block to block is fp32, every branch computes in the 16-bit model dtype.  These entry points are the memory-bound layers that READ
 * the stream: same arithmetic as their 16-bit namesakes, x (ed_layernorm_s32, ed_groupnorm_nhwc_s32) or b and sum_out
 * (ed_add_layernorm_s32: sum_out = a + b in fp32, unrounded) are fp32, everything else -- gamma, beta, a, out -- has `dtype`.
 * ed_groupnorm_nhwc_s32 takes no folded biases; workspace: ed_groupnorm_nhwc_workspace.  Same shape limits as the namesakes.
 */
int ed_layernorm_s32(const void* x, const void* gamma, const void* beta, void* out, int dtype, int64_t M, int D, float eps,
                     void* stream);
int ed_add_layernorm_s32(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* out,
                         int dtype, int64_t M, int D, float eps, void* stream);
int ed_groupnorm_nhwc_s32(const void* x, const void* gamma, const void* beta, void* out, float* workspace, int dtype, int N, int C,
                          int HW, int G, float eps, int act_silu, void* stream);

/*
 * ed_add_layernorm -- residual add + LayerNorm in one pass (BasicTransformerBlock: `x = attn(norm(x)) + x` and the
 * `norm(x)` that follows): sum_out = round16(a + b), out = LayerNorm(sum_out).  [M, D] 16-bit, same limits as
 * ed_layernorm; sum_out may alias a or b.
 */
int ed_add_layernorm(const void* a, const void* b, const void* gamma, const void* beta, void* sum_out, void* out,
                     int dtype, int64_t M, int D, float eps, void* stream);

/*
 * ed_tokens_add_nchw -- out[n,c,p] = x[n,c,p] + tokens[n,p,c]: the residual add that closes a Transformer2DModel
 * (token layout back to NCHW), through a 64 x 64 LDS tile so both layouts are read / written with 16-byte vectors.
 *   x, out dtype [N,C,HW]; tokens dtype [N,HW,C]; C % 64 == 0, HW % 64 == 0; dtype = ED_F16 | ED_BF16.
 */
int ed_tokens_add_nchw(const void* x, const void* tokens, void* out, int dtype, int N, int C, int HW, void* stream);

/*
 * ed_groupnorm_nhwc -- the same GroupNorm [+ SiLU] for channels-last activations: x / out dtype [N, HW, C] (the memory
 * of an NCHW tensor in torch.channels_last format, which is also the transformer's token layout).  Three launches
 * (partial sums, finalise in double, vectorised apply); `workspace` is caller-owned fp32 scratch of
 * ed_groupnorm_nhwc_workspace(N, C, HW, G) bytes.  conv_bias [C] / chan_bias [N,C] (optional) as in ed_groupnorm.
 * C % 8 == 0, C % G == 0, C / G >= 8, G <= 256; dtype = ED_F16 | ED_BF16.
 */
int ed_groupnorm_nhwc(const void* x, const void* gamma, const void* beta, const void* conv_bias, const void* chan_bias,
                      void* out, float* workspace, int dtype, int N, int C, int HW, int G, float eps, int act_silu,
                      void* stream);
int64_t ed_groupnorm_nhwc_workspace(int N, int C, int HW, int G);

/*
 * ed_groupnorm_nhwc_cat -- ed_groupnorm_nhwc of cat([x1, x2], channel axis) without the concatenated tensor ever existing: the up
 * blocks' `ResnetBlock2D(torch.cat([hidden, skip], 1))` (diffusers' CrossAttnUpBlock2D / UpBlock2D behind ED:393-432) reads the two
 * channels-last sources in place.  x1 dtype [N, HW, C1], x2 dtype [N, HW, C2], out dtype [N, HW, C1 + C2]; C1 % 8 == 0 (a 16-byte
 * vector never straddles the sources; a GROUP may: the statistics are per-column sums).  No folded biases.  Same limits on
 * C = C1 + C2 and G as ed_groupnorm_nhwc; workspace: ed_groupnorm_nhwc_workspace(N, C1 + C2, HW, G).  Bit-identical to
 * ed_groupnorm_nhwc on the materialised concatenation.
 */
int ed_groupnorm_nhwc_cat(const void* x1, const void* x2, const void* gamma, const void* beta, void* out, float* workspace, int dtype,
                          int N, int C1, int C2, int HW, int G, float eps, int act_silu, void* stream);

/*
 * ed_assemble_rows -- ed_pick_assemble + ed_gather_views in one launch: all rows of one fused model batch (K CFG pairs
 * of the randomly picked reduced latent + V context crops).  Arguments as in those two entry points ("g" = the global /
 * pick part with its own PH x PW and offsets, "v" = the view part); bit-identical to calling them one after the other.
 */
int ed_assemble_rows(const float* latent, int B, int C, int H, int W, const uint8_t* idx, const int32_t* src_row,
                     const int32_t* src_col, const float* gframe, void* g_rows, float* low, int K, int h, int w, int gPH,
                     int gPW, int g_off_y, int g_off_x, const int32_t* win_y0, const int32_t* win_x0,
                     const float* vframe, void* v_rows, int V, int Sh, int Sw, int vPH, int vPW, int v_off_y, int v_off_x,
                     int dtype, void* stream);

/*
 * ed_phase_epilogue -- everything between the model call and the next latent in one launch: ed_unpad_direction +
 * ed_fill_directions + ed_scatter_centres + ed_cfg_ddim_step and, when x_next != NULL, ed_rrg_update
 * (ED:429-443, 633-647 x K, 688, 852-861, 1031/1053 + scheduler.step, 886-940 + 1078).  Every output element is a gather
 * over the model output rows, so no intermediate (dirs / direction / local) ever goes through HBM.  Bit-identical to the
 * chain of separate entry points (same fp32 operation order).
 *   g_out dtype [(K*2*B),C,gPH,gPW], v_out dtype [(V*B),C,vPH,vPW]: model output rows; x f32 [B,C,H,W] the latent the
 *   phase started from; stamp / inv_* / up_* / down_* as ed_fill_directions; row_blk .. col_src as ed_scatter_centres;
 *   prev, x0 f32 [B,C,H,W] (always written); low_dir, uncond_last f32 [B,C,h,w] and direction, local f32 [B,C,H,W] are
 *   optional by-products (NULL = skip); x_next f32 [B,C,H,W] (optional) = prev + RRG term, needs low_latent f32 [B,C,h,w]
 *   (the last picked reduced latent, ed_pick_assemble's `low[K-1]`), rrg_norm = 2/(C*H*W) and rrg_weight.
 */
int ed_phase_epilogue(const void* g_out, const void* v_out, int dtype, const float* x, const int8_t* stamp,
                      const int32_t* inv_row, const int32_t* inv_col, const int32_t* up_row, const int32_t* up_col,
                      const int32_t* down_row, const int32_t* down_col, const int32_t* row_blk, const int32_t* row_src,
                      const int32_t* col_blk, const int32_t* col_src, const float* low_latent, float* prev, float* x0,
                      float* x_next, float* low_dir, float* uncond_last, float* direction, float* local, int K, int B,
                      int C, int H, int W, int h, int w, int gPH, int gPW, int g_off_y, int g_off_x, int vPH, int vPW,
                      int n_col_blocks, float g, float sqrt_beta_t, float sqrt_alpha_t, float sqrt_alpha_prev,
                      float sqrt_1m_alpha_prev, float rrg_norm, float rrg_weight, void* stream);

/*
 * ed_flash_attention -- fused attention forward of the UNet's transformer blocks (what diffusers' AttnProcessor2_0
 * does with F.scaled_dot_product_attention inside `self.unet(...)`, ED:422-426): out = softmax(scale * Q K^T) V per
 * (batch, head), never materialising the Nq x Nk score matrix.  MFMA 32x32x16 (bf16 / f16), online softmax in fp32.
 *   q   dtype [B, Nq, H, 64]   element strides q_sb (batch), q_sn (token); head stride 64, unit d stride
 *   k,v dtype [B, Nk, H, 64]   strides likewise (q/k/v may be column slices of one fused projection output)
 *   out dtype [B, Nq, H, 64]   strides o_sb, o_sn
 *   head_dim = 64 (SDXL, SD 2.x: every v_path below) or 40 / 80 / 160 (SD 1.x's 8 heads: one generic kernel with the head
 *   dimension zero-padded to a multiple of 32 inside the kernel, v_path ignored); dtype = ED_F16 | ED_BF16; q/k/v 16-byte aligned with strides % 8 == 0, out 8-byte aligned
 *   with strides % 4 == 0.  Nk need not be a multiple of the 64-key tile (cross-attention: 77 text tokens).
 *   v_path: kernel variant.  0 = V transposed on the fly by ds_read_b64_tr_b16, 1 = V^T tile staged in LDS; +2 = 64
 *   query rows per wave (0..3 give bit-identical results).  4 = software-pipelined kernel (softmax of tile t issued in
 *   the shadow of the MFMAs of tiles t+1 / t-1, deferred O rescale; K / V addressed with 32-bit offsets: returns
 *   hipErrorInvalidValue when (Nk + 128) * max(k_sn, v_sn) * 2 >= 2^31).  5 = 4 without the per-tile row maximum after the
 *   first tile (numerators against the standing reference; exact redo of a tile whose sum exceeds 2^6).  8 = small-KV kernel for Nk <= 96 (cross
 *   attention on the 77 text tokens: K / V staged once per 512 query rows, single pass, no online rescale).
 *   4, 5 and 8 agree with 0..3 to the rounding of P (same fp32 accumulation, different summation grouping).
 *   6 = 5 for EXPONENT-DOMAIN queries: the caller has multiplied q by scale * log2(e) (the model folds it into the query
 *   projection weights), `scale` is ignored and out = sum_k 2^(q.k) v / sum_k 2^(q.k).  The row reference -m is the initial
 *   value of the S accumulators (the MFMA's C operand), so a numerator is one v_exp_f32 of the accumulator: no FMA, no
 *   per-tile maximum (exact maximum of the first tile, then the lazy check of 5).  Same limits as 4 / 5; Nk >= 64.
 *   7 = 4 with every LDS operand read issued three MFMAs ahead of its use instead of one (a 4-deep register ring).
 *   9 / 10 = 4 / 5 with 8 waves (256 query rows) per workgroup: each K / V tile is staged once for twice the query rows.
 *   K / V tile loads of 4 .. 7, 9, 10 carry their whole byte offset in the per-lane offset: rows past Nk read as zeros by the
 *   buffer range check, which does not cover a scalar offset.  6, 7, 9 and 10 return hipErrorInvalidValue for Nk < 64 (less than one
 *   full key tile); the exponent-domain contract of 6 (q pre-multiplied, `scale` ignored) can only be honoured by the caller -- the
 *   Python wrapper is what checks it.
 */
int ed_flash_attention(const void* q, const void* k, const void* v, void* out, int dtype, int B, int H, int Nq, int Nk,
                       int head_dim, int64_t q_sb, int64_t q_sn, int64_t k_sb, int64_t k_sn, int64_t v_sb, int64_t v_sn,
                       int64_t o_sb, int64_t o_sn, float scale, int v_path, void* stream);

/*
 * ed_groupnorm_f32 -- GroupNorm (+SiLU) of an fp32 NCHW activation: the VAE's normalisation layers (AutoencoderKL inside
 * ED:270 decode / ED:350 encode; fp32 like the reference keeps it, ED:328).  HW % 4 == 0.  Two launches: partial sums over
 * 64 K-element chunks of every (sample, group), then an apply pass whose blocks combine their group's partials in double.
 *   x, out f32 [N, C, HW]; gamma, beta f32 [C]; workspace: ed_groupnorm_f32_workspace(N, C, HW, G) bytes
 */
int64_t ed_groupnorm_f32_workspace(int N, int C, int HW, int G);
int ed_groupnorm_f32(const void* x, const void* gamma, const void* beta, void* out, float* workspace, int N, int C, int HW,
                     int G, float eps, int act_silu, void* stream);

/*
 * ed_softmax_rows -- x[r, :] = softmax(scale * x[r, :]) in place, fp32, rows x cols contiguous (cols % 4 == 0).  The VAE
 * mid-block attention (AutoencoderKL inside ED:270 decode / ED:350 encode; one 512-wide head, fp32) runs as
 * Q K^T (library fp32 GEMM) -> ed_softmax_rows -> S V (library fp32 GEMM) instead of the AOTriton SDPA kernel.
 */
int ed_softmax_rows(void* x, int64_t rows, int64_t cols, float scale, void* stream);

/*
 * ---- dense contractions inside the UNet (csrc/gemm_kernels.hip): one 256 x 256 x 64, 8-wave, 8-phase MFMA main loop
 * (LDS-DMA staging with counted vmcnt, st_16x32-swizzled LDS image) behind three entry points.  16-bit I/O, fp32
 * accumulation, ONE rounding of the epilogue's fp32 result.  All pointers 16-byte aligned; byte sizes of x and w below
 * 2^31 (32-bit buffer offsets); hipErrorInvalidValue for a shape outside these limits (the caller keeps the library call).
 *
 * ed_geglu_gemm -- diffusers GEGLU.forward (`h, gate = proj(x).chunk(2, -1); h * gelu(gate)`) in one kernel:
 *   out[m, n] = (x[m,:] . w[n,:] + bias[n]) * gelu(x[m,:] . w[I+n,:] + bias[I+n])
 *   x [M, K], w [2I, K] (torch Linear weight), bias [2I] or NULL, out [M, I]; K % 64 == 0, I % 128 == 0.
 *   gelu = x Phi(x) with erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7), evaluated on the fp32 accumulator: the
 *   projection output is never rounded to 16 bits and never touches HBM.
 */
int ed_geglu_gemm(const void* x, const void* w, const void* bias, void* out, int dtype, int64_t M, int K, int I, void* stream);

/*
 * ed_linear -- torch.nn.functional.linear with an optional fused residual: out = x w^T + bias (+ residual)
 *   x [M, K], w [N, K], bias [N] or NULL, residual [M, N] or NULL, out [M, N]; K % 64 == 0, N % 8 == 0.
 */
int ed_linear(const void* x, const void* w, const void* bias, const void* residual, void* out, int dtype, int64_t M, int K, int N,
              void* stream);

/*
 * ed_conv3x3_nhwc -- Conv2d(Cin, N, 3, stride 1, padding 1) on a channels-last image as an implicit GEMM (K = 9 Cin; the A
 * operand row of output pixel m at tap (dy, dx) is the Cin vector of pixel m + dy W + dx, zeros outside the image), with
 * ResnetBlock2D's adds in the epilogue:
 *   out[b,y,x,n] = sum_{dy,dx,c} x[b,y+dy,x+dx,c] w[n,dy,dx,c] + bias[n] + sample_bias[b,n] + residual[b,y,x,n]
 *   x [B,H,W,Cin], w [N,3,3,Cin] (a torch Conv2d weight in channels_last memory format), bias [N] / sample_bias [B,N] /
 *   residual [B,H,W,N] or NULL, out [B,H,W,N]; Cin % 64 == 0, N % 8 == 0.
 */
int ed_conv3x3_nhwc(const void* x, const void* w, const void* bias, const void* sample_bias, const void* residual, void* out,
                    int dtype, int B, int H, int W, int Cin, int N, void* stream);

/*
 * ed_conv3x3_nhwc_up2x -- Upsample2D (diffusers: F.interpolate(x, scale_factor = 2, mode = "nearest") + conv 3x3, behind ED:393-432) as ONE
 * launch: ed_conv3x3_nhwc whose A operand is the 2x nearest-neighbour upsampling of x, never written -- output pixel (y, x) at tap
 * (dy, dx) reads source pixel ((y + dy) >> 1, (x + dx) >> 1).  x dtype [B, H/2, W/2, Cin] channels-last, H and W = the OUTPUT size (even),
 * w / bias / out as in ed_conv3x3_nhwc; no sample bias, no residual.  Same products in the same order as ed_conv3x3_nhwc on the
 * materialised upsampling: bit-identical to it.
 */
int ed_conv3x3_nhwc_up2x(const void* x, const void* w, const void* bias, void* out, int dtype, int B, int H, int W, int Cin, int N,
                         void* stream);

/*
 * ed_conv3x3_nhwc_s2 -- Downsample2D of the UNet (diffusers: conv 3x3, stride 2, padding 1, behind ED:393-432) on the same main loop: output
 * pixel (y, x) at tap (dy, dx) reads input pixel (2 y + dy, 2 x + dx), zeros outside.  x dtype [B, 2H, 2W, Cin] channels-last, H and W =
 * the OUTPUT size, w [N, 3, 3, Cin] / bias [N] / out [B, H, W, N] as in ed_conv3x3_nhwc; no sample bias, no residual.
 */
int ed_conv3x3_nhwc_s2(const void* x, const void* w, const void* bias, void* out, int dtype, int B, int H, int W, int Cin, int N,
                       void* stream);

/*
 * ---- the fp32 VAE's ResnetBlock convolutions on split 16-bit operands (csrc/vae_kernels.hip + the fp32-output epilogue of the
 * GEMM main loop; round 5).  The reference runs the VAE in fp32 (elastic_diffusion.py:267-310 decode, :327-364 pad-strip encodes,
 * kept out of autocast at :328); these entry points keep fp32 accuracy (~3e-7 relative per convolution) on the 16-bit MFMA pipe:
 * an fp32 value v is carried as hi = fp16(v), lo = fp16(v - hi), and  x.w = xh.wh + xl.wh + xh.wl  is ONE 16-bit convolution over
 * 3 Cin channels, x'' = [xh | xl | xh] per pixel against w'' = [wh | wh | wl] per tap, accumulated in fp32 by the MFMA.
 *
 * ed_groupnorm_nhwc_f32 -- GroupNorm [+ SiLU] of an fp32 channels-last activation x [N, HW, C] (statistics in fp32 partial sums
 *   combined in double, fixed order):  split16 = 0: out fp32 [N, HW, C];  split16 = 1: out fp16 [N, HW, 3 C] = [hi | lo | hi], the
 *   A operand of ed_conv3x3_nhwc_f32out (after GroupNorm + SiLU the values are bounded by the affine parameters: fp16's range is
 *   safe; hi and lo nevertheless SATURATE at +-65504 -- a value beyond the range gives a finite, clamped operand, never inf - inf).  (C / G) % 4 == 0, G <= 256; workspace: ed_groupnorm_nhwc_f32_workspace bytes; pointers 16-byte aligned.
 * ed_conv3x3_nhwc_f32out -- ed_conv3x3_nhwc's main loop with an fp32 epilogue:
 *   out[b,y,x,n] = out_scale * sum_{dy,dx,c} x[b,y+dy,x+dx,c] w[n,dy,dx,c] + bias[n] + residual[b,y,x,n]      (no 16-bit rounding)
 *   x fp16 [B,H,W,Cin'] and w fp16 [N,3,3,Cin'] (Cin' = 3 Cin for split operands; any Cin' % 64 == 0 works), bias fp32 [N] or NULL,
 *   residual fp32 [B,H,W,N] or NULL, out fp32 [B,H,W,N]; dtype must be ED_F16; out_scale: a power of two that undoes the
 *   pre-scaling of the split weights (which keeps wl out of fp16's subnormal range); act_absmax: NULL, or the device float that
 *   ed_split_f32_nhwc scaled the activation by (the result is multiplied by 2^e, see there).  N % 8 == 0, 32-bit operand offsets as above.
 * ed_split_f32_nhwc -- the same (hi, lo) split for a RAW fp32 channels-last activation x [N, H, W, C] (the VAE decoder's upsampler
 *   convolutions, whose input is the un-normalised stream): out fp16 [N, U H, U W, 3 C] = [hi | lo | hi], U = 2 with upsample2x (nearest-
 *   neighbour upsampling folded into the write), else 1.  The raw stream is NOT bounded (the real SDXL decoder stream leaves fp16's
 *   range -- it is why that VAE produces NaNs in fp16), so the split is made exact over the whole fp32 range by a per-tensor power of
 *   two (round 6):  absmax -> e = max(0, exponent(absmax) - 14);  the kernel splits 2^-e v (|2^-e v| < 2^15: hi never
 *   saturates) and ed_conv3x3_nhwc_f32out multiplies its result by 2^e (its `act_absmax` argument: the SAME device float).  Scaling
 *   by a power of two is exact, so hi + lo = v to 2^-22 |v| as in the bounded case.  absmax = NULL: no scaling; hi AND lo then saturate
 *   at +-65504 (finite, never inf - inf; exact up to |v| = 65504, 11 bits up to 1.3e5).  C % 4 == 0; pointers 16-byte aligned.
 * ed_absmax_f32 -- out[0] = max |x[i]| over n fp32 values (bit pattern maximum: a NaN input yields a NaN, an Inf an Inf), the
 *   per-tensor figure above.  One device float, written by this call in stream order (memset + atomic max of the non-negative bit
 *   patterns: order-independent, bit-reproducible).
 */
int ed_absmax_f32(const void* x, int64_t n, float* out, void* stream);
int ed_split_f32_nhwc(const void* x, void* out, int N, int C, int H, int W, int upsample2x, const float* absmax, void* stream);
int64_t ed_groupnorm_nhwc_f32_workspace(int N, int C, int HW, int G);
int ed_groupnorm_nhwc_f32(const void* x, const void* gamma, const void* beta, void* out, float* workspace, int N, int C, int HW, int G,
                          float eps, int act_silu, int split16, void* stream);
int ed_conv3x3_nhwc_f32out(const void* x, const void* w, const float* bias, const float* residual, float* out, int dtype, int B, int H, int W,
                           int Cin, int N, float out_scale, const float* act_absmax, void* stream);

/*
 * ed_conv3x3_nhwc_f32out_s2 -- the VAE encoder's Downsample2D (diffusers: F.pad(x, (0, 1, 0, 1)) + conv 3x3, stride 2, padding 0; AutoencoderKL
 * behind ED:327-364) on the split-operand main loop at fp32 accuracy: output pixel (y, x) at tap (ty, tx) in 0..2 reads input pixel
 * (2 y + ty, 2 x + tx), zeros past the bottom / right edge.  x fp16 [B, 2H, 2W, Cin'] = the split operand of the raw fp32 stream
 * (ed_split_f32_nhwc; Cin' = 3 Cin), H and W = the OUTPUT size, w / bias / out / out_scale / act_absmax as in ed_conv3x3_nhwc_f32out; no residual.
 */
int ed_conv3x3_nhwc_f32out_s2(const void* x, const void* w, const float* bias, float* out, int dtype, int B, int H, int W, int Cin, int N,
                              float out_scale, const float* act_absmax, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ELASTIC_HIP_H */
