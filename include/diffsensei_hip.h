/* diffsensei_hip.h — C ABI of the MI355X (gfx950) DiffSensei sampling kernels.
 *
 * This is the drop-in boundary for ONE path of jianzongwu/DiffSensei: the SDXL-UNet denoising loop with
 * region-masked IP-Adapter cross-attention (reference src/pipelines/pipeline_diffsensei.py:310-337,
 * src/models/unet.py:116-347, src/models/attention_processor.py, src/models/resampler.py).
 * The reference has no FFI of its own (pure Python over torch/diffusers); these entry points are what a
 * ctypes binding for that path binds (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch.Tensor.data_ptr()); the library never
 *     allocates or frees device memory and keeps no reference after the call returns (plans excepted: a plan
 *     borrows every pointer baked into its ops until ds_plan_destroy);
 *   - activations are fp16, channels-last: images [B, H*W, C], token matrices [rows, C];
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *     asynchronous on it, no hidden synchronisation;
 *   - return value: 0 = ok, negative = error (message via ds_last_error()); nothing throws or aborts.
 */
#ifndef DIFFSENSEI_HIP_H
#define DIFFSENSEI_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* ds_last_error(void);
int ds_version(void);
/* number of HIP devices visible / properties of the current one (sanity for loaders) */
int ds_device_info(int* cu_count, int* lds_bytes, char* arch_name, int arch_name_len);
/* Tuning / test knobs.  THREAD-LOCAL since round 5 (no global mutable state beside the last-error slot - which is
 * thread-local too): a value set here changes only the launches the CALLING THREAD issues afterwards; other threads, and so
 * other serving handles of the process, keep the automatic dispatch.  Every knob defaults to 0 = automatic, which is what
 * production runs.
 * They exist so that A/B runs and the parity tests can force a kernel variant the automatic rule would only pick at
 * large problem sizes.  Variants of one operator that share their arithmetic are bit-identical and the tests assert torch.equal
 * between them (GEMM families, conv block sizes, flash attention <1> / <2>, ip_attn variants, hipGraph vs eager); the ones that
 * re-associate a sum - self_attn_sp_kernel vs the flash kernels, gn_variant 1's chunking - are compared at a stated tolerance.
 *   "gemm_variant"       0 auto | 1 register-staged only | 2, 7 two-buffer LDS-DMA | 8, 9 one-buffer LDS-DMA |
 *                        3 256x256 ping-pong (gemm_pp_kernel) | 10 halo-patch conv | 11 64x160 tiles (gemm_t160_kernel)
 *   "gemm_t160"          0 (default) small-batch projections whose 64x160 (or, failing that, 128x160) grid is one block per CU
 *                        run gemm_t160_kernel | 1 never (A/B) | 2 never its 128-row tile | 3 128-row tiles wherever it runs (tests)
 *   "gemm_g320"          0 (default) ds_gemm_g320_fits follows its shape rule | 1 it answers 0 (A/B: planners then keep the
 *                        128-row GEGLU packing and the 128 x 128 / 256 x 256 kernels)
 *   "gemm_pp_even"       1 (default) gemm_pp_kernel's persistent grid = ceil(tiles / rounds) blocks, every round full |
 *                        0 one block per CU with a partial last round
 *   "gemm_ring"          0 (default) grids of <= 512 64x128 blocks (num_samples 1) use the ring-buffered
 *                        gemm_glds_kernel<64,false,3|4> | 1 never (A/B: profiles/r02_ring_in_pipeline_ab.txt)
 *   "gemm_pp_narrow"     0 (default) the N, K <= 640 projections run gemm_pp_kernel where its ragged last tile column takes the
 *                        branch-free epilogue (plain epilogue, N % 64 == 0, M % 256 == 0) | 1 never: the 128x128 kernels, the
 *                        rule of rounds 1-5 (A/B: profiles/r06_pp_narrow_ab.txt)
 *   "conv_halo_variant"  0 auto (16x16-pixel blocks from 1024 blocks on; the ring-buffered 8x16 kernel for grids of at most
 *                        "conv_deep_blocks" blocks per CU) | 1 force conv_halo_kernel (8x16 pixels) | 2 force conv_halo256_kernel
 *                        (16x16 pixels) | 3 force conv_halo_deep_kernel (8x16 pixels, W ring of three buffers: small grids) |
 *                        4 auto without conv_halo_deep_kernel (A/B)
 *   "conv_deep_blocks"   grids of <= this many 8x16 blocks per CU run conv_halo_deep_kernel (default 1; 0 = never)
 *   "attn_variant"       0 auto (>= 128 blocks of 256 query rows: the software-pipelined self_attn_sp_kernel - every UNet shape
 *                        at every batch; smaller grids: self_attn_kernel<1>) | 1 force self_attn_kernel<1> (32 rows per wave) |
 *                        2 force self_attn_kernel<2> (64 rows per wave) | 3 force self_attn_sp_kernel | 4 the same with the
 *                        plain block order instead of the per-XCD head grouping (A/B: profiles/r03_self_attn_sp.txt)
 *   "ip_attn_min_blocks" grid size below which ip_attn_kernel stops doubling its query tiles per block (default 1024)
 *   "ip_attn_variant"    0 / 1 the register-staged ip_attn_kernel<4,false> | 2 force the 8-wave LDS-DMA ring kernel
 *                        ip_attn_kernel<8,true> where N % 256 == 0 - bit-identical results; faster back to back, slower inside the
 *                        UNet forward, so never automatic (A/B: profiles/r04_ipattn_ring_ab.txt, r04_forward_option_ab.txt) |
 *                        3 the register-staged kernel WITHOUT the round-6 specialisation that never issues the work of the padding
 *                        keys 80..95 (taken automatically when both key counts lie in (64, 80]; bit-identical; A/B)
 *   "gn_variant"         0 (default) GroupNorm on 512-thread blocks with >= 64 rows per block | 1 the round-3 geometry
 *                        (256 threads, 8-row chunks; A/B: profiles/r04_gn_geometry_ab.txt)
 *   "llm_gemv_variant"   0 auto (software-pipelined persistent GEMV) | 1 one column per wavefront | 2 un-pipelined streaming GEMV
 *   "gemm_debug"         bits 0..7: ablation builds of gemm_pp_kernel (only in a library built with -DDS_ABLATION; 0 otherwise) |
 *                        bit 8 (256): gemm_pp_kernel drains a tile's C stores before the next tile's first k-tile instead of
 *                        letting them retire under it - same bits out (tests/test_gpu_ops.py::test_gemm_pingpong_tile_handover) |
 *                        bit 10 (1024): halo-patch convs use the row-index patch swizzle (2-way LDS bank conflicts; A/B only)
 * returns 0, or -1 (ds_last_error()) for an unknown key / out-of-range value */
int ds_set_option(const char* key, int value);
/* Debug counters kept by the kernels on the current device (test instrumentation; SYNCHRONISES the device - never on a serving path).
 *   "attn_sp_recentre"  how many times self_attn_sp_kernel's rare re-centring branch ran (per wave and query block) since the
 *                       last reset: lets a model-level test prove that its logits crossed the 2^14 threshold
 *                       (tests/test_gpu_outlier_magnitudes.py).
 * *value receives the count; reset != 0 clears it afterwards.  Returns 0, or -1 for an unknown name. */
int ds_debug_counter(const char* name, int reset, long long* value);
/* Host-side query of the dispatch rule (no GPU work): 1 when a plain f16 GEMM of this shape runs gemm_t160_kernel (64 x 160
 * tiles, one block per CU: the M = 2048, N = 1280 projections of a batch-1 request at 1024 x 1024).  A launch planner asks
 * before it requests that kernel's LayerNorm statistics format (DsOp GEMM i[11] = 160: three partial pairs per row and
 * 160-column tile - columns 0..63, 64..127, 128..159) and tells the consumers to add 3 N / 160 entries (i[10]); a direct
 * ds_gemm_ln_* call always gets the 64-column format. */
int ds_gemm_t160_fits(int M, int N, int K, int batch);
/* Host-side query of the dispatch rule (no GPU work): 1 when the GEGLU projection [M, N packed] x K of a small-batch request
 * should run gemm_g320_kernel (256 x 320 tiles, exactly one block per CU: M = 2048, N = 10240, K = 1280 at UNet batch 2,
 * 1024 x 1024 - diffusers' GEGLU [3P] reached from reference src/models/unet.py:244-338).  That kernel reads W / bias / c
 * packed in groups of 320 rows (160 hidden rows, then their 160 gate rows) and is selected by epilogue code 4 (below);
 * a planner asks BEFORE packing.  Epilogue codes of the GEMM entry points: 0 none, 1 GEGLU (128-row groups: 64 hidden + 64
 * gates), 2 GELU, 3 QuickGELU, 4 GEGLU in 320-row groups. */
int ds_gemm_g320_fits(int M, int N, int K, int batch);
/* Host-side query: partial-sum chunks per image that a stride-1 3x3 convolution of this shape writes for the GroupNorm behind
 * it (DsOp CONV3X3 p[6] = the GroupNorm workspace; DsOp GROUPNORM i[6] = this number: the GroupNorm then skips its statistics
 * pass).  0: this convolution cannot (not a halo-patch kernel shape, or more than 128 pixel tiles per image).  Replaces the
 * first read of conv1's output by diffusers' ResnetBlock2D.norm2 [3P], reached from reference src/models/unet.py:244-338. */
int ds_conv3x3_gn_chunks(int B, int H, int W, int Cin, int Cout);

/* ------------------------------------------------------------------------------------------------
 * Operators.  Each replaces the torch call(s) named in its comment.
 * ---------------------------------------------------------------------------------------------- */

/* y[M,N] = act(x[M,K] @ w[N,K]^T + bias[N]) (+residual[M,N]).  epilogue: 0 none, 2 GELU(erf), 3 quick-GELU,
 * 1 GEGLU: w rows are packed per 128-row tile as 64 "hidden" + 64 "gate" rows and y[M,N/2] = hidden * gelu(gate).
 * replaces nn.Linear at reference src/models/attention_processor.py:56,63,64,84,207,225,226,245,246,261 and the
 * diffusers GEGLU/FeedForward/proj_in/proj_out linears reached from src/models/unet.py:244-338.
 * x2/k1: optional second A source for columns k >= k1 (channel concat of two tensors without a copy). */
int ds_gemm_f16(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int k1, const void* w, int64_t ldw,
                const void* bias, const void* residual, int64_t ldr, void* y, int64_t ldy, int M, int N, int K,
                int epilogue, void* stream);

/* LayerNorm fused into the GEMM pair around it (round 4).  Replaces the three `nn.LayerNorm` passes of diffusers'
 * BasicTransformerBlock [3P] that the reference reaches from src/models/unet.py:244-338 (norm2 -> attn2.to_q,
 * norm3 -> ff.net.0 GEGLU; attention_processor.py:209 takes the normalised hidden states as its query input):
 *   producer  y = x w^T + bias (+ residual), and `stats_out` [N/64][M] float2 receives the (sum, sum of squares) of every
 *             stored row per 64-column strip;
 *   ds_ln_finalize: partial sums -> stats [M] float2 (mean, rstd = rsqrt(var + eps));
 *   consumer  y = rstd_m (x_m . gw_n - mean_m c_n) + b'_n on the RAW x, with gw = gamma (.) w (f16), ln_c [N][2] f16 =
 *             (-c hi, -c lo), c_n = sum_k gw_nk, bias_ln = bias + w beta - all packed once at load time (GEGLU: in the packed
 *             row order).  epilogue 0 or 1 (GEGLU).  Either role may be absent (null pointers), both may be combined.
 * Two implementations, one statistics format.  ds_gemm_ln_fusable(M, N, K, epilogue, batch) says which one ds_gemm_f16's
 * dispatch gives the shape: 1 = the 256 x 256 persistent kernel (M, N multiples of 256, K of 128; consumers take the
 * finalised ln_stats of ds_ln_finalize through ds_gemm_ln_f16), 2 = the 128-wide kernels of small batches and of the
 * 640-channel level (N a multiple of 128, K of 64; producers through ds_gemm_ln_f16 with ln_stats = NULL, consumers through
 * ds_gemm_ln_partial_f16, which sums the K/64 partials of its own rows in the epilogue - no finalize launch), 0 = neither
 * (callers keep ds_layernorm_f16 + ds_gemm_f16).  Producers and consumers of either kind combine. */
int ds_gemm_ln_f16(const void* x, int64_t ldx, const void* gw, int64_t ldw, const void* bias_ln, const float* ln_stats,
                   const void* ln_c, const void* residual, int64_t ldr, void* y, int64_t ldy, float* stats_out, int M, int N,
                   int K, int epilogue, void* stream);
int ds_gemm_ln_partial_f16(const void* x, int64_t ldx, const void* gw, int64_t ldw, const void* bias_ln, const float* ln_partial,
                           float eps, const void* ln_c, const void* residual, int64_t ldr, void* y, int64_t ldy, float* stats_out,
                           int M, int N, int K, int epilogue, void* stream);
int ds_ln_finalize(const float* partial, float* stats, int M, int strips, int C, float eps, void* stream);
/* operand-swapped consumer (norm1 -> attn1.to_v, produced transposed: y[b] = a @ LN(x[b])^T, a = gamma (.) Wv [M,K] shared,
 * x[b] the raw rows [N,K] of batch item b): the normalised rows index the OUTPUT COLUMNS, so ln_stats is read at
 * [b * ln_bstride + n] and ln_cb is [M][4] f16 = (-c hi, -c lo, b' hi, b' lo) per output row. */
int ds_gemm_ln_swapped_f16(const void* a, int64_t lda, const void* x, int64_t ldx, int64_t sx, const float* ln_stats,
                           int64_t ln_bstride, const void* ln_cb, void* y, int64_t ldy, int64_t sy, int M, int N, int K,
                           int batch, void* stream);
/* the same on the 128-wide kernels: ln_partial = the producer's partial sums [K/64][ln_rows] float2 (ln_rows = rows of the
 * normalised matrix over all batch items); every block finalises the statistics of its 128 output columns itself. */
int ds_gemm_ln_swapped_partial_f16(const void* a, int64_t lda, const void* x, int64_t ldx, int64_t sx, const float* ln_partial,
                                   float eps, int64_t ln_rows, int64_t ln_bstride, const void* ln_cb, void* y, int64_t ldy,
                                   int64_t sy, int M, int N, int K, int batch, void* stream);
int ds_gemm_ln_fusable(int M, int N, int K, int epilogue, int batch);

/* batched variant: grid.z = batch with element strides (0 = shared operand); used for V^T = Wv @ X_b^T */
int ds_gemm_f16_batched(const void* x, int64_t ldx, int64_t sx, const void* w, int64_t ldw, int64_t sw, void* y,
                        int64_t ldy, int64_t sy, int M, int N, int K, int batch, void* stream);

/* ---- VAE decoder (bf16 storage, fp32 accumulate): replaces AutoencoderKL.decode reached from reference
 * src/pipelines/pipeline_diffsensei.py:339-367 (the reference upcasts the VAE to fp32 because fp16 overflows).
 * Same layouts as the f16 entry points; all tensors bf16 unless noted. */
/* y = conv3x3(x[B,H,W,Cin], w[Cout,3,3,Cin]) + bias (+ residual); stride 1; upsample != 0 = nearest x2 in front.
 * Cin % 64 == 0.  ResnetBlock2D.conv1/conv2, Upsample2D.conv of the decoder. */
int ds_conv3x3_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y, int B, int H,
                    int W, int Cin, int Cout, int upsample, void* stream);
/* y[M,N] = x[M,K] @ w[N,K]^T + bias (+ residual); M, N % 16 == 0, K % 128 == 0: conv_shortcut (1x1), to_q/k/out */
int ds_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, const void* residual,
                 int64_t ldr, void* y, int64_t ldy, int M, int N, int K, void* stream);
/* batched (element strides, 0 = shared operand): V^T[b] = Wv @ X_b^T */
int ds_gemm_bf16_batched(const void* x, int64_t ldx, int64_t sx, const void* w, int64_t ldw, int64_t sw, void* y,
                         int64_t ldy, int64_t sy, int M, int N, int K, int batch, void* stream);
/* GroupNorm (+SiLU) over [B,HW,C]; ws: ds_groupnorm_workspace_bytes(B, C) */
int ds_groupnorm_bf16(const void* x, void* y, const void* gamma, const void* beta, void* ws, int B, int HW, int C,
                      int groups, float eps, int silu, void* stream);
/* single-head attention, head dim 512: q,k,o [B,N,512], vt [B,512,N]; N % 8 == 0.  Decoder mid_block.attentions.0.
 * n_valid (0 = N): only keys [0, n_valid) take part - rows [n_valid, N) are the caller's padding (latent sizes whose
 * token count is not a multiple of 16 are padded on the host side of the VAE engine). */
int ds_wide_attn_bf16(const void* q, const void* k, const void* vt, void* o, int B, int N, int n_valid, float scale,
                      void* stream);
/* y[B,H,W,C] = conv_in(post_quant_conv(latents / scaling_factor)); latents fp32 NCHW [B,4,H,W]; post_quant_w [4,4],
 * post_quant_b [4] fp32; w [C,3,3,4], bias [C] bf16 */
int ds_vae_conv_in_bf16(const float* latents, const float* post_quant_w, const float* post_quant_b, const void* w,
                        const void* bias, void* y, int B, int H, int W, int C, float scaling_factor, void* stream);
/* image fp32 NCHW [B,3,H,W] = conv_out(x[B,H,W,C]); w [3,3,3,C], bias [3] bf16; denormalize != 0 also applies
 * VaeImageProcessor.denormalize, (x / 2 + 0.5).clamp(0, 1) (pipeline_diffsensei.py:367 postprocess) */
int ds_vae_conv_out_bf16(const void* x, const void* w, const void* bias, float* image, int B, int H, int W, int C,
                         int denormalize, void* stream);

/* ---- the same decoder in scaled fp16 (diffsensei_amd/vae.py, precision "fp16-scaled": what runs when the VAE config has
 * force_upcast, i.e. where the reference upcasts the VAE to fp32 "as it overflows in float16", pipeline_diffsensei.py:339-344).
 * All tensors f16.  Conv outputs and the residual stream are stored multiplied by S = 2^-6 (range 4.2e6, fp16's 11-bit
 * mantissa instead of bf16's 8), conv INPUTS are O(1): GroupNorm+SiLU outputs, pre-multiplied by S through `out_scale` so
 * that every accumulator already carries the factor; biases are pre-scaled on the host; GroupNorm of a scaled tensor is
 * exact with eps * S^2.  The 3x3 convs and linears go through ds_conv3x3_f16 / ds_gemm_f16 / ds_gemm_f16_batched. */
/* GroupNorm (+SiLU) over x[B,HW,C], y = act(norm(x)) * out_scale; ws: ds_groupnorm_workspace_bytes(B, C) */
int ds_groupnorm_scaled_f16(const void* x, void* y, const void* gamma, const void* beta, void* ws, int B, int HW, int C,
                            int groups, float eps, int silu, float out_scale, void* stream);
/* ds_wide_attn_bf16 / ds_vae_conv_in_bf16 / ds_vae_conv_out_bf16 with f16 tensors (same arguments) */
int ds_wide_attn_f16(const void* q, const void* k, const void* vt, void* o, int B, int N, int n_valid, float scale,
                     void* stream);
int ds_vae_conv_in_f16(const float* latents, const float* post_quant_w, const float* post_quant_b, const void* w,
                       const void* bias, void* y, int B, int H, int W, int C, float scaling_factor, void* stream);
int ds_vae_conv_out_f16(const void* x, const void* w, const void* bias, float* image, int B, int H, int W, int C,
                        int denormalize, void* stream);

/* 3x3 convolution, pad 1, NHWC: y[B,Ho,Wo,Cout] = conv(x[B,H,W,Cin], w[Cout,3,3,Cin]) + bias
 * (+ rowbias[b, :] per image — the resnet time_emb_proj term) (+ residual).  stride in {1,2};
 * upsample != 0 fuses a nearest x2 upsample in front (diffusers Upsample2D).  Cin % 64 == 0.
 * replaces ResnetBlock2D.conv1/conv2, Downsample2D.conv, Upsample2D.conv reached from src/models/unet.py:244-332. */
int ds_conv3x3_f16(const void* x, const void* w, const void* bias, const void* rowbias, int64_t rowbias_ld,
                   const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int stride, int upsample,
                   void* stream);
/* Upsample2D with an explicit output size (diffusers resizes to the skip tensor's size when a latent side is not a
 * multiple of 4 - `forward_upsample_size` [3P]): nearest resize of x [B,H,W,Cin] to Hout x Wout exactly as
 * F.interpolate(size=..., mode="nearest") indexes it (src = min(floor(dst * float(in)/out), in-1)), then the 3x3 conv.
 * The resized tensor never exists in HBM. */
int ds_conv3x3_resize_f16(const void* x, const void* w, const void* bias, const void* rowbias, int64_t rowbias_ld,
                          const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int Hout, int Wout,
                          void* stream);

/* GroupNorm (+SiLU) over NHWC; x2 (may be NULL) supplies channels C1..C1+C2 (skip concat).  ws: device scratch
 * of ds_groupnorm_workspace_bytes(B, C1+C2) bytes. */
size_t ds_groupnorm_workspace_bytes(int B, int C);
int ds_groupnorm_f16(const void* x1, const void* x2, void* y, const void* gamma, const void* beta, void* ws, int B,
                     int HW, int C1, int C2, int groups, float eps, int silu, void* stream);
int ds_layernorm_f16(const void* x, void* y, const void* gamma, const void* beta, int rows, int C, float eps,
                     void* stream);

/* Flash self-attention, head_dim 64.  q,k: [B,N,ld] with head h at column h*64; vt: [B,heads,64,ldv] (V stored
 * key-contiguous); o: [B,N,ldo].  replaces F.scaled_dot_product_attention at
 * reference src/models/attention_processor.py:76-78. */
int ds_self_attn_f16(const void* q, int64_t ldq, int64_t sq, const void* k, int64_t ldk, int64_t sk, const void* vt,
                     int64_t ldv, void* o, int64_t ldo, int64_t so, int B, int heads, int Nq, int Nk, float scale,
                     void* stream);

/* (Rounds 2-5 carried an OCP e4m3 variant of this call for BASELINE.json configs[4] - ds_quantize_fp8_e4m3_f16 /
 * ds_self_attn_fp8_f16 on v_mfma_f32_32x32x64_f8f6f4.  Removed in round 6: 5.4e-2 per-op error on white noise - 25x the fp16
 * kernels' - for +3.7 % at 2048 x 2048, and the fp16 Q K^T / fp8 P V hybrid cannot go below 3.7e-2
 * (tests/test_fp8_error_floor.py).  configs[4] is served by the fp16 kernels.) */

/* Fused text + region-masked IP cross-attention core of MaskedIPAttnProcessor2_0
 * (reference src/models/attention_processor.py:235-258 incl. prepare_attention_mask_ip :115-169):
 *   o = softmax(q kt^T * s) vt  +  ip_scale * softmax(q ki^T * s + M(bbox)) vi
 * kt/ki: [B,96,C] key panels (rows >= Lt / Li are padding), vtt/vti: [B,C,96] transposed value panels,
 * bbox: [B,max_ips,4] fp32 relative boxes, (mask_h, mask_w): the grid the reference infers from (N, aspect_ratio).
 * ip_scale_dev: optional device float overriding ip_scale (lets a captured graph follow set_ip_scale).
 * ldk/sk: row / batch stride of the key panels, sv: batch stride of the value panels (elements; 0 = dense) —
 * the engine projects the text/IP tokens for all 70 layers with one stacked GEMM and hands out column slices. */
int ds_masked_ip_attn_f16(const void* q, int64_t ldq, const void* kt, const void* vtt, const void* ki,
                          const void* vti, const float* bbox, void* o, int64_t ldo, int B, int heads, int N, int Lt,
                          int Li, int n_dummy, int tok_per_ip, int max_ips, int mask_h, int mask_w, float qk_scale,
                          float ip_scale, const float* ip_scale_dev, int64_t ldk, int64_t sk, int64_t sv,
                          void* stream);
/* debug/test hook: bit k of flags[b*N+i] = token i inside box k (the reference's inside_bbox_mask) */
int ds_ip_region_flags(const float* bbox, uint8_t* flags, int B, int N, int max_ips, int mask_h, int mask_w,
                       void* stream);

/* generic small attention (head_dim <= 256, arbitrary lengths): encoders + perceiver resampler
 * (reference src/models/resampler.py:67-72).  q/k/v/o: [B,N,ld] with head h at column h*D. */
int ds_small_attn_f16(const void* q, int64_t ldq, int64_t sq, const void* k, int64_t ldk, int64_t sk, const void* v,
                      int64_t ldv, int64_t sv, void* o, int64_t ldo, int64_t so, int B, int heads, int Nq, int Nk,
                      int D, float scale, void* stream);

/* causal variant (token t attends to tokens <= t): the SDXL CLIP text encoders reached through `encode_prompt`
 * (reference src/pipelines/pipeline_diffsensei.py:237-245 -> transformers CLIPTextModel [3P]) */
int ds_small_attn_causal_f16(const void* q, int64_t ldq, int64_t sq, const void* k, int64_t ldk, int64_t sk, const void* v,
                             int64_t ldv, int64_t sv, void* o, int64_t ldo, int64_t so, int B, int heads, int N, int D,
                             float scale, void* stream);
/* out[b,t,:] = tok_emb[ids[b,t],:] + pos_emb[t,:]  (CLIPTextEmbeddings [3P]); ids int32 [B,T] */
int ds_embed_tokens_f16(const int32_t* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int T, int D,
                        int vocab, void* stream);

/* conv_in (Cin=4) fused with UNetMangaModel.encode_dialog_bbox (reference src/models/unet.py:206-210, :88-114).
 * dialog_boxes: int32 [B,ndialog,4] pixel boxes (x1,y1,x2,y2), already truncated/clamped like the reference. */
int ds_conv_in_dialog_f16(const void* x, const void* w, const void* bias, const int32_t* dialog_boxes,
                          const void* dialog_emb, void* y, int B, int H, int W, int Cin, int Cout, int ndialog,
                          void* stream);
/* conv_out (Cout=4) on an already GroupNorm+SiLU'd input (reference src/models/unet.py:335-338) */
int ds_conv_out_f16(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin, int Cout,
                    void* stream);

/* y[M,N] = act_out(act_in(x)[M,K] @ w[N,K]^T + bias + addend), M small (time-embedding MLPs, time_emb_proj) */
int ds_skinny_linear_f16(const void* x, const void* w, const void* bias, const void* addend, void* y, int M, int N,
                         int K, int silu_in, int silu_out, void* stream);

/* Per-step scalar table (device, fp32, 8 floats per row):
 *   {timestep, c_in_div, k0, k1, k2, k3, c_in_div_next, guidance_scale}
 *   Euler: k0 = sigma_i, k1 = sigma_{i+1};  DDIM: k0..k3 = sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)
 * `step_ctr` (device int32, may be NULL = row 0) selects the row. */
int ds_timestep_embed_f16(const float* table, const int32_t* step_ctr, void* out, int B, int dim, int flip_sin_to_cos,
                          float freq_shift, void* stream);
int ds_add_time_ids_f16(const void* text_embeds, const void* time_ids, void* out, int B, int pooled_dim, int n_ids,
                        int dim, int flip_sin_to_cos, float freq_shift, void* stream);
/* classifier-free guidance + scheduler.step + next scale_model_input in one launch
 * (reference src/pipelines/pipeline_diffsensei.py:315-317, :333-337).  eps: NHWC [2ns,HW,4] (uncond first);
 * latents: NCHW [ns,4,HW] updated in place; model_in: NHWC [2ns,HW,4].  kind: 0 Euler, 1 DDIM. */
int ds_cfg_sampler_step_f16(const void* eps, void* latents, void* model_in, const float* table,
                            const int32_t* step_ctr, int ns, int HW, int kind, int do_cfg, void* stream);
int ds_prepare_model_input_f16(const void* latents, void* model_in, const float* table, const int32_t* step_ctr,
                               int ns, int HW, int do_cfg, void* stream);
int ds_nhwc_to_nchw_f16(const void* x, void* y, int B, int HW, int C, void* stream);
int ds_nchw_to_nhwc_f16(const void* x, void* y, int B, int HW, int C, void* stream);
int ds_pad_rows_f16(const void* x, void* y, int B, int rows_in, int rows_out, int row_off, int total_rows, int C,
                    void* stream);
/* tail of `image_processor.postprocess(image, output_type="pil")` (reference src/pipelines/pipeline_diffsensei.py:367;
 * diffusers pt_to_numpy + numpy_to_pil [3P]): image fp32 NCHW [B,3,H,W] in [0,1] -> out uint8 NHWC [B,H,W,3] =
 * (x * 255).round() with numpy's round-half-to-even.  H*W must be a multiple of 4. */
int ds_image_f32_to_u8_nhwc(const float* image, uint8_t* out, int B, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------
 * MLLM pre-pass: LLaMA greedy decoding with a KV cache.
 * Replaces, for batch 1, `self.llm.generate(...)` as driven by reference src/models/mllm/seed_x.py:121-136
 * (LlamaForCausalLM of src/models/mllm/modeling_llama_xformer.py:612-, logits processor of
 * src/models/mllm/generation.py:19-30).  All kernels read the step-varying scalars from a device-side state block
 *     int32 state[8] = {rows in the KV cache, tokens generated, finished flag, current token id,
 *                       max_new_tokens of this call, eos token id, spare, spare}
 * so one token step is a static launch list (DS_OP_LLM_* above) that replays as a hipGraph.
 * ---------------------------------------------------------------------------------------------- */
/* y[M,N] = x'[M,K] @ w[N,K]^T (+ residual).  `rms` != 0 puts a LlamaRMSNorm in front of the projection
 * (modeling_llama_xformer.py:77-82 + the q/k/v or gate/up projections :277-296): r_m = rsqrt(mean(x_m^2)+eps) and
 *   rms_gain [K] given:  x'[m][k] = f16(rms_gain[k] * f16(x[m][k] * r_m))   - the reference's rounding points exactly;
 *   rms_gain NULL:       y = r_m * (x @ w^T)                                  - for a caller that folded the gain into w.
 * `swiglu`: w is [2N,K] (gate rows, then up rows) and y = silu(x'.w_gate) * (x'.w_up) (LlamaMLP.forward :166-167).
 * M <= 16 rows per pass internally; HBM-bound weight streaming, one wavefront per output column. */
int ds_llm_gemv_f16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, const void* residual,
                    int64_t ldr, int M, int N, int K, int rms, const void* rms_gain, int swiglu, float eps, void* stream);
/* rotary embedding + KV-cache append + attention for M (<= 16) new rows at positions state[0] .. state[0]+M-1
 * (LlamaAttention.forward :192-244: causal inside the prompt chunk, every cached key for a single new token).
 * qkv: [M,(heads+2*kv_heads)*D] un-rotated q|k|v; caches [T_max,kv_heads*D]; rope tables fp32 [T_max,D/2];
 * D in {64,128}.  Does NOT advance state[0] (ds_llm_select_f16 / ds_llm_advance do, once per step). */
int ds_llm_attn_f16(const void* qkv, int64_t ldqkv, void* k_cache, void* v_cache, int64_t ldc, const float* rope_cos,
                    const float* rope_sin, void* out, int64_t ldo, const int32_t* state, int M, int heads,
                    int kv_heads, int D, int T_max, float scale, void* stream);
/* LlamaRMSNorm with its gain (the final `model.norm`, :595).  `feat` (may be NULL; M == 1): the row is also stored as
 * row state[1]-1 of a [max_out,H] buffer - the per-token hidden states seed_x.py:143 gathers. */
int ds_llm_rmsnorm_f16(const void* x, int64_t ldx, const void* gamma, void* y, int64_t ldy, void* feat,
                       const int32_t* state, int M, int H, int max_out, float eps, void* stream);
/* out[0,:] = embed_tokens[state[3]] */
int ds_llm_embed_f16(const void* table, const int32_t* state, void* out, int H, int vocab, void* stream);
/* greedy choice with AutoImageTokenGenerationProcessor (generation.py:19-30) folded in; chain = [<img>, <img_0> ..
 * <img_{n-1}>, </img>] (n_chain ids, 0 = no processor).  Appends to out_ids[state[1]++] (capacity out_cap), sets
 * state[3], adds `adv` to state[0], raises state[2] on the eos id (state[5]) or after state[4] ids.  A no-op once
 * state[2] is set. */
int ds_llm_select_f16(const void* logits, int V, const int32_t* chain, int n_chain, int out_cap, int adv,
                      int32_t* state, int32_t* out_ids, void* stream);
int ds_llm_advance(int32_t* state, int rows, void* stream); /* state[0] += rows (prompt chunks before the last) */
/* out = a*scale + b*(1-scale), n elements (multiple of 8): `img_gen_feat * mllm_scale + image_embeds * (1 - mllm_scale)`
 * (reference scripts/demo/gradio.py:108-109) */
int ds_blend_f16(const void* a, const void* b, void* out, int64_t n, float scale, void* stream);
/* act[M,I] = silu(gate_up[:, :I]) * gate_up[:, I:] - the prompt pass, where gate|up comes out of ds_gemm_f16 as [M,2I]
 * (LlamaMLP.forward, modeling_llama_xformer.py:166-167) */
int ds_llm_swiglu_f16(const void* gate_up, void* act, int M, int I, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Character-reference pre-processing on the device.  Replaces the Pillow resize that
 * `CLIPImageProcessor()` / `ViTImageProcessor()` run on the host (reference src/pipelines/pipeline_diffsensei.py:125-126)
 * bit for bit: 8-bit separable resample with 22-bit fixed-point taps (libImaging Resample.c), horizontal pass, 8-bit
 * intermediate, vertical pass; then centre crop, * rescale, (x - mean) / std, CHW fp32.
 * Tables (device int32, built once per (in, out, filter) by the host with Pillow's double arithmetic):
 *   first[o] = first source index of output o, count[o] = taps used, taps[o*ksize + k] = round(w * 2^22).
 * Images: RGB uint8, HWC, device.  mean3 / std3: HOST arrays of 3 floats.
 * ---------------------------------------------------------------------------------------------- */
/* dst[H,out_w,3] = horizontal pass of src[H,W,3] */
int ds_resize_h_u8(const uint8_t* src, int H, int W, const int32_t* first, const int32_t* count, const int32_t* taps,
                   int ksize, int out_w, uint8_t* dst, void* stream);
/* vertical pass of tmp[Ht,Wt,3] restricted to the crop window rows [top, top+out_h) x columns [left, left+out_w),
 * then out_f32[3,out_h,out_w] = (u8 * scale - mean) / std; out_u8 (may be NULL) receives the cropped bytes [out_h,out_w,3] */
int ds_resize_v_norm_u8(const uint8_t* tmp, int Ht, int Wt, const int32_t* first, const int32_t* count,
                        const int32_t* taps, int ksize, int top, int left, int out_h, int out_w, float scale,
                        const float* mean3, const float* std3, float* out_f32, uint8_t* out_u8, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Plans: a static launch list (one UNet forward, or forward + CFG + scheduler step) built once by the host
 * and replayed with zero host arithmetic — optionally as a captured hipGraph.
 * ---------------------------------------------------------------------------------------------- */
enum ds_opcode {
    DS_OP_GEMM = 1,          /* p: x, x2, w, y, bias, rowbias, residual, ln_stats, ln_c, stats_out (fused LayerNorm, see ds_gemm_ln_f16; i[8] = operand-swapped form, l[10] = ln_bstride; i[9] = ln_stats holds PARTIAL sums, f[0] = eps, l[11] = ln_rows: ds_gemm_ln_partial_f16 / ds_gemm_ln_swapped_partial_f16)   l: ldx ldx2 ldw ldy ldr sx sx2 sw sy sr
                                i: M N K K1 epilogue (0 none, 1 GEGLU in 128-row groups, 2 GELU, 3 QuickGELU, 4 GEGLU in 320-row groups: ds_gemm_g320_fits)
                                batch rowbias_ld rows_per_group; i[10] = strips a consumer of partial sums adds per
                                row (0 = K / 64), i[11] = statistics format a producer emits (0 = one entry per 64 columns; 160: ds_gemm_t160_fits) */
    DS_OP_CONV3X3 = 2,       /* p: x, w, y, bias, rowbias, residual, gn_partial (optional: GroupNorm workspace, see ds_conv3x3_gn_chunks)
                                i: B H W Cin Cout stride upsample rowbias_ld Hout Wout (upsample only; 0 0 = 2H x 2W) */
    DS_OP_GROUPNORM = 3,     /* p: x1, x2, y, gamma, beta, ws             i: B HW C1 C2 groups silu pre_chunks (0, or the number of
                                partial-sum chunks per image the producing convolution left in ws)   f: eps */
    DS_OP_LAYERNORM = 4,     /* p: x, y, gamma, beta                      i: rows C                   f: eps */
    DS_OP_SELF_ATTN = 5,     /* p: q, k, vt, o   l: ldq ldk ldv ldo sq sk so   i: B heads Nq Nk       f: scale */
    DS_OP_IP_ATTN = 6,       /* p: q, kt, vtt, ki, vti, bbox, o, ip_scale_dev   l: ldq ldo ldk sk sv
                                i: B heads N Lt Li n_dummy tok_per_ip max_ips mask_h mask_w   f: qk_scale ip_scale */
    DS_OP_CONV_IN = 7,       /* p: x, w, bias, boxes, demb, y             i: B H W Cin Cout ndialog */
    DS_OP_CONV_OUT = 8,      /* p: x, w, bias, y                          i: B H W Cin Cout */
    DS_OP_SKINNY = 9,        /* p: x, w, bias, addend, y                  i: M N K silu_in silu_out */
    DS_OP_TIMESTEP_EMBED = 10, /* p: table, ctr, out                      i: B dim flip               f: freq_shift */
    DS_OP_ADD_TIME_IDS = 11, /* p: text_embeds, time_ids, out             i: B pooled n_ids dim flip  f: freq_shift */
    DS_OP_SAMPLER_STEP = 12, /* p: eps, latents, model_in, table, ctr     i: ns HW kind do_cfg */
    DS_OP_PREP_INPUT = 13,   /* p: latents, model_in, table, ctr          i: ns HW do_cfg */
    DS_OP_ADVANCE = 14,      /* p: ctr */
    DS_OP_NHWC2NCHW = 15,    /* p: x, y                                   i: B HW C */
    DS_OP_NCHW2NHWC = 16,    /* p: x, y                                   i: B HW C */
    DS_OP_PAD_ROWS = 17,     /* p: x, y                                   i: B rows_in rows_out row_off total_rows C */
    DS_OP_SMALL_ATTN = 18,   /* p: q, k, v, o   l: ldq ldk ldv ldo sq sk sv so   i: B heads Nq Nk D   f: scale */
    DS_OP_LLM_GEMV = 19,     /* p: x, w, y, residual, rms_gain   l: ldx ldy ldr   i: M N K rms swiglu  f: eps */
    DS_OP_LLM_ATTN = 20,     /* p: qkv, k_cache, v_cache, rope_cos, rope_sin, out, state   l: ldqkv ldc ldo
                                i: M heads kv_heads D T_max                                           f: scale */
    DS_OP_LLM_RMSNORM = 21,  /* p: x, gamma, y, feat, state l: ldx ldy       i: M H max_out           f: eps */
    DS_OP_LLM_EMBED = 22,    /* p: table, state, out                         i: H vocab */
    DS_OP_LLM_SELECT = 23,   /* p: logits, chain, state, out_ids             i: V n_chain out_cap adv */
    DS_OP_LLM_ADVANCE = 24,  /* p: state                                     i: rows */
    /* 25, 26: retired (fp8 attention, rounds 2-5) */
    DS_OP_LN_FINALIZE = 27   /* p: partial, stats                            i: M strips C               f: eps */
};

typedef struct ds_op {
    int32_t code;
    int32_t i[16];
    float f[4];
    int64_t l[12];
    void* p[10];
} ds_op;

typedef struct ds_plan ds_plan;
int ds_op_run(const ds_op* op, void* stream); /* run one op immediately */
/* roofline accounting: kernel the op dispatches to (rocprofv3 spelling), algorithmic flops (2*MAC) and HBM bytes */
int ds_op_describe(const ds_op* op, char* name, int name_len, double* flops, double* bytes);
int ds_plan_create(const ds_op* ops, int n_ops, ds_plan** out);
int ds_plan_num_ops(const ds_plan* plan);
int ds_plan_run(ds_plan* plan, void* stream);            /* eager: one launch per op */
int ds_plan_capture(ds_plan* plan, void* stream);        /* capture the launch list into a hipGraph (once) */
int ds_plan_replay(ds_plan* plan, void* stream);         /* hipGraphLaunch of the captured graph */
int ds_plan_destroy(ds_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSENSEI_HIP_H */
