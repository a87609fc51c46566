// Internal launcher interface between the C-ABI layer / plan executor and the HIP kernels.
#pragma once
#include "ds_common.h"

// EPI_GEGLU320: GEGLU whose packed W rows come in groups of 320 (160 hidden + their 160 gates): gemm_g320_kernel only
enum { EPI_NONE = 0, EPI_GEGLU = 1, EPI_GELU = 2, EPI_QUICK_GELU = 3, EPI_GEGLU320 = 4 };

struct GemmParams {
    const half_t* A = nullptr;   // plain: [M,K] rows (lda);  conv: NHWC input [B,Hin,Win,Cin]
    const half_t* A2 = nullptr;  // plain only: second source for k >= K1 (channel concat without a copy)
    const half_t* W = nullptr;   // [N,K] rows (ldw), K contiguous; conv: K = (ky*3+kx)*Cin + ci
    half_t* C = nullptr;         // [M,N] rows (ldc)   (GEGLU: [M,N/2])
    const half_t* bias = nullptr;      // [N]
    const half_t* rowbias = nullptr;   // [M/rows_per_group, rowbias_ld]  (resnet time-embedding projection)
    const half_t* residual = nullptr;  // [M,N] rows (ldr), added after rounding to f16
    long lda = 0, lda2 = 0, ldw = 0, ldc = 0, ldr = 0;
    long sA = 0, sA2 = 0, sW = 0, sC = 0, sR = 0;  // batch strides (elements), grid.z
    int M = 0, N = 0, K = 0, K1 = 0;
    int rowbias_ld = 0, rows_per_group = 1;
    int epi = EPI_NONE;
    int conv = 0;
    int Hin = 0, Win = 0, Cin = 0, Hout = 0, Wout = 0, cstride = 1, upsample = 0;
    float up_sy = 0.5f, up_sx = 0.5f;  // upsample: nearest-resize scales float(Hin)/Hout, float(Win)/Wout (ATen's definition)
    int tiles_m = 0, tiles_n = 0;
    int nbatch = 1;  // gemm_pp_kernel only: batch items folded into the persistent tile walk (id -> item, tile); others use grid.z
    // ---- LayerNorm fused across a producer / consumer pair of gemm_pp_kernel launches (gemm_pp.hip header, "LayerNorm"):
    // producer: per-row partial sums of the STORED output, one (sum, sum of squares) pair per row and 64-column strip,
    //           laid out [N / 64][M] (float2); consumer: y = rstd_m * (x_m . gw_n - mu_m c_n) + b'_n with gw = gamma (.) W
    //           packed at load time, (mu, rstd) per row from ds_launch_ln_finalize, c_n = sum_k gw_nk as a NEGATED f16
    //           (hi, lo) pair per output column, b' = bias + W beta passed as `bias`.
    float* stats_out = nullptr;        // producer: [N/64][M] float2 partials (fast plain epilogue, nbatch 1)
    const float* ln_stats = nullptr;   // consumer: [M] float2 (mu, rstd)
    const half_t* ln_c = nullptr;      // consumer: [N][2] f16 (-c hi, -c lo), in the (packed) row order of W
    int ln_swapped = 0;                // consumer, operand-swapped form (V^T = Wv X_b^T): the normalised rows are the rows of W
                                       // (tile columns): ln_stats[b * ln_bstride + n], ln_c = [M][4] f16 (-c hi, -c lo, b' hi, b' lo)
    long ln_bstride = 0;               //   rows of the normalised matrix per batch item
    int ln_partial = 0;                // consumer on the 128-wide LDS-DMA kernels (gemm.hip, "Fused LayerNorm"): ln_stats points at the
    float ln_eps = 1e-5f;              //   producer's PARTIALS [K/64][rows] float2 and the epilogue finalises its own rows (eps = ln_eps)
    long ln_rows = 0;                  //   operand-swapped form only: rows of the normalised matrix over all batch items (stride of the partials)
    int ln_nstrips = 0;                // consumer of partials: entries per row to sum (0 = K / 64; 3 K / 160 behind a gemm_t160_kernel producer)
    int stats_strip = 0;               // producer: statistics format (0 = one entry per 64 columns).  160 is what gemm_t160_kernel emits - its
                                       //   160-column tiles hold no whole 64-column strips -: three entries per tile (columns 0..63, 64..127,
                                       //   128..159), [3 N / 160][M] float2; the launch planner requests it explicitly (and tells the
                                       //   consumers: ln_nstrips), a direct caller never gets it
    // ---- GroupNorm statistics out of the producing convolution (conv_halo.hip, round 6): the halo-patch kernels also write, per
    // (image, pixel tile, output channel), the (sum, sum of squares) of the f16 values they store - the partial-sum layout of
    // gn_stats_kernel, [B][gn_chunks][Cout] float2 - so the GroupNorm that follows skips its statistics pass (one read of x less)
    float* gn_partial = nullptr;
    int gn_chunks = 0;                 // pixel tiles per image of the kernel variant the dispatch picks (ds_conv_halo_gn_chunks)
    int dtype = DS_DTYPE_F16;  // element type of A / W / C / bias / residual (the pointers are 2-byte opaque): bf16 = VAE path
    int debug = 0;  // ablation only (ds_set_option "gemm_debug"): 1 skip MFMA, 2 skip tile loads — results are garbage
};
int ds_launch_gemm(const GemmParams& p, int batch, hipStream_t stream);
int ds_gemm_conv_gn_chunks(const GemmParams& p);  // gemm.hip: GroupNorm partial chunks per image the conv dispatch would emit (0: none)
int ds_gemm_ln_kind(int M, int N, int K, int batch, int epi);  // fused LayerNorm: 1 = gemm_pp_kernel, 2 = 128-wide kernels, 0 = none
bool ds_gemm_pp_applicable(const GemmParams& p);  // gemm_pp.hip: 256 x 256 ping-pong kernel takes this shape
int ds_launch_gemm_pp(const GemmParams& p, int batch, hipStream_t stream);
bool ds_conv_halo_applicable(const GemmParams& p);  // conv_halo.hip: halo-patch 3x3 convolution takes this shape
int ds_launch_conv_halo(const GemmParams& p, hipStream_t stream);
int ds_conv_halo_gn_chunks(const GemmParams& p);  // pixel tiles per image of the variant ds_launch_conv_halo would run; 0: no statistics (too many tiles / not applicable)
void ds_conv_halo_set_variant(int v);  // 0 auto, 1 8x16-pixel blocks, 2 16x16-pixel blocks, 3 ring-buffered 8x16 blocks, 4 auto without the ring-buffered kernel
void ds_conv_halo_set_deep_blocks(int v);  // the ring-buffered 8x16 kernel takes grids of <= v blocks per CU (default 1)
const char* ds_gemm_kernel_name(const GemmParams& p, int batch);  // the instantiation ds_launch_gemm dispatches to
void ds_gemm_set_debug(int v);
void ds_gemm_set_ring(int v);     // 0 auto (ring-buffered kernel for small grids), 1 never
void ds_gemm_set_pp_narrow(int v);  // 0 auto, 1: N, K <= 640 projections never on gemm_pp_kernel (A/B)
void ds_gemm_pp_set_even(int v);  // experiment: 1 = persistent grid shrunk so that every round of tiles is full
void ds_gemm_set_variant(int v);  // 0 auto, 1 register staging only, 2 glds (BM <= 128), 3 glds, BM = 256 when large
// gemm_t160.hip: 64 x 160 tiles, one block per CU, for the small-batch projections of the 1280-channel level
bool ds_gemm_t160_shape(int M, int N, int K, int batch);           // the automatic dispatch rule (host logic only)
int ds_gemm_t160_rows(int M, int N, int K, int batch);             // rows per tile the launcher uses there: 64 | 128 (0: the rule does not pick the kernel -> 64 when forced)
bool ds_gemm_t160_possible(const GemmParams& p, int batch);        // what the kernel can run at all (gemm_variant 11)
bool ds_gemm_t160_applicable(const GemmParams& p, int batch);      // possible && shape rule
int ds_launch_gemm_t160(const GemmParams& p, hipStream_t stream);
void ds_gemm_set_t160(int v);     // 0 auto, 1 never
// gemm_g320.hip: 256 x 320 tiles, one block per CU, for the GEGLU projection of a small-batch request (epi == EPI_GEGLU320)
bool ds_gemm_g320_shape(int M, int N, int K, int batch);           // the automatic dispatch rule (host logic only)
bool ds_gemm_g320_possible(const GemmParams& p, int batch);
bool ds_gemm_g320_plain_applicable(const GemmParams& p, int batch);   // the plain-epilogue form: possible && shape rule (q|k at M = 8192, N = 2560)
int ds_launch_gemm_g320(const GemmParams& p, hipStream_t stream);
void ds_gemm_set_g320(int v);     // 0 auto, 1 never

// ---- VAE decoder only (vae.hip) ---------------------------------------------------------------------
int ds_launch_wide_attn(const void* q, const void* k, const void* vt, void* o, int B, int N, int n_valid, int dtype,
                        float scale, hipStream_t stream);  // one head of dim 512: q,k,o [B,N,512]; vt [B,512,N]; keys >= n_valid masked
int ds_launch_vae_conv_in(const float* lat, const float* wpq, const float* bpq, const void* w, const void* bias, void* y,
                          int B, int H, int W, int C, float scaling_factor, int dtype, hipStream_t stream);
int ds_launch_vae_conv_out(const void* x, const void* w, const void* bias, float* img, int B, int H, int W, int C,
                           int denorm, int dtype, hipStream_t stream);

// ---- normalisation ---------------------------------------------------------------------------------
struct GroupNormParams {
    const half_t* x1 = nullptr;  // [B,HW,C1]
    const half_t* x2 = nullptr;  // [B,HW,C2] or null  (channels C1..C1+C2)
    half_t* y = nullptr;         // [B,HW,C1+C2]
    const half_t* gamma = nullptr;
    const half_t* beta = nullptr;
    float* ws = nullptr;         // workspace, ds_groupnorm_ws_floats() floats
    int B = 0, HW = 0, C1 = 0, C2 = 0, groups = 32;
    float eps = 1e-5f;
    int silu = 0;
    int dtype = DS_DTYPE_F16;  // bf16: VAE decoder path (x, y, gamma, beta are 2-byte opaque pointers)
    float out_scale = 1.0f;    // y = act(norm(x)) * out_scale (the VAE decoder's scaled-fp16 mode; exact for powers of two)
    int pre_chunks = 0;        // > 0: ws already holds [B][pre_chunks][C] float2 partial sums (written by the producing convolution,
                               //      GemmParams::gn_partial): no statistics launch
};
size_t ds_groupnorm_ws_floats(int B, int C);
void ds_groupnorm_set_variant(int v);  // 0 auto (round-4 geometry), 1 round-3 geometry (A/B)
int ds_launch_groupnorm(const GroupNormParams& p, hipStream_t stream);
int ds_launch_layernorm(const half_t* x, half_t* y, const half_t* gamma, const half_t* beta, int rows, int C,
                        float eps, hipStream_t stream);
// row statistics of a fused LayerNorm: partial [strips][M] float2 (sum, sum of squares) -> stats [M] float2 (mean, rstd)
int ds_launch_ln_finalize(const float* partial, float* stats, int M, int strips, int C, float eps, hipStream_t stream);
bool ds_gemm_pp_fast_path(int M, int N, int K, int batch, int epi);  // gemm.hip: such a GEMM runs gemm_pp_kernel's branch-free epilogues

// ---- attention -------------------------------------------------------------------------------------
struct SelfAttnParams {
    const half_t* q = nullptr;   // [B,N,*] rows (ldq), head h at column h*64
    const half_t* k = nullptr;   // [B,N,*] rows (ldk)
    const half_t* vt = nullptr;  // [B,heads,64,ldv]  (V transposed: keys contiguous)
    half_t* o = nullptr;         // [B,N,*] rows (ldo)
    long ldq = 0, ldk = 0, ldv = 0, ldo = 0;
    long sq = 0, sk = 0, so = 0;  // per-batch strides (elements)
    int B = 0, heads = 0, Nq = 0, Nk = 0;
    float scale = 0.125f;
    int xcd_map = 1;  // self_attn_sp_kernel: 1 = all query blocks of a head on one XCD (0: plain order, A/B via attn_variant 4)
};
int ds_launch_self_attn(const SelfAttnParams& p, hipStream_t stream);
int ds_launch_self_attn_sp(const SelfAttnParams& p, hipStream_t stream);
int ds_attn_sp_recentre_count(int reset, long long* value);   // debug counter of the rare re-centring branch (attention_sp.hip)
const char* ds_self_attn_kernel_name(int B, int heads, int Nq, int Nk);  // which kernel ds_launch_self_attn picks for this shape  // attention_sp.hip: software-pipelined variant
void ds_attn_set_variant(int v);  // 0 auto, 1 force 32 query rows per wave, 2 force 64 rows per wave, 3 force the software-pipelined kernel
void ds_ip_attn_set_min_blocks(int v);
void ds_ip_attn_set_variant(int v);  // 0 auto, 1 four-wave register-staged kernel, 2 eight-wave LDS-DMA ring kernel (N % 256 == 0), 3 four-wave without the T16 specialisation (A/B)

struct IPAttnParams {
    const half_t* q = nullptr;     // [B,N,C] rows (ldq)
    const half_t* kt = nullptr;    // text keys   [B,LP,C]      (rows >= Lt are padding)
    const half_t* vtt = nullptr;   // text values [B,C,LP]      (transposed)
    const half_t* ki = nullptr;    // ip keys     [B,LP,C]
    const half_t* vti = nullptr;   // ip values   [B,C,LP]
    const float* bbox = nullptr;   // [B,max_ips,4]
    half_t* o = nullptr;           // [B,N,C] rows (ldo)
    long ldq = 0, ldo = 0;
    long ldk = 0, sk = 0;          // key panels: row stride / batch stride (elements); 0 -> dense [B,LP,C]
    long sv = 0;                   // value panels: batch stride (elements); 0 -> dense [B,C,LP]
    int B = 0, heads = 0, N = 0, C = 0;
    int Lt = 77, Li = 80, LP = 96;
    int n_dummy = 16, tok_per_ip = 16, max_ips = 4;
    int mask_h = 0, mask_w = 0;    // grid the reference infers from (N, aspect_ratio)
    float qk_scale = 0.125f, ip_scale = 1.0f;
    const float* ip_scale_ptr = nullptr;  // device scalar; overrides ip_scale when set (graph-replay safe)
};
int ds_launch_ip_attn(const IPAttnParams& p, hipStream_t stream);
int ds_launch_small_attn(const half_t* q, const half_t* k, const half_t* v, half_t* o, long ldq, long ldk, long ldv,
                         long ldo, long sq, long sk, long sv, long so, int B, int heads, int Nq, int Nk, int D,
                         float scale, hipStream_t stream, int causal = 0);
int ds_launch_ip_region_flags(const float* bbox, uint8_t* flags, int B, int N, int max_ips, int mask_h, int mask_w,
                              hipStream_t stream);

// ---- small / elementwise ------------------------------------------------------------------------------
int ds_launch_conv_in(const half_t* x, const half_t* w, const half_t* bias, const int* dialog_boxes,
                      const half_t* dialog_emb, half_t* y, int B, int H, int W, int Cin, int Cout, int ndialog,
                      hipStream_t stream);
int ds_launch_conv_out(const half_t* x, const half_t* w, const half_t* bias, half_t* y, int B, int H, int W, int Cin,
                       int Cout, hipStream_t stream);
int ds_launch_skinny_linear(const half_t* x, const half_t* w, const half_t* bias, const half_t* addend, half_t* y,
                            int M, int N, int K, int act_in, int act_out, hipStream_t stream);
// Per-step scalar table: row i (8 floats) = {timestep, c_in_div, k0, k1, k2, k3, c_in_div_next, guidance};
// `ctr` is a device int selecting the row (null = row 0) so a captured hipGraph is step-independent.
int ds_launch_timestep_embed(const float* table, const int* ctr, half_t* out, int B, int dim, int flip,
                             float freq_shift, hipStream_t stream);
int ds_launch_add_time_ids(const half_t* text_embeds, const half_t* time_ids, half_t* out, int B, int pooled_dim,
                           int n_ids, int dim, int flip, float freq_shift, hipStream_t stream);

struct SamplerStepParams {
    const half_t* eps = nullptr;    // UNet output, NHWC [2*ns, HW, 4]: rows [0,ns) uncond, [ns,2ns) cond
    half_t* latents = nullptr;      // NCHW [ns,4,H,W], updated in place (fp16 between steps, like the reference)
    half_t* model_in = nullptr;     // NHWC [2*ns, HW, 4]: next step's scaled input (both CFG halves)
    const float* coef = nullptr;    // per-step scalar table (see elementwise.hip)
    int ns = 0, HW = 0, C = 4;
    int kind = 0;                   // 0 Euler, 1 DDIM
    int do_cfg = 1;
};
int ds_launch_sampler_step(const SamplerStepParams& p, const int* ctr, hipStream_t stream);
int ds_launch_prepare_model_input(const half_t* latents, half_t* model_in, const float* table, const int* ctr, int ns,
                                  int HW, int C, int do_cfg, hipStream_t stream);
int ds_launch_advance_counter(int* ctr, hipStream_t stream);
int ds_launch_nhwc_to_nchw(const half_t* x, half_t* y, int B, int HW, int C, hipStream_t stream);
int ds_launch_image_to_u8(const float* img, uint8_t* out, int B, int H, int W, hipStream_t stream);
int ds_launch_nchw_to_nhwc(const half_t* x, half_t* y, int B, int HW, int C, hipStream_t stream);
int ds_launch_embed_tokens(const int* ids, const half_t* tok_emb, const half_t* pos_emb, half_t* out, int B, int T,
                           int D, int vocab, hipStream_t stream);
int ds_launch_pad_rows(const half_t* x, half_t* y, int B, int rows_in, int rows_out, int row_off, int total_rows,
                       int C, hipStream_t stream);

// ---- MLLM pre-pass: LLaMA greedy decoding (llm.hip) ---------------------------------------------------
// device-side state block (int32[8]): {tokens in the KV cache, tokens generated, finished, current token,
//                                      max_new_tokens, eos id, spare, spare}
struct LlmGemvParams {
    const half_t* x = nullptr;         // [M,K] rows (ldx)
    const half_t* w = nullptr;         // [N,K] row-major; SwiGLU: [2N,K] = gate rows then up rows
    half_t* y = nullptr;               // [M,N] rows (ldy)
    const half_t* residual = nullptr;  // [M,N] rows (ldr) added after the fp16 rounding; may alias y
    long ldx = 0, ldy = 0, ldr = 0;
    int M = 0, N = 0, K = 0;
    int rms = 0;                       // LlamaRMSNorm in front of the projection: row m is scaled by rsqrt(mean(x[m]^2) + eps)
    const half_t* gain = nullptr;      // [K] RMSNorm weight.  Given: x' = f16(gain * f16(x * r)) exactly like the reference
                                       // (normalise in fp32, round to fp16, multiply by the fp16 gain, round) is what meets
                                       // w; null: the caller folded the gain into w and only r is applied to the dot
    int swiglu = 0;                    // y = silu(x.w[n]) * (x.w[n+N])
    float eps = 1e-6f;
};
int ds_launch_llm_gemv(const LlmGemvParams& p, hipStream_t stream);
void ds_llm_gemv_set_variant(int v);   // 0 auto (pipelined), 1 one column per wavefront, 2 streaming without pipelining

struct LlmAttnParams {
    const half_t* qkv = nullptr;       // [M, (heads + 2 kv_heads) * D] rows (ldqkv): q | k | v, not yet rotated
    half_t* kc = nullptr;              // key cache   [T_max, kv_heads*D] rows (ldc), rotated keys
    half_t* vc = nullptr;              // value cache [T_max, kv_heads*D]
    const float* rope_cos = nullptr;   // [T_max, D/2]
    const float* rope_sin = nullptr;
    half_t* out = nullptr;             // [M, heads*D] rows (ldo)
    const int* state = nullptr;        // state[0] = rows already in the cache
    long ldqkv = 0, ldc = 0, ldo = 0;
    int M = 0, heads = 0, kv_heads = 0, D = 0, T_max = 0;
    float scale = 1.0f;
};
int ds_launch_llm_attn(const LlmAttnParams& p, hipStream_t stream);
int ds_launch_llm_rmsnorm(const half_t* x, long ldx, const half_t* gamma, half_t* y, long ldy, half_t* feat,
                          const int* state, int M, int H, int max_out, float eps, hipStream_t stream);
int ds_launch_llm_embed(const half_t* table, const int* state, half_t* out, int H, int vocab, hipStream_t stream);
int ds_launch_llm_select(const half_t* logits, int V, const int* chain, int n_chain, int out_cap, int adv, int* state,
                         int* out_ids, hipStream_t stream);
int ds_launch_llm_advance(int* state, int rows, hipStream_t stream);
int ds_launch_blend(const half_t* a, const half_t* b, half_t* out, long n, float s, hipStream_t stream);
int ds_launch_llm_swiglu(const half_t* gu, half_t* act, int M, int I, hipStream_t stream);

// ---- character-reference pre-processing (preprocess.hip): Pillow's 8-bit separable resize + crop + normalise ----
int ds_launch_resize_h(const uint8_t* src, int H, int W, const int* first, const int* count, const int* taps, int ksize,
                       int out_w, uint8_t* dst, hipStream_t stream);
int ds_launch_resize_v_norm(const uint8_t* tmp, int Ht, int Wt, const int* first, const int* count, const int* taps,
                            int ksize, int top, int left, int out_h, int out_w, float scale, const float* mean3,
                            const float* std3, float* out_f32, uint8_t* out_u8, hipStream_t stream);
