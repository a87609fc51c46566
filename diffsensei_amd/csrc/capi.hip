// C ABI (include/diffsensei_hip.h) over the kernel launchers, plus the plan executor: a flat launch list
// that replays one UNet forward (+CFG+scheduler step) with no host arithmetic, eagerly or as a hipGraph.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/diffsensei_hip.h"
#include "ds_common.h"
#include "ds_kernels.h"

static thread_local char g_err[512] = "";
void ds_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define H(x) reinterpret_cast<const half_t*>(x)
#define HM(x) reinterpret_cast<half_t*>(x)
#define S(x) reinterpret_cast<hipStream_t>(x)

extern "C" {

const char* ds_last_error(void) { return g_err; }
int ds_version(void) { return 100; }

int ds_device_info(int* cu_count, int* lds_bytes, char* arch_name, int arch_name_len) {
    int dev = 0;
    DS_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    DS_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)prop.sharedMemPerBlock;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return 0;
}

int ds_set_option(const char* key, int value) {
    DS_REQUIRE(key != nullptr, "ds_set_option: null key");
    if (strcmp(key, "gemm_variant") == 0) {
        DS_REQUIRE(value >= 0 && value <= 11, "gemm_variant must be 0..11");
        ds_gemm_set_variant(value);
        return 0;
    }
    if (strcmp(key, "ip_attn_min_blocks") == 0) {
        ds_ip_attn_set_min_blocks(value);
        return 0;
    }
    if (strcmp(key, "gn_variant") == 0) {
        DS_REQUIRE(value >= 0 && value <= 1, "gn_variant must be 0 (auto) or 1 (round-3 geometry)");
        ds_groupnorm_set_variant(value);
        return 0;
    }
    if (strcmp(key, "ip_attn_variant") == 0) {
        DS_REQUIRE(value >= 0 && value <= 3, "ip_attn_variant must be 0 (auto), 1 (register-staged), 2 (LDS-DMA ring) or 3 (register-staged, padding keys 80..95 not skipped)");
        ds_ip_attn_set_variant(value);
        return 0;
    }
    if (strcmp(key, "attn_variant") == 0) {
        DS_REQUIRE(value >= 0 && value <= 4, "attn_variant must be 0..4");
        ds_attn_set_variant(value);
        return 0;
    }
    if (strcmp(key, "conv_halo_variant") == 0) {
        DS_REQUIRE(value >= 0 && value <= 4, "conv_halo_variant must be 0..4");
        ds_conv_halo_set_variant(value);
        return 0;
    }
    if (strcmp(key, "conv_deep_blocks") == 0) {
        DS_REQUIRE(value >= 0 && value <= 8, "conv_deep_blocks must be 0..8");
        ds_conv_halo_set_deep_blocks(value);
        return 0;
    }
    if (strcmp(key, "llm_gemv_variant") == 0) {
        DS_REQUIRE(value >= 0 && value <= 2, "llm_gemv_variant must be 0..2");
        ds_llm_gemv_set_variant(value);
        return 0;
    }
    if (strcmp(key, "gemm_t160") == 0) {
        DS_REQUIRE(value >= 0 && value <= 3, "gemm_t160 must be 0 (auto), 1 (off), 2 (no 128-row tiles) or 3 (128-row tiles wherever it runs)");
        ds_gemm_set_t160(value);
        return 0;
    }
    if (strcmp(key, "gemm_g320") == 0) {
        DS_REQUIRE(value >= 0 && value <= 1, "gemm_g320 must be 0 (auto) or 1 (off)");
        ds_gemm_set_g320(value);
        return 0;
    }
    if (strcmp(key, "gemm_ring") == 0) {
        DS_REQUIRE(value >= 0 && value <= 1, "gemm_ring must be 0 (auto) or 1 (off)");
        ds_gemm_set_ring(value);
        return 0;
    }
    if (strcmp(key, "gemm_pp_narrow") == 0) {
        DS_REQUIRE(value >= 0 && value <= 1, "gemm_pp_narrow must be 0 (auto) or 1 (N, K <= 640 never on gemm_pp_kernel)");
        ds_gemm_set_pp_narrow(value);
        return 0;
    }
    if (strcmp(key, "gemm_pp_even") == 0) {
        ds_gemm_pp_set_even(value);
        return 0;
    }
    if (strcmp(key, "gemm_debug") == 0) {
        ds_gemm_set_debug(value);
        return 0;
    }
    ds_set_error("ds_set_option: unknown key '%s'", key);
    return -1;
}

int ds_gemm_t160_fits(int M, int N, int K, int batch) { return ds_gemm_t160_shape(M, N, K, batch) ? 1 : 0; }
int ds_gemm_g320_fits(int M, int N, int K, int batch) { return ds_gemm_g320_shape(M, N, K, batch) ? 1 : 0; }

int ds_debug_counter(const char* name, int reset, long long* value) {
    DS_REQUIRE(name != nullptr, "ds_debug_counter: null name");
    if (strcmp(name, "attn_sp_recentre") == 0) return ds_attn_sp_recentre_count(reset, value);
    ds_set_error("ds_debug_counter: unknown counter '%s'", name);
    return -1;
}

int ds_gemm_f16(const void* x, int64_t ldx, const void* x2, int64_t ldx2, int k1, const void* w, int64_t ldw,
                const void* bias, const void* residual, int64_t ldr, void* y, int64_t ldy, int M, int N, int K,
                int epilogue, void* stream) {
    GemmParams p;
    p.A = H(x); p.lda = ldx; p.A2 = H(x2); p.lda2 = ldx2; p.K1 = x2 ? k1 : K;
    p.W = H(w); p.ldw = ldw; p.bias = H(bias); p.residual = H(residual); p.ldr = ldr;
    p.C = HM(y); p.ldc = ldy; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    return ds_launch_gemm(p, 1, S(stream));
}

int ds_gemm_ln_f16(const void* x, int64_t ldx, const void* gw, int64_t ldw, const void* bias_ln, const float* ln_stats,
                   const void* ln_c, const void* residual, int64_t ldr, void* y, int64_t ldy, float* stats_out, int M, int N,
                   int K, int epilogue, void* stream) {
    GemmParams p;
    p.A = H(x); p.lda = ldx; p.K1 = K; p.W = H(gw); p.ldw = ldw; p.bias = H(bias_ln); p.residual = H(residual); p.ldr = ldr;
    p.C = HM(y); p.ldc = ldy; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    p.ln_stats = ln_stats; p.ln_c = H(ln_c); p.stats_out = stats_out;
    DS_REQUIRE(ln_stats || stats_out, "ds_gemm_ln_f16: neither row statistics to consume nor to emit - use ds_gemm_f16");
    return ds_launch_gemm(p, 1, S(stream));
}

int ds_gemm_ln_partial_f16(const void* x, int64_t ldx, const void* gw, int64_t ldw, const void* bias_ln, const float* ln_partial,
                           float eps, const void* ln_c, const void* residual, int64_t ldr, void* y, int64_t ldy, float* stats_out,
                           int M, int N, int K, int epilogue, void* stream) {
    GemmParams p;
    p.A = H(x); p.lda = ldx; p.K1 = K; p.W = H(gw); p.ldw = ldw; p.bias = H(bias_ln); p.residual = H(residual); p.ldr = ldr;
    p.C = HM(y); p.ldc = ldy; p.M = M; p.N = N; p.K = K; p.epi = epilogue;
    p.ln_stats = ln_partial; p.ln_partial = 1; p.ln_eps = eps; p.ln_c = H(ln_c); p.stats_out = stats_out;
    DS_REQUIRE(ln_partial && ln_c, "ds_gemm_ln_partial_f16: the partial sums and c are required");
    return ds_launch_gemm(p, 1, S(stream));
}

int ds_ln_finalize(const float* partial, float* stats, int M, int strips, int C, float eps, void* stream) {
    return ds_launch_ln_finalize(partial, stats, M, strips, C, eps, S(stream));
}

int ds_gemm_ln_swapped_f16(const void* a, int64_t lda, const void* x, int64_t ldx, int64_t sx, const float* ln_stats,
                           int64_t ln_bstride, const void* ln_cb, void* y, int64_t ldy, int64_t sy, int M, int N, int K,
                           int batch, void* stream) {
    GemmParams p;
    p.A = H(a); p.lda = lda; p.sA = 0; p.K1 = K; p.W = H(x); p.ldw = ldx; p.sW = sx;
    p.C = HM(y); p.ldc = ldy; p.sC = sy; p.M = M; p.N = N; p.K = K;
    p.ln_stats = ln_stats; p.ln_c = H(ln_cb); p.ln_swapped = 1; p.ln_bstride = ln_bstride;
    DS_REQUIRE(ln_stats && ln_cb, "ds_gemm_ln_swapped_f16: statistics and (c, b') are required");
    return ds_launch_gemm(p, batch, S(stream));
}

int ds_gemm_ln_swapped_partial_f16(const void* a, int64_t lda, const void* x, int64_t ldx, int64_t sx, const float* ln_partial,
                                   float eps, int64_t ln_rows, int64_t ln_bstride, const void* ln_cb, void* y, int64_t ldy,
                                   int64_t sy, int M, int N, int K, int batch, void* stream) {
    GemmParams p;
    p.A = H(a); p.lda = lda; p.sA = 0; p.K1 = K; p.W = H(x); p.ldw = ldx; p.sW = sx;
    p.C = HM(y); p.ldc = ldy; p.sC = sy; p.M = M; p.N = N; p.K = K;
    p.ln_stats = ln_partial; p.ln_partial = 1; p.ln_eps = eps; p.ln_rows = ln_rows; p.ln_c = H(ln_cb); p.ln_swapped = 1;
    p.ln_bstride = ln_bstride;
    DS_REQUIRE(ln_partial && ln_cb, "ds_gemm_ln_swapped_partial_f16: partial sums and (c, b') are required");
    return ds_launch_gemm(p, batch, S(stream));
}

int ds_gemm_ln_fusable(int M, int N, int K, int epilogue, int batch) { return ds_gemm_ln_kind(M, N, K, batch, epilogue); }

int ds_gemm_f16_batched(const void* x, int64_t ldx, int64_t sx, const void* w, int64_t ldw, int64_t sw, void* y,
                        int64_t ldy, int64_t sy, int M, int N, int K, int batch, void* stream) {
    GemmParams p;
    p.A = H(x); p.lda = ldx; p.sA = sx; p.K1 = K; p.W = H(w); p.ldw = ldw; p.sW = sw;
    p.C = HM(y); p.ldc = ldy; p.sC = sy; p.M = M; p.N = N; p.K = K;
    return ds_launch_gemm(p, batch, S(stream));
}

static int conv3x3_impl(const void* x, const void* w, const void* bias, const void* rowbias, int64_t rowbias_ld,
                        const void* residual, void* y, int B, int H_, int W_, int Cin, int Cout, int stride,
                        int upsample, hipStream_t stream, int dtype = DS_DTYPE_F16, int out_h = 0, int out_w = 0,
                        float* gn_partial = nullptr, int* gn_chunks_out = nullptr) {
    DS_REQUIRE(stride == 1 || stride == 2, "conv3x3: stride must be 1 or 2");
    DS_REQUIRE(!(upsample && stride != 1), "conv3x3: upsample with stride 2 is not a thing");
    DS_REQUIRE((out_h == 0 && out_w == 0) || (upsample && out_h > 0 && out_w > 0),
               "conv3x3: an explicit output size needs the upsample flag");
    GemmParams p;
    p.conv = 1;
    p.A = H(x); p.W = H(w); p.ldw = 9L * Cin; p.bias = H(bias); p.rowbias = H(rowbias); p.rowbias_ld = (int)rowbias_ld;
    p.residual = H(residual); p.ldr = Cout; p.C = HM(y); p.ldc = Cout;
    p.Hin = H_; p.Win = W_; p.Cin = Cin; p.cstride = stride; p.upsample = upsample;
    p.Hout = upsample ? (out_h ? out_h : 2 * H_) : (stride == 2 ? (H_ + 1) / 2 : H_);
    p.Wout = upsample ? (out_w ? out_w : 2 * W_) : (stride == 2 ? (W_ + 1) / 2 : W_);
    if (upsample) {  // ATen: scale = float(input_size) / output_size
        p.up_sy = (float)H_ / (float)p.Hout;
        p.up_sx = (float)W_ / (float)p.Wout;
    }
    p.M = B * p.Hout * p.Wout; p.N = Cout; p.K = 9 * Cin; p.K1 = p.K;
    p.rows_per_group = p.Hout * p.Wout;
    p.dtype = dtype;
    p.gn_partial = gn_partial;
    if (gn_chunks_out) {   // host query only (ds_conv3x3_gn_chunks): no launch
        *gn_chunks_out = ds_gemm_conv_gn_chunks(p);
        return 0;
    }
    return ds_launch_gemm(p, 1, stream);
}

int ds_conv3x3_gn_chunks(int B, int H_, int W_, int Cin, int Cout) {
    int n = 0;
    if (B <= 0 || H_ <= 0 || W_ <= 0 || Cin <= 0 || Cout <= 0) return 0;
    conv3x3_impl(nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, B, H_, W_, Cin, Cout, 1, 0, nullptr, DS_DTYPE_F16, 0, 0, nullptr, &n);
    return n;
}

int ds_conv3x3_f16(const void* x, const void* w, const void* bias, const void* rowbias, int64_t rowbias_ld,
                   const void* residual, void* y, int B, int H_, int W_, int Cin, int Cout, int stride, int upsample,
                   void* stream) {
    return conv3x3_impl(x, w, bias, rowbias, rowbias_ld, residual, y, B, H_, W_, Cin, Cout, stride, upsample, S(stream));
}

int ds_conv3x3_resize_f16(const void* x, const void* w, const void* bias, const void* rowbias, int64_t rowbias_ld,
                          const void* residual, void* y, int B, int H_, int W_, int Cin, int Cout, int Hout, int Wout,
                          void* stream) {
    return conv3x3_impl(x, w, bias, rowbias, rowbias_ld, residual, y, B, H_, W_, Cin, Cout, 1, 1, S(stream), DS_DTYPE_F16, Hout,
                        Wout);
}

// ---- bf16 entry points: the VAE decoder (fp16 overflows there; the reference runs it in fp32) ----------------------
int ds_conv3x3_bf16(const void* x, const void* w, const void* bias, const void* residual, void* y, int B, int H_,
                    int W_, int Cin, int Cout, int upsample, void* stream) {
    return conv3x3_impl(x, w, bias, nullptr, 0, residual, y, B, H_, W_, Cin, Cout, 1, upsample, S(stream), DS_DTYPE_BF16);
}

int ds_gemm_bf16(const void* x, int64_t ldx, const void* w, int64_t ldw, const void* bias, const void* residual,
                 int64_t ldr, void* y, int64_t ldy, int M, int N, int K, void* stream) {
    GemmParams p;
    p.A = H(x); p.lda = ldx; p.K1 = K; p.W = H(w); p.ldw = ldw; p.bias = H(bias); p.residual = H(residual); p.ldr = ldr;
    p.C = HM(y); p.ldc = ldy; p.M = M; p.N = N; p.K = K; p.dtype = DS_DTYPE_BF16;
    return ds_launch_gemm(p, 1, S(stream));
}

int ds_gemm_bf16_batched(const void* x, int64_t ldx, int64_t sx, const void* w, int64_t ldw, int64_t sw, void* y,
                         int64_t ldy, int64_t sy, int M, int N, int K, int batch, void* stream) {
    GemmParams p;
    p.A = H(x); p.lda = ldx; p.sA = sx; p.K1 = K; p.W = H(w); p.ldw = ldw; p.sW = sw;
    p.C = HM(y); p.ldc = ldy; p.sC = sy; p.M = M; p.N = N; p.K = K; p.dtype = DS_DTYPE_BF16;
    return ds_launch_gemm(p, batch, S(stream));
}

int ds_groupnorm_bf16(const void* x, void* y, const void* gamma, const void* beta, void* ws, int B, int HW, int C,
                      int groups, float eps, int silu, void* stream) {
    GroupNormParams p;
    p.x1 = H(x); p.y = HM(y); p.gamma = H(gamma); p.beta = H(beta); p.ws = reinterpret_cast<float*>(ws);
    p.B = B; p.HW = HW; p.C1 = C; p.C2 = 0; p.groups = groups; p.eps = eps; p.silu = silu; p.dtype = DS_DTYPE_BF16;
    return ds_launch_groupnorm(p, S(stream));
}

int ds_wide_attn_bf16(const void* q, const void* k, const void* vt, void* o, int B, int N, int n_valid, float scale,
                      void* stream) {
    return ds_launch_wide_attn(q, k, vt, o, B, N, n_valid, DS_DTYPE_BF16, scale, S(stream));
}

int ds_vae_conv_in_bf16(const float* latents, const float* post_quant_w, const float* post_quant_b, const void* w,
                        const void* bias, void* y, int B, int H_, int W_, int C, float scaling_factor, void* stream) {
    return ds_launch_vae_conv_in(latents, post_quant_w, post_quant_b, w, bias, y, B, H_, W_, C, scaling_factor,
                                 DS_DTYPE_BF16, S(stream));
}

int ds_vae_conv_out_bf16(const void* x, const void* w, const void* bias, float* image, int B, int H_, int W_, int C,
                         int denormalize, void* stream) {
    return ds_launch_vae_conv_out(x, w, bias, image, B, H_, W_, C, denormalize, DS_DTYPE_BF16, S(stream));
}

// ---- f16 twins of the decoder entry points (the scaled-fp16 "upcast" mode of diffsensei_amd/vae.py)
int ds_groupnorm_scaled_f16(const void* x, void* y, const void* gamma, const void* beta, void* ws, int B, int HW, int C,
                            int groups, float eps, int silu, float out_scale, void* stream) {
    GroupNormParams p;
    p.x1 = H(x); p.y = HM(y); p.gamma = H(gamma); p.beta = H(beta); p.ws = reinterpret_cast<float*>(ws);
    p.B = B; p.HW = HW; p.C1 = C; p.C2 = 0; p.groups = groups; p.eps = eps; p.silu = silu; p.out_scale = out_scale;
    return ds_launch_groupnorm(p, S(stream));
}

int ds_wide_attn_f16(const void* q, const void* k, const void* vt, void* o, int B, int N, int n_valid, float scale,
                     void* stream) {
    return ds_launch_wide_attn(q, k, vt, o, B, N, n_valid, DS_DTYPE_F16, scale, S(stream));
}

int ds_vae_conv_in_f16(const float* latents, const float* post_quant_w, const float* post_quant_b, const void* w,
                       const void* bias, void* y, int B, int H_, int W_, int C, float scaling_factor, void* stream) {
    return ds_launch_vae_conv_in(latents, post_quant_w, post_quant_b, w, bias, y, B, H_, W_, C, scaling_factor,
                                 DS_DTYPE_F16, S(stream));
}

int ds_vae_conv_out_f16(const void* x, const void* w, const void* bias, float* image, int B, int H_, int W_, int C,
                        int denormalize, void* stream) {
    return ds_launch_vae_conv_out(x, w, bias, image, B, H_, W_, C, denormalize, DS_DTYPE_F16, S(stream));
}

size_t ds_groupnorm_workspace_bytes(int B, int C) { return ds_groupnorm_ws_floats(B, C) * sizeof(float); }

int ds_groupnorm_f16(const void* x1, const void* x2, void* y, const void* gamma, const void* beta, void* ws, int B,
                     int HW, int C1, int C2, int groups, float eps, int silu, void* stream) {
    GroupNormParams p;
    p.x1 = H(x1); p.x2 = H(x2); p.y = HM(y); p.gamma = H(gamma); p.beta = H(beta); p.ws = reinterpret_cast<float*>(ws);
    p.B = B; p.HW = HW; p.C1 = C1; p.C2 = x2 ? C2 : 0; p.groups = groups; p.eps = eps; p.silu = silu;
    return ds_launch_groupnorm(p, S(stream));
}

int ds_layernorm_f16(const void* x, void* y, const void* gamma, const void* beta, int rows, int C, float eps,
                     void* stream) {
    return ds_launch_layernorm(H(x), HM(y), H(gamma), H(beta), rows, C, eps, S(stream));
}

int ds_self_attn_f16(const void* q, int64_t ldq, int64_t sq, const void* k, int64_t ldk, int64_t sk, const void* vt,
                     int64_t ldv, void* o, int64_t ldo, int64_t so, int B, int heads, int Nq, int Nk, float scale,
                     void* stream) {
    SelfAttnParams p;
    p.q = H(q); p.k = H(k); p.vt = H(vt); p.o = HM(o);
    p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.sq = sq; p.sk = sk; p.so = so;
    p.B = B; p.heads = heads; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    return ds_launch_self_attn(p, S(stream));
}

int ds_masked_ip_attn_f16(const void* q, int64_t ldq, const void* kt, const void* vtt, const void* ki,
                          const void* vti, const float* bbox, void* o, int64_t ldo, int B, int heads, int N, int Lt,
                          int Li, int n_dummy, int tok_per_ip, int max_ips, int mask_h, int mask_w, float qk_scale,
                          float ip_scale, const float* ip_scale_dev, int64_t ldk, int64_t sk, int64_t sv,
                          void* stream) {
    IPAttnParams p;
    p.ldk = ldk; p.sk = sk; p.sv = sv;
    p.q = H(q); p.kt = H(kt); p.vtt = H(vtt); p.ki = H(ki); p.vti = H(vti); p.bbox = bbox; p.o = HM(o);
    p.ldq = ldq; p.ldo = ldo; p.B = B; p.heads = heads; p.N = N; p.C = heads * 64;
    p.Lt = Lt; p.Li = Li; p.LP = 96; p.n_dummy = n_dummy; p.tok_per_ip = tok_per_ip; p.max_ips = max_ips;
    p.mask_h = mask_h; p.mask_w = mask_w; p.qk_scale = qk_scale; p.ip_scale = ip_scale; p.ip_scale_ptr = ip_scale_dev;
    return ds_launch_ip_attn(p, S(stream));
}

int ds_ip_region_flags(const float* bbox, uint8_t* flags, int B, int N, int max_ips, int mask_h, int mask_w,
                       void* stream) {
    return ds_launch_ip_region_flags(bbox, flags, B, N, max_ips, mask_h, mask_w, S(stream));
}

int ds_small_attn_f16(const void* q, int64_t ldq, int64_t sq, const void* k, int64_t ldk, int64_t sk, const void* v,
                      int64_t ldv, int64_t sv, void* o, int64_t ldo, int64_t so, int B, int heads, int Nq, int Nk,
                      int D, float scale, void* stream) {
    return ds_launch_small_attn(H(q), H(k), H(v), HM(o), ldq, ldk, ldv, ldo, sq, sk, sv, so, B, heads, Nq, Nk, D, scale,
                                S(stream));
}

int ds_small_attn_causal_f16(const void* q, int64_t ldq, int64_t sq, const void* k, int64_t ldk, int64_t sk, const void* v,
                             int64_t ldv, int64_t sv, void* o, int64_t ldo, int64_t so, int B, int heads, int N, int D,
                             float scale, void* stream) {
    return ds_launch_small_attn(H(q), H(k), H(v), HM(o), ldq, ldk, ldv, ldo, sq, sk, sv, so, B, heads, N, N, D, scale,
                                S(stream), 1);
}

int ds_embed_tokens_f16(const int32_t* ids, const void* tok_emb, const void* pos_emb, void* out, int B, int T, int D,
                        int vocab, void* stream) {
    return ds_launch_embed_tokens(ids, H(tok_emb), H(pos_emb), HM(out), B, T, D, vocab, S(stream));
}

int ds_conv_in_dialog_f16(const void* x, const void* w, const void* bias, const int32_t* dialog_boxes,
                          const void* dialog_emb, void* y, int B, int H_, int W_, int Cin, int Cout, int ndialog,
                          void* stream) {
    return ds_launch_conv_in(H(x), H(w), H(bias), dialog_boxes, H(dialog_emb), HM(y), B, H_, W_, Cin, Cout, ndialog,
                             S(stream));
}

int ds_conv_out_f16(const void* x, const void* w, const void* bias, void* y, int B, int H_, int W_, int Cin, int Cout,
                    void* stream) {
    return ds_launch_conv_out(H(x), H(w), H(bias), HM(y), B, H_, W_, Cin, Cout, S(stream));
}

int ds_skinny_linear_f16(const void* x, const void* w, const void* bias, const void* addend, void* y, int M, int N,
                         int K, int silu_in, int silu_out, void* stream) {
    return ds_launch_skinny_linear(H(x), H(w), H(bias), H(addend), HM(y), M, N, K, silu_in, silu_out, S(stream));
}

// ---- MLLM pre-pass (llm.hip): LLaMA greedy decoding
static int llm_gemv_impl(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, const void* residual,
                         int64_t ldr, int M, int N, int K, int rms, int swiglu, float eps, hipStream_t st,
                         const void* rms_gain = nullptr) {
    LlmGemvParams g;
    g.x = H(x); g.w = H(w); g.y = HM(y); g.residual = H(residual); g.gain = rms ? H(rms_gain) : nullptr;
    g.ldx = ldx; g.ldy = ldy; g.ldr = ldr; g.M = M; g.N = N; g.K = K; g.rms = rms; g.swiglu = swiglu; g.eps = eps;
    return ds_launch_llm_gemv(g, st);
}

static int llm_attn_impl(const void* qkv, int64_t ldqkv, void* kc, void* vc, int64_t ldc, const float* rope_cos,
                         const float* rope_sin, void* out, int64_t ldo, const int32_t* state, int M, int heads,
                         int kv_heads, int D, int T_max, float scale, hipStream_t st) {
    LlmAttnParams a;
    a.qkv = H(qkv); a.kc = HM(kc); a.vc = HM(vc); a.rope_cos = rope_cos; a.rope_sin = rope_sin; a.out = HM(out);
    a.state = state; a.ldqkv = ldqkv; a.ldc = ldc; a.ldo = ldo;
    a.M = M; a.heads = heads; a.kv_heads = kv_heads; a.D = D; a.T_max = T_max; a.scale = scale;
    return ds_launch_llm_attn(a, st);
}

int ds_llm_gemv_f16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, const void* residual,
                    int64_t ldr, int M, int N, int K, int rms, const void* rms_gain, int swiglu, float eps, void* stream) {
    return llm_gemv_impl(x, ldx, w, y, ldy, residual, ldr, M, N, K, rms, swiglu, eps, S(stream), rms_gain);
}

int ds_llm_attn_f16(const void* qkv, int64_t ldqkv, void* k_cache, void* v_cache, int64_t ldc, const float* rope_cos,
                    const float* rope_sin, void* out, int64_t ldo, const int32_t* state, int M, int heads,
                    int kv_heads, int D, int T_max, float scale, void* stream) {
    return llm_attn_impl(qkv, ldqkv, k_cache, v_cache, ldc, rope_cos, rope_sin, out, ldo, state, M, heads, kv_heads, D,
                         T_max, scale, S(stream));
}

int ds_llm_rmsnorm_f16(const void* x, int64_t ldx, const void* gamma, void* y, int64_t ldy, void* feat,
                       const int32_t* state, int M, int Hd, int max_out, float eps, void* stream) {
    return ds_launch_llm_rmsnorm(H(x), ldx, H(gamma), HM(y), ldy, HM(feat), state, M, Hd, max_out, eps, S(stream));
}

int ds_llm_embed_f16(const void* table, const int32_t* state, void* out, int Hd, int vocab, void* stream) {
    return ds_launch_llm_embed(H(table), state, HM(out), Hd, vocab, S(stream));
}

int ds_llm_select_f16(const void* logits, int V, const int32_t* chain, int n_chain, int out_cap, int adv,
                      int32_t* state, int32_t* out_ids, void* stream) {
    return ds_launch_llm_select(H(logits), V, chain, n_chain, out_cap, adv, state, out_ids, S(stream));
}

int ds_llm_advance(int32_t* state, int rows, void* stream) { return ds_launch_llm_advance(state, rows, S(stream)); }

int ds_llm_swiglu_f16(const void* gate_up, void* act, int M, int I, void* stream) {
    return ds_launch_llm_swiglu(H(gate_up), HM(act), M, I, S(stream));
}

int ds_resize_h_u8(const uint8_t* src, int H_, int W_, const int32_t* first, const int32_t* count, const int32_t* taps,
                   int ksize, int out_w, uint8_t* dst, void* stream) {
    return ds_launch_resize_h(src, H_, W_, first, count, taps, ksize, out_w, dst, S(stream));
}

int ds_resize_v_norm_u8(const uint8_t* tmp, int Ht, int Wt, const int32_t* first, const int32_t* count,
                        const int32_t* taps, int ksize, int top, int left, int out_h, int out_w, float scale,
                        const float* mean3, const float* std3, float* out_f32, uint8_t* out_u8, void* stream) {
    return ds_launch_resize_v_norm(tmp, Ht, Wt, first, count, taps, ksize, top, left, out_h, out_w, scale, mean3, std3,
                                   out_f32, out_u8, S(stream));
}

int ds_blend_f16(const void* a, const void* b, void* out, int64_t n, float scale, void* stream) {
    return ds_launch_blend(H(a), H(b), HM(out), (long)n, scale, S(stream));
}

int ds_timestep_embed_f16(const float* table, const int32_t* step_ctr, void* out, int B, int dim, int flip,
                          float freq_shift, void* stream) {
    return ds_launch_timestep_embed(table, step_ctr, HM(out), B, dim, flip, freq_shift, S(stream));
}

int ds_add_time_ids_f16(const void* text_embeds, const void* time_ids, void* out, int B, int pooled_dim, int n_ids,
                        int dim, int flip, float freq_shift, void* stream) {
    return ds_launch_add_time_ids(H(text_embeds), H(time_ids), HM(out), B, pooled_dim, n_ids, dim, flip, freq_shift,
                                  S(stream));
}

int ds_cfg_sampler_step_f16(const void* eps, void* latents, void* model_in, const float* table,
                            const int32_t* step_ctr, int ns, int HW, int kind, int do_cfg, void* stream) {
    SamplerStepParams p;
    p.eps = H(eps); p.latents = HM(latents); p.model_in = HM(model_in); p.coef = table;
    p.ns = ns; p.HW = HW; p.kind = kind; p.do_cfg = do_cfg;
    return ds_launch_sampler_step(p, step_ctr, S(stream));
}

int ds_prepare_model_input_f16(const void* latents, void* model_in, const float* table, const int32_t* step_ctr,
                               int ns, int HW, int do_cfg, void* stream) {
    return ds_launch_prepare_model_input(H(latents), HM(model_in), table, step_ctr, ns, HW, 4, do_cfg, S(stream));
}

int ds_nhwc_to_nchw_f16(const void* x, void* y, int B, int HW, int C, void* stream) {
    return ds_launch_nhwc_to_nchw(H(x), HM(y), B, HW, C, S(stream));
}
int ds_nchw_to_nhwc_f16(const void* x, void* y, int B, int HW, int C, void* stream) {
    return ds_launch_nchw_to_nhwc(H(x), HM(y), B, HW, C, S(stream));
}
int ds_image_f32_to_u8_nhwc(const float* image, uint8_t* out, int B, int H_, int W_, void* stream) {
    return ds_launch_image_to_u8(image, out, B, H_, W_, S(stream));
}
int ds_pad_rows_f16(const void* x, void* y, int B, int rows_in, int rows_out, int row_off, int total_rows, int C,
                    void* stream) {
    return ds_launch_pad_rows(H(x), HM(y), B, rows_in, rows_out, row_off, total_rows, C, S(stream));
}

// ---------------------------------------------------------------------------------------- plan executor
static int run_op(const ds_op& o, hipStream_t st) {
    const int32_t* i = o.i;
    const int64_t* l = o.l;
    void* const* p = o.p;
    switch (o.code) {
        case DS_OP_GEMM: {
            GemmParams g;
            g.A = H(p[0]); g.A2 = H(p[1]); g.W = H(p[2]); g.C = HM(p[3]); g.bias = H(p[4]); g.rowbias = H(p[5]);
            g.residual = H(p[6]);
            g.lda = l[0]; g.lda2 = l[1]; g.ldw = l[2]; g.ldc = l[3]; g.ldr = l[4];
            g.sA = l[5]; g.sA2 = l[6]; g.sW = l[7]; g.sC = l[8]; g.sR = l[9];
            g.M = i[0]; g.N = i[1]; g.K = i[2]; g.K1 = g.A2 ? i[3] : i[2];
            g.epi = i[4];
            g.rowbias_ld = i[6]; g.rows_per_group = i[7] > 0 ? i[7] : 1;
            g.ln_stats = reinterpret_cast<const float*>(p[7]); g.ln_c = H(p[8]); g.stats_out = reinterpret_cast<float*>(p[9]);
            g.ln_swapped = i[8]; g.ln_bstride = l[10];
            g.ln_partial = i[9]; g.ln_eps = i[9] ? o.f[0] : g.ln_eps; g.ln_rows = l[11];
            g.ln_nstrips = i[10]; g.stats_strip = i[11];
            return ds_launch_gemm(g, i[5] > 0 ? i[5] : 1, st);
        }
        case DS_OP_LN_FINALIZE:
            return ds_launch_ln_finalize(reinterpret_cast<const float*>(p[0]), reinterpret_cast<float*>(p[1]), i[0], i[1], i[2],
                                         o.f[0], st);
        case DS_OP_CONV3X3:
            return conv3x3_impl(p[0], p[1], p[3], p[4], i[7], p[5], p[2], i[0], i[1], i[2], i[3], i[4], i[5], i[6], st,
                                DS_DTYPE_F16, i[8], i[9], reinterpret_cast<float*>(p[6]));
        case DS_OP_GROUPNORM: {
            GroupNormParams g;
            g.x1 = H(p[0]); g.x2 = H(p[1]); g.y = HM(p[2]); g.gamma = H(p[3]); g.beta = H(p[4]);
            g.ws = reinterpret_cast<float*>(p[5]);
            g.B = i[0]; g.HW = i[1]; g.C1 = i[2]; g.C2 = p[1] ? i[3] : 0; g.groups = i[4]; g.silu = i[5]; g.eps = o.f[0];
            g.pre_chunks = i[6];
            return ds_launch_groupnorm(g, st);
        }
        case DS_OP_LAYERNORM:
            return ds_launch_layernorm(H(p[0]), HM(p[1]), H(p[2]), H(p[3]), i[0], i[1], o.f[0], st);
        case DS_OP_SELF_ATTN: {
            SelfAttnParams a;
            a.q = H(p[0]); a.k = H(p[1]); a.vt = H(p[2]); a.o = HM(p[3]);
            a.ldq = l[0]; a.ldk = l[1]; a.ldv = l[2]; a.ldo = l[3]; a.sq = l[4]; a.sk = l[5]; a.so = l[6];
            a.B = i[0]; a.heads = i[1]; a.Nq = i[2]; a.Nk = i[3]; a.scale = o.f[0];
            return ds_launch_self_attn(a, st);
        }
        case DS_OP_IP_ATTN: {
            IPAttnParams a;
            a.q = H(p[0]); a.kt = H(p[1]); a.vtt = H(p[2]); a.ki = H(p[3]); a.vti = H(p[4]);
            a.bbox = reinterpret_cast<const float*>(p[5]); a.o = HM(p[6]);
            a.ip_scale_ptr = reinterpret_cast<const float*>(p[7]);
            a.ldq = l[0]; a.ldo = l[1]; a.ldk = l[2]; a.sk = l[3]; a.sv = l[4];
            a.B = i[0]; a.heads = i[1]; a.N = i[2]; a.C = i[1] * 64; a.Lt = i[3]; a.Li = i[4]; a.LP = 96;
            a.n_dummy = i[5]; a.tok_per_ip = i[6]; a.max_ips = i[7]; a.mask_h = i[8]; a.mask_w = i[9];
            a.qk_scale = o.f[0]; a.ip_scale = o.f[1];
            return ds_launch_ip_attn(a, st);
        }
        case DS_OP_CONV_IN:
            return ds_launch_conv_in(H(p[0]), H(p[1]), H(p[2]), reinterpret_cast<const int*>(p[3]), H(p[4]), HM(p[5]),
                                     i[0], i[1], i[2], i[3], i[4], i[5], st);
        case DS_OP_CONV_OUT:
            return ds_launch_conv_out(H(p[0]), H(p[1]), H(p[2]), HM(p[3]), i[0], i[1], i[2], i[3], i[4], st);
        case DS_OP_SKINNY:
            return ds_launch_skinny_linear(H(p[0]), H(p[1]), H(p[2]), H(p[3]), HM(p[4]), i[0], i[1], i[2], i[3], i[4], st);
        case DS_OP_TIMESTEP_EMBED:
            return ds_launch_timestep_embed(reinterpret_cast<const float*>(p[0]), reinterpret_cast<const int*>(p[1]),
                                            HM(p[2]), i[0], i[1], i[2], o.f[0], st);
        case DS_OP_ADD_TIME_IDS:
            return ds_launch_add_time_ids(H(p[0]), H(p[1]), HM(p[2]), i[0], i[1], i[2], i[3], i[4], o.f[0], st);
        case DS_OP_SAMPLER_STEP: {
            SamplerStepParams s;
            s.eps = H(p[0]); s.latents = HM(p[1]); s.model_in = HM(p[2]); s.coef = reinterpret_cast<const float*>(p[3]);
            s.ns = i[0]; s.HW = i[1]; s.kind = i[2]; s.do_cfg = i[3];
            return ds_launch_sampler_step(s, reinterpret_cast<const int*>(p[4]), st);
        }
        case DS_OP_PREP_INPUT:
            return ds_launch_prepare_model_input(H(p[0]), HM(p[1]), reinterpret_cast<const float*>(p[2]),
                                                 reinterpret_cast<const int*>(p[3]), i[0], i[1], 4, i[2], st);
        case DS_OP_ADVANCE:
            return ds_launch_advance_counter(reinterpret_cast<int*>(p[0]), st);
        case DS_OP_NHWC2NCHW:
            return ds_launch_nhwc_to_nchw(H(p[0]), HM(p[1]), i[0], i[1], i[2], st);
        case DS_OP_NCHW2NHWC:
            return ds_launch_nchw_to_nhwc(H(p[0]), HM(p[1]), i[0], i[1], i[2], st);
        case DS_OP_PAD_ROWS:
            return ds_launch_pad_rows(H(p[0]), HM(p[1]), i[0], i[1], i[2], i[3], i[4], i[5], st);
        case DS_OP_SMALL_ATTN:
            return ds_launch_small_attn(H(p[0]), H(p[1]), H(p[2]), HM(p[3]), l[0], l[1], l[2], l[3], l[4], l[5], l[6],
                                        l[7], i[0], i[1], i[2], i[3], i[4], o.f[0], st);
        case DS_OP_LLM_GEMV:
            return llm_gemv_impl(p[0], l[0], p[1], p[2], l[1], p[3], l[2], i[0], i[1], i[2], i[3], i[4], o.f[0], st, p[4]);
        case DS_OP_LLM_ATTN:
            return llm_attn_impl(p[0], l[0], p[1], p[2], l[1], reinterpret_cast<const float*>(p[3]),
                                 reinterpret_cast<const float*>(p[4]), p[5], l[2], reinterpret_cast<const int32_t*>(p[6]),
                                 i[0], i[1], i[2], i[3], i[4], o.f[0], st);
        case DS_OP_LLM_RMSNORM:
            return ds_launch_llm_rmsnorm(H(p[0]), l[0], H(p[1]), HM(p[2]), l[1], HM(p[3]),
                                         reinterpret_cast<const int*>(p[4]), i[0], i[1], i[2], o.f[0], st);
        case DS_OP_LLM_EMBED:
            return ds_launch_llm_embed(H(p[0]), reinterpret_cast<const int*>(p[1]), HM(p[2]), i[0], i[1], st);
        case DS_OP_LLM_SELECT:
            return ds_launch_llm_select(H(p[0]), i[0], reinterpret_cast<const int*>(p[1]), i[1], i[2], i[3],
                                        reinterpret_cast<int*>(p[2]), reinterpret_cast<int*>(p[3]), st);
        case DS_OP_LLM_ADVANCE:
            return ds_launch_llm_advance(reinterpret_cast<int*>(p[0]), i[0], st);
        default:
            ds_set_error("plan: unknown opcode %d", o.code);
            return -4;
    }
}

// Static description of one op for roofline accounting: the gfx950 kernel it dispatches to (same spelling as the
// rocprofv3 kernel trace), its ALGORITHMIC flops (2*MAC) and its algorithmic HBM bytes (each operand once).
int ds_op_describe(const ds_op* op, char* name, int name_len, double* flops, double* bytes) {
    DS_REQUIRE(op != nullptr, "ds_op_describe: null op");
    const int32_t* i = op->i;
    const char* nm = "other";
    double fl = 0, by = 0;
    switch (op->code) {
        case DS_OP_GEMM: {
            GemmParams g;
            g.M = i[0]; g.N = i[1]; g.K = i[2];
            g.ln_stats = reinterpret_cast<const float*>(op->p[7]); g.stats_out = reinterpret_cast<float*>(op->p[9]);
            g.ln_c = H(op->p[8]); g.residual = H(op->p[6]);
            g.ln_swapped = i[8]; g.epi = i[4]; g.ln_partial = i[9]; g.stats_strip = i[11];
            g.A2 = H(op->p[1]); g.rowbias = H(op->p[5]);
            g.lda = g.ldw = g.K1 = g.K; g.ldc = g.N;
            const int batch = i[5] > 0 ? i[5] : 1;
            nm = ds_gemm_kernel_name(g, batch);
            fl = 2.0 * i[0] * (double)i[1] * i[2] * batch;
            by = 2.0 * batch * ((double)i[0] * i[2] + (double)i[1] * i[2] + (double)i[0] * (i[4] == 1 || i[4] == 4 ? i[1] / 2 : i[1]));
            break;
        }
        case DS_OP_CONV3X3: {
            GemmParams g;
            const int Ho = i[6] ? (i[8] ? i[8] : 2 * i[1]) : (i[5] == 2 ? (i[1] + 1) / 2 : i[1]);
            const int Wo = i[6] ? (i[9] ? i[9] : 2 * i[2]) : (i[5] == 2 ? (i[2] + 1) / 2 : i[2]);
            g.M = i[0] * Ho * Wo; g.N = i[4]; g.K = 9 * i[3];
            g.conv = 1;
            g.Hin = i[1]; g.Win = i[2]; g.Cin = i[3]; g.Hout = Ho; g.Wout = Wo; g.cstride = i[5]; g.upsample = i[6];
            nm = ds_gemm_kernel_name(g, 1);
            fl = 2.0 * g.M * (double)g.N * g.K;
            by = 2.0 * ((double)i[0] * i[1] * i[2] * i[3] + (double)g.N * g.K + (double)g.M * g.N);
            break;
        }
        case DS_OP_GROUPNORM: nm = "groupnorm(3 kernels)"; by = 2.0 * 2.0 * i[0] * (double)i[1] * (i[2] + i[3]); break;
        case DS_OP_LAYERNORM: nm = "layernorm_kernel"; by = 2.0 * 2.0 * i[0] * (double)i[1]; break;
        case DS_OP_LN_FINALIZE: nm = "ln_finalize_kernel"; by = 8.0 * i[0] * (double)(i[1] + 1); break;
        case DS_OP_SELF_ATTN:
            nm = ds_self_attn_kernel_name(i[0], i[1], i[2], i[3]);
            fl = 4.0 * i[0] * (double)i[1] * i[2] * (double)i[3] * 64;
            by = 2.0 * i[0] * (double)i[1] * 64 * (2.0 * i[2] + 2.0 * i[3]);
            break;
        case DS_OP_IP_ATTN:
            nm = "ip_attn_kernel";
            fl = 4.0 * i[0] * (double)i[1] * i[2] * (double)(i[3] + i[4]) * 64;
            by = 2.0 * i[0] * (double)i[1] * 64 * (2.0 * i[2] + 2.0 * (i[3] + i[4]));
            break;
        case DS_OP_SMALL_ATTN: nm = "small_attn_kernel"; fl = 4.0 * i[0] * (double)i[1] * i[2] * (double)i[3] * i[4]; break;
        case DS_OP_CONV_IN: nm = "conv_in_kernel"; fl = 2.0 * i[0] * (double)i[1] * i[2] * 9 * i[3] * i[4]; break;
        case DS_OP_CONV_OUT: nm = "conv_out_kernel"; fl = 2.0 * i[0] * (double)i[1] * i[2] * 9 * i[3] * i[4]; break;
        case DS_OP_SKINNY: nm = "skinny_linear_kernel"; fl = 2.0 * i[0] * (double)i[1] * i[2]; by = 2.0 * i[1] * (double)i[2]; break;
        case DS_OP_SAMPLER_STEP: nm = "sampler_step_kernel"; break;
        case DS_OP_LLM_GEMV:
            nm = "llm_gemv_kernel";
            fl = 2.0 * i[0] * (double)i[1] * i[2] * (i[4] ? 2 : 1);
            by = 2.0 * ((double)i[1] * i[2] * (i[4] ? 2 : 1) + (double)i[0] * i[2] + (double)i[0] * i[1]);
            break;
        case DS_OP_LLM_ATTN: nm = "llm_attn_kernel"; break;
        case DS_OP_LLM_RMSNORM: nm = "llm_rmsnorm_kernel"; by = 4.0 * i[0] * (double)i[1]; break;
        case DS_OP_LLM_SELECT: nm = "llm_select_kernel"; by = 2.0 * i[0]; break;
        default: break;
    }
    if (name && name_len > 0) {
        strncpy(name, nm, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (flops) *flops = fl;
    if (bytes) *bytes = by;
    return 0;
}

struct ds_plan {
    std::vector<ds_op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

int ds_op_run(const ds_op* op, void* stream) {
    DS_REQUIRE(op != nullptr, "ds_op_run: null op");
    return run_op(*op, S(stream));
}

int ds_plan_create(const ds_op* ops, int n_ops, ds_plan** out) {
    DS_REQUIRE(ops && n_ops > 0 && out, "ds_plan_create: bad arguments");
    ds_plan* pl = new ds_plan();
    pl->ops.assign(ops, ops + n_ops);
    *out = pl;
    return 0;
}

int ds_plan_num_ops(const ds_plan* plan) { return plan ? (int)plan->ops.size() : -1; }

int ds_plan_run(ds_plan* plan, void* stream) {
    DS_REQUIRE(plan, "ds_plan_run: null plan");
    for (size_t k = 0; k < plan->ops.size(); ++k) {
        const int rc = run_op(plan->ops[k], S(stream));
        if (rc != 0) {
            char msg[600];
            snprintf(msg, sizeof(msg), "plan op %zu (code %d): %s", k, plan->ops[k].code, g_err);
            ds_set_error("%s", msg);
            return rc;
        }
    }
    return 0;
}

int ds_plan_capture(ds_plan* plan, void* stream) {
    DS_REQUIRE(plan, "ds_plan_capture: null plan");
    if (plan->exec) return 0;
    hipStream_t st = S(stream);
    DS_REQUIRE(st != nullptr, "ds_plan_capture: needs a non-default stream");
    DS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    const int rc = ds_plan_run(plan, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc != 0) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) {
        ds_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
        return -2;
    }
    plan->graph = g;
    DS_HIP(hipGraphInstantiate(&plan->exec, g, nullptr, nullptr, 0));
    return 0;
}

int ds_plan_replay(ds_plan* plan, void* stream) {
    DS_REQUIRE(plan && plan->exec, "ds_plan_replay: plan not captured");
    DS_HIP(hipGraphLaunch(plan->exec, S(stream)));
    return 0;
}

int ds_plan_destroy(ds_plan* plan) {
    if (!plan) return 0;
    if (plan->exec) (void)hipGraphExecDestroy(plan->exec);
    if (plan->graph) (void)hipGraphDestroy(plan->graph);
    delete plan;
    return 0;
}

}  // extern "C"
