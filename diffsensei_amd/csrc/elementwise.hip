// Small, bandwidth/latency-bound kernels of the sampling loop: conv_in (+ dialog-box embedding), conv_out,
// skinny (M <= 16) linears for the time-embedding MLPs, sinusoidal embeddings, the fused
// classifier-free-guidance + scheduler step, and layout helpers.
//
// reference call sites:
//   conv_in + encode_dialog_bbox      src/models/unet.py:206-210, :88-114
//   time / added-cond embeddings      src/models/unet.py:190-199  (diffusers get_time_embed / get_aug_embed [3P])
//   conv_norm_out -> conv_out         src/models/unet.py:335-338
//   CFG + scheduler.step + next scale_model_input   src/pipelines/pipeline_diffsensei.py:315-317, :333-337
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

// Per-step scalar table: row i (8 floats) = {timestep, c_in_div, k0, k1, k2, k3, c_in_div_next, guidance}.
// `ctr` (device int, may be null = row 0) selects the row, so one captured hipGraph serves every step.
__device__ __forceinline__ const float* coef_row(const float* table, const int* ctr) {
    return table + (ctr ? (long)(*ctr) * 8 : 0);
}

// ---------------------------------------------------------------- conv_in: 3x3, Cin = 4, NHWC, + dialog boxes
__global__ __launch_bounds__(256) void conv_in_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                      const half_t* __restrict__ bias, const int* __restrict__ boxes,
                                                      const half_t* __restrict__ demb, half_t* __restrict__ y, int B,
                                                      int H, int W, int Cout, int ndialog) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half_t* sw = reinterpret_cast<half_t*>(smem_raw);  // [9 taps][2 channel pairs][Cout] of (w[ci = 2 pair], w[ci = 2 pair + 1])
    for (int i = threadIdx.x; i < 36 * Cout; i += 256) {
        const int j = i & 1, co = (i >> 1) % Cout, tp = (i >> 1) / Cout;   // tp = tap * 2 + pair
        sw[i] = w[co * 36 + tp * 2 + j];
    }
    __syncthreads();
    const int ncc = Cout >> 3;
    const long total = (long)B * H * W * ncc;
    // grid-stride: the 36 x Cout weight panel (a transposing gather) is staged once per resident block, not once per 256
    // outputs (at batch 32 that was 82 000 stagings of 23 KB: 1.1 ms for a kernel whose output takes 70 us to write)
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
    const int cc = (int)(id % ncc);
    const long pix = id / ncc;
    const int b = (int)(pix / (H * W));
    const int rem = (int)(pix - (long)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const h4 xv = *reinterpret_cast<const h4*>(x + ((long)(b * H + iy) * W + ix) * 4);
            // v_dot2_f32_f16: two input channels per instruction, f16 products summed in f32 - no converts (the f32 fma form
            // spent 2/3 of its VALU instructions on them: 1.0 ms per batch-64 launch for 0.7 GB of output)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const h2 xp = {xv[2 * pr], xv[2 * pr + 1]};
                const half_t* wp = sw + (((ky * 3 + kx) * 2 + pr) * Cout + cc * 8) * 2;
                const h8 w0 = *reinterpret_cast<const h8*>(wp), w1 = *reinterpret_cast<const h8*>(wp + 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] = __builtin_amdgcn_fdot2(xp, (h2){w0[2 * e], w0[2 * e + 1]}, acc[e], false);
                    acc[4 + e] = __builtin_amdgcn_fdot2(xp, (h2){w1[2 * e], w1[2 * e + 1]}, acc[4 + e], false);
                }
            }
        }
    bool inside = false;
    for (int j = 0; j < ndialog; ++j) {
        const int* bx = boxes + ((long)b * ndialog + j) * 4;
        inside |= (ox >= bx[0]) & (ox < bx[2]) & (oy >= bx[1]) & (oy < bx[3]);
    }
    const h8 bv = *reinterpret_cast<const h8*>(bias + cc * 8);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[e] + (float)bv[e]);
    if (inside) {
        const h8 dv = *reinterpret_cast<const h8*>(demb + cc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)dv[e]);
    }
    *reinterpret_cast<h8*>(y + pix * Cout + cc * 8) = o;
    }
}

// ---------------------------------------------------------------- conv_out: 3x3, Cout = 4, 8 lanes per pixel
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                       const half_t* __restrict__ bias, half_t* __restrict__ y, int B,
                                                       int H, int W, int Cin) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    half_t* sw = reinterpret_cast<half_t*>(smem_raw);  // [4][9*Cin]
    const int K = 9 * Cin;
    for (int i = threadIdx.x; i < 4 * K / 8; i += 256)
        reinterpret_cast<h8*>(sw)[i] = reinterpret_cast<const h8*>(w)[i];
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const long npix = (long)B * H * W;
    // grid-stride over groups of 32 pixels: the weight panel is staged once per resident block
    for (long pix0 = (long)blockIdx.x * 32; pix0 < npix; pix0 += (long)gridDim.x * 32) {
    const long pix = pix0 + (threadIdx.x >> 3);
    const bool live = pix < npix;
    const long pc = live ? pix : 0;
    const int b = (int)(pc / (H * W));
    const int rem = (int)(pc - (long)b * H * W);
    const int oy = rem / W, ox = rem - oy * W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nch = Cin >> 6;  // 8-lane groups of 16-byte chunks: Cin/64 chunks per lane per tap
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        const half_t* xp = x + ((long)(b * H + iy) * W + ix) * Cin;
        for (int i = 0; i < nch; ++i) {
            const int c = (i * 8 + sub) * 8;
            const h8 xv = *reinterpret_cast<const h8*>(xp + c);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                const h8 wv = *reinterpret_cast<const h8*>(sw + co * K + tap * Cin + c);
#pragma unroll
                for (int e = 0; e < 8; e += 2)   // v_dot2_f32_f16: two channels per instruction, no converts
                    acc[co] = __builtin_amdgcn_fdot2((h2){xv[e], xv[e + 1]}, (h2){wv[e], wv[e + 1]}, acc[co], false);
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        acc[co] += __shfl_xor(acc[co], 1, 64);
        acc[co] += __shfl_xor(acc[co], 2, 64);
        acc[co] += __shfl_xor(acc[co], 4, 64);
    }
    if (live && sub == 0) {
        h4 o;
#pragma unroll
        for (int co = 0; co < 4; ++co) o[co] = (half_t)(acc[co] + (float)bias[co]);
        *reinterpret_cast<h4*>(y + pix * 4) = o;
    }
    }
}

// ---------------------------------------------------------------- skinny linear: one wavefront per output column
// y[m][n] = act_out( sum_k act_in(x[m][k]) * w[n][k] + bias[n] + addend[m][n] ),  M <= 16 per pass.
template <int MC>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w,
                                                            const half_t* __restrict__ bias,
                                                            const half_t* __restrict__ addend, half_t* __restrict__ y,
                                                            int M, int N, int K, int act_in, int act_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int m_base = blockIdx.y * MC;
    float acc[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = 0.f;
    const half_t* wr = w + (long)n * K;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const h8 wv = *reinterpret_cast<const h8*>(wr + k);
        float wf[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) wf[e] = (float)wv[e];
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            if (m_base + m < M) {
                const h8 xv = *reinterpret_cast<const h8*>(x + (long)(m_base + m) * K + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float xf = (float)xv[e];
                    if (act_in) xf = (float)(half_t)ds_silu(xf);  // the reference rounds SiLU(emb) to f16 first
                    acc[m] = fmaf(xf, wf[e], acc[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = wave_sum(acc[m]);
    if (lane == 0) {
        const float bf = bias ? (float)bias[n] : 0.f;
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            if (m_base + m < M) {
                half_t o = (half_t)(acc[m] + bf);
                if (addend) o = (half_t)((float)o + (float)addend[(long)(m_base + m) * N + n]);
                if (act_out) o = (half_t)ds_silu((float)o);
                y[(long)(m_base + m) * N + n] = o;
            }
        }
    }
}

// ---------------------------------------------------------------- sinusoidal embeddings (diffusers Timesteps [3P])
__device__ __forceinline__ float sinusoid(float t, int j, int dim, int flip, float freq_shift) {
    const int half = dim >> 1;
    const bool first = j < half;
    const int i = first ? j : j - half;
    const float freq = expf(-9.210340371976184f * (float)i / ((float)half - freq_shift));  // ln(10000)
    const float a = t * freq;
    const bool use_cos = flip ? first : !first;
    return use_cos ? cosf(a) : sinf(a);
}

__global__ void timestep_embed_kernel(const float* table, const int* ctr, half_t* out, int B, int dim, int flip,
                                      float freq_shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * dim) return;
    const float t = coef_row(table, ctr)[0];
    out[i] = (half_t)sinusoid(t, i % dim, dim, flip, freq_shift);
}

// out[b] = cat(text_embeds[b] (pooled_dim), sinusoid(time_ids[b][0..n_ids)) (n_ids*dim))
__global__ void add_time_ids_kernel(const half_t* text_embeds, const half_t* time_ids, half_t* out, int B,
                                    int pooled_dim, int n_ids, int dim, int flip, float freq_shift) {
    const int width = pooled_dim + n_ids * dim;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * width) return;
    const int b = i / width, j = i - b * width;
    if (j < pooled_dim) {
        out[i] = text_embeds[b * pooled_dim + j];
    } else {
        const int jj = j - pooled_dim, id = jj / dim;
        out[i] = (half_t)sinusoid((float)time_ids[b * n_ids + id], jj - id * dim, dim, flip, freq_shift);
    }
}

// ---------------------------------------------------------------- CFG + scheduler step + next model input
// One thread per (sample, pixel): 4 latent channels.  All arithmetic mirrors the reference's rounding points:
// CFG in fp16 (noise_pred tensors are fp16), Euler update in fp32, latents stored fp16 between steps.
__global__ __launch_bounds__(256) void sampler_step_kernel(SamplerStepParams p, const int* ctr) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)p.ns * p.HW;
    if (i >= total) return;
    const int n = (int)(i / p.HW), pix = (int)(i - (long)n * p.HW);
    const float* cf = coef_row(p.coef, ctr);
    const h4 eu = *reinterpret_cast<const h4*>(p.eps + ((long)n * p.HW + pix) * 4);
    h4 e = eu;
    if (p.do_cfg) {
        const h4 ec = *reinterpret_cast<const h4*>(p.eps + ((long)(p.ns + n) * p.HW + pix) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const half_t d = (half_t)((float)ec[c] - (float)eu[c]);
            const half_t gd = (half_t)(cf[7] * (float)d);
            e[c] = (half_t)((float)eu[c] + (float)gd);
        }
    }
    h4 xin;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        half_t* lp = p.latents + ((long)n * 4 + c) * p.HW + pix;
        const float x = (float)*lp;
        const float ef = (float)e[c];
        float xn;
        if (p.kind == 0) {  // Euler: k0 = sigma, k1 = sigma_next
            const float s = cf[2], sn = cf[3];
            const float pred_x0 = x - s * ef;
            const float deriv = (x - pred_x0) / s;
            xn = x + deriv * (sn - s);
        } else {            // DDIM eta=0: k0 = sqrt(a_t), k1 = sqrt(1-a_t), k2 = sqrt(a_prev), k3 = sqrt(1-a_prev)
            const float pred_x0 = (x - cf[3] * ef) / cf[2];
            xn = cf[4] * pred_x0 + cf[5] * ef;
        }
        const half_t xh = (half_t)xn;
        *lp = xh;
        xin[c] = (half_t)((float)xh / cf[6]);
    }
    *reinterpret_cast<h4*>(p.model_in + ((long)n * p.HW + pix) * 4) = xin;
    if (p.do_cfg) *reinterpret_cast<h4*>(p.model_in + ((long)(p.ns + n) * p.HW + pix) * 4) = xin;
}

__global__ __launch_bounds__(256) void prepare_model_input_kernel(const half_t* latents, half_t* model_in,
                                                                  const float* table, const int* ctr, int ns, int HW,
                                                                  int do_cfg) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)ns * HW) return;
    const int n = (int)(i / HW), pix = (int)(i - (long)n * HW);
    const float div = coef_row(table, ctr)[1];
    h4 xin;
#pragma unroll
    for (int c = 0; c < 4; ++c) xin[c] = (half_t)((float)latents[((long)n * 4 + c) * HW + pix] / div);
    *reinterpret_cast<h4*>(model_in + ((long)n * HW + pix) * 4) = xin;
    if (do_cfg) *reinterpret_cast<h4*>(model_in + ((long)(ns + n) * HW + pix) * 4) = xin;
}

__global__ void advance_counter_kernel(int* ctr) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *ctr += 1;
}

// ---------------------------------------------------------------- layout helpers
__global__ void nhwc_to_nchw_kernel(const half_t* x, half_t* y, int B, int HW, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW * C) return;
    const int c = (int)(i % C);
    const long r = i / C;
    const int pix = (int)(r % HW), b = (int)(r / HW);
    y[((long)b * C + c) * HW + pix] = x[i];
}
__global__ void nchw_to_nhwc_kernel(const half_t* x, half_t* y, int B, int HW, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * HW * C) return;
    const int c = (int)(i % C);
    const long r = i / C;
    const int pix = (int)(r % HW), b = (int)(r / HW);
    y[i] = x[((long)b * C + c) * HW + pix];
}
// y[b][r][:] = r < rows_in ? x[b][row_off + r][:] : 0   (x is [B, total_rows, C])
__global__ void pad_rows_kernel(const half_t* x, half_t* y, int B, int rows_in, int rows_out, int row_off,
                                int total_rows, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = C >> 3;
    if (i >= (long)B * rows_out * c8) return;
    const int cc = (int)(i % c8);
    const long rr = i / c8;
    const int r = (int)(rr % rows_out), b = (int)(rr / rows_out);
    h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < rows_in) v = *reinterpret_cast<const h8*>(x + ((long)b * total_rows + row_off + r) * C + cc * 8);
    *reinterpret_cast<h8*>(y + ((long)b * rows_out + r) * C + cc * 8) = v;
}

// out[b,t,:] = tok_emb[ids[b,t],:] + pos_emb[t,:]   (CLIP text embeddings; ids clamped to the vocabulary)
__global__ void embed_tokens_kernel(const int* ids, const half_t* tok_emb, const half_t* pos_emb, half_t* out, int B,
                                    int T, int D, int vocab) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int d8 = D >> 3;
    if (i >= (long)B * T * d8) return;
    const int c = (int)(i % d8);
    const long bt = i / d8;
    const int t = (int)(bt % T);
    int id = ids[bt];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const h8 a = *reinterpret_cast<const h8*>(tok_emb + (long)id * D + c * 8);
    const h8 b = *reinterpret_cast<const h8*>(pos_emb + (long)t * D + c * 8);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + (float)b[e]);
    *reinterpret_cast<h8*>(out + bt * D + c * 8) = o;
}

// VaeImageProcessor.postprocess(output_type="pil") tail (reference src/pipelines/pipeline_diffsensei.py:367 ->
// pt_to_numpy + numpy_to_pil [3P]): [B,3,H,W] fp32 in [0,1] -> [B,H,W,3] uint8 = (x * 255).round() (numpy rounds half
// to even = v_rndne_f32).  One thread per 4 pixels: 3 x 16-byte plane reads, one 12-byte interleaved store.
__global__ __launch_bounds__(256) void image_to_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ out, int B,
                                                          long plane) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;  // group of 4 pixels
    const long groups = plane / 4;
    const int b = blockIdx.y;
    if (q >= groups) return;
    const float* ip = img + (long)b * 3 * plane + q * 4;
    const float4 r = *reinterpret_cast<const float4*>(ip);
    const float4 g = *reinterpret_cast<const float4*>(ip + plane);
    const float4 bl = *reinterpret_cast<const float4*>(ip + 2 * plane);
    auto cv = [](float v) -> unsigned {
        return (unsigned)__builtin_rintf(fminf(fmaxf(v * 255.0f, 0.f), 255.f));
    };
    const unsigned px[12] = {cv(r.x), cv(g.x), cv(bl.x), cv(r.y), cv(g.y), cv(bl.y),
                             cv(r.z), cv(g.z), cv(bl.z), cv(r.w), cv(g.w), cv(bl.w)};
    uint3 w;
    w.x = px[0] | (px[1] << 8) | (px[2] << 16) | (px[3] << 24);
    w.y = px[4] | (px[5] << 8) | (px[6] << 16) | (px[7] << 24);
    w.z = px[8] | (px[9] << 8) | (px[10] << 16) | (px[11] << 24);
    *reinterpret_cast<uint3*>(out + ((long)b * plane + q * 4) * 3) = w;
}

}  // namespace

int ds_launch_embed_tokens(const int* ids, const half_t* tok_emb, const half_t* pos_emb, half_t* out, int B, int T,
                           int D, int vocab, hipStream_t stream) {
    DS_REQUIRE(D % 8 == 0 && B > 0 && T > 0 && vocab > 0, "embed_tokens: bad shape");
    const long total = (long)B * T * (D / 8);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, ids, tok_emb,
                       pos_emb, out, B, T, D, vocab);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_conv_in(const half_t* x, const half_t* w, const half_t* bias, const int* dialog_boxes,
                      const half_t* dialog_emb, half_t* y, int B, int H, int W, int Cin, int Cout, int ndialog,
                      hipStream_t stream) {
    DS_REQUIRE(Cin == 4, "conv_in: Cin must be 4 (got %d)", Cin);
    DS_REQUIRE(Cout % 8 == 0, "conv_in: Cout must be a multiple of 8");
    DS_REQUIRE(ndialog == 0 || (dialog_boxes && dialog_emb), "conv_in: dialog boxes without embedding");
    const long total = (long)B * H * W * (Cout / 8);
    const long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(conv_in_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), (size_t)36 * Cout * 2, stream,
                       x, w, bias, dialog_boxes, dialog_emb, y, B, H, W, Cout, ndialog);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_conv_out(const half_t* x, const half_t* w, const half_t* bias, half_t* y, int B, int H, int W, int Cin,
                       int Cout, hipStream_t stream) {
    DS_REQUIRE(Cout == 4, "conv_out: Cout must be 4 (got %d)", Cout);
    DS_REQUIRE(Cin % 64 == 0, "conv_out: Cin must be a multiple of 64");
    const size_t lds = (size_t)4 * 9 * Cin * 2;
    DS_REQUIRE(lds <= 64 * 1024, "conv_out: Cin %d too large for the LDS weight panel", Cin);
    const long npix = (long)B * H * W;
    const long blocks = (npix + 31) / 32;
    hipLaunchKernelGGL(conv_out_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), lds, stream, x, w, bias, y,
                       B, H, W, Cin);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_skinny_linear(const half_t* x, const half_t* w, const half_t* bias, const half_t* addend, half_t* y,
                            int M, int N, int K, int act_in, int act_out, hipStream_t stream) {
    DS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 8 == 0, "skinny_linear: bad shape M=%d N=%d K=%d", M, N, K);
    if (M <= 4) {
        hipLaunchKernelGGL(skinny_linear_kernel<4>, dim3((N + 3) / 4, (M + 3) / 4), dim3(256), 0, stream, x, w, bias,
                           addend, y, M, N, K, act_in, act_out);
    } else {
        hipLaunchKernelGGL(skinny_linear_kernel<16>, dim3((N + 3) / 4, (M + 15) / 16), dim3(256), 0, stream, x, w,
                           bias, addend, y, M, N, K, act_in, act_out);
    }
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_timestep_embed(const float* table, const int* ctr, half_t* out, int B, int dim, int flip,
                             float freq_shift, hipStream_t stream) {
    DS_REQUIRE(dim % 2 == 0, "timestep_embed: odd dim");
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, stream, table, ctr, out, B, dim,
                       flip, freq_shift);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_add_time_ids(const half_t* text_embeds, const half_t* time_ids, half_t* out, int B, int pooled_dim,
                           int n_ids, int dim, int flip, float freq_shift, hipStream_t stream) {
    const int width = pooled_dim + n_ids * dim;
    hipLaunchKernelGGL(add_time_ids_kernel, dim3((B * width + 255) / 256), dim3(256), 0, stream, text_embeds, time_ids,
                       out, B, pooled_dim, n_ids, dim, flip, freq_shift);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_sampler_step(const SamplerStepParams& p, const int* ctr, hipStream_t stream) {
    DS_REQUIRE(p.C == 4, "sampler_step: latent channels must be 4");
    DS_REQUIRE(p.ns > 0 && p.HW > 0 && p.coef, "sampler_step: bad arguments");
    const long total = (long)p.ns * p.HW;
    hipLaunchKernelGGL(sampler_step_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, ctr);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_prepare_model_input(const half_t* latents, half_t* model_in, const float* table, const int* ctr, int ns,
                                  int HW, int C, int do_cfg, hipStream_t stream) {
    DS_REQUIRE(C == 4, "prepare_model_input: latent channels must be 4");
    const long total = (long)ns * HW;
    hipLaunchKernelGGL(prepare_model_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, latents,
                       model_in, table, ctr, ns, HW, do_cfg);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_advance_counter(int* ctr, hipStream_t stream) {
    hipLaunchKernelGGL(advance_counter_kernel, dim3(1), dim3(64), 0, stream, ctr);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_nhwc_to_nchw(const half_t* x, half_t* y, int B, int HW, int C, hipStream_t stream) {
    const long total = (long)B * HW * C;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, B, HW, C);
    DS_LAUNCH_CHECK();
    return 0;
}
int ds_launch_nchw_to_nhwc(const half_t* x, half_t* y, int B, int HW, int C, hipStream_t stream) {
    const long total = (long)B * HW * C;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, B, HW, C);
    DS_LAUNCH_CHECK();
    return 0;
}
int ds_launch_pad_rows(const half_t* x, half_t* y, int B, int rows_in, int rows_out, int row_off, int total_rows,
                       int C, hipStream_t stream) {
    DS_REQUIRE(C % 8 == 0, "pad_rows: C must be a multiple of 8");
    const long total = (long)B * rows_out * (C / 8);
    hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, x, y, B, rows_in,
                       rows_out, row_off, total_rows, C);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_image_to_u8(const float* img, uint8_t* out, int B, int H, int W, hipStream_t stream) {
    const long plane = (long)H * W;
    DS_REQUIRE(B > 0 && plane > 0 && plane % 4 == 0, "image_to_u8: H*W (%ld) must be a positive multiple of 4", plane);
    hipLaunchKernelGGL(image_to_u8_kernel, dim3((unsigned)((plane / 4 + 255) / 256), B), dim3(256), 0, stream, img, out, B,
                       plane);
    DS_LAUNCH_CHECK();
    return 0;
}
