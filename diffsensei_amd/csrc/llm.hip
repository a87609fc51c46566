// MLLM pre-pass: LLaMA greedy decoding with a KV cache for gfx950.
// Reference: src/models/mllm/seed_x.py:90-171 (generate), src/models/mllm/modeling_llama_xformer.py:97-314 (rotary
// embedding, attention with past_key_value, MLP, decoder layer), src/models/mllm/generation.py:19-30 (logits processor).
//
// Batch-1 greedy decoding streams every weight once per token: the path is HBM-bound (13 B parameters = 26 GB per
// token), so the kernels below are wide vector loads + fp32 dot products on the VALU, not MFMA tiles.  The same kernels
// take up to 16 rows (causal inside the chunk), which is how the attention of the prompt runs; the prompt's projections
// go through the MFMA GEMMs (mllm.py: _prompt_mfma; 16-row passes of these kernels remain as an A/B path).  The token
// loop is a static launch list whose only varying inputs live in a small device-side state block, so it replays as a
// hipGraph:
//     state[0] = tokens in the KV cache   state[1] = tokens generated   state[2] = finished   state[3] = current token
//     state[4] = max_new_tokens of this call   state[5] = eos token id                      (int32[8], two spare)
//
// HBM layout: weights [N,K] row-major fp16 (one wavefront streams one row); the RMSNorm gains are folded into the
// following projection on the host (W' = W diag(g)), so a projection only needs the row's 1/rms, which every wavefront
// recomputes from the x it reads anyway; KV cache [T_max, kv_heads*D] fp16 per layer, keys stored rotated.
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

__device__ __forceinline__ float dot8(const h8 a, const h8 b, float acc) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        h2 x = {a[e], a[e + 1]}, y = {b[e], b[e + 1]};
        acc = __builtin_amdgcn_fdot2(x, y, acc, false);
    }
    return acc;
}

// LlamaRMSNorm output for 8 elements, with the reference's rounding points (modeling_llama_xformer.py:77-82):
// hidden = x(fp32) * r -> .to(fp16) -> weight(fp16) * hidden -> fp16
__device__ __forceinline__ h8 rms_gain8(const h8& x, const h8& g, float r) {
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)g[e] * (float)(half_t)((float)x[e] * r));
    return o;
}

// ---------------------------------------------------------------------------------------------------------------
// y[m][n] = r_m * sum_k x[m][k] w[n][k]  (+ residual[m][n]),  r_m = rsqrt(mean_k x[m][k]^2 + eps) if rms else 1
// (with p.gain: y[m][n] = sum_k f16(gain[k] * f16(x[m][k] r_m)) w[n][k], the reference's RMSNorm roundings)
// SWIGLU:  y[m][n] = silu(r_m * x.w[n]) * (r_m * x.w[n+N])     (gate rows [0,N), up rows [N,2N))
// One wavefront per output column; MC rows of x per pass (x comes from L1/L2, it is MC*K*2 bytes).
// ---------------------------------------------------------------------------------------------------------------
template <int MC, int SWIGLU>
__global__ __launch_bounds__(256) void llm_gemv_kernel(LlmGemvParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    const int m0 = blockIdx.y * MC;
    if (n >= p.N) return;
    const int K = p.K;
    float r[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) r[m] = 1.0f;
    if (p.rms) {
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            float ss = 0.f;
            if (m0 + m < p.M) {
                const half_t* xr = p.x + (long)(m0 + m) * p.ldx;
                for (int k = lane * 8; k < K; k += 512) {
                    const h8 v = *reinterpret_cast<const h8*>(xr + k);
                    ss = dot8(v, v, ss);
                }
            }
            r[m] = __builtin_amdgcn_rsqf(wave_sum(ss) / (float)K + p.eps);
        }
    }
    float acc[MC], acu[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) acc[m] = acu[m] = 0.f;
    const half_t* wg = p.w + (long)n * K;
    const half_t* wu = p.w + ((long)n + p.N) * K;
    constexpr int U = (MC <= 4) ? 4 : 2;  // independent 16-byte loads in flight per lane
    for (int k0 = lane * 8; k0 < K; k0 += 512 * U) {
        h8 wv[U], uv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(k0 + 512 * u, K - 8);  // clamped re-read past the end, masked below
            wv[u] = *reinterpret_cast<const h8*>(wg + k);
            if (SWIGLU) uv[u] = *reinterpret_cast<const h8*>(wu + k);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 512 * u;
            if (k < K) {
#pragma unroll
                for (int m = 0; m < MC; ++m) {
                    if (m0 + m < p.M) {
                        h8 xv = *reinterpret_cast<const h8*>(p.x + (long)(m0 + m) * p.ldx + k);
                        if (p.gain) xv = rms_gain8(xv, *reinterpret_cast<const h8*>(p.gain + k), r[m]);
                        acc[m] = dot8(xv, wv[u], acc[m]);
                        if (SWIGLU) acu[m] = dot8(xv, uv[u], acu[m]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const float rr = p.gain ? 1.0f : r[m];  // with a gain the scale is already inside x'
        acc[m] = wave_sum(acc[m]) * rr;
        if (SWIGLU) acu[m] = wave_sum(acu[m]) * rr;
    }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            if (m0 + m < p.M) {
                float o;
                if (SWIGLU) {  // fp16 roundings of the reference: act_fn(gate_proj(x)) * up_proj(x)
                    const float g = (float)(half_t)acc[m], u = (float)(half_t)acu[m];
                    o = (float)(half_t)ds_silu(g) * u;
                } else {
                    o = (float)(half_t)acc[m];
                    if (p.residual) o += (float)p.residual[(long)(m0 + m) * p.ldr + n];
                }
                p.y[(long)(m0 + m) * p.ldy + n] = (half_t)o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention of chunk row r (absolute position pos0 + r) over keys 0 .. pos0 + r: cache rows for j < pos0, the chunk's
// own (still un-rotated) k/v rows of the fused qkv buffer for j >= pos0.  The block of (head h, row r) also appends
// its rotated key and its value to the cache, so one launch per layer does rope + append + attention.
// Lanes are laid out [64/LPK keys][LPK chunks of 8 dims]: a wavefront reads whole 2*D-byte key rows.
// ---------------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void rope_chunk(const half_t* row, int c, const float* cs, const float* sn, float out[8]) {
    constexpr int LPK = D / 8, HALF = D / 2;
    const h8 a = *reinterpret_cast<const h8*>(row + 8 * c);
    const h8 b = *reinterpret_cast<const h8*>(row + 8 * (c ^ (LPK / 2)));
    const int d0 = (8 * c) % HALF;
    const float sgn = (8 * c < HALF) ? -1.0f : 1.0f;  // rotate_half = cat(-x2, x1)
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = (float)a[e] * cs[d0 + e] + sgn * (float)b[e] * sn[d0 + e];
}

constexpr int ATT_WAVES = 16;  // one key group per wavefront per iteration: the loop is a chain of dependent loads

template <int D>
__global__ __launch_bounds__(64 * ATT_WAVES) void llm_attn_kernel(LlmAttnParams p) {
    constexpr int LPK = D / 8, KPW = 64 / LPK, HALF = D / 2;
    extern __shared__ float sm[];
    float* sc = sm;                    // [T_max] scores, then probabilities
    float* red = sm + p.T_max;         // [ATT_WAVES][D]
    float* misc = red + ATT_WAVES * D;  // [2 * ATT_WAVES]
    const int h = blockIdx.x, r = blockIdx.y;
    const int hkv = h / (p.heads / p.kv_heads);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane % LPK, ks = lane / LPK;
    const int pos0 = p.state[0];
    const int T = pos0 + r + 1;
    if (T > p.T_max) return;  // the host sizes the cache; never write past it
    const long kcol = (long)(p.heads + hkv) * D, vcol = (long)(p.heads + p.kv_heads + hkv) * D;
    const half_t* qrow = p.qkv + (long)r * p.ldqkv;

    float qf[8];
    rope_chunk<D>(qrow + (long)h * D, c, p.rope_cos + (long)(pos0 + r) * HALF, p.rope_sin + (long)(pos0 + r) * HALF, qf);

    // ---- scores
    float mx = -3.0e38f;
    for (int g = wave; g * KPW < T; g += ATT_WAVES) {
        const int j = g * KPW + ks;
        float s = 0.f;
        if (j < T) {
            float kf[8];
            if (j < pos0) {
                const h8 kv = *reinterpret_cast<const h8*>(p.kc + (long)j * p.ldc + (long)hkv * D + 8 * c);
#pragma unroll
                for (int e = 0; e < 8; ++e) kf[e] = (float)kv[e];
            } else {
                rope_chunk<D>(p.qkv + (long)(j - pos0) * p.ldqkv + kcol, c, p.rope_cos + (long)j * HALF,
                              p.rope_sin + (long)j * HALF, kf);
#pragma unroll
                for (int e = 0; e < 8; ++e) kf[e] = (float)(half_t)kf[e];  // same rounding as the cached copy
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qf[e], kf[e], s);
        }
#pragma unroll
        for (int o = 1; o < LPK; o <<= 1) s += __shfl_xor(s, o, 64);
        s *= p.scale;
        if (j < T) {
            if (c == 0) sc[j] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = wave_max(mx);
    if (lane == 0) misc[wave] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) mx = fmaxf(mx, misc[w]);
    float sum = 0.f;
    for (int j = tid; j < T; j += 64 * ATT_WAVES) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    if (lane == 0) misc[ATT_WAVES + wave] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < ATT_WAVES; ++w) tot += misc[ATT_WAVES + w];
    const float inv = 1.0f / tot;

    // ---- P V
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int g = wave; g * KPW < T; g += ATT_WAVES) {
        const int j = g * KPW + ks;
        if (j < T) {
            const float pj = (float)(half_t)(sc[j] * inv);  // the reference casts the probabilities to fp16
            const half_t* vp = (j < pos0) ? p.vc + (long)j * p.ldc + (long)hkv * D + 8 * c
                                          : p.qkv + (long)(j - pos0) * p.ldqkv + vcol + 8 * c;
            const h8 vv = *reinterpret_cast<const h8*>(vp);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(pj, (float)vv[e], acc[e]);
        }
    }
#pragma unroll
    for (int o = LPK; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (ks == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * D + 8 * c + e] = acc[e];
    }
    __syncthreads();
    if (tid < D) {
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < ATT_WAVES; ++w) o += red[w * D + tid];
        p.out[(long)r * p.ldo + (long)h * D + tid] = (half_t)o;
    }
    // ---- append this row's rotated key and its value (one query head per kv head does it)
    if (h % (p.heads / p.kv_heads) == 0 && wave == 0 && ks == 0) {
        float kf[8];
        rope_chunk<D>(qrow + kcol, c, p.rope_cos + (long)(pos0 + r) * HALF, p.rope_sin + (long)(pos0 + r) * HALF, kf);
        h8 ko;
#pragma unroll
        for (int e = 0; e < 8; ++e) ko[e] = (half_t)kf[e];
        const long off = (long)(pos0 + r) * p.ldc + (long)hkv * D + 8 * c;
        *reinterpret_cast<h8*>(p.kc + off) = ko;
        *reinterpret_cast<h8*>(p.vc + off) = *reinterpret_cast<const h8*>(qrow + vcol + 8 * c);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// y[m] = g * (x[m] / rms(x[m]))  (LlamaRMSNorm: normalise in fp32, cast to fp16, then multiply by the gain).
// `feat` (token loop only, M = 1): the same row is also stored as row state[1]-1 of the per-token feature buffer —
// the hidden state seed_x.py:143 collects for every generated token that was fed back.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void llm_rmsnorm_kernel(const half_t* __restrict__ x, long ldx,
                                                          const half_t* __restrict__ gamma, half_t* __restrict__ y,
                                                          long ldy, half_t* __restrict__ feat,
                                                          const int* __restrict__ state, int H, int max_out,
                                                          float eps) {
    __shared__ float part[4];
    const int m = blockIdx.x, tid = threadIdx.x;
    const half_t* xr = x + (long)m * ldx;
    float ss = 0.f;
    for (int k = tid * 8; k < H; k += 2048) {
        const h8 v = *reinterpret_cast<const h8*>(xr + k);
        ss = dot8(v, v, ss);
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) part[tid >> 6] = ss;
    __syncthreads();
    const float r = __builtin_amdgcn_rsqf((part[0] + part[1] + part[2] + part[3]) / (float)H + eps);
    half_t* fr = nullptr;
    if (feat) {
        const int row = state[1] - 1 + m;
        if (row >= 0 && row < max_out && !state[2]) fr = feat + (long)row * H;
    }
    for (int k = tid * 8; k < H; k += 2048) {
        const h8 v = *reinterpret_cast<const h8*>(xr + k);
        const h8 g = *reinterpret_cast<const h8*>(gamma + k);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)g[e] * (float)(half_t)((float)v[e] * r));
        *reinterpret_cast<h8*>(y + (long)m * ldy + k) = o;
        if (fr) *reinterpret_cast<h8*>(fr + k) = o;
    }
}

// h[0,:] = embed_tokens[state[3]]
__global__ __launch_bounds__(256) void llm_embed_kernel(const half_t* __restrict__ table, const int* __restrict__ state,
                                                        half_t* __restrict__ out, int H, int vocab) {
    const int tok = min(max(state[3], 0), vocab - 1);
    const int k = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (k < H) *reinterpret_cast<h8*>(out + k) = *reinterpret_cast<const h8*>(table + (long)tok * H + k);
}

// ---------------------------------------------------------------------------------------------------------------
// Greedy choice with the reference's AutoImageTokenGenerationProcessor folded in (generation.py:19-30):
//   chain = [<img>, <img_00000> .. <img_{n-1}>, </img>];  previous token in chain[:-1] -> the next chain id wins
//   (the reference lifts its score to max + 10); otherwise the scores of chain[1:] are set to exactly 0.0 first.
// argmax ties go to the lowest id (torch.argmax).  Then: append the id, make it the current token, advance the cache
// length by `adv` rows, and raise `finished` on EOS (state[5]) or when state[4] ids are out.  One block.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void llm_select_kernel(const half_t* __restrict__ logits, int V,
                                                          const int* __restrict__ chain, int n_chain, int out_cap,
                                                          int adv, int* __restrict__ state,
                                                          int* __restrict__ out_ids) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    __shared__ int forced, lo, hi;
    const int tid = threadIdx.x;
    if (state[2]) return;  // finished: replays of the captured step are no-ops
    const int prev = state[3];
    if (tid == 0) {
        forced = -1;
        lo = 0x7fffffff;
        hi = -1;
    }
    __syncthreads();
    if (tid < n_chain - 1 && chain[tid] == prev) atomicMax(&forced, chain[tid + 1]);
    if (tid >= 1 && tid < n_chain) {  // id range of chain[1:], so the membership scan below runs for few ids only
        atomicMin(&lo, chain[tid]);
        atomicMax(&hi, chain[tid]);
    }
    __syncthreads();
    const int zlo = lo, zhi = hi;
    float best = -3.0e38f;
    int idx = 0x7fffffff;
    for (int v = tid; v < V; v += 1024) {
        float s = (float)logits[v];
        if (v >= zlo && v <= zhi)
            for (int q = 1; q < n_chain; ++q)
                if (chain[q] == v) s = 0.0f;
        if (s > best) {  // strided ascending scan: the first maximum of this thread is its lowest id
            best = s;
            idx = v;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ob > best || (ob == best && oi < idx)) {
            best = ob;
            idx = oi;
        }
    }
    if ((tid & 63) == 0) {
        bv[tid >> 6] = best;
        bi[tid >> 6] = idx;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) {
                best = bv[w];
                idx = bi[w];
            }
        const int next = forced >= 0 ? forced : idx;
        const int n_out = state[1], max_new = state[4], eos = state[5];
        if (n_out < out_cap) out_ids[n_out] = next;
        state[1] = n_out + 1;
        state[3] = next;
        state[0] += adv;
        if (next == eos || n_out + 1 >= max_new) state[2] = 1;
    }
}

// act[m][n] = silu(gu[m][n]) * gu[m][I+n]  (prompt pass: the gate|up projection comes out of the MFMA GEMM as [M,2I])
__global__ __launch_bounds__(256) void llm_swiglu_kernel(const half_t* __restrict__ gu, half_t* __restrict__ act,
                                                         int M, int I) {
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (idx >= (long)M * I) return;
    const long m = idx / I, n = idx - m * I;
    const h8 g = *reinterpret_cast<const h8*>(gu + m * 2 * I + n);
    const h8 u = *reinterpret_cast<const h8*>(gu + m * 2 * I + I + n);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)(half_t)ds_silu((float)g[e]) * (float)u[e]);
    *reinterpret_cast<h8*>(act + idx) = o;
}

__global__ void llm_advance_kernel(int* state, int rows) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[0] += rows;
}

// out = a * s + b * (1 - s): the MLLM / encoder blend of scripts/demo/gradio.py:108-109 (n % 8 == 0)
__global__ __launch_bounds__(256) void blend_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                    half_t* __restrict__ out, long n, float s) {
    const long k = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (k >= n) return;
    const h8 av = *reinterpret_cast<const h8*>(a + k), bv = *reinterpret_cast<const h8*>(b + k);
    h8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)(half_t)((float)av[e] * s) + (float)(half_t)((float)bv[e] * (1.0f - s)));
    *reinterpret_cast<h8*>(out + k) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// Token-loop variant (M <= 4): the kernel above keeps only ~4 KB of weights in flight per wavefront and spends a third
// of each wavefront's life on the x prologue - measured 3.9 TB/s on the 13 B model.  Here the block stages x (and the
// row scales) in LDS once, every wavefront walks several output columns, and each column iteration has 8 KB (plain) /
// 2 x 4 KB (SwiGLU) of independent 16-byte weight loads in flight before the first dot product.
// ---------------------------------------------------------------------------------------------------------------
template <int MC, int SWIGLU>
__global__ __launch_bounds__(256) void llm_gemv_stream_kernel(LlmGemvParams p) {
    extern __shared__ char smem_raw[];
    half_t* xs = reinterpret_cast<half_t*>(smem_raw);               // [MC][K]
    float* rs = reinterpret_cast<float*>(smem_raw + (size_t)MC * p.K * 2);  // [MC] row scales, then [MC][4] partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K;
    {
        float ss[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            ss[m] = 0.f;
            for (int k = tid * 8; k < K; k += 2048) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (m < p.M) v = *reinterpret_cast<const h8*>(p.x + (long)m * p.ldx + k);
                *reinterpret_cast<h8*>(xs + (long)m * K + k) = v;
                ss[m] = dot8(v, v, ss[m]);
            }
            ss[m] = wave_sum(ss[m]);
            if (lane == 0) rs[MC + m * 4 + wave] = ss[m];
        }
        __syncthreads();
        if (tid < MC) {
            const float s = rs[MC + tid * 4] + rs[MC + tid * 4 + 1] + rs[MC + tid * 4 + 2] + rs[MC + tid * 4 + 3];
            rs[tid] = p.rms ? __builtin_amdgcn_rsqf(s / (float)K + p.eps) : 1.0f;
        }
        __syncthreads();
        if (p.rms && p.gain) {  // the staged rows become the RMSNorm OUTPUT (reference roundings); no scale left for the dots
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                const float r = rs[m];
                for (int k = tid * 8; k < K; k += 2048) {
                    h8* xp = reinterpret_cast<h8*>(xs + (long)m * K + k);
                    *xp = rms_gain8(*xp, *reinterpret_cast<const h8*>(p.gain + k), r);
                }
            }
            __syncthreads();
            if (tid < MC) rs[tid] = 1.0f;
            __syncthreads();
        }
    }
    constexpr int U = SWIGLU ? 4 : 8;
    for (int n = blockIdx.x * 4 + wave; n < p.N; n += gridDim.x * 4) {
        float acc[MC], acu[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) acc[m] = acu[m] = 0.f;
        const half_t* wg = p.w + (long)n * K;
        const half_t* wu = p.w + ((long)n + p.N) * K;
        for (int k0 = lane * 8; k0 < K; k0 += 512 * U) {
            h8 wv[U], uv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = min(k0 + 512 * u, K - 8);
                wv[u] = *reinterpret_cast<const h8*>(wg + k);
                if (SWIGLU) uv[u] = *reinterpret_cast<const h8*>(wu + k);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 512 * u;
                if (k < K) {
#pragma unroll
                    for (int m = 0; m < MC; ++m) {
                        const h8 xv = *reinterpret_cast<const h8*>(xs + (long)m * K + k);
                        acc[m] = dot8(xv, wv[u], acc[m]);
                        if (SWIGLU) acu[m] = dot8(xv, uv[u], acu[m]);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            acc[m] = wave_sum(acc[m]) * rs[m];
            if (SWIGLU) acu[m] = wave_sum(acu[m]) * rs[m];
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                if (m < p.M) {
                    float o;
                    if (SWIGLU) {
                        const float g = (float)(half_t)acc[m], u = (float)(half_t)acu[m];
                        o = (float)(half_t)ds_silu(g) * u;
                    } else {
                        o = (float)(half_t)acc[m];
                        if (p.residual) o += (float)p.residual[(long)m * p.ldr + n];
                    }
                    p.y[(long)m * p.ldy + n] = (half_t)o;
                }
            }
        }
    }
}

// Same data path, software-pipelined: the weight loads of the NEXT k-block (or of the next column's first k-block) are
// issued before the dot products of the current one, and the first block is requested before x is staged, so a
// wavefront always has 8 KB of weights in flight - no load bubble at block start, between k-blocks or between columns.
template <int MC, int SWIGLU>
__global__ __launch_bounds__(256) void llm_gemv_pipe_kernel(LlmGemvParams p) {
    extern __shared__ char smem_raw[];
    half_t* xs = reinterpret_cast<half_t*>(smem_raw);
    float* rs = reinterpret_cast<float*>(smem_raw + (size_t)MC * p.K * 2);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = p.K, N = p.N;
    constexpr int U = SWIGLU ? 4 : 8;
    const int NIT = (K + 512 * U - 1) / (512 * U);
    const int stride = gridDim.x * 4;
    int n = blockIdx.x * 4 + wave;
    h8 wa[U], ua[U], wb[U], ub[U];
    auto issue = [&](int col, int it, h8(&wv)[U], h8(&uv)[U]) {
        const half_t* wg = p.w + (long)col * K;
        const half_t* wu = p.w + ((long)col + N) * K;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = min(it * 512 * U + 512 * u + lane * 8, K - 8);
            wv[u] = *reinterpret_cast<const h8*>(wg + k);
            if (SWIGLU) uv[u] = *reinterpret_cast<const h8*>(wu + k);
        }
    };
    if (n < N) issue(n, 0, wa, ua);
    {
        float ss[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            ss[m] = 0.f;
            for (int k = tid * 8; k < K; k += 2048) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (m < p.M) v = *reinterpret_cast<const h8*>(p.x + (long)m * p.ldx + k);
                *reinterpret_cast<h8*>(xs + (long)m * K + k) = v;
                ss[m] = dot8(v, v, ss[m]);
            }
            ss[m] = wave_sum(ss[m]);
            if (lane == 0) rs[MC + m * 4 + wave] = ss[m];
        }
        __syncthreads();
        if (tid < MC) {
            const float s = rs[MC + tid * 4] + rs[MC + tid * 4 + 1] + rs[MC + tid * 4 + 2] + rs[MC + tid * 4 + 3];
            rs[tid] = p.rms ? __builtin_amdgcn_rsqf(s / (float)K + p.eps) : 1.0f;
        }
        __syncthreads();
        if (p.rms && p.gain) {  // the staged rows become the RMSNorm OUTPUT (reference roundings); no scale left for the dots
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                const float r = rs[m];
                for (int k = tid * 8; k < K; k += 2048) {
                    h8* xp = reinterpret_cast<h8*>(xs + (long)m * K + k);
                    *xp = rms_gain8(*xp, *reinterpret_cast<const h8*>(p.gain + k), r);
                }
            }
            __syncthreads();
            if (tid < MC) rs[tid] = 1.0f;
            __syncthreads();
        }
    }
    while (n < N) {
        float acc[MC], acu[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) acc[m] = acu[m] = 0.f;
        for (int it = 0; it < NIT; ++it) {
            const bool last = it + 1 == NIT;
            const int ncol = last ? n + stride : n;
            if (ncol < N) issue(ncol, last ? 0 : it + 1, wb, ub);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = it * 512 * U + 512 * u + lane * 8;
                if (k < K) {
#pragma unroll
                    for (int m = 0; m < MC; ++m) {
                        const h8 xv = *reinterpret_cast<const h8*>(xs + (long)m * K + k);
                        acc[m] = dot8(xv, wa[u], acc[m]);
                        if (SWIGLU) acu[m] = dot8(xv, ua[u], acu[m]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                wa[u] = wb[u];
                if (SWIGLU) ua[u] = ub[u];
            }
        }
#pragma unroll
        for (int m = 0; m < MC; ++m) {
            acc[m] = wave_sum(acc[m]) * rs[m];
            if (SWIGLU) acu[m] = wave_sum(acu[m]) * rs[m];
        }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MC; ++m) {
                if (m < p.M) {
                    float o;
                    if (SWIGLU) {
                        const float g = (float)(half_t)acc[m], u = (float)(half_t)acu[m];
                        o = (float)(half_t)ds_silu(g) * u;
                    } else {
                        o = (float)(half_t)acc[m];
                        if (p.residual) o += (float)p.residual[(long)m * p.ldr + n];
                    }
                    p.y[(long)m * p.ldy + n] = (half_t)o;
                }
            }
        }
        n += stride;
    }
}

static thread_local int g_llm_gemv_variant = 0;  // 0 auto (pipelined), 1 one-column-per-wavefront kernel, 2 un-pipelined streaming kernel

template <int MC, int SWIGLU>
int launch_gemv_stream(const LlmGemvParams& p, hipStream_t stream) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        DS_HIP(hipGetDevice(&dev));
        DS_HIP(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount;
    }
    const size_t lds = (size_t)MC * p.K * 2 + (size_t)MC * 5 * sizeof(float);
    // resident blocks per CU: 8 x 4 wavefronts at <= 64 VGPRs (streaming kernel), 5 at ~90 VGPRs (pipelined kernel);
    // launching exactly the resident set makes every wavefront walk the same number of columns (no tail round)
    const size_t by_regs = g_llm_gemv_variant == 2 ? 8 : 5;
    const int per_cu = (int)min(by_regs, (size_t)(160 * 1024) / (lds + 512));
    // (N = 5120 = exactly one column per resident wavefront for the o / down projections.  Fewer, longer-lived wavefronts
    // with 2-4 columns each to pipeline were measured and are slower: 5.03 -> 5.13 / 5.22 / 5.56 ms per token,
    // profiles/r02_mllm_min_cols_ab.jsonl - resident wavefronts matter more than per-wavefront prefetch here.)
    const int blocks = min((p.N + 3) / 4, cus * max(per_cu, 1));
    if (g_llm_gemv_variant == 2) hipLaunchKernelGGL((llm_gemv_stream_kernel<MC, SWIGLU>), dim3(blocks), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((llm_gemv_pipe_kernel<MC, SWIGLU>), dim3(blocks), dim3(256), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

template <int SWIGLU>
int launch_gemv(const LlmGemvParams& p, hipStream_t stream) {
    const int mcs = p.M == 1 ? 1 : p.M == 2 ? 2 : 4;  // LDS copy of x must fit the default 64 KiB dynamic limit
    if (p.M <= 4 && g_llm_gemv_variant != 1 && (size_t)mcs * p.K * 2 + 128 <= 64 * 1024) {
        if (p.M == 1) return launch_gemv_stream<1, SWIGLU>(p, stream);
        if (p.M == 2) return launch_gemv_stream<2, SWIGLU>(p, stream);
        return launch_gemv_stream<4, SWIGLU>(p, stream);
    }
    const int mc = p.M <= 1 ? 1 : p.M <= 2 ? 2 : p.M <= 4 ? 4 : p.M <= 8 ? 8 : 16;
    const dim3 grid((p.N + 3) / 4, (p.M + mc - 1) / mc);
    switch (mc) {
        case 1: hipLaunchKernelGGL((llm_gemv_kernel<1, SWIGLU>), grid, dim3(256), 0, stream, p); break;
        case 2: hipLaunchKernelGGL((llm_gemv_kernel<2, SWIGLU>), grid, dim3(256), 0, stream, p); break;
        case 4: hipLaunchKernelGGL((llm_gemv_kernel<4, SWIGLU>), grid, dim3(256), 0, stream, p); break;
        case 8: hipLaunchKernelGGL((llm_gemv_kernel<8, SWIGLU>), grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((llm_gemv_kernel<16, SWIGLU>), grid, dim3(256), 0, stream, p); break;
    }
    DS_LAUNCH_CHECK();
    return 0;
}

}  // namespace

void ds_llm_gemv_set_variant(int v) { g_llm_gemv_variant = v; }

int ds_launch_llm_gemv(const LlmGemvParams& p, hipStream_t stream) {
    DS_REQUIRE(p.M > 0 && p.N > 0 && p.K >= 8 && p.K % 8 == 0, "llm_gemv: bad shape M=%d N=%d K=%d", p.M, p.N, p.K);
    DS_REQUIRE(p.ldx % 8 == 0 && p.ldx >= p.K, "llm_gemv: ldx %ld must be a multiple of 8 and >= K", p.ldx);
    DS_REQUIRE(p.x && p.w && p.y, "llm_gemv: null operand");
    DS_REQUIRE(!(p.swiglu && p.residual), "llm_gemv: the SwiGLU epilogue takes no residual");
    return p.swiglu ? launch_gemv<1>(p, stream) : launch_gemv<0>(p, stream);
}

int ds_launch_llm_attn(const LlmAttnParams& p, hipStream_t stream) {
    DS_REQUIRE(p.D == 64 || p.D == 128, "llm_attn: head_dim %d unsupported (64 or 128)", p.D);
    DS_REQUIRE(p.M > 0 && p.M <= 16, "llm_attn: %d rows per pass (1..16)", p.M);
    DS_REQUIRE(p.heads > 0 && p.kv_heads > 0 && p.heads % p.kv_heads == 0, "llm_attn: heads %d / kv_heads %d", p.heads,
               p.kv_heads);
    DS_REQUIRE(p.T_max > 0 && p.T_max <= 8192, "llm_attn: cache length %d (max 8192)", p.T_max);
    DS_REQUIRE(p.ldqkv % 8 == 0 && p.ldc % 8 == 0, "llm_attn: row strides must be multiples of 8");
    DS_REQUIRE(p.qkv && p.kc && p.vc && p.rope_cos && p.rope_sin && p.out && p.state, "llm_attn: null operand");
    const size_t lds = (size_t)(p.T_max + ATT_WAVES * p.D + 2 * ATT_WAVES) * sizeof(float);
    const dim3 grid(p.heads, p.M);
    if (p.D == 128) hipLaunchKernelGGL(llm_attn_kernel<128>, grid, dim3(64 * ATT_WAVES), lds, stream, p);
    else hipLaunchKernelGGL(llm_attn_kernel<64>, grid, dim3(64 * ATT_WAVES), lds, stream, p);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_llm_rmsnorm(const half_t* x, long ldx, const half_t* gamma, half_t* y, long ldy, half_t* feat,
                          const int* state, int M, int H, int max_out, float eps, hipStream_t stream) {
    DS_REQUIRE(M > 0 && H >= 8 && H % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "llm_rmsnorm: bad shape M=%d H=%d", M, H);
    DS_REQUIRE(!feat || (state && M == 1), "llm_rmsnorm: the feature tap is for the 1-row token loop");
    hipLaunchKernelGGL(llm_rmsnorm_kernel, dim3(M), dim3(256), 0, stream, x, ldx, gamma, y, ldy, feat, state, H, max_out,
                       eps);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_llm_embed(const half_t* table, const int* state, half_t* out, int H, int vocab, hipStream_t stream) {
    DS_REQUIRE(H % 8 == 0 && vocab > 0, "llm_embed: bad shape H=%d vocab=%d", H, vocab);
    hipLaunchKernelGGL(llm_embed_kernel, dim3((H / 8 + 255) / 256), dim3(256), 0, stream, table, state, out, H, vocab);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_llm_select(const half_t* logits, int V, const int* chain, int n_chain, int out_cap, int adv, int* state,
                         int* out_ids, hipStream_t stream) {
    DS_REQUIRE(V > 0 && n_chain >= 0 && n_chain <= 1024 && out_cap > 0 && adv >= 0, "llm_select: bad arguments");
    DS_REQUIRE(n_chain == 0 || chain, "llm_select: chain ids missing");
    hipLaunchKernelGGL(llm_select_kernel, dim3(1), dim3(1024), 0, stream, logits, V, chain, n_chain, out_cap, adv, state,
                       out_ids);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_llm_swiglu(const half_t* gu, half_t* act, int M, int I, hipStream_t stream) {
    DS_REQUIRE(M > 0 && I >= 8 && I % 8 == 0 && gu && act, "llm_swiglu: bad shape M=%d I=%d", M, I);
    const long n8 = (long)M * I / 8;
    hipLaunchKernelGGL(llm_swiglu_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, gu, act, M, I);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_blend(const half_t* a, const half_t* b, half_t* out, long n, float s, hipStream_t stream) {
    DS_REQUIRE(n > 0 && n % 8 == 0 && a && b && out, "blend: n (%ld) must be a positive multiple of 8", n);
    hipLaunchKernelGGL(blend_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, stream, a, b, out, n, s);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_llm_advance(int* state, int rows, hipStream_t stream) {
    hipLaunchKernelGGL(llm_advance_kernel, dim3(1), dim3(64), 0, stream, state, rows);
    DS_LAUNCH_CHECK();
    return 0;
}
