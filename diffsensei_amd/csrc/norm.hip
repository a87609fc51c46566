// GroupNorm(+SiLU) over NHWC fp16 activations and LayerNorm over token rows.  HBM-bound kernels:
// every access is a 16-byte vector per lane, reductions are wavefront shuffles + one LDS pass.
//
// Replaces the diffusers GroupNorm/SiLU of ResnetBlock2D / Transformer2DModel / conv_norm_out reached from
// reference src/models/unet.py:244-338 and the three LayerNorms of each BasicTransformerBlock [3P].
//
// GroupNorm is three launches: (1) per-(row-chunk, channel) partial sums, deterministic (no atomics);
// (2) per-(batch, group) mean/rstd folded with gamma/beta into a per-(batch, channel) scale/shift table;
// (3) y = silu(a*x + s).  The input may be the channel-concat of two tensors (UNet up-path skip joins,
// reference src/models/unet.py:304-332 -> torch.cat in diffusers up blocks) — the concat is never
// materialised un-normalised.
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr int GN_MAX_CHUNKS = 128;

__host__ __device__ inline int gn_chunks(int HW) {
    int n = (HW + 7) / 8;
    return n < GN_MAX_CHUNKS ? (n < 1 ? 1 : n) : GN_MAX_CHUNKS;
}

struct GnGeom {
    int W, R, cc, c, row_start, row_end;
    bool active;
};
// A block covers up to 256 16-byte channel chunks (W lanes per row) and as many rows at a time as its threads allow
// (R = blockDim / W).  512-thread blocks since round 4: with 256 threads the 1280-channel tensors (W = 160 -> R = 1) left 96
// of 256 lanes idle; 512 threads give R = 3 (480 of 512 lanes active), and 480 or 512 active lanes for every width of the UNet
// (320, 640, 960, 1280, 1920, 2560 channels).
__device__ __forceinline__ GnGeom gn_geom(int C, int HW) {
    GnGeom g;
    const int ncc = C >> 3;
    const int slab0 = blockIdx.x * 256;
    g.W = min(ncc - slab0, 256);
    g.R = (int)blockDim.x / g.W;
    const int t = threadIdx.x;
    const int cw = t % g.W, r = t / g.W;
    g.active = r < g.R;
    g.cc = slab0 + cw;
    g.c = g.cc * 8;
    const int nch = gridDim.y;
    const int rpc = (HW + nch - 1) / nch;
    g.row_start = blockIdx.y * rpc + r;
    g.row_end = min(HW, (int)(blockIdx.y + 1) * rpc);
    return g;
}

// (Eight loads in flight per lane instead of four - gn_variant 2 of round 5 - measured +1...2 % alone and noise inside the forward,
// profiles/r05_gn_8_in_flight_ab.txt; removed in round 6.)
template <typename T>
__global__ __launch_bounds__(512) void gn_stats_kernel(GroupNormParams p) {
    __shared__ float red[512 * 16];
    const int C = p.C1 + p.C2;
    const GnGeom g = gn_geom(C, p.HW);
    const int b = blockIdx.z;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = ss[e] = 0.f;
    if (g.active) {
        const bool first = g.c < p.C1;
        const T* src = reinterpret_cast<const T*>(first ? p.x1 + (long)b * p.HW * p.C1 + g.c : p.x2 + (long)b * p.HW * p.C2 + (g.c - p.C1));
        const int ld = first ? p.C1 : p.C2;
        int row = g.row_start;
        const long step = (long)g.R * ld;
        for (; row + 3 * g.R < g.row_end; row += 4 * g.R) {  // 4 independent 16-byte loads in flight per lane
            const T* q = src + (long)row * ld;
            const typename Elt<T>::v8 v0 = *reinterpret_cast<const typename Elt<T>::v8*>(q);
            const typename Elt<T>::v8 v1 = *reinterpret_cast<const typename Elt<T>::v8*>(q + step);
            const typename Elt<T>::v8 v2 = *reinterpret_cast<const typename Elt<T>::v8*>(q + 2 * step);
            const typename Elt<T>::v8 v3 = *reinterpret_cast<const typename Elt<T>::v8*>(q + 3 * step);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f0 = (float)v0[e], f1 = (float)v1[e], f2 = (float)v2[e], f3 = (float)v3[e];
                s[e] += (f0 + f1) + (f2 + f3);
                ss[e] += fmaf(f0, f0, f1 * f1) + fmaf(f2, f2, f3 * f3);
            }
        }
        for (; row < g.row_end; row += g.R) {
            const typename Elt<T>::v8 v = *reinterpret_cast<const typename Elt<T>::v8*>(src + (long)row * ld);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                s[e] += f;
                ss[e] = fmaf(f, f, ss[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        red[threadIdx.x * 16 + e] = s[e];
        red[threadIdx.x * 16 + 8 + e] = ss[e];
    }
    __syncthreads();
    if (threadIdx.x < g.W) {
        float a[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = 0.f;
        for (int r = 0; r < g.R; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) a[e] += red[(r * g.W + threadIdx.x) * 16 + e];
        // partial[b][chunk][c][2]
        float* dst = p.ws + (((long)b * gridDim.y + blockIdx.y) * C + g.c) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dst[2 * e] = a[e];
            dst[2 * e + 1] = a[8 + e];
        }
    }
}

// one block per (batch item, group): 256 lanes walk the group's nchunks x cpg partial pairs, four independent 8-byte loads
// in flight per lane (one wave per group with one load in flight took 35 us per launch - 80 dependent round trips - whatever
// the batch: 5 % of a batch-2 forward)
template <typename T>
__global__ __launch_bounds__(256) void gn_finalize_kernel(GroupNormParams p, int nchunks) {
    __shared__ float red[8];
    const int C = p.C1 + p.C2;
    const int b = blockIdx.x, g = blockIdx.y;
    const int cpg = C / p.groups;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* part = p.ws + (long)b * nchunks * C * 2 + (long)g * cpg * 2;
    float* tabA = p.ws + (long)gridDim.x * nchunks * C * 2 + (long)b * C * 2;
    float* tabS = tabA + C;
    const float inv_n = 1.0f / ((float)cpg * (float)p.HW);
    const int items = nchunks * cpg;
    auto at = [&](int i) {
        const int ch = i / cpg, cc = i - ch * cpg;
        return *reinterpret_cast<const f32x2*>(part + ((long)ch * C + cc) * 2);
    };
    float s = 0.f, ss = 0.f;
    int i = tid;
    for (; i + 768 < items; i += 1024) {
        const f32x2 q0 = at(i), q1 = at(i + 256), q2 = at(i + 512), q3 = at(i + 768);
        s += (q0[0] + q1[0]) + (q2[0] + q3[0]);
        ss += (q0[1] + q1[1]) + (q2[1] + q3[1]);
    }
    for (; i < items; i += 256) {
        const f32x2 q = at(i);
        s += q[0];
        ss += q[1];
    }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lane == 0) {
        red[wave] = s;
        red[4 + wave] = ss;
    }
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    ss = (red[4] + red[5]) + (red[6] + red[7]);
    const float mean = s * inv_n;
    const float var = fmaxf(ss * inv_n - mean * mean, 0.f);
    const float rstd = rsqrtf(var + p.eps);
    for (int cc = tid; cc < cpg; cc += 256) {
        const int c = g * cpg + cc;
        const float a = rstd * (float)reinterpret_cast<const T*>(p.gamma)[c];
        tabA[c] = a;
        tabS[c] = (float)reinterpret_cast<const T*>(p.beta)[c] - mean * a;
    }
}

template <typename T>
__global__ __launch_bounds__(512) void gn_apply_kernel(GroupNormParams p, int nchunks_stats) {
    const int C = p.C1 + p.C2;
    const GnGeom g = gn_geom(C, p.HW);
    const int b = blockIdx.z;
    if (!g.active) return;
    const float* tabA = p.ws + (long)gridDim.z * nchunks_stats * C * 2 + (long)b * C * 2;
    const float* tabS = tabA + C;
    float a[8], s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = tabA[g.c + e];
        s[e] = tabS[g.c + e];
    }
    const bool first = g.c < p.C1;
    const T* src = reinterpret_cast<const T*>(first ? p.x1 + (long)b * p.HW * p.C1 + g.c : p.x2 + (long)b * p.HW * p.C2 + (g.c - p.C1));
    const int ld = first ? p.C1 : p.C2;
    T* dst = reinterpret_cast<T*>(p.y) + (long)b * p.HW * C + g.c;
    auto norm8 = [&](const typename Elt<T>::v8& v) {
        typename Elt<T>::v8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = fmaf(a[e], (float)v[e], s[e]);
            if (p.silu) f = ds_silu(f);
            o[e] = (T)(f * p.out_scale);
        }
        return o;
    };
    int row = g.row_start;
    const long step = (long)g.R * ld, dstep = (long)g.R * C;
    for (; row + 3 * g.R < g.row_end; row += 4 * g.R) {  // 4 independent 16-byte loads in flight per lane
        const T* q = src + (long)row * ld;
        const typename Elt<T>::v8 v0 = *reinterpret_cast<const typename Elt<T>::v8*>(q);
        const typename Elt<T>::v8 v1 = *reinterpret_cast<const typename Elt<T>::v8*>(q + step);
        const typename Elt<T>::v8 v2 = *reinterpret_cast<const typename Elt<T>::v8*>(q + 2 * step);
        const typename Elt<T>::v8 v3 = *reinterpret_cast<const typename Elt<T>::v8*>(q + 3 * step);
        T* d = dst + (long)row * C;
        *reinterpret_cast<typename Elt<T>::v8*>(d) = norm8(v0);
        *reinterpret_cast<typename Elt<T>::v8*>(d + dstep) = norm8(v1);
        *reinterpret_cast<typename Elt<T>::v8*>(d + 2 * dstep) = norm8(v2);
        *reinterpret_cast<typename Elt<T>::v8*>(d + 3 * dstep) = norm8(v3);
    }
    for (; row < g.row_end; row += g.R) {
        const typename Elt<T>::v8 v = *reinterpret_cast<const typename Elt<T>::v8*>(src + (long)row * ld);
        *reinterpret_cast<typename Elt<T>::v8*>(dst + (long)row * C) = norm8(v);
    }
}

// Fused LayerNorm (gemm_pp.hip): the producing GEMM's epilogue left one (sum, sum of squares) pair per row and 64-column
// strip, [strips][M] float2; this turns them into (mean, rstd) per row for the consuming GEMM's epilogue.  Sums are fp32
// over f16 values: var = E[x^2] - mean^2 keeps ~2e-6 (1 + mean^2 / var) relative accuracy - far inside f16's 5e-4 for any
// row whose mean is not tens of standard deviations (the stand-alone kernel below is two-pass).
__global__ __launch_bounds__(256) void ln_finalize_kernel(const f32x2* __restrict__ part, f32x2* __restrict__ stats, int M,
                                                          int strips, float inv_c, float eps) {
    // four lanes per row (lane & 3 = which strips: q, q + 4, ...), all of a lane's loads independent and in flight together;
    // the four partial sums meet in a 2-step DPP butterfly - a fixed order, so the result does not depend on anything but
    // the data (one thread per row walking 20 strips four at a time took 8 us per launch, 1.5 ms per forward)
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = t >> 2, q = t & 3;
    float s = 0.f, ss = 0.f;
    if (row < M) {
#pragma unroll 8
        for (int j = q; j < strips; j += 4) {
            const f32x2 v = part[(long)j * M + row];
            s += v[0];
            ss += v[1];
        }
    }
    s += __shfl_xor(s, 1, 64);
    ss += __shfl_xor(ss, 1, 64);
    s += __shfl_xor(s, 2, 64);
    ss += __shfl_xor(ss, 2, 64);
    if (row < M && q == 0) {
        const float mean = s * inv_c;
        const float var = fmaxf(fmaf(-mean, mean, ss * inv_c), 0.f);
        f32x2 o = {mean, rsqrtf(var + eps)};
        stats[row] = o;
    }
}

// LayerNorm: one wavefront per row, the row lives in registers (two-pass variance, like torch).
template <int MAXCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, half_t* __restrict__ y,
                                                        const half_t* __restrict__ gamma,
                                                        const half_t* __restrict__ beta, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int ncc = C >> 3;
    const half_t* xr = x + (long)row * C;
    h8 v[MAXCH];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < ncc) {
            v[i] = *reinterpret_cast<const h8*>(xr + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)v[i][e];
        }
    }
    s = wave_sum(s);
    const float mean = s / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < ncc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = (float)v[i][e] - mean;
                ss = fmaf(d, d, ss);
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)C + eps);
    half_t* yr = y + (long)row * C;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
        const int cc = lane + 64 * i;
        if (cc < ncc) {
            const h8 gv = *reinterpret_cast<const h8*>(gamma + cc * 8);
            const h8 bv = *reinterpret_cast<const h8*>(beta + cc * 8);
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)v[i][e] - mean) * rstd * (float)gv[e] + (float)bv[e]);
            *reinterpret_cast<h8*>(yr + cc * 8) = o;
        }
    }
}

}  // namespace

static thread_local int g_gn_variant = 0;  // 0: 512-thread blocks, >= 64 rows per block; 1: the round-3 geometry (256 threads, 8-row chunks)
void ds_groupnorm_set_variant(int v) { g_gn_variant = v; }

size_t ds_groupnorm_ws_floats(int B, int C) { return (size_t)B * GN_MAX_CHUNKS * C * 2 + (size_t)B * C * 2; }

int ds_launch_groupnorm(const GroupNormParams& p, hipStream_t stream) {
    const int C = p.C1 + p.C2;
    DS_REQUIRE(p.B > 0 && p.HW > 0 && C > 0, "groupnorm: empty input");
    DS_REQUIRE(C % p.groups == 0, "groupnorm: C (%d) not divisible by groups (%d)", C, p.groups);
    DS_REQUIRE(p.C1 % 8 == 0 && p.C2 % 8 == 0, "groupnorm: channel counts must be multiples of 8");
    DS_REQUIRE(p.ws != nullptr, "groupnorm: workspace missing");
    const int nslab = ((C >> 3) + 255) / 256;
    // Row chunks per image: enough blocks for the chip (>= 1024) first, then >= 64 rows per block.  (The old rule - 8 rows per
    // block up to 128 chunks - gave the 32 x 32-token level at UNet batch 64 blocks of 20 KiB that wrote 10 KiB of partial
    // sums each: 84 MB of partials beside a 168 MB tensor.)  g_gn_variant 1 = the round-3 geometry (A/B).
    int nch = gn_chunks(p.HW), threads = 256;
    // Statistics already in the workspace (written by the producing convolution's epilogue, conv_halo.hip: one partial pair per
    // channel and pixel tile): no statistics launch - the finalize launch adds up `pre_chunks` partials per channel, and the
    // normalisation pass is the only one that reads x.
    const int stat_chunks = p.pre_chunks > 0 ? p.pre_chunks : 0;
    DS_REQUIRE(stat_chunks <= GN_MAX_CHUNKS && (stat_chunks == 0 || (p.x2 == nullptr && p.dtype == DS_DTYPE_F16)),
               "groupnorm: precomputed statistics need a single f16 source and <= %d chunks per image (got %d)", GN_MAX_CHUNKS, stat_chunks);
    if (g_gn_variant != 1) {
        const int want = (1024 + p.B * nslab - 1) / (p.B * nslab), by_rows = (p.HW + 63) / 64;
        nch = min(nch, max(want, by_rows));
        threads = 512;
    }
    dim3 grid(nslab, nch, p.B);
    if (p.dtype == DS_DTYPE_BF16) {
        hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(threads), 0, stream, p);
        hipLaunchKernelGGL(gn_finalize_kernel<bf16_t>, dim3(p.B, p.groups), dim3(256), 0, stream, p, nch);
        hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(threads), 0, stream, p, nch);
    } else if (stat_chunks > 0) {
        hipLaunchKernelGGL(gn_finalize_kernel<half_t>, dim3(p.B, p.groups), dim3(256), 0, stream, p, stat_chunks);
        hipLaunchKernelGGL(gn_apply_kernel<half_t>, grid, dim3(threads), 0, stream, p, stat_chunks);
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<half_t>, grid, dim3(threads), 0, stream, p);
        hipLaunchKernelGGL(gn_finalize_kernel<half_t>, dim3(p.B, p.groups), dim3(256), 0, stream, p, nch);
        hipLaunchKernelGGL(gn_apply_kernel<half_t>, grid, dim3(threads), 0, stream, p, nch);
    }
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_layernorm(const half_t* x, half_t* y, const half_t* gamma, const half_t* beta, int rows, int C,
                        float eps, hipStream_t stream) {
    DS_REQUIRE(rows > 0 && C > 0 && C % 8 == 0, "layernorm: bad shape rows=%d C=%d", rows, C);
    DS_REQUIRE(C <= 16 * 64 * 8, "layernorm: C (%d) > 8192 unsupported", C);
    dim3 grid((rows + 3) / 4);
    const int ncc = C >> 3;
    if (ncc <= 64 * 2)
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, C, eps);
    else if (ncc <= 64 * 4)
        hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, C, eps);
    else if (ncc <= 64 * 8)
        hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, C, eps);
    else  // the MLLM's input QwenResampler normalises LLaMA-width rows (5120)
        hipLaunchKernelGGL(layernorm_kernel<16>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, C, eps);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_ln_finalize(const float* partial, float* stats, int M, int strips, int C, float eps, hipStream_t stream) {
    DS_REQUIRE(M > 0 && strips > 0 && C > 0, "ln_finalize: empty problem M=%d strips=%d C=%d", M, strips, C);
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((M * 4 + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const f32x2*>(partial),
                       reinterpret_cast<f32x2*>(stats), M, strips, 1.0f / (float)C, eps);
    DS_LAUNCH_CHECK();
    return 0;
}
