// Kernels that exist only for the VAE decoder (reference src/pipelines/pipeline_diffsensei.py:339-367 ->
// diffusers AutoencoderKL.decode [3P]).  The decoder runs in bf16 storage / fp32 accumulate:
// fp16 overflows inside it (the reference upcasts the whole VAE to fp32 for that reason, :340-344).  Its 3x3
// convolutions, 1x1 shortcuts / attention projections and GroupNorms reuse conv_halo.hip, gemm_pp.hip and norm.hip
// instantiated for bf16_t; this file adds what has no UNet counterpart:
//
//   wide_attn_kernel    the mid-block self-attention: ONE head of dim 512 over (H/8 * W/8) tokens (16384 at 1024^2).
//                       Flash attention with the softmax on the lane (S^T = K Q^T, as attention.hip), but 512 output
//                       columns would need 256 accumulator registers, so the output is cut into four 128-column
//                       slices (grid.y) and each slice recomputes the full-depth scores: 2.5x the MFMA work of a
//                       single pass for one layer that is < 0.3 % of a panel's time - in exchange no cross-wave
//                       reduction and no score matrix in HBM (it would be 1 GB per image in fp32).
//   vae_conv_in_kernel  post_quant_conv (1x1, 4 -> 4) + 1/scaling_factor + conv_in (3x3, 4 -> C) on the fp32 NCHW latents
//   vae_conv_out_kernel conv_out (3x3, C -> 3) of the normalised activations -> fp32 NCHW image
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr int DH = 512;        // head dim
constexpr int DS = 128;        // output columns per block (grid.y slices)
constexpr int KT = 32;         // keys per tile
constexpr int KROW = DH * 2;   // bytes per key row in LDS
constexpr int VSTR = KT * 2 + 8;  // bytes per V^T row in LDS: 64 + 8 pad -> conflict-free 8-byte reads

// 16-byte chunk c (0..63) of key row r sits at chunk c ^ (r & 15): the 16 lanes of a ds_read_b128 group read 16
// different rows at one c, i.e. 16 different slots of the 256-byte bank window
__device__ __forceinline__ int kswz(int row, int chunk) { return row * KROW + ((chunk ^ (row & 15)) << 4); }

template <typename T>
__global__ __launch_bounds__(256, 1) void wide_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                           const T* __restrict__ vt, T* __restrict__ o, int N,
                                                           int n_valid, float scale) {
    typedef typename Elt<T>::v8 V8;
    typedef typename Elt<T>::v4 V4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto sKb = [&](int buf) { return smem + buf * (KT * KROW); };                      // K tiles: 2 x 32 KiB
    auto sVb = [&](int buf) { return smem + 2 * KT * KROW + buf * (DS * VSTR); };      // V^T slices: 2 x 9 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.z, slice = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;

    // Q fragments (B operand of S^T = K Q^T): lane holds Q[q][kk*16 + lhi*8 .. +7], the whole depth stays in registers
    V8 qf[DH / 16];
    {
        const int qrow = min(q0 + l31, N - 1);
        const T* qp = q + ((long)b * N + qrow) * DH + lhi * 8;
#pragma unroll
        for (int kk = 0; kk < DH / 16; ++kk) qf[kk] = *reinterpret_cast<const V8*>(qp + kk * 16);
    }
    const T* kbase = k + (long)b * N * DH;
    const T* vbase = vt + ((long)b * DH + slice * DS) * N;

    // tile loads.  K tile (32 keys x 512 = 32 KiB): LDS-DMA, one instruction = one 1-KiB key row, lane = destination
    // slot, so the lane fetches the chunk the swizzle maps there (no registers: the 128 VGPRs of Q leave none to spare).
    // V^T slice (128 x 32 keys = 8 KiB, padded rows): 2 x 16 B per thread through registers.
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    V8 rv[2];
    auto load_tile = [&](int t, int buf) {
        const int key0 = t * KT;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = wave * 8 + j;
            const int key = min(key0 + row, N - 1);
            const T* src = kbase + (long)key * DH + ((lane ^ (row & 15)) << 3);
            __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(sKb(buf) + row * KROW), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = tid + 256 * j;        // 0 .. 511: (d row, 8-key chunk)
            const int row = id >> 2, c = id & 3;
            const int kc = min(key0 + c * 8, N - 8);
            rv[j] = *reinterpret_cast<const V8*>(vbase + (long)row * N + kc);
        }
    };
    auto store_tile = [&](int buf) {  // the V^T half of the tile (K went straight to LDS)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int id = tid + 256 * j;
            const int row = id >> 2, c = id & 3;
            V4 lo, hi;
#pragma unroll
            for (int e = 0; e < 4; ++e) lo[e] = rv[j][e], hi[e] = rv[j][4 + e];
            *reinterpret_cast<V4*>(sVb(buf) + row * VSTR + c * 16) = lo;
            *reinterpret_cast<V4*>(sVb(buf) + row * VSTR + c * 16 + 8) = hi;
        }
    };

    f32x16 ot[DS / 32];
#pragma unroll
    for (int d = 0; d < DS / 32; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[d][r] = 0.f;
    float m_run = NEG_BIG, l_part = 0.f;
    const float c = scale * LOG2E;
    const int nt = (n_valid + KT - 1) / KT;  // keys [n_valid, N) are padding rows of the caller's token matrices
    load_tile(0, 0);
    store_tile(0);
    __syncthreads();  // drains vmcnt: the K rows have landed
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) load_tile(t + 1, buf ^ 1);  // buf^1 was last read before the barrier that ended iteration t-1
        // ---- S^T (32 keys x 32 queries) over the full depth
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < DH / 16; ++kk) {
            const V8 kf = *reinterpret_cast<const V8*>(sKb(buf) + kswz(l31, kk * 2 + lhi));
            st = Elt<T>::mfma(kf, qf[kk], st);
        }
        if (t * KT + KT > n_valid) {  // ragged last tile: clamped / padding rows must not contribute
            int n_here = n_valid;
            asm volatile("" : "+s"(n_here)::"memory");
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = t * KT + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (key >= n_here) st[r] = NEG_BIG;
                asm volatile("" : "+v"(st[r]));
            }
        }
        // ---- online softmax: query row = lane & 31, the other 16 keys of the tile live on lane ^ 32
        float mloc = NEG_BIG;
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mloc);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        const bool moved = __builtin_amdgcn_ballot_w64(m_new != m_run) != 0;
        m_run = m_new;
        const float mc = m_new * c;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(fmaf(st[r], c, -mc));
            st[r] = e;
            psum += e;
        }
        l_part = fmaf(l_part, alpha, psum);
        if (moved) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int d = 0; d < DS / 32; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[d][r] *= alpha;
        }
        // ---- O^T (128 columns of this slice) += V^T P^T
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            V8 pf;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = (T)st[hb * 8 + e];
#pragma unroll
            for (int db = 0; db < DS / 32; ++db) {
                const int row = db * 32 + l31;
                const V4 v0 = *reinterpret_cast<const V4*>(sVb(buf) + row * VSTR + hb * 32 + 8 * lhi);
                const V4 v1 = *reinterpret_cast<const V4*>(sVb(buf) + row * VSTR + hb * 32 + 16 + 8 * lhi);
                V8 vf;
#pragma unroll
                for (int e = 0; e < 4; ++e) vf[e] = v0[e], vf[4 + e] = v1[e];
                ot[db] = Elt<T>::mfma(vf, pf, ot[db]);
            }
        }
        if (t + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
    }
    const float l = l_part + __shfl_xor(l_part, 32, 64);
    const float inv = 1.0f / l;
    const int qrow = q0 + l31;
    if (qrow < N) {
        T* op = o + ((long)b * N + qrow) * DH + slice * DS;
#pragma unroll
        for (int db = 0; db < DS / 32; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                V4 ov;
#pragma unroll
                for (int e = 0; e < 4; ++e) ov[e] = (T)(ot[db][4 * g + e] * inv);
                *reinterpret_cast<V4*>(op + db * 32 + 8 * g + 4 * lhi) = ov;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// z = post_quant_conv(latents / scaling_factor) (1x1, zero outside the image AFTER the affine map, as the padded
// conv_in sees it), y = conv_in(z): 3x3, 4 -> C.  latents fp32 NCHW [B,4,H,W]; y NHWC T.  One thread = one pixel x
// 8 output channels; weights (C x 36, fp32) live in LDS; a block walks 64-pixel groups.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vae_conv_in_kernel(const float* __restrict__ lat, const float* __restrict__ wpq,
                                                          const float* __restrict__ bpq, const T* __restrict__ w,
                                                          const T* __restrict__ bias, T* __restrict__ y, int B, int H,
                                                          int W, int C, float inv_sf, int px_per_block) {
    typedef typename Elt<T>::v8 V8;
    extern __shared__ float sw[];  // [C][36] conv_in weights (ky, kx, ci) + [C] bias
    for (int i = threadIdx.x; i < C * 36; i += 256) sw[i] = (float)w[i];
    for (int i = threadIdx.x; i < C; i += 256) sw[C * 36 + i] = (float)bias[i];
    float pq[16], pb[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) pq[i] = wpq[i] * inv_sf;
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = bpq[i];
    __syncthreads();
    const int groups = C / 8;                  // channel groups of 8
    const int ppb = 256 / groups;              // pixels handled per pass
    const int cg = threadIdx.x % groups, pl = threadIdx.x / groups;
    const long total = (long)B * H * W;
    const long first = (long)blockIdx.x * px_per_block;
    for (long base = first; base < first + px_per_block && base < total; base += ppb) {
        const long pix = base + pl;
        if (pl >= ppb || pix >= total) continue;
        const int b = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)b * H * W);
        const int oy = rem / W, ox = rem - oy * W;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = sw[C * 36 + cg * 8 + e];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy + ky - 1, ix = ox + kx - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                float l4[4], z[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) l4[ci] = lat[(((long)b * 4 + ci) * H + iy) * W + ix];
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    z[co] = pb[co] + pq[co * 4 + 0] * l4[0] + pq[co * 4 + 1] * l4[1] + pq[co * 4 + 2] * l4[2] + pq[co * 4 + 3] * l4[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float* wr = sw + (cg * 8 + e) * 36 + (ky * 3 + kx) * 4;
                    acc[e] += wr[0] * z[0] + wr[1] * z[1] + wr[2] * z[2] + wr[3] * z[3];
                }
            }
        V8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) ov[e] = (T)acc[e];
        *reinterpret_cast<V8*>(y + pix * C + cg * 8) = ov;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// conv_out: 3x3, C -> 3, NHWC T in, fp32 NCHW image out.  Eight lanes per output pixel: lane s takes the 16-byte channel chunks
// s, s + 8, ... of every tap, so a wave reads 8 adjacent pixels x 128 contiguous bytes per instruction, and the three sums are
// folded over the 8 lanes at the end.  (Round 1-2 had one THREAD per pixel: 2.3 KiB of 256-byte-strided reads and three
// dependent chains of 1152 FMAs each - 2.4 ms per 1024 x 1024 image for 0.27 GB of input, 9.6 ms per launch of four images in
// profiles/r03_rocprof_kernel_stats/.)  fp32 accumulation as before; the weight panel sits in LDS in its storage type.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vae_conv_out_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                           const T* __restrict__ bias, float* __restrict__ img, int B,
                                                           int H, int W, int C, int denorm) {
    typedef typename Elt<T>::v8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* sw = reinterpret_cast<T*>(smem_raw);  // [3][9][C]
    for (int i = threadIdx.x; i < 27 * C / 8; i += 256) reinterpret_cast<V8*>(sw)[i] = reinterpret_cast<const V8*>(w)[i];
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const long plane = (long)H * W, npix = (long)B * plane;
    const float bsel = (float)bias[sub < 3 ? sub : 0];
    // grid-stride over groups of 32 pixels (consecutive along x): the weight panel is staged once per resident block
    for (long pix0 = (long)blockIdx.x * 32; pix0 < npix; pix0 += (long)gridDim.x * 32) {
        const long pix = pix0 + (threadIdx.x >> 3);
        const bool live = pix < npix;
        const long pc = live ? pix : 0;
        const int b = (int)(pc / plane);
        const int rem = (int)(pc - (long)b * plane);
        const int oy = rem / W, ox = rem - oy * W;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int iy = oy + ky - 1, ix = ox + kx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const T* xp = x + (((long)b * H + iy) * W + ix) * C;
            for (int c = sub * 8; c < C; c += 64) {
                const V8 xv = *reinterpret_cast<const V8*>(xp + c);
#pragma unroll
                for (int co = 0; co < 3; ++co) {
                    const V8 wv = *reinterpret_cast<const V8*>(sw + (co * 9 + tap) * C + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[co] = fmaf((float)xv[e], (float)wv[e], acc[co]);
                }
            }
        }
#pragma unroll
        for (int co = 0; co < 3; ++co) {
            acc[co] += __shfl_xor(acc[co], 1, 64);
            acc[co] += __shfl_xor(acc[co], 2, 64);
            acc[co] += __shfl_xor(acc[co], 4, 64);
        }
        if (live && sub < 3) {  // lanes 0..2 of the pixel's group write the three colour planes
            float v = (sub == 0 ? acc[0] : sub == 1 ? acc[1] : acc[2]) + bsel;
            if (denorm) v = fminf(fmaxf(v * 0.5f + 0.5f, 0.f), 1.f);  // VaeImageProcessor.denormalize: (x / 2 + 0.5).clamp(0, 1)
            img[((long)b * 3 + sub) * plane + rem] = v;
        }
    }
}

}  // namespace

int ds_launch_wide_attn(const void* q, const void* k, const void* vt, void* o, int B, int N, int n_valid, int dtype,
                        float scale, hipStream_t stream) {
    DS_REQUIRE(B > 0 && N >= 8 && N % 8 == 0, "wide_attn: token count (%d) must be a positive multiple of 8", N);
    if (n_valid <= 0) n_valid = N;
    DS_REQUIRE(n_valid <= N, "wide_attn: n_valid (%d) exceeds the row count (%d)", n_valid, N);
    const size_t lds = 2 * KT * KROW + 2 * DS * VSTR;  // 64 KiB + 18 KiB
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_attn_kernel<bf16_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wide_attn_kernel<half_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid((N + 127) / 128, DH / DS, B);
    if (dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL(wide_attn_kernel<bf16_t>, grid, dim3(256), lds, stream, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)vt, (bf16_t*)o, N, n_valid, scale);
    else
        hipLaunchKernelGGL(wide_attn_kernel<half_t>, grid, dim3(256), lds, stream, (const half_t*)q, (const half_t*)k,
                           (const half_t*)vt, (half_t*)o, N, n_valid, scale);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_vae_conv_in(const float* lat, const float* wpq, const float* bpq, const void* w, const void* bias, void* y,
                          int B, int H, int W, int C, float scaling_factor, int dtype, hipStream_t stream) {
    DS_REQUIRE(C >= 8 && C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0,
               "vae_conv_in: C (%d) must be 8 x a divisor of 256", C);
    const size_t lds = (size_t)(C * 36 + C) * sizeof(float);
    DS_REQUIRE(lds <= 160 * 1024, "vae_conv_in: weights (%zu B) exceed LDS", lds);
    static unsigned long long attr_devs = 0;
    if (ds_first_on_device(attr_devs)) {
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_conv_in_kernel<bf16_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        DS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vae_conv_in_kernel<half_t>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int ppb = 256 / (C / 8);
    const int px_per_block = ppb * 16;  // the weight stage-in is amortised over 16 passes
    const long total = (long)B * H * W;
    dim3 grid((unsigned)((total + px_per_block - 1) / px_per_block));
    const float inv_sf = 1.0f / scaling_factor;
    if (dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL(vae_conv_in_kernel<bf16_t>, grid, dim3(256), lds, stream, lat, wpq, bpq, (const bf16_t*)w,
                           (const bf16_t*)bias, (bf16_t*)y, B, H, W, C, inv_sf, px_per_block);
    else
        hipLaunchKernelGGL(vae_conv_in_kernel<half_t>, grid, dim3(256), lds, stream, lat, wpq, bpq, (const half_t*)w,
                           (const half_t*)bias, (half_t*)y, B, H, W, C, inv_sf, px_per_block);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_vae_conv_out(const void* x, const void* w, const void* bias, float* img, int B, int H, int W, int C,
                           int denorm, int dtype, hipStream_t stream) {
    DS_REQUIRE(C % 8 == 0 && C > 0, "vae_conv_out: C (%d) must be a multiple of 8", C);
    const size_t lds = (size_t)27 * C * 2;
    DS_REQUIRE(lds <= 64 * 1024, "vae_conv_out: weights (%zu B) exceed 64 KiB of LDS", lds);
    const long groups = ((long)B * H * W + 31) / 32;
    dim3 grid((unsigned)(groups < 4096 ? groups : 4096));
    if (dtype == DS_DTYPE_BF16)
        hipLaunchKernelGGL(vae_conv_out_kernel<bf16_t>, grid, dim3(256), lds, stream, (const bf16_t*)x, (const bf16_t*)w,
                           (const bf16_t*)bias, img, B, H, W, C, denorm);
    else
        hipLaunchKernelGGL(vae_conv_out_kernel<half_t>, grid, dim3(256), lds, stream, (const half_t*)x, (const half_t*)w,
                           (const half_t*)bias, img, B, H, W, C, denorm);
    DS_LAUNCH_CHECK();
    return 0;
}
