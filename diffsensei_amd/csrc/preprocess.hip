// Character-reference pre-processing on the device: the 8-bit separable resize behind
// `CLIPImageProcessor()` / `ViTImageProcessor()` (reference src/pipelines/pipeline_diffsensei.py:125-126, which runs it
// in Pillow on the host) + centre crop + rescale + normalisation, so the reference images are uploaded once as raw RGB
// bytes and never come back.
//
// Integer work, HBM/latency-bound, bit-exact by construction: the host computes Pillow's coefficient tables (double
// arithmetic, rounded to 22-bit fixed point exactly like libImaging's normalize_coeffs_8bpc) and the kernels do what
// ImagingResampleHorizontal_8bpc / Vertical_8bpc do: int32 accumulate from 1 << 21, arithmetic shift by 22, clamp to a
// byte, with the 8-bit intermediate image between the passes.  Nothing here wants MFMA: a 224 x 224 x 3 output is 150 K
// dot products of <= ~20 taps.
#include "ds_common.h"
#include "ds_kernels.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int acc) {
    const int v = acc >> PRECISION_BITS;  // arithmetic shift, like the C source
    return (uint8_t)min(max(v, 0), 255);
}

// dst[y][xo][c] = clip8(2^21 + sum_k src[y][first[xo] + k][c] * taps[xo][k]);  one thread per (xo, c), one row per blockIdx.y
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, int W, const int* __restrict__ first,
                                                       const int* __restrict__ count, const int* __restrict__ taps,
                                                       int ksize, int out_w, uint8_t* __restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_w * 3) return;
    const int xo = idx / 3, c = idx - xo * 3, y = blockIdx.y;
    const uint8_t* row = src + ((long)y * W + first[xo]) * 3 + c;
    const int* k = taps + (long)xo * ksize;
    const int n = count[xo];
    int acc = 1 << (PRECISION_BITS - 1);
    for (int i = 0; i < n; ++i) acc += (int)row[i * 3] * k[i];
    dst[((long)y * out_w + xo) * 3 + c] = clip8(acc);
}

// vertical pass restricted to the crop window, then rescale + normalise:
//   u = clip8(2^21 + sum_k tmp[first[yo+top] + k][xo+left][c] * taps[yo+top][k])
//   out_f32[c][yo][xo] = (u * scale - mean[c]) / std[c]          out_u8[yo][xo][c] = u   (optional, for the parity test)
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, int Wt,
                                                            const int* __restrict__ first, const int* __restrict__ count,
                                                            const int* __restrict__ taps, int ksize, int top, int left,
                                                            int out_h, int out_w, float scale, float m0, float m1, float m2,
                                                            float s0, float s1, float s2, float* __restrict__ out_f32,
                                                            uint8_t* __restrict__ out_u8) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= out_w * 3) return;
    const int xo = idx / 3, c = idx - xo * 3, yo = blockIdx.y;
    const int ys = yo + top;
    const uint8_t* col = tmp + ((long)first[ys] * Wt + xo + left) * 3 + c;
    const int* k = taps + (long)ys * ksize;
    const int n = count[ys];
    int acc = 1 << (PRECISION_BITS - 1);
    for (int i = 0; i < n; ++i) acc += (int)col[(long)i * Wt * 3] * k[i];
    const uint8_t u = clip8(acc);
    const float mean = c == 0 ? m0 : c == 1 ? m1 : m2, sd = c == 0 ? s0 : c == 1 ? s1 : s2;
    out_f32[((long)c * out_h + yo) * out_w + xo] = ((float)u * scale - mean) / sd;
    if (out_u8) out_u8[((long)yo * out_w + xo) * 3 + c] = u;
}

}  // namespace

int ds_launch_resize_h(const uint8_t* src, int H, int W, const int* first, const int* count, const int* taps, int ksize,
                       int out_w, uint8_t* dst, hipStream_t stream) {
    DS_REQUIRE(H > 0 && W > 0 && out_w > 0 && ksize > 0, "resize_h: bad shape H=%d W=%d out_w=%d ksize=%d", H, W, out_w, ksize);
    DS_REQUIRE(src && first && count && taps && dst, "resize_h: null operand");
    hipLaunchKernelGGL(resize_h_kernel, dim3((out_w * 3 + 255) / 256, H), dim3(256), 0, stream, src, W, first, count, taps,
                       ksize, out_w, dst);
    DS_LAUNCH_CHECK();
    return 0;
}

int ds_launch_resize_v_norm(const uint8_t* tmp, int Ht, int Wt, const int* first, const int* count, const int* taps,
                            int ksize, int top, int left, int out_h, int out_w, float scale, const float* mean3,
                            const float* std3, float* out_f32, uint8_t* out_u8, hipStream_t stream) {
    DS_REQUIRE(Ht > 0 && Wt > 0 && out_h > 0 && out_w > 0 && ksize > 0, "resize_v: bad shape");
    DS_REQUIRE(top >= 0 && left >= 0 && left + out_w <= Wt, "resize_v: crop window [%d,%d)+%d outside width %d", left,
               left + out_w, top, Wt);
    DS_REQUIRE(tmp && first && count && taps && mean3 && std3 && out_f32, "resize_v: null operand");
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((out_w * 3 + 255) / 256, out_h), dim3(256), 0, stream, tmp, Wt, first,
                       count, taps, ksize, top, left, out_h, out_w, scale, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                       std3[2], out_f32, out_u8);
    DS_LAUNCH_CHECK();
    return 0;
}
