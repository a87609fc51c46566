// Shared device/host helpers for the DiffSensei gfx950 kernels.
// Everything here is written for CDNA4 only: wave = 64 lanes, MFMA 32x32x16 f16, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16_t;  // VAE decoder path: fp16 overflows there (the reference upcasts the VAE to fp32)
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

enum { DS_DTYPE_F16 = 0, DS_DTYPE_BF16 = 1 };

// element-type traits of the kernels that exist for both 16-bit formats (same bytes moved, different MFMA / converts)
template <typename T>
struct Elt;
template <>
struct Elt<half_t> {
    typedef h8 v8;
    typedef h4 v4;
    typedef h2 v2;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Elt<bf16_t> {
    typedef b8 v8;
    typedef b4 v4;
    typedef b2 v2;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

#define DS_WAVE 64

// ---- error plumbing (C ABI never throws; last error is a thread-local string) -------------------
void ds_set_error(const char* fmt, ...);
#define DS_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            ds_set_error(__VA_ARGS__);   \
            return -1;                   \
        }                                \
    } while (0)
#define DS_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) {                                            \
            ds_set_error("%s failed: %s", #call, hipGetErrorString(e__));   \
            return -2;                                                      \
        }                                                                   \
    } while (0)
#define DS_LAUNCH_CHECK()                                                   \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            ds_set_error("kernel launch failed: %s", hipGetErrorString(e__)); \
            return -3;                                                      \
        }                                                                   \
    } while (0)

// One-off per-DEVICE set-up at a launch site (hipFuncSetAttribute for > 64 KiB of dynamic LDS): function attributes
// are per device, and ops.bind_device can move a process to another device, so a process-wide `static bool` is not
// enough.  `seen` is the call site's own static bit mask (devices 0..63).
inline bool ds_first_on_device(unsigned long long& seen) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;   // unknown: redo the set-up, it is idempotent
    const unsigned long long bit = 1ull << dev;
    if (seen & bit) return false;
    seen |= bit;
    return true;
}

// ---- device helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float ds_silu(float x) { return x / (1.0f + __expf(-x)); }
// Exact (erf) GELU, branch-free: gelu(x) = x * Phi(x) with Phi(-|x|) = erfc(z)/2, z = |x|/sqrt(2), and
// erfc(z) = t P(t) exp(-z^2), t = 1/(1 + 0.37 z) - degree-7 fit of erfc(z) exp(z^2), max relative error 3.9e-7 on
// z in [0, 8] (tools/fit_gelu.py prints these coefficients and the error statistics).  Half the instructions of
// 0.5 x (1 + erff(x/sqrt 2)) (ocml's erff is two divergent branches), and no 1 + erf cancellation in the negative
// tail: after rounding to f16 it differs from the exact value in 4e-5 of cases (that form: 1.6e-2).  In the 256 x 256
// GEGLU GEMM the erf was 13-20 % of the kernel (profiles/r01_gemm_pp_microbench.txt).
__device__ __forceinline__ float ds_gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.26162950903902255f, 1.0f));
    float P = -7.287154991e-02f;
    P = fmaf(P, t, 2.236382245e-01f);
    P = fmaf(P, t, -1.017244238e-01f);
    P = fmaf(P, t, 1.651046857e-01f);
    P = fmaf(P, t, 7.372602999e-02f);
    P = fmaf(P, t, 1.079816715e-01f);
    P = fmaf(P, t, 1.041452194e-01f);
    const float u = ax * 0.8493218002880191f;                  // u^2 = z^2 log2(e)
    const float r = (P * t) * __builtin_amdgcn_exp2f(-(u * u));  // erfc(z) / 2 = Phi(-|x|)
    return x * (x < 0.f ? r : 1.0f - r);
}

// Source index of nearest-neighbour resizing exactly as ATen computes it (F.interpolate(mode="nearest"), the op behind
// diffusers' Upsample2D [3P]): min(int(floorf(dst * scale)), in - 1) with scale = float(in) / out (0.5 for the plain x2
// case, where it equals dst >> 1).  The reference reaches the general case whenever a latent side is not a multiple of 4:
// diffusers then resizes to the skip tensor's size (forward_upsample_size).
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    return min((int)floorf((float)dst * scale), in_size - 1);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: the dispatcher places block b on XCD b % 8, so hand
// each XCD one contiguous chunk of the logical tile order (neighbouring tiles share operand panels in
// that XCD's L2).  Speed only — correctness never depends on it.
// Logical tile id -> (tm, tn) in GROUP x GROUP super-tiles (column-major inside a group of GROUP tile rows), so the
// ~64 tiles an XCD has in flight touch ~8 A panels + ~8 W panels instead of 1 + 64: the per-XCD L2 (4 MiB) then
// serves most panel re-reads (measured on the FF GEMM: L2 hit rate 49 % -> see DESIGN.md).
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int& tm, int& tn) {
    constexpr int GROUP = 8;
    const int per_group = GROUP * tiles_n;
    const int group = id / per_group;
    const int first_m = group * GROUP;
    const int rows = min(tiles_m - first_m, GROUP);
    const int in_group = id - group * per_group;
    tm = first_m + in_group % rows;
    tn = in_group / rows;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int NX = 8;
    if (nblk < NX * 2) return bid;
    int xcd = bid % NX, local = bid / NX;
    int q = nblk / NX, r = nblk % NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + local;
}
