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
// Exact (erf) GELU, branch-free, ONE transcendental: gelu(x) = max(x, 0) - |x| Phi(-|x|), and
// log2 Phi(-a) = log2(erfcx(a / sqrt 2) / 2) - a^2 log2(e) / 2 is smooth enough on a in [0, 5.75] for one degree-10 polynomial
// (max error 5e-7 in the exponent = 3.5e-7 RELATIVE in Phi(-a), so the negative tail keeps its relative accuracy - no
// 1 + erf cancellation; beyond 5.75 |gelu - max(x, 0)| is below half an f16 subnormal, |x| is clamped at 8 and the fit is only
// kept monotone).  tools/fit_gelu.py derives the coefficients and checks the f32 evaluation order below on ALL finite f16
// inputs (tests/test_gelu_fit.py repeats it on the constants parsed from this file): after rounding to f16 it differs from
// the exact value on 3e-4 of them (1.1e-4 / 5.7e-5 of N(0,1) / N(0,2) samples), always by one ulp;
// 0.5 x (1 + erff(x / sqrt 2)) - what torch's fp32 kernel computes - differs on 5e-3 of them (1.6e-2 of N(0,2)), by up to two.
// Round 4: replaces the round-1 form t P(t) exp(-z^2), t = 1 / (1 + 0.37 z) (same accuracy class: 4e-5) - v_rcp_f32 and
// v_exp_f32 are both quarter-rate, and in the 256 x 256 GEGLU GEMM the 64 evaluations per lane were ~4 us of every
// 41-us tile with the matrix pipe idle: 10 fma (5 v_pk_fma_f32 per pair) + v_exp_f32 + 3 instead of
// ~15 + v_rcp_f32 + v_exp_f32.  A NaN gate comes out as 0 (v_max / v_min drop it); the hidden value it multiplies in GEGLU
// is NaN then anyway (same input row).
__device__ __forceinline__ float ds_gelu_erf(float x) {
    const float a = fminf(fabsf(x), 8.0f);
    float q = -9.521883721e-09f;
    q = fmaf(q, a, 3.357761849e-07f);
    q = fmaf(q, a, -4.994478671e-06f);
    q = fmaf(q, a, 3.903887141e-05f);
    q = fmaf(q, a, -1.394536987e-04f);
    q = fmaf(q, a, -3.214741642e-04f);
    q = fmaf(q, a, 7.390159313e-03f);
    q = fmaf(q, a, -5.278805848e-02f);
    q = fmaf(q, a, -4.590846261e-01f);
    q = fmaf(q, a, -1.151124720e+00f);
    q = fmaf(q, a, -9.999994968e-01f);
    float y = fmaf(-a, __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));  // q = log2 Phi(-a)
    // The f32 result is pinned in a register: callers round it to f16, and a compiler free to fold that rounding into this fma
    // (v_fma_mixlo_f16 rounds the exact sum once; fma + convert rounds twice) does so for some call sites and not for others
    // - the bits of an element would then depend on where in a tile it sits (profiles/r04_determinism_bisect.txt).
    asm("" : "+v"(y));
    return y;
}
// Two values at once, the Horner chain spelled as <2 x float> fmas: left to itself the compiler evaluates the scalar form with
// v_fmaak_f32 (literal addend, one element per instruction); as vector fmas the chain is 10 v_pk_fma_f32 per PAIR with the
// coefficients splat from registers.  Bit-identical to ds_gelu_erf per element (same operations in the same order).
__device__ __forceinline__ f32x2 ds_gelu_erf2(f32x2 x) {
    const f32x2 a = {fminf(fabsf(x[0]), 8.0f), fminf(fabsf(x[1]), 8.0f)};
    auto k = [](float c) { return f32x2{c, c}; };
    f32x2 q = k(-9.521883721e-09f);
    q = __builtin_elementwise_fma(q, a, k(3.357761849e-07f));
    q = __builtin_elementwise_fma(q, a, k(-4.994478671e-06f));
    q = __builtin_elementwise_fma(q, a, k(3.903887141e-05f));
    q = __builtin_elementwise_fma(q, a, k(-1.394536987e-04f));
    q = __builtin_elementwise_fma(q, a, k(-3.214741642e-04f));
    q = __builtin_elementwise_fma(q, a, k(7.390159313e-03f));
    q = __builtin_elementwise_fma(q, a, k(-5.278805848e-02f));
    q = __builtin_elementwise_fma(q, a, k(-4.590846261e-01f));
    q = __builtin_elementwise_fma(q, a, k(-1.151124720e+00f));
    q = __builtin_elementwise_fma(q, a, k(-9.999994968e-01f));
    const f32x2 r = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2 relu = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    f32x2 y = __builtin_elementwise_fma(-a, r, relu);
    asm("" : "+v"(y));  // as in ds_gelu_erf: the rounding to f16 stays a separate instruction
    return y;
}

// Source index of nearest-neighbour resizing exactly as ATen computes it (F.interpolate(mode="nearest"), the op behind
// diffusers' Upsample2D [3P]): min(int(floorf(dst * scale)), in - 1) with scale = float(in) / out (0.5 for the plain x2
// case, where it equals dst >> 1).  The reference reaches the general case whenever a latent side is not a multiple of 4:
// diffusers then resizes to the skip tensor's size (forward_upsample_size).
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    return min((int)floorf((float)dst * scale), in_size - 1);
}

// v_mov_b32 with a DPP control: 0xB1 = quad_perm [1,0,3,2] (lane ^ 1), 0x4E = quad_perm [2,3,0,1] (lane ^ 2),
// 0x141 = row_half_mirror (lane j of every 8 <-> lane 7 - j)
template <int CTRL>
__device__ __forceinline__ float ds_dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
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
// One LDS-DMA piece (16 bytes per lane: lane l lands at lds_addr + 16 l) issued from inline asm: global address = uniform 64-bit
// base + the lane's 32-bit byte offset, LDS base through M0.  Why not __builtin_amdgcn_global_load_lds: the compiler models that
// builtin as a FLAT access that may touch LDS and so may bump lgkmcnt out of order ("pending flat"): while one is in flight -
// and the counted vmcnt waits of these kernels are inline asm it cannot see, so for it one always is - EVERY LDS wait it places is
// s_waitcnt lgkmcnt(0), which kills any fragment read-ahead (found in round 5 on self_attn_sp_kernel: half of its MFMAs sat behind
// a full LDS round trip).  The hardware tracks LDS-DMA with vmcnt only (gemm_pp_kernel's lgkmcnt(0) waits would otherwise wait
// microseconds for its staging), so counted lgkmcnt waits beside DMA in flight are exact.  "memory": nothing moves across it.
__device__ __forceinline__ void lds_dma16(const void* base, unsigned byte_off, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(byte_off), "s"(base), "s"(lds_addr) : "memory", "m0");
}

// the same with a per-lane 64-bit global address (the halo-patch pieces: image rows for some lanes, the zero page for others)
__device__ __forceinline__ void lds_dma16_v(const void* lane_ptr, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lane_ptr), "s"(lds_addr) : "memory", "m0");
}

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
