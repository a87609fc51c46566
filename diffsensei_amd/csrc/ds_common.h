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

// ---- device helpers ------------------------------------------------------------------------------
__device__ __forceinline__ float ds_silu(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float ds_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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
